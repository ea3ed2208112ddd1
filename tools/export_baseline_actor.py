#!/usr/bin/env python3
"""Export the reference's low-level controller weights (envs/JSBSim/model/baseline_model.pt, a data file) to the flat fp32 blob
the HIP controller kernel and the CPU checker read: aircombat-selfplay_amd/data/baseline_actor.f32.

Layout (little-endian float32, row-major [out][in], offsets in floats; BaselineActor of baseline_actor.py:91-110):
  W1[128][12] b1[128] g1[128] be1[128]          MLP layer 1: Linear, ReLU, LayerNorm
  W2[128][128] b2[128] g2[128] be2[128]         MLP layer 2
  Wih[384][128] Whh[384][128] bih[384] bhh[384] GRU, gate order r, z, n (torch.nn.GRU)
  g3[128] be3[128]                              LayerNorm on the GRU output
  Wa[153][128] ba[153]                          the four Categorical heads [41, 41, 41, 30] stacked
Runs only where /root/reference exists (the build container)."""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ["base.mlp.fc.0.weight", "base.mlp.fc.0.bias", "base.mlp.fc.2.weight", "base.mlp.fc.2.bias",
         "base.mlp.fc.3.weight", "base.mlp.fc.3.bias", "base.mlp.fc.5.weight", "base.mlp.fc.5.bias",
         "rnn.gru.weight_ih_l0", "rnn.gru.weight_hh_l0", "rnn.gru.bias_ih_l0", "rnn.gru.bias_hh_l0",
         "rnn.norm.weight", "rnn.norm.bias"]
HEADS_W = [f"act.action_outs.{i}.logits_net.weight" for i in range(4)]
HEADS_B = [f"act.action_outs.{i}.logits_net.bias" for i in range(4)]


def main():
    sd = torch.load(os.path.join(REF, "envs/JSBSim/model/baseline_model.pt"), map_location="cpu")
    parts = [sd[k].numpy().astype(np.float32).ravel() for k in ORDER]
    parts += [np.concatenate([sd[k].numpy().astype(np.float32) for k in HEADS_W], axis=0).ravel()]
    parts += [np.concatenate([sd[k].numpy().astype(np.float32) for k in HEADS_B], axis=0).ravel()]
    blob = np.concatenate(parts)
    assert blob.size == 137753, blob.size
    out = os.path.join(ROOT, "aircombat-selfplay_amd", "data", "baseline_actor.f32")
    blob.tofile(out)
    print(out, blob.size, "floats")


if __name__ == "__main__":
    sys.exit(main())
