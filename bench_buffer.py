#!/usr/bin/env python3
"""Measurement of the device rollout buffer (not the headline metric, which bench.py reports).

Workload: the reference's training shape (scripts/train_*.sh: --buffer-size 3000, hidden 128, 5 mini-batches, chunk length 8)
at the BASELINE batch (4096 envs x 2 agents = 8192 columns). Prints ONE JSON line:
  compute_returns  GAE + proper time limits: algorithmic bytes = T*N*(4 reads + 1 write)*4 B, kernel time from HIP events on the
                   buffer's stream -> GB/s against the 8 TB/s HBM peak
  minibatch        one recurrent_generator mini-batch gathered on the device (device-resident outputs): bytes written / time
  cpu_baseline     the numpy oracle's compute_returns on a bounded sample of columns, scaled to GB/s of the same accounting
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffer-size", type=int, default=3000)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=2)
    ap.add_argument("--obs-dim", type=int, default=15)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--repeats", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import aircombat_selfplay_amd as pkg
    T, E, A, H = args.buffer_size, args.envs, args.agents, args.hidden
    N = E * A
    cfg = types.SimpleNamespace(buffer_size=T, n_rollout_threads=E, gamma=0.99, use_proper_time_limits=True, use_gae=True, gae_lambda=0.95,
                                recurrent_hidden_size=H, recurrent_hidden_layers=1)
    buf = pkg.DeviceReplayBuffer(cfg, A, args.obs_dim, 4)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    # fill the arrays the recurrence reads, in place on the device
    for name, fn in (("rewards", lambda v: v.normal_(generator=g)), ("value_preds", lambda v: v.normal_(generator=g)),
                     ("masks", lambda v: v.copy_((torch.rand(v.shape, device=v.device, generator=g) > 0.01).float())),
                     ("bad_masks", lambda v: v.copy_((torch.rand(v.shape, device=v.device, generator=g) > 0.01).float())),
                     ("obs", lambda v: v.normal_(generator=g))):
        fn(buf.device_tensor(name))
    torch.cuda.synchronize()
    nv = torch.randn(N, device="cuda:0", generator=g)
    times = []
    for _ in range(args.repeats + 2):
        buf.compute_returns(nv, on_device=True)
        times.append(buf.last_returns_kernel_ms())
    ms = float(np.mean(times[2:]))
    alg_bytes = T * N * 5 * 4
    out = {"metric": "rollout-buffer compute_returns (GAE, proper time limits)", "unit": "GB/s", "value": alg_bytes / (ms * 1e-3) / 1e9,
           "dtype": "f32", "data": "synthetic", "higher_is_better": True,
           "config": {"workload": "reference training shape at the BASELINE batch", "buffer_size": T, "envs": E, "agents": A, "columns": N,
                      "hidden_size": H, "obs_dim": args.obs_dim},
           "roofline": {"bound": "hbm", "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": "rbuf::returns_kernel<true, true, 32, 128>", "kernel_ms": ms,
                        "algorithmic_bytes_per_launch": alg_bytes, "traffic": None}}
    # one mini-batch gathered into device-resident outputs
    L, MB = 8, 5
    chunks_total = E * T // L
    per = chunks_total // MB
    order = np.random.default_rng(3).permutation(chunks_total)[:per].astype(np.int32)
    lib = buf.lib
    lib.ac_buffer_advantages(buf._h)
    widths = {"obs": args.obs_dim, "actions": 4, "masks": 1, "action_log_probs": 1, "advantages": 1, "returns": 1, "value_preds": 1}
    outs = {k: torch.empty(L * per * w, device="cuda:0") for k, w in widths.items()}
    outs["rnn_states_actor"] = torch.empty(per * H, device="cuda:0")
    outs["rnn_states_critic"] = torch.empty(per * H, device="cuda:0")
    batch = pkg.capi.AcBufferBatch(**{k: v.data_ptr() for k, v in outs.items()})
    gathered = sum(v.numel() for v in outs.values()) * 4
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        rc = lib.ac_buffer_minibatch(buf._h, order.ctypes.data, per, L, C.byref(batch), 1)
        assert rc == 0, lib.last_error()
    dt = (time.perf_counter() - t0) / reps
    out["minibatch"] = {"chunks": int(per), "chunk_len": L, "rows": int(per * L), "bytes_out": int(gathered), "ms": dt * 1e3,
                        "GBps_out": gathered / dt / 1e9, "note": "11 gather launches + chunk upload, wall time incl. launch overhead"}
    if not args.no_cpu_baseline:
        from oracle.rollout_buffer import OracleRolloutBuffer
        Es = 64
        ref = OracleRolloutBuffer(T, Es, A, 1, 1, 1, 1, 0.99, 0.95, True, True)
        rng = np.random.default_rng(0)
        ref.rewards[:] = rng.normal(size=ref.rewards.shape); ref.value_preds[:] = rng.normal(size=ref.value_preds.shape)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 10.0:
            ref.compute_returns(rng.normal(size=(Es, A, 1)).astype(np.float32))
            n += 1
        dt = (time.perf_counter() - t0) / n
        out["cpu_baseline"] = {"value": T * Es * A * 20 / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": f"numpy oracle (the reference's own loop structure), {T} steps x {Es * A} columns, {n} repeats in 10 s"}
    buf.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
