"""Rollout buffer: the oracle against golden vectors from the reference's own ReplayBuffer / SharedReplayBuffer
(CPU), and the HIP buffer behind include/aircombat_buffer.h against the oracle (GPU).

Bars: compute_returns is float32 arithmetic in a fixed order -> BIT-EXACT (oracle vs reference, HIP vs oracle); data movement
(insert, after_update, mini-batch gather) is exact; the normalised advantages involve a mean / std reduction whose summation order
differs (numpy pairwise float32 vs fp64 block sums) -> |d| <= 2e-6 + 2e-6*|x|.
"""
import os
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = [(shared, proper, gae) for shared in (False, True) for proper in (False, True) for gae in (False, True)]


def key_of(shared, proper, gae):
    return f"{'shared' if shared else 'single'}_{'proper' if proper else 'plain'}_{'gae' if gae else 'mc'}"


def feed(buf, d, shared, set0):
    """The golden input stream through insert(); `set0(name, value)` seeds slot 0 like the runners do after reset."""
    T = int(d["dims"][0])
    set0("obs", d["in_obs"][0]); set0("rnn_states_actor", d["in_rnn_a"][0]); set0("rnn_states_critic", d["in_rnn_c"][0])
    if shared:
        set0("share_obs", d["in_share_obs"][0])
    for t in range(T):
        kw = dict(obs=d["in_obs"][t + 1], actions=d["in_actions"][t], rewards=d["in_rewards"][t], masks=d["in_masks"][t],
                  action_log_probs=d["in_logp_shared" if shared else "in_logp"][t], value_preds=d["in_values"][t],
                  rnn_states_actor=d["in_rnn_a"][t + 1], rnn_states_critic=d["in_rnn_c"][t + 1], bad_masks=d["in_bad_masks"][t])
        if shared:
            kw.update(share_obs=d["in_share_obs"][t + 1], active_masks=d["in_active_masks"][t])
        buf.insert(**kw)


def make_oracle(d, shared, proper, gae):
    from oracle.rollout_buffer import OracleRolloutBuffer
    T, E, A, OBS, SH, ACT, _, H = (int(x) for x in d["dims"])
    return OracleRolloutBuffer(T, E, A, OBS, ACT, 1, H, 0.99, 0.95, gae, proper, share_dim=SH if shared else 0)


@pytest.mark.parametrize("shared,proper,gae", MODES)
def test_oracle_matches_reference_buffer(shared, proper, gae):
    d = np.load(os.path.join(G, "rollout_buffer.npz"))
    b = make_oracle(d, shared, proper, gae)
    feed(b, d, shared, lambda name, v: getattr(b, name).__setitem__(0, v))
    assert b.step == 0
    b.compute_returns(d["in_next_value"])
    key = key_of(shared, proper, gae)
    assert (b.returns == d[key + "_returns"]).all()                    # bit-exact
    assert (b.advantages == d[key + "_advantages"]).all()
    if proper and gae:
        assert (b.masks == d[key + "_masks"]).all() and (b.bad_masks == d[key + "_bad_masks"]).all()
        if shared:
            assert (b.bad_masks == 1).all()                            # the shared insert drops bad_masks (buffer.py:343)
        L, MB = (int(x) for x in d["chunk"])
        for k, batch in enumerate(b.minibatches(d["perm"], MB, L)):
            assert len(batch) == (11 if shared else 9)
            for j, arr in enumerate(batch):
                want = d[f"{key}_batch{k}_{j}"]
                assert arr.shape == want.shape and (arr == want).all(), (k, j)
        b.after_update()
        assert (b.obs[0] == d[key + "_after_obs0"]).all() and (b.masks[0] == d[key + "_after_masks0"]).all()


def _args(T, E, H, proper, gae, gamma=0.99, lam=0.95):
    return types.SimpleNamespace(buffer_size=T, n_rollout_threads=E, gamma=gamma, use_proper_time_limits=proper, use_gae=gae, gae_lambda=lam,
                                 recurrent_hidden_size=H, recurrent_hidden_layers=1)


def make_device(pkg, d, shared, proper, gae):
    T, E, A, OBS, SH, ACT, _, H = (int(x) for x in d["dims"])
    if shared:
        return pkg.DeviceSharedReplayBuffer(_args(T, E, H, proper, gae), A, OBS, SH, ACT)
    return pkg.DeviceReplayBuffer(_args(T, E, H, proper, gae), A, OBS, ACT)


@pytest.mark.gpu
@pytest.mark.parametrize("shared,proper,gae", MODES)
def test_device_buffer_matches_golden_and_oracle(pkg, shared, proper, gae):
    d = np.load(os.path.join(G, "rollout_buffer.npz"))
    dev, ref = make_device(pkg, d, shared, proper, gae), make_oracle(d, shared, proper, gae)
    feed(dev, d, shared, lambda name, v: dev.set_slot(name, 0, v))
    feed(ref, d, shared, lambda name, v: getattr(ref, name).__setitem__(0, v))
    assert dev.step == 0
    dev.compute_returns(d["in_next_value"]); ref.compute_returns(d["in_next_value"])
    key = key_of(shared, proper, gae)
    for name in ("obs", "actions", "rewards", "masks", "bad_masks", "action_log_probs", "value_preds", "rnn_states_actor", "rnn_states_critic") + \
            (("share_obs", "active_masks") if shared else ()):
        assert (dev.array(name) == getattr(ref, name)).all(), name
    assert (dev.array("returns") == d[key + "_returns"]).all()         # bit-exact against the reference's numpy recurrence
    adv = dev.advantages
    assert (np.abs(adv - d[key + "_advantages"]) <= 2e-6 + 2e-6 * np.abs(d[key + "_advantages"])).all()
    L, MB = (int(x) for x in d["chunk"])
    names = dev._BATCH + ("rnn_states_actor", "rnn_states_critic")
    for k, (got, want) in enumerate(zip(dev.recurrent_generator(dev.advantages, MB, L, chunk_order=d["perm"]) if shared else
                                        pkg.DeviceReplayBuffer.recurrent_generator(dev, MB, L, chunk_order=d["perm"]),
                                        ref.minibatches(d["perm"], MB, L))):
        for name, a, w in zip(names, got, want):
            assert a.shape == w.shape, (k, name, a.shape, w.shape)
            if name == "advantages":
                assert (np.abs(a - w) <= 2e-6 + 2e-6 * np.abs(w)).all()
            else:
                assert (a == w).all(), (k, name)
    dev.after_update(); ref.after_update()
    for name in ("obs", "masks", "bad_masks", "rnn_states_actor"):
        assert (dev.array(name)[0] == getattr(ref, name)[0]).all(), name
    dev.clear()
    assert dev.step == 0 and (dev.array("masks") == 1).all() and (dev.array("returns") == 0).all()
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("proper,gae", [(True, True), (False, False)])
def test_device_buffer_large_and_ragged_shapes(pkg, proper, gae):
    """Sizes that do not divide the kernel's blocking (T = 37 steps against 8-step load blocks, 3 x 43 = 129 columns against
    64-lane workgroups), device-side insert, and a chunk length that does not divide T (chunks straddle columns)."""
    import torch
    from oracle.rollout_buffer import OracleRolloutBuffer
    T, E, A, OBS, ACT, H = 37, 43, 3, 15, 4, 128
    rng = np.random.default_rng(5)
    dev = pkg.DeviceReplayBuffer(_args(T, E, H, proper, gae, gamma=0.97, lam=0.9), A, OBS, ACT)
    ref = OracleRolloutBuffer(T, E, A, OBS, ACT, 1, H, 0.97, 0.9, gae, proper)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    for t in range(T):
        kw = dict(obs=f(E, A, OBS), actions=f(E, A, ACT), rewards=f(E, A, 1), masks=(rng.random((E, A, 1)) > 0.1).astype(np.float32),
                  action_log_probs=f(E, A, 1), value_preds=f(E, A, 1), rnn_states_actor=f(E, A, 1, H), rnn_states_critic=f(E, A, 1, H),
                  bad_masks=(rng.random((E, A, 1)) > 0.05).astype(np.float32))
        ref.insert(**kw)
        if t % 2:   # every other step through device pointers (the env handle's buffers in production)
            dev.insert(on_device=True, **{k: torch.from_numpy(v).to("cuda:0") for k, v in kw.items()})
        else:
            dev.insert(**kw)
    nv = f(E, A, 1)
    ref.compute_returns(nv); dev.compute_returns(torch.from_numpy(nv).to("cuda:0"), on_device=True)
    assert (dev.array("returns") == ref.returns).all()
    L, MB = 5, 4
    order = rng.permutation(E * T // L)
    for got, want in zip(pkg.DeviceReplayBuffer.recurrent_generator(dev, MB, L, chunk_order=order), ref.minibatches(order, MB, L)):
        for j, (a, w) in enumerate(zip(got, want)):
            assert a.shape == w.shape
            assert (np.abs(a - w) <= 2e-6 + 2e-6 * np.abs(w)).all() if j == 4 else (a == w).all(), j
    view = dev.device_tensor("returns")
    assert tuple(view.shape) == (T + 1, E, A, 1) and (view.cpu().numpy() == ref.returns).all()
    dev.close()


@pytest.mark.gpu
def test_device_buffer_rejects_bad_arguments(pkg):
    dev = pkg.DeviceReplayBuffer(_args(4, 2, 8, True, True), 2, 5, 4)
    with pytest.raises(RuntimeError, match="chunk index outside"):
        dev.minibatch([100], 2)
    import ctypes as C
    ptr, n = C.c_void_p(), C.c_int64()
    assert dev.lib.ac_buffer_device_ptr(dev._h, pkg.capi.AC_BUF_SHARE_OBS, C.byref(ptr), C.byref(n)) != 0
    assert "no such field" in dev.lib.last_error()
    dev.close()


@pytest.mark.gpu
def test_env_to_buffer_rollout_stays_on_the_device(pkg):
    """N2 + N4 together: a rollout in which nothing crosses PCIe. The env handle steps on device-resident actions
    (step_device), its own HBM output buffers (obs, rewards, dones) go into the rollout buffer device-to-device, and the returns
    are computed in place. Checked against the same rollout collected through the host interface and the numpy oracle buffer."""
    import torch
    from oracle.rollout_buffer import OracleRolloutBuffer
    T, E, H = 24, 33, 16
    cfg = pkg.default_config("singlecombat")
    env_d, env_h = pkg.HipVecEnv(cfg, E, seed=2), pkg.HipVecEnv(cfg, E, seed=2)
    A, OBS = env_d.num_agents, env_d.obs_dim
    dev = pkg.DeviceReplayBuffer(_args(T, E, H, True, True), A, OBS, 4)
    ref = OracleRolloutBuffer(T, E, A, OBS, 4, 1, H, 0.99, 0.95, True, True)
    env_d.reset(); obs0 = env_h.reset()
    act_d, obs_d, rew_d, done_d, _ = env_d.device_tensors()
    dev.set_slot("obs", 0, obs0); ref.obs[0] = obs0
    rng = np.random.default_rng(12)
    zeros_h = np.zeros((E, A, 1, H), dtype=np.float32)
    zeros_d = torch.zeros((E, A, 1, H), device="cuda:0")
    for t in range(T):
        a = rand_actions(rng, E, A)
        logp, val = rng.normal(size=(E, A, 1)).astype(np.float32), rng.normal(size=(E, A, 1)).astype(np.float32)
        # device side: actions written in place, step, outputs inserted without leaving HBM
        act_d.copy_(torch.from_numpy(a))
        env_d.step_device(act_d.data_ptr()); env_d.sync()
        masks_d = (1.0 - done_d.float()).contiguous()
        dev.insert(obs_d, act_d, rew_d, masks_d, torch.from_numpy(logp).cuda(), torch.from_numpy(val).cuda(), zeros_d, zeros_d, on_device=True)
        # host side: the reference's call sequence (runner/jsbsim_runner.py:116-133)
        o, r, d, _ = env_h.step(a)
        ref.insert(o, a, r, (1.0 - d).astype(np.float32), logp, val, zeros_h, zeros_h)
    nv = rng.normal(size=(E, A, 1)).astype(np.float32)
    dev.compute_returns(torch.from_numpy(nv).cuda(), on_device=True); ref.compute_returns(nv)
    for name in ("obs", "actions", "rewards", "masks", "value_preds", "returns"):
        assert (dev.array(name) == getattr(ref, name)).all(), name     # two handles, same seed, same actions: bit-identical rollouts
    dev.close(); env_d.close(); env_h.close()


def rand_actions(rng, E, A):
    return np.stack([rng.integers(0, n, size=(E, A)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32)


@pytest.mark.gpu
def test_returns_at_training_size_properties(pkg):
    """The reference's training shape (buffer_size 3000) at the BASELINE batch (8192 columns), checked through properties that do
    not need the slow reference loop: with gamma = lambda = 1, masks = bad_masks = 1 and V = 0 the GAE return is the suffix sum
    of the rewards plus the bootstrap value; sampled columns equal the numpy oracle bit for bit with random masks, in both tilings
    of the kernel (32-column workgroups below 16 384 columns, 64-column workgroups from there)."""
    import torch
    from oracle.rollout_buffer import OracleRolloutBuffer
    T, E, A, H = 3000, 4096, 2, 1
    g = torch.Generator(device="cuda:0").manual_seed(7)
    buf = pkg.DeviceReplayBuffer(_args(T, E, H, True, True, gamma=1.0, lam=1.0), A, 1, 1)
    rew = buf.device_tensor("rewards").normal_(generator=g)
    buf.device_tensor("value_preds").zero_()
    nv = torch.randn(E * A, device="cuda:0", generator=g)
    buf.compute_returns(nv, on_device=True)
    R = buf.device_tensor("returns")[:T, ..., 0].double()
    want = torch.flip(torch.cumsum(torch.flip(rew[..., 0].double(), [0]), 0), [0]) + nv.view(E, A).double()
    assert float((R - want).abs().max()) <= 2e-2        # 3000 float32 additions of N(0,1) terms: |sum| ~ 55, ulp 4e-6, error random-walks
    buf.close()
    # random masks / values at gamma 0.99: sampled columns against the oracle, bit for bit
    buf = pkg.DeviceReplayBuffer(_args(T, E, H, True, True), A, 1, 1)
    for name in ("rewards", "value_preds"):
        buf.device_tensor(name).normal_(generator=g)
    for name in ("masks", "bad_masks"):
        t = buf.device_tensor(name)
        t.copy_((torch.rand(t.shape, device="cuda:0", generator=g) > 0.02).float())
    buf.compute_returns(nv, on_device=True)
    cols = [0, 1, 31, 32, 63, 64, 4095, 8191]
    ref = OracleRolloutBuffer(T, len(cols), 1, 1, 1, 1, H, 0.99, 0.95, True, True)
    pick = lambda name: buf.device_tensor(name).view(-1, E * A, 1)[:, cols].cpu().numpy().reshape(-1, len(cols), 1, 1)
    ref.rewards[:], ref.value_preds[:], ref.masks[:], ref.bad_masks[:] = pick("rewards"), pick("value_preds"), pick("masks"), pick("bad_masks")
    ref.compute_returns(nv[cols].cpu().numpy().reshape(len(cols), 1, 1))
    assert (pick("returns") == ref.returns).all()
    buf.close()
    # 16 384 columns select the kernel's other tiling (64-column workgroups, 64-step tiles): same check on a shorter buffer
    T2, E2 = 200, 8192
    buf = pkg.DeviceReplayBuffer(_args(T2, E2, H, False, True), A, 1, 1)
    for name in ("rewards", "value_preds"):
        buf.device_tensor(name).normal_(generator=g)
    t = buf.device_tensor("masks")
    t.copy_((torch.rand(t.shape, device="cuda:0", generator=g) > 0.05).float())
    nv2 = torch.randn(E2 * A, device="cuda:0", generator=g)
    buf.compute_returns(nv2, on_device=True)
    cols = [0, 63, 64, 8191, 8192, 16383]
    ref = OracleRolloutBuffer(T2, len(cols), 1, 1, 1, 1, H, 0.99, 0.95, True, False)
    pick = lambda name: buf.device_tensor(name).view(-1, E2 * A, 1)[:, cols].cpu().numpy().reshape(-1, len(cols), 1, 1)
    ref.rewards[:], ref.value_preds[:], ref.masks[:] = pick("rewards"), pick("value_preds"), pick("masks")
    ref.compute_returns(nv2[cols].cpu().numpy().reshape(len(cols), 1, 1))
    assert (pick("returns") == ref.returns).all()
    buf.close()
