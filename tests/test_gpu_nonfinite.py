"""A non-finite aircraft state must surface as an error, never fly on silently. The reference traps NaN in
BaseEnv._pack with a pdb prompt (env_base.py:277-281) and raises RuntimeError("JSBSim failed.") when the FDM gives up
(simulatior.py:223-225); its ExtremeState condition (catalog.py:386-416) is all `>=` comparisons, which a NaN fails. Here a NaN / Inf in
any integrator state makes the aircraft an ExtremeState termination AND fails the step with RuntimeError naming the env and agent;
reset() clears the condition."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def actions(env, rng):
    E, A = env.num_envs, env.num_agents
    a = np.stack([rng.integers(0, n, size=(E, A)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32)
    if env.act_dim > 4:
        a = np.concatenate([a, np.zeros((E, A, env.act_dim - 4), dtype=np.float32)], axis=-1)
    return a


@pytest.mark.parametrize("task,field,value", [("singlecombat", "wq", np.nan), ("singlecombat", "vx", np.inf), ("singlecombat", "rx", np.nan),
                                              ("singlecombat", "tank0", np.nan), ("singlecombat", "q2", np.nan), ("heading", "wp", np.nan),
                                              ("multiplecombat", "vz", np.nan), ("singlecombat_shoot", "wr", np.nan), ("scenario1", "vy", np.nan),
                                              ("scenario_nvn", "wq", np.nan), ("wvr_lowlevel", "q0", np.nan), ("multiplecombat_dodge_missile", "vx", np.nan),
                                              ("multiplecombat_shoot", "wq", np.nan)])
def test_non_finite_state_fails_the_step_and_terminates_the_aircraft(pkg, task, field, value):
    _poisoned_aircraft_is_named(pkg, task, field, value, None)


@pytest.mark.parametrize("task,field,bad_agent", [("singlecombat", "wq", 0), ("singlecombat", "rx", 0), ("multiplecombat", "vz", 1), ("scenario_nvn", "wq", 0),
                                                  ("scenario1", "vy", 0), ("singlecombat_shoot", "wr", 0)])
def test_the_error_names_the_faulty_aircraft_not_an_opponent_whose_reward_inherited_the_nan(pkg, task, field, bad_agent):
    """A NaN pose makes the OTHER aircraft's posture reward NaN too (orientation of a NaN angle), so several lanes of the env report in
    the same step; whichever store lands last must not decide. The error word is merged with a system-scope max in which an aircraft
    whose own state probe fired outranks one that only inherited the NaN: the message names the poisoned aircraft also when it is the
    LOWEST lane of its env."""
    _poisoned_aircraft_is_named(pkg, task, field, np.nan, bad_agent)


def _poisoned_aircraft_is_named(pkg, task, field, value, which):
    cfg = pkg.default_config(task)
    A = cfg.n_agents
    E = 70                       # ragged last workgroup
    cls = pkg.HipShareVecEnv if A > 2 else pkg.HipVecEnv
    env = cls(cfg, E, seed=3, copy=False)
    env.reset()
    rng = np.random.default_rng(1)
    for _ in range(3):
        env.step(actions(env, rng))                     # a healthy batch steps without complaint
    bad_env, bad_agent = 37, (A - 1 if which is None else which)
    ix = env.lib.state_field_names().index(field)
    st = env.get_state(bad_env, bad_agent)
    st[ix] = value
    env.set_state(bad_env, bad_agent, st)
    with pytest.raises(RuntimeError) as err:
        for _ in range(3):                              # (a poisoned tank reaches the load factors one tick later than a poisoned rate)
            env.step(actions(env, rng))
    assert "JSBSim failed" in str(err.value) and f"env {bad_env}, agent {bad_agent}" in str(err.value), str(err.value)
    # the step itself completed: the aircraft was terminated like an ExtremeState, every other env's outputs are sound
    res = env._results[env._cur]
    obs, rew, done = res[0], res[-3], res[-2]
    assert done[bad_env, bad_agent, 0]
    others = np.ones(E, dtype=bool); others[bad_env] = False
    assert np.isfinite(obs[others]).all() and np.isfinite(rew[others]).all()
    # sticky until reset() ...
    with pytest.raises(RuntimeError):
        env.step(actions(env, rng))
    # ... which clears it
    assert np.isfinite(env.reset() if A <= 2 else env.reset()[0]).all()
    for _ in range(3):
        out = env.step(actions(env, rng))
        assert np.isfinite(out[0]).all()
    env.close()


def test_default_step_returns_arrays_the_caller_owns(pkg):
    """The reference's VecEnv returns fresh arrays every step (np.stack, env_wrappers.py:276-282): with the default copy=True what a
    caller keeps from step t is untouched by later steps, reset() and close(); copy=False hands out the live buffer views."""
    cfg = pkg.default_config("heading")
    env = pkg.HipVecEnv(cfg, 8, seed=5)
    env.reset()
    rng = np.random.default_rng(2)
    kept = []
    for k in range(6):
        obs, rew, done, infos = env.step(actions(env, rng))
        kept.append((obs, obs.copy(), rew, rew.copy(), done, done.copy(), infos, [dict(d) for d in infos]))
    for obs, obs0, rew, rew0, done, done0, infos, infos0 in kept:
        assert (obs == obs0).all() and (rew == rew0).all() and (done == done0).all() and list(infos) == infos0
    assert [i[0]["current_step"] for i in [k[6] for k in kept]] == [1, 2, 3, 4, 5, 6]
    env.reset()
    env.close()
    assert (kept[-1][0] == kept[-1][1]).all()           # still readable after close(): the caller's own memory
    view = pkg.HipVecEnv(cfg, 8, seed=5, copy=False)
    view.reset()
    a = view.step(actions(view, rng))[0]
    view.step(actions(view, rng))
    c = view.step(actions(view, rng))[0]
    assert a is c                                        # two alternating buffer sets
    view.close()


def test_default_step_hands_out_owned_arrays_without_copying(pkg):
    """The default mode's ring of page-locked result sets: in a rollout loop (step t's arrays dropped before step t + 2 begins) the
    arrays step() returns ARE the buffers the kernel wrote -- a handful of addresses, no copy; anything the caller still holds in any
    form (a slice, a reshaped view, a torch tensor sharing the memory) keeps its set out of the ring, and when more results are
    alive than the ring has sets step() falls back to fresh copies; a set still held at close() outlives the handle."""
    import gc
    cfg = pkg.default_config("singlecombat")
    env = pkg.HipVecEnv(cfg, 64, seed=5)
    env.reset()
    rng = np.random.default_rng(2)
    addr = lambda a: a.__array_interface__["data"][0]
    seen = set()
    res = None
    for k in range(12):                                  # rollout pattern: the previous result is alive while the next step runs
        res = env.step(actions(env, rng))
        seen.add(addr(res[0]))
        assert not res[0].flags.owndata                  # a view of a page-locked set, not a copy
    assert len(seen) == 2, seen                          # two sets take turns
    piece, flat = res[0][3:5, 1], res[1].reshape(-1)     # all the caller keeps of step 12: a slice of obs, a reshaped view of the rewards
    piece0, flat0, held = piece.copy(), flat.copy(), addr(res[0])
    del res
    for k in range(10):
        out = env.step(actions(env, rng))
        assert addr(out[0]) != held                      # that set is never stepped into while any view of it lives
        del out
    assert (piece == piece0).all() and (flat == flat0).all()
    del piece, flat
    gc.collect()
    later = {addr(env.step(actions(env, rng))[0]) for k in range(8)}
    assert held in later                                 # dropped: back in the ring
    hoard = [env.step(actions(env, rng)) for k in range(10)]          # ten live results, four sets: the later ones are copies
    assert len({addr(h[0]) for h in hoard}) == 10
    assert sum(h[0].flags.owndata for h in hoard) == 10 - 4
    snap = [(h[0].copy(), h[1].copy(), h[2].copy(), [dict(d) for d in h[3]]) for h in hoard]
    for k in range(6):
        env.step(actions(env, rng))
    for h, c in zip(hoard, snap):
        assert (h[0] == c[0]).all() and (h[1] == c[1]).all() and (h[2] == c[2]).all() and [dict(d) for d in h[3]] == c[3]
    env.close()                                          # four sets are still held: detached, not freed
    for h, c in zip(hoard, snap):
        assert (h[0] == c[0]).all()
    del hoard, h
    gc.collect()                                         # ... and freed with their last array (no crash, nothing to assert)
    share = pkg.HipShareVecEnv(pkg.default_nvn_config(2), 16, seed=1)
    share.reset()
    a = np.tile(np.array([20, 19, 20, 0], dtype=np.float32), (16, 4, 1))
    obs, sh, rew, done, infos = share.step(a)
    first = addr(obs)
    del obs, rew, done, infos                            # only the broadcast share_obs view is kept: it is a view of the obs buffer
    share.step(a)
    third = share.step(a)
    assert addr(third[0]) != first and np.shares_memory(sh, sh) and not sh.flags.owndata
    del sh, third
    share.close()
