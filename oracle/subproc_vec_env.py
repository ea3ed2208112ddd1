"""ORACLE — TEST INFRASTRUCTURE ONLY.

The CPU restatement behind the reference's two VecEnv shapes, for bench.py's ``cpu_baseline`` leg and for tests:

* ``OracleBlockVecEnv``   — every env in the calling process, one thread (the reference's DummyVecEnv, envs/env_wrappers.py:48-180;
                            the authors train with n_rollout_threads = 32);
* ``OracleSubprocVecEnv`` — worker processes that each own a contiguous block of envs and talk to the parent over
                            ``multiprocessing.Pipe`` with the reference's protocol (``('step', actions)`` -> ``(obs, rewards, dones,
                            infos)``, ``'reset'``, ``'close'``; envs/env_wrappers.py:182-229,231-320), the auto-reset done inside the
                            worker (:191-204) and the parent stacking what comes back (:276-282). The reference starts one process
                            per env; here a worker steps its whole block per message, which is the generous reading for a CPU baseline
                            (fewer, larger messages).

The product never imports this module.
"""
import ctypes as C
import multiprocessing as mp

import numpy as np

from . import oracle as O


def _lib():
    L = O.lib()
    if not getattr(L, "_block_ready", False):
        L.or_block_create.restype = C.c_void_p
        L.or_block_create.argtypes = [C.POINTER(O.OrEnvConfig), C.c_int, C.c_uint64]
        L.or_block_destroy.argtypes = [C.c_void_p]
        L.or_block_obs_dim.argtypes = [C.c_void_p]
        L.or_block_act_dim.argtypes = [C.c_void_p]
        L.or_block_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.or_block_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L._block_ready = True
    return L


class OracleBlockVecEnv:
    def __init__(self, cfg, num_envs, chaff_seed=0):
        self.L = _lib()
        if cfg.hierarchical:
            O.actor_load()
        self.b = self.L.or_block_create(C.byref(cfg), int(num_envs), int(chaff_seed))
        self.num_envs, self.num_agents = int(num_envs), int(cfg.n_aircraft)
        self.obs_dim, self.act_dim = self.L.or_block_obs_dim(self.b), self.L.or_block_act_dim(self.b)

    def reset(self):
        obs = np.empty((self.num_envs, self.num_agents, self.obs_dim), dtype=np.float32)
        self.L.or_block_reset(self.b, obs.ctypes.data)
        return obs

    def step(self, actions):
        E, A = self.num_envs, self.num_agents
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(E, A, self.act_dim)
        obs = np.empty((E, A, self.obs_dim), dtype=np.float32)
        rew = np.empty((E, A, 1), dtype=np.float32)
        done = np.empty((E, A, 1), dtype=np.uint8)
        info = np.empty((E, 4), dtype=np.int32)
        self.L.or_block_step(self.b, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, info.ctypes.data)
        return obs, rew, done.astype(bool), info

    def close(self):
        if self.b:
            self.L.or_block_destroy(self.b)
            self.b = None


def _worker(remote, parent_remote, cfg_bytes, count, chaff_seed):
    """worker() of envs/env_wrappers.py:182-229 for a block of envs."""
    parent_remote.close()
    cfg = O.OrEnvConfig.from_buffer_copy(cfg_bytes)
    env = OracleBlockVecEnv(cfg, count, chaff_seed)
    try:
        while True:
            cmd, data = remote.recv()
            if cmd == "step":
                remote.send(env.step(data))
            elif cmd == "reset":
                remote.send(env.reset())
            elif cmd == "close":
                remote.close()
                break
            else:
                raise NotImplementedError(cmd)
    finally:
        env.close()


class OracleSubprocVecEnv:
    def __init__(self, cfg, num_envs, num_workers, chaff_seed=0):
        self.num_envs, self.num_agents = int(num_envs), int(cfg.n_aircraft)
        num_workers = max(1, min(int(num_workers), self.num_envs))
        base, rem = divmod(self.num_envs, num_workers)
        self.counts = [base + (1 if w < rem else 0) for w in range(num_workers)]
        self.starts = np.concatenate([[0], np.cumsum(self.counts)[:-1]]).astype(int)
        ctx = mp.get_context("fork")       # the workers only run the C oracle: no GPU state is inherited or used
        self.remotes, work_remotes = zip(*[ctx.Pipe() for _ in range(num_workers)])
        raw = bytes(cfg)
        self.ps = [ctx.Process(target=_worker, args=(wr, r, raw, n, chaff_seed + int(s)), daemon=True)
                   for wr, r, n, s in zip(work_remotes, self.remotes, self.counts, self.starts)]
        for p in self.ps:
            p.start()
        for wr in work_remotes:
            wr.close()
        self.waiting = False
        self.closed = False

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32)
        for remote, s, n in zip(self.remotes, self.starts, self.counts):
            remote.send(("step", a[s:s + n]))
        self.waiting = True

    def step_wait(self):
        results = [remote.recv() for remote in self.remotes]
        self.waiting = False
        obs, rew, done, info = zip(*results)
        return np.concatenate(obs), np.concatenate(rew), np.concatenate(done), np.concatenate(info)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def reset(self):
        for remote in self.remotes:
            remote.send(("reset", None))
        return np.concatenate([remote.recv() for remote in self.remotes])

    def close(self):
        if self.closed:
            return
        if self.waiting:
            for remote in self.remotes:
                remote.recv()
        for remote in self.remotes:
            remote.send(("close", None))
        for p in self.ps:
            p.join()
        self.closed = True
