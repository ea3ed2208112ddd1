"""HipVecEnv — the reference's VecEnv surface on top of libaircombat_hip.so.

Mirrors ``SubprocVecEnv`` / ``DummyVecEnv`` of the reference (envs/env_wrappers.py:48-320): attributes ``num_envs``,
``num_agents``, ``observation_space``, ``action_space``; methods ``reset()``, ``step(actions)``, ``step_async`` /
``step_wait``, ``close()``; the same return shapes ``obs[E,A,obs_dim]``, ``rewards[E,A,1]``, ``dones[E,A,1]`` (bool),
``infos`` (ndarray of dict), and the same auto-reset rule (env_wrappers.py:191-204). Outputs are float32
(the buffers cast to float32 on insert, algorithms/utils/buffer.py:52-58).

Ownership of what ``step`` returns. The reference hands back fresh arrays every step (``np.stack``, env_wrappers.py:276-282), so a
caller may keep ``obs`` / ``dones`` / ``infos`` of step t for as long as it likes. That is the default here too (``copy=True``), and
it costs no copy in the usual rollout loop: the library keeps a small ring of page-locked result sets (the step kernel writes its
outputs straight into one), and ``step`` picks a set whose arrays NOBODY outside holds any more -- the caller's names, slices, views,
tensors made from them all count as holding (reference counts) -- so an array the caller still has is never written again. When the
caller hoards more results than the ring has sets, ``step`` falls back to handing out fresh copies. The arrays outlive ``close()``
(a set still held then is detached from the handle and freed with its last array). ``copy=False`` (constructor argument) returns
live views of two alternating sets instead, without any check: valid until the step after next overwrites them, never after
``close()``. A non-finite aircraft state surfaces as ``RuntimeError`` (the reference's ``RuntimeError("JSBSim failed.")``,
core/simulatior.py:223-225; its ``pdb`` NaN trap in ``_pack``, env_base.py:277-281, is not reproduced).
"""
import ctypes as C
import os
import sys

import numpy as np

from .capi import (AcConfig, AC_STATE_LEN, AC_TASK_HEADING, AC_TASK_SINGLECOMBAT, AC_TASK_SHOOT_MISSILE, AC_TASK_MULTICOMBAT, AC_TASK_SCENARIO1,
                   AC_TASK_SCENARIO_NVN, load_library)
from .config import config_from_yaml, default_config

DONE_MESSAGES = {
    0: "", 1: "altitude is too low", 2: "is on an extreme state", 3: "acceleration is too high", 4: "has been shot down",
    5: "has crashed", 6: "mission completed", 7: "step limits", 8: "unreached heading",
}


class _Box:
    """Minimal stand-in for gymnasium.spaces.Box when gymnasium is not installed (attributes the runners read)."""

    def __init__(self, low, high, shape):
        self.low = np.full(shape, low, dtype=np.float32)
        self.high = np.full(shape, high, dtype=np.float32)
        self.shape = tuple(shape)
        self.dtype = np.float32

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, float32)"


class _MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.int64

    def __repr__(self):
        return f"MultiDiscrete({self.nvec.tolist()})"


class _Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64


class _Tuple(tuple):
    pass


class LazyInfos:
    """`infos` of VecEnv.step: a read-only sequence of one info dict per env (`current_step`, and on an episode end
    `done_condition` / `heading_turn_counts`), equal element by element to the object array np.stack makes of the workers' dicts
    (env_wrappers.py:276-282). The dicts are built from the kernel's int32 codes when an element is read: the reference's runners
    only iterate them in the heading task (runner/jsbsim_runner.py:55-57), and building 4096 fresh dicts per step otherwise costs
    more host time than the whole device step. np.asarray(infos) gives the reference's object array."""

    __slots__ = ("_codes",)

    def __init__(self, codes, snapshot=False):
        # [E, 4] int32 rows (current_step, done code, heading_turn_counts, env-reset flag), or the packed words [E] of ac_host_buffers
        # (AC_INFO_* in include/aircombat.h: bits 0-15, 16-23, 24-30, 31). snapshot: keep a private copy of the words (16 KB at 4096
        # envs), so that the dicts read later are still this step's
        self._codes = np.array(codes, copy=True) if snapshot else codes

    def __len__(self):
        return len(self._codes)

    shape = property(lambda self: (len(self._codes),))      # what callers read off the reference's ndarray
    ndim, dtype = 1, np.dtype(object)

    def _one(self, i):
        if self._codes.ndim == 1:
            w = int(self._codes[i]) & 0xFFFFFFFF
            step, code, turns = w & 0xFFFF, (w >> 16) & 0xFF, (w >> 24) & 0x7F
        else:
            step, code, turns = (int(v) for v in self._codes[i, :3])
        d = {"current_step": step}
        if code:
            d["done_condition"] = DONE_MESSAGES.get(code, "")
            if code == 8:   # UnreachHeading reports its curriculum stage when it ends the episode (unreach_heading.py:60-63)
                d["heading_turn_counts"] = turns
        return d

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._one(k) for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._one(i)

    def __iter__(self):
        return (self._one(i) for i in range(len(self)))

    def __array__(self, dtype=None, copy=None):
        out = np.empty(len(self), dtype=object)
        out[:] = list(self)
        return out

    def __repr__(self):
        return f"LazyInfos({len(self)} envs)"


def _spaces():
    try:
        from gymnasium import spaces  # the reference's own dependency, used when present
        return spaces.Box, spaces.MultiDiscrete, spaces.Discrete, spaces.Tuple
    except Exception:  # not installed in this image
        return (lambda low, high, shape: _Box(low, high, shape)), _MultiDiscrete, _Discrete, (lambda xs: _Tuple(xs))


CONTROLLER_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "baseline_actor.f32")


RING_SETS = 4        # result sets the default mode cycles through (the library allows AC_HOST_SETS = 8; one more is the staging set)


class _HostSetOwner:
    """One set of the library's page-locked buffers. While the handle lives the library owns the memory; a set whose arrays the caller
    still holds at close() is detached (ac_host_set_detach) and this object frees it when the last array referring to it is gone."""

    def __init__(self, lib, ptrs):
        self.lib, self.ptrs, self.detached = lib, [p.value for p in ptrs], False

    def __del__(self):
        if self.detached:
            try:
                self.lib.ac_host_set_free(*self.ptrs)
            except Exception:       # interpreter shutdown: the process's memory goes with it
                pass


class _Mapped:
    """Array-interface view of one buffer of a set; numpy keeps it (and through it the set's owner) alive as the arrays' base."""

    def __init__(self, owner, ptr, shape, typestr):
        self.owner = owner
        self.__array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 3}


class HipVecEnv:
    """E parallel 1v1 air-combat envs advanced by one HIP kernel launch per ``step``."""

    def __init__(self, config, num_envs, device_id=0, seed=0, lib=None, copy=True):
        if not isinstance(config, AcConfig):
            raise TypeError("config must be an AcConfig (see config_from_yaml / default_config)")
        self.lib = lib or load_library()
        self.config = config
        self.copy = bool(copy)
        self.num_envs = int(num_envs)
        self.num_agents = int(config.n_agents)
        handle = C.c_void_p()
        self.lib.check(self.lib.ac_create(C.byref(config), self.num_envs, int(device_id), int(seed), C.byref(handle)), "ac_create")
        self._h = handle
        self.closed = False
        self.waiting = False
        self.obs_dim = self.lib.ac_obs_dim(self._h)
        self.act_dim = self.lib.ac_act_dim(self._h)
        Box, MultiDiscrete, Discrete, Tuple = _spaces()
        self.observation_space = Box(low=-10, high=10.0, shape=(self.obs_dim,))
        self.hierarchical = bool(config.hierarchical)
        if config.task == AC_TASK_HEADING:
            # env.seed(seed + 1000 i) -> seeding.np_random (env_base.py:252-258, train_jsbsim.py:33): hand numpy's own PCG64
            # state of every env to the device, so resets and UnreachHeading draw exactly numpy's stream
            self._seed_heading(int(seed))
        if self.hierarchical:
            # BaselineActor() + load_state_dict(model/baseline_model.pt) of HierarchicalSingleCombatTask.__init__
            # (singlecombat_task.py:211-219): the exported weights go to the device once
            w = np.fromfile(CONTROLLER_WEIGHTS, dtype=np.float32)
            self.lib.check(self.lib.ac_load_controller(self._h, w.ctypes.data, int(w.size)), "ac_load_controller")
        if self.hierarchical and config.task in (AC_TASK_SCENARIO1, AC_TASK_SCENARIO_NVN):
            self.action_space = Tuple([MultiDiscrete([3, 5, 3]), MultiDiscrete([2, 2, 2, 2])])     # scenario1_task.py:29-31
        elif self.hierarchical and (config.legacy_obs or config.task == AC_TASK_SHOOT_MISSILE):
            self.action_space = Tuple([MultiDiscrete([3, 5, 3]), Discrete(2)])                     # *_with_missile_task.py:221-223
        elif self.hierarchical:
            self.action_space = MultiDiscrete([3, 5, 3])                                           # singlecombat_task.py:221-222
        elif config.task == AC_TASK_SHOOT_MISSILE or config.legacy_obs and config.task == AC_TASK_MULTICOMBAT:
            self.action_space = Tuple([MultiDiscrete([41, 41, 41, 30]), Discrete(2)])                # *_with_missile_task.py:176-178
        elif config.task in (AC_TASK_SCENARIO1, AC_TASK_SCENARIO_NVN):
            # low-level controls + [gun, AIM-9M, AIM-120B, chaff] (scenario1_task.py:29-31 with the controller net bypassed)
            self.action_space = Tuple([MultiDiscrete([41, 41, 41, 30]), MultiDiscrete([2, 2, 2, 2])])
        else:
            self.action_space = MultiDiscrete([41, 41, 41, 30])
        # Sets of page-locked host buffers owned by the library and mapped into the device (ac_host_buffers): the step kernel reads
        # the actions from, and writes observations / rewards / dones / info into, the set of the step. copy=False alternates two
        # sets, so what step() returns stays untouched while the next step runs; the default mode cycles through a ring of sets and
        # only steps into one whose arrays the caller has dropped (module docstring).
        self._sets = []
        for k in range(2):
            self._add_set()
        self._cur = 0
        self._pending = None
        self._staging = None         # default mode's fallback: a set that is never handed out (results are copied out of it)
        self._actions = self._sets[0]["actions"]
        # what step() hands back for each set, built once: the arrays are views of fixed buffers, and LazyInfos reads its codes when asked
        self._results = [self._result(st) for st in self._sets]
        self._step_host = self.lib.ac_step_host
        self._ref0 = self._refs(self._sets[0])   # what a set's arrays count when only this object holds them

    def _add_set(self):
        k = len(self._sets)
        E, A = self.num_envs, self.num_agents
        ptrs = [C.c_void_p() for _ in range(5)]
        self.lib.check(self.lib.ac_host_buffers(self._h, k, *[C.byref(p) for p in ptrs]), "ac_host_buffers")
        owner = _HostSetOwner(self.lib, ptrs)

        def arr(ptr, shape, typestr):
            return np.asarray(_Mapped(owner, ptr.value, shape, typestr))

        st = {"actions": arr(ptrs[0], (E, A, self.act_dim), "<f4"), "owner": owner, "index": k,
              # (kept in ONE tuple and nowhere else, so that their reference counts tell whether anybody outside holds them; the info
              # words count too: the LazyInfos handed out with a step reads them when it is asked)
              "out": (arr(ptrs[1], (E, A, self.obs_dim), "<f4"), arr(ptrs[2], (E, A, 1), "<f4"), arr(ptrs[3], (E, A, 1), "|b1"),   # the kernel writes 0 / 1
                      arr(ptrs[4], (E,), "<i4"))}
        self._extend_set(st)
        self._sets.append(st)
        return st

    def _extend_set(self, st):
        """Hook for subclasses that hand out one more array per set (HipShareVecEnv: the share_obs view)."""

    def _held(self, st):
        """Does anybody outside this object hold one of the set's result arrays (a name, a slice, a view, a tensor sharing its memory)?"""
        return self._refs(st) != self._ref0

    @staticmethod
    def _refs(st):
        t = st["out"]
        if len(t) == 4:
            return sys.getrefcount(t[0]), sys.getrefcount(t[1]), sys.getrefcount(t[2]), sys.getrefcount(t[3])
        return sys.getrefcount(t[0]), sys.getrefcount(t[1]), sys.getrefcount(t[2]), sys.getrefcount(t[3]), sys.getrefcount(t[4])

    def _next_set(self):
        """copy=False: the other one of two sets. Default: the next set of the ring that nobody holds, one more set while the ring may
        grow, else the staging set (its results are then copied out, like the reference's np.stack)."""
        if not self.copy:
            self._cur ^= 1
            return self._sets[self._cur], False
        n = len(self._sets)
        for i in range(1, n + 1):
            st = self._sets[(self._cur + i) % n]
            if not self._held(st):
                self._cur = st["index"]
                return st, False
        if n < RING_SETS:
            st = self._add_set()
            self._results.append(self._result(st))
            self._cur = st["index"]
            return st, False
        if self._staging is None:
            self._staging = self._add_set()
            self._sets.pop()                     # (not part of the ring)
        return self._staging, True

    # ---- reference surface
    def seed(self, seed=None):
        """env.seed(seed) of every env (env_base.py:252-258; env i gets seed + 1000 i like make_train_env): only the heading task
        draws from env.np_random, so only it has something to re-seed."""
        if seed is not None and self.config.task == AC_TASK_HEADING:
            self._seed_heading(int(seed))

    def _seed_heading(self, seed):
        st = np.zeros((self.num_envs, 4), dtype=np.uint64)
        m = (1 << 64) - 1
        for i in range(self.num_envs):
            b = np.random.PCG64(seed + 1000 * i).state["state"]
            st[i] = (b["state"] >> 64, b["state"] & m, b["inc"] >> 64, b["inc"] & m)
        self.lib.check(self.lib.ac_seed_envs(self._h, st.ctypes.data), "ac_seed_envs")

    def reset(self):
        self._assert_not_closed()
        obs = np.empty((self.num_envs, self.num_agents, self.obs_dim), dtype=np.float32)     # the caller's own, like the reference's np.stack
        self.lib.check(self.lib.ac_reset(self._h, obs.ctypes.data), "ac_reset")
        return obs

    def _hand_over(self, actions):
        """Pick the set of this step and put the actions into its mapped buffer (the kernel reads them from there)."""
        st, staged = self._next_set()
        dst = st["actions"]
        a = actions if isinstance(actions, np.ndarray) else np.asarray(actions, dtype=np.float32)
        if a.shape != dst.shape:
            a = a.reshape(dst.shape)  # nested lists [E][A][act_dim] from the runners
        np.copyto(dst, a)
        return st, staged

    def step_async(self, actions):
        """SubprocVecEnv.step_async (env_wrappers.py:269-273): hand the actions over and start the step."""
        self._assert_not_closed()
        self._pending = self._hand_over(actions)
        self.lib.check(self.lib.ac_step_host_async(self._h, self._pending[0]["index"]), "ac_step_host_async")
        self.waiting = True

    def step_wait(self):
        """SubprocVecEnv.step_wait (env_wrappers.py:275-282): float32 arrays the caller owns (the buffers cast on insert,
        algorithms/utils/buffer.py:52-58), or with copy=False views of the step's buffer set, valid until the step after next."""
        self._assert_not_closed()
        self.lib.check(self.lib.ac_step_host_wait(self._h), "ac_step_host_wait")
        self.waiting = False
        st, staged = self._pending
        return self._returned(st, staged)

    def _returned(self, st, staged):
        if not self.copy:
            return self._results[st["index"]]
        return self._fresh(st) if staged else self._owned(st)

    def _result(self, st):
        o = st["out"]
        return o[0], o[1], o[2], LazyInfos(o[3])

    def _owned(self, st):
        """The set's arrays themselves: nobody else held them when the step began, and they are not written again while the caller does
        (the info words included: the LazyInfos keeps them, and with them the set, for as long as the caller keeps it)."""
        o = st["out"]
        return o[0], o[1], o[2], LazyInfos(o[3])

    def _fresh(self, st):
        o = st["out"]
        return o[0].copy(), o[1].copy(), o[2].copy(), LazyInfos(o[3], snapshot=True)

    def step(self, actions):
        """VecEnv.step = step_async + step_wait (env_wrappers.py:30-42), through one library call."""
        assert not self.closed, "Trying to operate on a HipVecEnv after calling close()"
        st, staged = self._hand_over(actions)
        if self._step_host(self._h, st["index"]) != 0:
            self.lib.check(-1, "ac_step_host")      # RuntimeError: a HIP failure, or a non-finite aircraft state ("JSBSim failed.")
        return self._returned(st, staged)

    def render(self, mode="txt", filepath="./JSBSimRecording.txt.acmi", env=0):
        """BaseEnv.render (env_base.py:207-250) for one env (the reference renders through DummyVecEnv, i.e. env 0): appends one
        Tacview ACMI frame to `filepath` -- every aircraft (env._jsbsims), every munition that has been launched (env._tempsims: a
        position record while it flies, removal + explosion once, then removal) and every chaff cloud (env._chaffsims), in the
        reference's order and text. A chaff record carries the pose its parent had at the release: the release happens at the end
        of an env step, so the frame rendered after that step still sees that pose (call render() after every step, as the
        reference's render loop does)."""
        from . import acmi
        if mode != "txt":
            raise NotImplementedError
        self._assert_not_closed()
        cfg = self.config
        step = int(self.get_state(env, 0)[self._ix("cur_step")])
        if not getattr(self, "_acmi_started", False):
            with open(filepath, mode="w", encoding="utf-8-sig") as f:
                f.write(acmi.HEADER)
            self._acmi_started, self._acmi_exploded, self._acmi_chaff, self._acmi_first, self._acmi_step = True, set(), {}, {}, -1
        if step <= self._acmi_step:      # the episode was reset: env.reset() clears _tempsims / _chaffsims (env_base.py:98-113)
            self._acmi_exploded, self._acmi_chaff, self._acmi_first = set(), {}, {}
        self._acmi_step = step
        center = (cfg.center_lon, cfg.center_lat, cfg.center_alt)
        uids = getattr(cfg, "uids", None) or [f"{'A' if a < cfg.n_ego else 'B'}0{(a if a < cfg.n_ego else a - cfg.n_ego) + 1}00" for a in range(self.num_agents)]
        color = lambda a: "Blue" if a < cfg.n_ego else "Red"
        msgs = []
        entities = [self.get_entity(env, a) for a in range(self.num_agents)]
        for a in range(self.num_agents):
            msgs.append(acmi.aircraft_record(uids[a], color(a), entities[a]))
        slots = {AC_TASK_SHOOT_MISSILE: 4, 2: 4, AC_TASK_SCENARIO1: 2, AC_TASK_SCENARIO_NVN: 2}.get(cfg.task, 0)
        base_missile = cfg.task in (AC_TASK_SHOOT_MISSILE, 2)       # MissileSimulator itself (300 m fuse); 2 = AC_TASK_DODGE_MISSILE
        if cfg.task == 2 and cfg.n_agents > 2:                      # multiplecombat_dodge_missile: two uids per aircraft, like the scenario tasks
            slots = 2
        # env._tempsims in dict order = first-launch order of the uids; a slot's uid is "agent + remaining count at the launch"
        # (scenario1_task.py:83,92; singlecombat_with_missile_task.py:199): slots are consumed from the highest count down
        flying = []
        for a in range(self.num_agents):
            nmis = min(int(cfg.num_missiles[a]), slots) if slots == 4 else slots
            for k in range(nmis):
                m = self.get_missile(env, a, k)
                if m[0] < 0:
                    self._acmi_exploded.discard((a, k))
                    continue
                flying.append((0.0, a, k, m, nmis))
        for _, a, k, _m, _n in flying:
            self._acmi_first.setdefault((a, k), (step, a))      # dict position of the uid: its first launch (step, agent order)
        flying.sort(key=lambda r: self._acmi_first[(r[1], r[2])])
        for _, a, k, m, nmis in flying:
            uid = f"{uids[a]}{nmis - k}"
            rec, boom = acmi.missile_records(uid, color(a), int(m[0]), m[1:4], m[7], m[8], center, (a, k) in self._acmi_exploded,
                                             300 if base_missile else 5, acmi.MISSILE_MODELS[int(m[11])])
            if boom:
                self._acmi_exploded.add((a, k))
            msgs.append(rec)
        # env._chaffsims: one ChaffSimulator per qualifying incoming missile of a release event, uid "agent + (remaining + 10)"
        if cfg.task in (AC_TASK_SCENARIO1, AC_TASK_SCENARIO_NVN):
            for a in range(self.num_agents):
                st = self.get_state(env, a)
                n_ch = int(st[self._ix("x_n_ch")])
                rem = int(cfg.num_missiles[a])
                for q in range(min(n_ch, 2)):
                    mult = int(st[self._ix(f"x_ch_mult{q}")])
                    alive = int(st[self._ix(f"x_ch_status{q}")]) == 0
                    for j in range(mult):
                        uid = f"{uids[a]}{rem + 10}"
                        rem -= 1
                        if uid not in self._acmi_chaff:
                            self._acmi_chaff[uid] = {"color": color(a), "pose": tuple(entities[a][:6])}
                        self._acmi_chaff[uid]["alive"] = alive
            for uid, ch in self._acmi_chaff.items():
                msgs.append(acmi.chaff_record(uid, ch["color"], ch["alive"], ch["pose"]))
        with open(filepath, mode="a", encoding="utf-8-sig") as f:
            f.write(f"#{step * cfg.agent_interaction_steps / cfg.sim_freq:.2f}\n")
            for msg in msgs:
                f.write(msg + "\n")

    def _ix(self, name):
        if not hasattr(self, "_names"):
            self._names = self.lib.state_field_names()
        return self._names.index(name)

    def close(self):
        """Frees the handle and its page-locked buffers. With copy=False the views handed out earlier point into those buffers: they
        must not be read after close(). In the default mode the arrays a caller still holds stay valid: their set is detached from the
        handle first and freed when the last of them is gone."""
        if self.closed:
            return
        if self.copy:
            for st in self._sets:
                if self._held(st):
                    self.lib.check(self.lib.ac_host_set_detach(self._h, st["index"]), "ac_host_set_detach")
                    st["owner"].detached = True
        self._sets, self._results, self._staging, self._pending = [], [], None, None    # views of library-owned memory: dropped before the handle frees it
        self._actions = None
        self.lib.ac_destroy(self._h)
        self._h = None
        self.closed = True

    def _assert_not_closed(self):
        assert not self.closed, "Trying to operate on a HipVecEnv after calling close()"

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device-resident path: obs / actions as torch tensors on the env's GPU
    def device_tensors(self):
        """torch views (no copy) of the handle's device buffers: actions, obs, rewards, dones, info."""
        import torch

        ptrs = [C.c_void_p() for _ in range(5)]
        self.lib.check(self.lib.ac_device_buffers(self._h, *[C.byref(p) for p in ptrs]), "ac_device_buffers")
        E, A = self.num_envs, self.num_agents

        def view(ptr, shape, typestr):
            holder = type("_DeviceBuffer", (), {})()
            holder.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr.value, False), "version": 2}
            return torch.as_tensor(holder, device="cuda")

        act = view(ptrs[0], (E, A, self.act_dim), "<f4")
        obs = view(ptrs[1], (E, A, self.obs_dim), "<f4")
        rew = view(ptrs[2], (E, A, 1), "<f4")
        done = view(ptrs[3], (E, A, 1), "|u1")
        info = view(ptrs[4], (E, 4), "<i4")
        return act, obs, rew, done, info

    def step_device(self, d_actions_ptr=None, stream=None):
        """Asynchronous step on device-resident actions; results stay in the device buffers.

        The handle's stream is non-blocking, i.e. NOT ordered against torch's streams. ``stream`` (a ``torch.cuda.Stream``, a raw
        ``hipStream_t`` value, or 0 for the default stream) names the stream that produced the actions and will consume the
        outputs: the step then waits for the work queued on it so far, and it waits for the step, without any host sync. With
        ``stream=None`` no ordering is added: the caller synchronises itself (``env.sync()`` / ``torch.cuda.synchronize()``)."""
        if stream is not None:
            raw = getattr(stream, "cuda_stream", stream)
            self.lib.check(self.lib.ac_order_after(self._h, raw), "ac_order_after")
        self.lib.check(self.lib.ac_step_async_device(self._h, d_actions_ptr), "ac_step_async_device")
        if stream is not None:
            self.lib.check(self.lib.ac_order_before(self._h, raw), "ac_order_before")

    def sync(self):
        self.lib.check(self.lib.ac_sync(self._h), "ac_sync")

    # ---- test / render access (mirrors env.agents[uid] reads and env.agents[uid].crash())
    def get_state(self, env, agent):
        out = (C.c_double * AC_STATE_LEN)()
        self.lib.check(self.lib.ac_get_state(self._h, env, agent, out), "ac_get_state")
        return np.array(out[:], dtype=np.float64)

    def set_state(self, env, agent, vec):
        buf = (C.c_double * AC_STATE_LEN)(*[float(v) for v in vec])
        self.lib.check(self.lib.ac_set_state(self._h, env, agent, buf), "ac_set_state")

    def set_status(self, env, agent, status):
        self.lib.check(self.lib.ac_set_status(self._h, env, agent, status), "ac_set_status")

    def get_entity(self, env, agent):
        out = (C.c_double * 12)()
        self.lib.check(self.lib.ac_get_entity(self._h, env, agent, out), "ac_get_entity")
        return np.array(out[:], dtype=np.float64)

    def get_heading_state(self, env):
        """HeadingTask bookkeeping of one env: sim_time, target heading / altitude / speed, next check time, turn count, last p, q."""
        out = (C.c_double * 8)()
        self.lib.check(self.lib.ac_get_heading_state(self._h, env, out), "ac_get_heading_state")
        return np.array(out[:])

    def get_controller_state(self, env, agent):
        """(hidden[128], low_action[act_low]) of the low-level controller for one aircraft (hierarchical tasks)."""
        hid = np.zeros(128, dtype=np.float32)
        low = np.zeros(8, dtype=np.float32)
        self.lib.check(self.lib.ac_get_controller_state(self._h, env, agent, hid.ctypes.data, low.ctypes.data), "ac_get_controller_state")
        return hid, low

    def set_controller_state(self, env, agent, hidden):
        hid = np.ascontiguousarray(hidden, dtype=np.float32)
        assert hid.size == 128
        self.lib.check(self.lib.ac_set_controller_state(self._h, env, agent, hid.ctypes.data), "ac_set_controller_state")

    def state_checksum(self):
        """64-bit order-independent digest of every aircraft's state (ac_state_checksum)."""
        out = C.c_uint64()
        self.lib.check(self.lib.ac_state_checksum(self._h, C.byref(out)), "ac_state_checksum")
        return int(out.value)

    def munitions_in_flight(self):
        """Munitions with status LAUNCHED over all envs (ac_munitions_in_flight)."""
        out = C.c_int32()
        self.lib.check(self.lib.ac_munitions_in_flight(self._h, C.byref(out)), "ac_munitions_in_flight")
        return int(out.value)

    def get_missile(self, env, agent, k):
        out = (C.c_double * 12)()
        self.lib.check(self.lib.ac_get_missile(self._h, env, agent, k, out), "ac_get_missile")
        return np.array(out[:], dtype=np.float64)


class HipShareVecEnv(HipVecEnv):
    """The Share* VecEnv family (envs/env_wrappers.py:323-462) for MultipleCombat: ``reset()`` -> ``(obs, share_obs)``,
    ``step()`` -> ``(obs, share_obs, rewards, dones, infos)``. ``share_obs[e, a]`` is the concatenation of all agents'
    observations of env ``e`` (BaseEnv.get_state, env_base.py:183-189), i.e. ``obs`` flattened per env; it is returned as a
    read-only broadcast view (the buffers copy on insert)."""

    def __init__(self, config, num_envs, device_id=0, seed=0, lib=None, copy=True):
        super().__init__(config, num_envs, device_id=device_id, seed=seed, lib=lib, copy=copy)
        Box = _spaces()[0]
        self.share_observation_space = Box(low=-10, high=10.0, shape=(self.num_agents * self.obs_dim,))

    def _share(self, obs):
        E, A, D = obs.shape
        return np.broadcast_to(obs.reshape(E, 1, A * D), (E, A, A * D))

    def reset(self):
        obs = super().reset()
        return obs, self._share(obs)

    def _extend_set(self, st):
        # the broadcast view of the set's observations, built once (np.broadcast_to + reshape cost ~4 us per step) and counted like the
        # set's other arrays: a caller holding only share_obs holds the set
        st["out"] = st["out"] + (self._share(st["out"][0]),)

    def _result(self, st):
        o = st["out"]
        return o[0], o[4], o[1], o[2], LazyInfos(o[3])

    def _owned(self, st):
        o = st["out"]
        return o[0], o[4], o[1], o[2], LazyInfos(o[3])

    def _fresh(self, st):
        o = st["out"]
        obs = o[0].copy()
        return obs, self._share(obs), o[1].copy(), o[2].copy(), LazyInfos(o[3], snapshot=True)


class MultiDeviceVecEnv:
    """One VecEnv over several GPUs from a single process: a contiguous block of envs per device
    (sharding.env_block), one handle, stream and pinned host slab per device, no collective; the caller sees one
    ``[E, A, ...]`` array. ``step`` runs the per-device ``ac_step`` calls concurrently (one thread per handle; ctypes releases the
    GIL for the duration of the call). Env ``i`` keeps the seed ``seed + 1000 i`` of a single-handle VecEnv (heading task).
    The multi-process form (one process per GPU, bench.py --gpus N) needs none of this."""

    def __init__(self, config, num_envs, device_ids, seed=0, share=None):
        from concurrent.futures import ThreadPoolExecutor
        from .sharding import env_block
        if not device_ids:
            raise ValueError("device_ids must name at least one GPU")
        share = (config.n_agents > 2) if share is None else share      # the MultipleCombatEnv family (share_obs)
        cls = HipShareVecEnv if share else HipVecEnv
        self.blocks = [env_block(r, len(device_ids), num_envs) for r in range(len(device_ids))]
        if min(c for _, c in self.blocks) < 1:
            raise ValueError("fewer envs than devices")
        # (the parts hand out views; _cat concatenates them into fresh arrays, so the caller of this class always owns what it gets)
        self.parts = [cls(config, count, device_id=dev, seed=seed + 1000 * start, copy=False) for dev, (start, count) in zip(device_ids, self.blocks)]
        self.share = share
        self.num_envs, self.num_agents = int(num_envs), self.parts[0].num_agents
        self.obs_dim, self.act_dim = self.parts[0].obs_dim, self.parts[0].act_dim
        self.observation_space, self.action_space = self.parts[0].observation_space, self.parts[0].action_space
        if share:
            self.share_observation_space = self.parts[0].share_observation_space
        self._pool = ThreadPoolExecutor(max_workers=len(self.parts))
        self._pending = None

    def _share(self, obs):
        E, A, D = obs.shape
        return np.broadcast_to(obs.reshape(E, 1, A * D), (E, A, A * D))

    def _cat(self, results):
        """One set of arrays for the caller: every column of the parts' results concatenated along the env axis, except share_obs,
        which is rebuilt as the broadcast view of the concatenated observations (the parts' share_obs are broadcast views themselves:
        concatenating them would materialise A copies of every row -- 66 MB per device and step at 4096 envs x 8 aircraft x 504)."""
        cols = list(zip(*results))
        obs = np.concatenate(cols[0], axis=0)
        rest = [np.concatenate(c, axis=0) for c in cols[(2 if self.share else 1):-1]]
        infos = LazyInfos(np.concatenate([i._codes for i in cols[-1]], axis=0))
        return ((obs, self._share(obs)) if self.share else (obs,)) + tuple(rest) + (infos,)

    def reset(self):
        obs = np.concatenate([(p.reset()[0] if self.share else p.reset()) for p in self.parts], axis=0)
        return (obs, self._share(obs)) if self.share else obs

    def render(self, mode="txt", filepath="./JSBSimRecording.txt.acmi", env=0):
        """BaseEnv.render for env `env` of the whole batch: the part that owns that env writes the frame (env_wrappers.py:167-169
        renders through DummyVecEnv, i.e. env 0 -- always the first part's)."""
        for p, (s, c) in zip(self.parts, self.blocks):
            if s <= env < s + c:
                return p.render(mode=mode, filepath=filepath, env=env - s)
        raise IndexError(f"env {env} of {self.num_envs}")

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.num_agents, self.act_dim)
        self._pending = [self._pool.submit(p.step, a[s:s + c]) for p, (s, c) in zip(self.parts, self.blocks)]

    def step_wait(self):
        res, self._pending = [f.result() for f in self._pending], None
        return self._cat(res)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        if seed is not None:
            for p, (s, _) in zip(self.parts, self.blocks):
                p.seed(seed + 1000 * s)

    def close(self):
        for p in self.parts:
            p.close()
        self._pool.shutdown(wait=True)


def make_env(scenario=None, num_envs=1, task=None, device_id=0, seed=0, copy=True):
    """``scenario``: path of a scenario YAML (reference format) or None for the 1v1 block of WVR_selfplay.yaml."""
    cfg = config_from_yaml(scenario, task=task) if scenario else default_config(task or "singlecombat")
    cls = HipShareVecEnv if cfg.n_agents > 2 else HipVecEnv         # the MultipleCombatEnv family (share_obs)
    return cls(cfg, num_envs, device_id=device_id, seed=seed, copy=copy)
