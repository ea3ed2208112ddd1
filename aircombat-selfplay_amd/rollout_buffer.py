"""Device-resident rollout buffers behind the reference's buffer interface.

``DeviceReplayBuffer`` / ``DeviceSharedReplayBuffer`` mirror ``ReplayBuffer`` / ``SharedReplayBuffer`` of the reference
(algorithms/utils/buffer.py:26-268, :270-448): same constructor arguments, same method names, same argument meaning and the same
array layouts, but every array lives in HBM behind ``include/aircombat_buffer.h`` (``libaircombat_hip.so``). ``insert`` takes
numpy arrays (the reference's call, runner/jsbsim_runner.py:133) or device pointers / torch tensors on the buffer's GPU
(the env handle's own output buffers: no host round trip); ``compute_returns`` runs the reference's float32 recurrence as a HIP
kernel and is bit-identical to the numpy code; ``recurrent_generator`` gathers each mini-batch on the device.

There is no CPU implementation here: without the HIP library the constructor raises ``HipExtensionMissing``.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (AcBufferBatch, AcBufferConfig, AcBufferStep, AC_BUF_ACTIONS, AC_BUF_ACTIVE_MASKS, AC_BUF_ADVANTAGES, AC_BUF_BAD_MASKS,
                   AC_BUF_LOGP, AC_BUF_MASKS, AC_BUF_OBS, AC_BUF_RETURNS, AC_BUF_REWARDS, AC_BUF_RNN_ACTOR, AC_BUF_RNN_CRITIC,
                   AC_BUF_SHARE_OBS, AC_BUF_VALUES)


def shape_from_space(space):
    """get_shape_from_space (algorithms/utils/utils.py:15-34) for gymnasium spaces or the stand-ins of vec_env.py; a plain int
    is taken as the flat dimension."""
    if isinstance(space, (int, np.integer)):
        return (int(space),)
    if isinstance(space, tuple) and len(space) == 2:          # Tuple(MultiDiscrete, Discrete | MultiDiscrete)
        second = space[1]
        return (len(space[0].nvec) + (len(second.nvec) if hasattr(second, "nvec") else 1),)
    if hasattr(space, "nvec") or (hasattr(space, "shape") and tuple(space.shape)):
        return tuple(space.shape)
    if hasattr(space, "n"):                                   # Discrete
        return (1,)
    raise NotImplementedError(f"Unsupported space type: {type(space)}!")


def _host(x):
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


class DeviceReplayBuffer:
    """ReplayBuffer(args, num_agents, obs_space, act_space) (buffer.py:36-69). ``args`` needs buffer_size, n_rollout_threads,
    gamma, use_proper_time_limits, use_gae, gae_lambda, recurrent_hidden_size, recurrent_hidden_layers."""

    _shared = False
    FIELDS = {"obs": AC_BUF_OBS, "share_obs": AC_BUF_SHARE_OBS, "actions": AC_BUF_ACTIONS, "rewards": AC_BUF_REWARDS, "masks": AC_BUF_MASKS,
              "bad_masks": AC_BUF_BAD_MASKS, "active_masks": AC_BUF_ACTIVE_MASKS, "action_log_probs": AC_BUF_LOGP, "value_preds": AC_BUF_VALUES,
              "returns": AC_BUF_RETURNS, "rnn_states_actor": AC_BUF_RNN_ACTOR, "rnn_states_critic": AC_BUF_RNN_CRITIC}

    def __init__(self, args, num_agents, obs_space, act_space, share_obs_space=None, device_id=0):
        self.lib = capi.load_library()
        self.buffer_size, self.n_rollout_threads, self.num_agents = int(args.buffer_size), int(args.n_rollout_threads), int(num_agents)
        self.gamma, self.gae_lambda = float(args.gamma), float(args.gae_lambda)
        self.use_gae, self.use_proper_time_limits = bool(args.use_gae), bool(args.use_proper_time_limits)
        self.recurrent_hidden_size, self.recurrent_hidden_layers = int(args.recurrent_hidden_size), int(args.recurrent_hidden_layers)
        self.obs_shape, self.act_shape = shape_from_space(obs_space), shape_from_space(act_space)
        self.share_obs_shape = shape_from_space(share_obs_space) if self._shared else None
        cfg = AcBufferConfig(self.buffer_size, self.n_rollout_threads, self.num_agents, int(np.prod(self.obs_shape)),
                             int(np.prod(self.share_obs_shape)) if self._shared else 0, int(np.prod(self.act_shape)),
                             int(np.prod(self.act_shape)) if self._shared else 1, self.recurrent_hidden_layers, self.recurrent_hidden_size,
                             int(self.use_gae), int(self.use_proper_time_limits), self.gamma, self.gae_lambda)
        self._h = self.lib.ac_buffer_create(C.byref(cfg), int(device_id))
        if not self._h:
            raise RuntimeError(f"ac_buffer_create failed: {self.lib.last_error()}")
        self.device_id = int(device_id)
        T, E, A = self.buffer_size, self.n_rollout_threads, self.num_agents
        hid = (self.recurrent_hidden_layers, self.recurrent_hidden_size)
        self._shapes = {"obs": (T + 1, E, A) + self.obs_shape, "actions": (T, E, A) + self.act_shape, "rewards": (T, E, A, 1),
                        "masks": (T + 1, E, A, 1), "bad_masks": (T + 1, E, A, 1),
                        "action_log_probs": (T, E, A) + (self.act_shape if self._shared else (1,)),
                        "value_preds": (T + 1, E, A, 1), "returns": (T + 1, E, A, 1),
                        "rnn_states_actor": (T + 1, E, A) + hid, "rnn_states_critic": (T + 1, E, A) + hid}
        if self._shared:
            self._shapes.update(share_obs=(T + 1, E, A) + self.share_obs_shape, active_masks=(T + 1, E, A, 1))

    # ---- plumbing
    def close(self):
        if getattr(self, "_h", None):
            self.lib.ac_buffer_destroy(self._h)
            self._h = None

    __del__ = close

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.last_error()}")

    @property
    def step(self):
        return self.lib.ac_buffer_step_index(self._h)

    def array(self, name):
        """Host copy of a whole array in the reference's shape (tests, logging); the training path never needs it."""
        fld = AC_BUF_ADVANTAGES if name == "advantages" else self.FIELDS[name]
        shape = (self.buffer_size, self.n_rollout_threads, self.num_agents, 1) if name == "advantages" else self._shapes[name]
        out = np.empty(shape, dtype=np.float32)
        self._ok(self.lib.ac_buffer_read(self._h, fld, out.ctypes.data), "ac_buffer_read")
        return out

    def set_slot(self, name, t, value):
        """buffer.<name>[t] = value (what the runners do once after reset: buffer.obs[0] = obs, runner/jsbsim_runner.py:36-42)."""
        v = _host(value)
        assert v.shape == self._shapes[name][1:], (v.shape, self._shapes[name][1:])
        self._ok(self.lib.ac_buffer_write_slot(self._h, self.FIELDS[name], int(t) % self._shapes[name][0], v.ctypes.data), "ac_buffer_write_slot")

    def device_tensor(self, name):
        """torch view of an array in HBM (no copy); requires torch with the ROCm runtime of the same process."""
        import torch
        fld = AC_BUF_ADVANTAGES if name == "advantages" else self.FIELDS[name]
        ptr, n = C.c_void_p(), C.c_int64()
        self._ok(self.lib.ac_buffer_device_ptr(self._h, fld, C.byref(ptr), C.byref(n)), "ac_buffer_device_ptr")
        shape = (self.buffer_size, self.n_rollout_threads, self.num_agents, 1) if name == "advantages" else self._shapes[name]
        iface = {"shape": (n.value,), "typestr": "<f4", "data": (ptr.value, False), "version": 2}
        holder = type("_View", (), {"__cuda_array_interface__": iface})()
        return torch.as_tensor(holder, device=f"cuda:{self.device_id}").view(shape)

    # ---- the reference's methods
    def insert(self, obs, actions, rewards, masks, action_log_probs, value_preds, rnn_states_actor, rnn_states_critic,
               bad_masks=None, share_obs=None, active_masks=None, available_actions=None, on_device=False, **kwargs):
        """buffer.py:77-111 / :313-343. With on_device=True every argument is a device pointer (int) or a torch tensor on the
        buffer's GPU holding contiguous float32."""
        keep = []

        def ptr(x):
            if x is None:
                return None
            if on_device:
                return int(x) if isinstance(x, (int, np.integer)) else int(x.data_ptr())
            a = _host(x)
            keep.append(a)
            return a.ctypes.data
        step = AcBufferStep(ptr(obs), ptr(actions), ptr(rewards), ptr(masks), ptr(action_log_probs), ptr(value_preds), ptr(rnn_states_actor),
                            ptr(rnn_states_critic), ptr(bad_masks), ptr(share_obs), ptr(active_masks))
        self._ok(self.lib.ac_buffer_insert(self._h, C.byref(step), int(on_device)), "ac_buffer_insert")

    def after_update(self):
        self._ok(self.lib.ac_buffer_after_update(self._h), "ac_buffer_after_update")

    def clear(self):
        self._ok(self.lib.ac_buffer_clear(self._h), "ac_buffer_clear")

    def compute_returns(self, next_value, on_device=False):
        if on_device:
            p = int(next_value) if isinstance(next_value, (int, np.integer)) else int(next_value.data_ptr())
            self._ok(self.lib.ac_buffer_compute_returns(self._h, p, 1), "ac_buffer_compute_returns")
        else:
            v = _host(next_value)
            assert v.size == self.n_rollout_threads * self.num_agents
            self._ok(self.lib.ac_buffer_compute_returns(self._h, v.ctypes.data, 0), "ac_buffer_compute_returns")

    def last_returns_kernel_ms(self):
        ms = C.c_float()
        self._ok(self.lib.ac_buffer_last_kernel_ms(self._h, C.byref(ms)), "ac_buffer_last_kernel_ms")
        return ms.value

    @property
    def advantages(self):
        """buffer.py:72-75 (host copy; the generator below reads the device array)."""
        self._ok(self.lib.ac_buffer_advantages(self._h), "ac_buffer_advantages")
        return self.array("advantages")

    # names of the per-step arrays of a mini-batch, in the order the reference yields them
    _BATCH = ("obs", "actions", "masks", "action_log_probs", "advantages", "returns", "value_preds")

    def minibatch(self, chunks, data_chunk_length):
        """One mini-batch for the given chunk indices (host numpy arrays in the reference's order and shapes)."""
        chunks = np.ascontiguousarray(chunks, dtype=np.int32)
        n, L = len(chunks), int(data_chunk_length)
        widths = {"obs": self.obs_shape, "share_obs": self.share_obs_shape, "actions": self.act_shape, "masks": (1,), "active_masks": (1,),
                  "action_log_probs": self._shapes["action_log_probs"][3:], "advantages": (1,), "returns": (1,), "value_preds": (1,)}
        outs = {k: np.empty((L * n,) + tuple(widths[k]), dtype=np.float32) for k in self._BATCH}
        hid = (self.recurrent_hidden_layers, self.recurrent_hidden_size)
        outs["rnn_states_actor"], outs["rnn_states_critic"] = np.empty((n,) + hid, np.float32), np.empty((n,) + hid, np.float32)
        batch = AcBufferBatch(**{k: v.ctypes.data for k, v in outs.items()})
        self._ok(self.lib.ac_buffer_minibatch(self._h, chunks.ctypes.data, n, L, C.byref(batch), 0), "ac_buffer_minibatch")
        return tuple(outs[k] for k in self._BATCH) + (outs["rnn_states_actor"], outs["rnn_states_critic"])

    @staticmethod
    def recurrent_generator(buffer, num_mini_batch, data_chunk_length, chunk_order=None):
        """ReplayBuffer.recurrent_generator(buffer, num_mini_batch, data_chunk_length) (buffer.py:169-268) for one buffer: the
        chunk permutation comes from torch.randperm like the reference's unless ``chunk_order`` is given (tests). The number of
        chunks follows the reference: n_rollout_threads * buffer_size // data_chunk_length (the agent axis is not counted)."""
        self = buffer
        if isinstance(buffer, (list, tuple)):
            if len(buffer) != 1:
                raise NotImplementedError("one device buffer per generator (the reference's list form concatenates host arrays)")
            self = buffer[0]
        L = int(data_chunk_length)
        assert self.n_rollout_threads * self.buffer_size >= L, "PPO requires n_rollout_threads * buffer_size >= data_chunk_length"
        data_chunks = self.n_rollout_threads * self.buffer_size // L
        mb = data_chunks // int(num_mini_batch)
        if chunk_order is None:
            import torch
            chunk_order = torch.randperm(data_chunks).numpy()
        chunk_order = np.asarray(chunk_order)
        self._ok(self.lib.ac_buffer_advantages(self._h), "ac_buffer_advantages")
        for i in range(int(num_mini_batch)):
            yield self.minibatch(chunk_order[i * mb:(i + 1) * mb], L)


class DeviceSharedReplayBuffer(DeviceReplayBuffer):
    """SharedReplayBuffer(args, num_agents, obs_space, share_obs_space, act_space) (buffer.py:272-311)."""

    _shared = True
    _BATCH = ("obs", "share_obs", "actions", "masks", "active_masks", "action_log_probs", "advantages", "returns", "value_preds")

    def __init__(self, args, num_agents, obs_space, share_obs_space, act_space, device_id=0):
        super().__init__(args, num_agents, obs_space, act_space, share_obs_space=share_obs_space, device_id=device_id)

    def insert(self, obs, share_obs, actions, rewards, masks, action_log_probs, value_preds, rnn_states_actor, rnn_states_critic,
               bad_masks=None, active_masks=None, available_actions=None, on_device=False):
        return super().insert(obs, actions, rewards, masks, action_log_probs, value_preds, rnn_states_actor, rnn_states_critic,
                              bad_masks=bad_masks, share_obs=share_obs, active_masks=active_masks, on_device=on_device)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length, chunk_order=None):   # noqa: signature of buffer.py:350
        """buffer.recurrent_generator(buffer.advantages, num_mini_batch, data_chunk_length) (mappo/ppo_trainer.py:90). The
        advantages argument is the buffer's own normalised advantages in the reference's only call; the device copy is used."""
        return DeviceReplayBuffer.recurrent_generator(self, num_mini_batch, data_chunk_length, chunk_order=chunk_order)
