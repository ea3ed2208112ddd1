"""ctypes binding of include/aircombat.h (the C ABI of libaircombat_hip.so).

The HIP library is the only implementation: there is no CPU fallback. If it is missing, loading raises
``HipExtensionMissing`` — build it with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes as C
import os

AC_MAX_AGENTS = 8
AC_MAX_MISSILES_PER_AGENT = 4
AC_STATE_LEN = 128

AC_TASK_HEADING, AC_TASK_SINGLECOMBAT, AC_TASK_DODGE_MISSILE, AC_TASK_SHOOT_MISSILE, AC_TASK_MULTICOMBAT = 0, 1, 2, 3, 4
AC_TASK_SCENARIO1, AC_TASK_SCENARIO_NVN, AC_TASK_WVR, AC_TASK_MANEUVER = 5, 6, 7, 8
AC_ALIVE, AC_CRASH, AC_SHOTDOWN = 0, 1, 2


class HipExtensionMissing(RuntimeError):
    pass


# ---- include/aircombat_buffer.h
(AC_BUF_OBS, AC_BUF_SHARE_OBS, AC_BUF_ACTIONS, AC_BUF_REWARDS, AC_BUF_MASKS, AC_BUF_BAD_MASKS, AC_BUF_ACTIVE_MASKS, AC_BUF_LOGP,
 AC_BUF_VALUES, AC_BUF_RETURNS, AC_BUF_RNN_ACTOR, AC_BUF_RNN_CRITIC, AC_BUF_ADVANTAGES, AC_BUF_NFIELDS) = range(14)


class AcBufferConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("buffer_size", "n_envs", "n_agents", "obs_dim", "share_obs_dim", "act_dim", "logp_dim",
                                         "hidden_layers", "hidden_size", "use_gae", "use_proper_time_limits")] + \
               [("gamma", C.c_double), ("gae_lambda", C.c_double)]


class AcBufferStep(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "actions", "rewards", "masks", "action_log_probs", "value_preds", "rnn_states_actor",
                                          "rnn_states_critic", "bad_masks", "share_obs", "active_masks")]


class AcBufferBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "share_obs", "actions", "masks", "active_masks", "action_log_probs", "advantages", "returns",
                                          "value_preds", "rnn_states_actor", "rnn_states_critic")]


class AcInitState(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("lon_deg", "lat_geod_deg", "h_sl_ft", "psi_deg", "u_fps", "v_fps", "w_fps",
                 "p_rad_sec", "q_rad_sec", "r_rad_sec")]


class AcConfig(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("n_agents", C.c_int32), ("n_ego", C.c_int32), ("sim_freq", C.c_int32),
        ("agent_interaction_steps", C.c_int32), ("max_steps", C.c_int32),
        ("center_lon", C.c_double), ("center_lat", C.c_double), ("center_alt", C.c_double),
        ("altitude_limit", C.c_double), ("acc_limit_x", C.c_double), ("acc_limit_y", C.c_double), ("acc_limit_z", C.c_double),
        ("init", AcInitState * AC_MAX_AGENTS), ("num_missiles", C.c_int32 * AC_MAX_AGENTS),
        ("posture_scale", C.c_double), ("posture_potential", C.c_int32),
        ("altitude_scale", C.c_double), ("altitude_potential", C.c_int32),
        ("event_scale", C.c_double), ("event_potential", C.c_int32),
        ("missile_posture_scale", C.c_double),
        ("shoot_penalty_scale", C.c_double), ("shoot_penalty_potential", C.c_int32),
        ("alt_safe", C.c_double), ("alt_danger", C.c_double), ("alt_kv", C.c_double),
        ("max_attack_angle", C.c_double), ("max_attack_distance", C.c_double), ("min_attack_interval", C.c_int32),
        ("use_artillery", C.c_int32),
        ("heading_scale", C.c_double), ("heading_potential", C.c_int32),
        ("max_heading_increment", C.c_double), ("max_altitude_increment", C.c_double),
        ("max_velocities_u_increment", C.c_double), ("check_interval", C.c_double),
        ("legacy_obs", C.c_int32),
        ("rwr", C.c_int32),
        ("use_baseline", C.c_int32),
        ("hierarchical", C.c_int32), ("approach", C.c_int32),
    ]


# every symbol include/aircombat.h declares: (restype, argtypes)
_p = C.c_void_p
SIGNATURES = {
    "ac_state_field_name": (C.c_char_p, [C.c_int]),
    "ac_create": (C.c_int, [C.POINTER(AcConfig), C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_p)]),
    "ac_destroy": (C.c_int, [_p]),
    "ac_obs_dim": (C.c_int, [_p]),
    "ac_act_dim": (C.c_int, [_p]),
    "ac_num_envs": (C.c_int, [_p]),
    "ac_num_agents": (C.c_int, [_p]),
    "ac_reset": (C.c_int, [_p, _p]),
    "ac_step": (C.c_int, [_p, _p, _p, _p, _p, _p]),
    "ac_host_buffers": (C.c_int, [_p, C.c_int32, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "ac_host_set_detach": (C.c_int, [_p, C.c_int32]),
    "ac_host_set_free": (None, [_p, _p, _p, _p, _p]),
    "ac_step_host_async": (C.c_int, [_p, C.c_int32]),
    "ac_step_host_wait": (C.c_int, [_p]),
    "ac_step_host": (C.c_int, [_p, C.c_int32]),
    "ac_order_after": (C.c_int, [_p, _p]),
    "ac_order_before": (C.c_int, [_p, _p]),
    "ac_step_async_device": (C.c_int, [_p, _p]),
    "ac_device_buffers": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "ac_stream": (_p, [_p]),
    "ac_sync": (C.c_int, [_p]),
    "ac_get_state": (C.c_int, [_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "ac_set_state": (C.c_int, [_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "ac_set_status": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32]),
    "ac_get_entity": (C.c_int, [_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "ac_get_missile": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "ac_timing_begin": (C.c_int, [_p]),
    "ac_timing_end": (C.c_int, [_p, C.POINTER(C.c_float)]),
    "ac_step_timed_device": (C.c_int, [_p, _p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ac_state_checksum": (C.c_int, [_p, C.POINTER(C.c_uint64)]),
    "ac_munitions_in_flight": (C.c_int, [_p, C.POINTER(C.c_int32)]),
    "ac_seed_envs": (C.c_int, [_p, _p]),
    "ac_get_heading_state": (C.c_int, [_p, C.c_int32, _p]),
    "ac_pin_host_buffer": (C.c_int, [_p, _p, C.c_int64]),
    "ac_unpin_host_buffer": (C.c_int, [_p, _p]),
    "ac_load_controller": (C.c_int, [_p, _p, C.c_int64]),
    "ac_split_f16x2": (C.c_int, [_p, C.c_int64, _p, _p]),
    "ac_selftest_missile_walk": (C.c_int, [C.c_int32, _p]),
    "ac_get_controller_state": (C.c_int, [_p, C.c_int32, C.c_int32, _p, _p]),
    "ac_set_controller_state": (C.c_int, [_p, C.c_int32, C.c_int32, _p]),
    "ac_last_error": (C.c_char_p, []),
    "ac_version": (C.c_char_p, []),
    # include/aircombat_buffer.h
    "ac_buffer_create": (_p, [C.POINTER(AcBufferConfig), C.c_int]),
    "ac_buffer_destroy": (None, [_p]),
    "ac_buffer_insert": (C.c_int, [_p, C.POINTER(AcBufferStep), C.c_int]),
    "ac_buffer_step_index": (C.c_int, [_p]),
    "ac_buffer_after_update": (C.c_int, [_p]),
    "ac_buffer_clear": (C.c_int, [_p]),
    "ac_buffer_compute_returns": (C.c_int, [_p, _p, C.c_int]),
    "ac_buffer_advantages": (C.c_int, [_p]),
    "ac_buffer_minibatch": (C.c_int, [_p, _p, C.c_int32, C.c_int32, C.POINTER(AcBufferBatch), C.c_int]),
    "ac_buffer_device_ptr": (C.c_int, [_p, C.c_int32, C.POINTER(_p), C.POINTER(C.c_int64)]),
    "ac_buffer_read": (C.c_int, [_p, C.c_int32, _p]),
    "ac_buffer_write_slot": (C.c_int, [_p, C.c_int32, C.c_int32, _p]),
    "ac_buffer_last_kernel_ms": (C.c_int, [_p, C.POINTER(C.c_float)]),
}


def library_path():
    override = os.environ.get("AIRCOMBAT_HIP_LIB")   # another build of the same extension (profiling variants); never a fallback
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libaircombat_hip.so")


class Lib:
    """Loaded libaircombat_hip.so with typed entry points."""

    def __init__(self, path=None):
        path = path or library_path()
        if not os.path.exists(path):
            raise HipExtensionMissing(
                f"{path} not found: the HIP extension is the only implementation of the step() path. "
                "Build it with __graft_entry__.build() (hipcc --offload-arch=gfx950).")
        try:
            self.dll = C.CDLL(path)
        except OSError as exc:  # e.g. libamdhip64 missing
            raise HipExtensionMissing(f"cannot load {path}: {exc}") from exc
        self.path = path
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def last_error(self):
        return (self.ac_last_error() or b"").decode()

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.last_error()}")

    def state_field_names(self):
        return [self.ac_state_field_name(i).decode() for i in range(AC_STATE_LEN)]


_LIB = None


def load_library(path=None):
    global _LIB
    # Kernel arguments in device memory: with them in host memory every kernel's first scalar load crosses PCIe (measured on this stack:
    # the BASELINE step kernel 16.3 -> 19.4 us with HIP_FORCE_DEV_KERNARG=0). ROCm 7 defaults to device memory; on a runtime that does not,
    # this asks for it -- it only takes effect if no HIP call has been made in the process yet, and never overrides the caller's setting.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    if _LIB is None or (path and _LIB.path != path):
        _LIB = Lib(path)
    return _LIB
