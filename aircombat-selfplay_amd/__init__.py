"""aircombat-selfplay_amd — MI355X-native vectorised air-combat ``step()``.

Host-side mirror of the reference's VecEnv surface (``envs/env_wrappers.py``) over the C ABI of
``include/aircombat.h`` (``libaircombat_hip.so``: hand-written HIP kernels for gfx950).

The directory name carries a hyphen, so import it as ``importlib.import_module("aircombat-selfplay_amd")``
or through the repo-root alias module ``aircombat_selfplay_amd``.
"""
from .capi import AcConfig, AcInitState, Lib, load_library, library_path, HipExtensionMissing  # noqa: F401
from .config import config_from_yaml, default_config, default_nvn_config, TASK_IDS  # noqa: F401
from .vec_env import HipVecEnv, HipShareVecEnv, MultiDeviceVecEnv, make_env  # noqa: F401
from .rollout_buffer import DeviceReplayBuffer, DeviceSharedReplayBuffer  # noqa: F401
from . import sharding  # noqa: F401

__all__ = ["AcConfig", "AcInitState", "Lib", "load_library", "library_path", "HipExtensionMissing",
           "config_from_yaml", "default_config", "default_nvn_config", "TASK_IDS", "HipVecEnv", "HipShareVecEnv", "MultiDeviceVecEnv", "make_env",
           "DeviceReplayBuffer", "DeviceSharedReplayBuffer"]
