"""gymnasium.vector.VectorEnv façade over the C ABI (include/cge_amd.h): shared plumbing.

The reference implements single `gym.Env`s (L2 in SURVEY.md) and is batched only by gymnasium's
SyncVectorEnv Python loop inside RLlib (smart_parking_env/examples/training.py:43-47).  The classes
built on this module occupy that SyncVectorEnv slot: same attribute surface, same
reset()/step() return tuples, but the batch lives in HBM and one step() is one kernel launch.
Observations/rewards/flags are torch tensors on the env's device (no host sync per step).
"""
import enum

import numpy as np
import torch

from . import _native
from ._spaces import VectorEnvBase

try:  # pragma: no cover
    from gymnasium.vector import AutoresetMode
except Exception:
    class AutoresetMode(enum.Enum):
        """Mirror of gymnasium.vector.AutoresetMode (gymnasium >= 1.1)."""
        NEXT_STEP = "NextStep"
        SAME_STEP = "SameStep"
        DISABLED = "Disabled"

_MODE_CODE = {"NEXT_STEP": _native.AUTORESET_NEXT_STEP, "SAME_STEP": _native.AUTORESET_SAME_STEP,
              "DISABLED": _native.AUTORESET_DISABLED}


def parse_autoreset_mode(mode):
    if isinstance(mode, str):
        key = mode.replace("-", "_").upper()
        key = {"NEXTSTEP": "NEXT_STEP", "SAMESTEP": "SAME_STEP"}.get(key, key)
        if key not in _MODE_CODE:
            raise ValueError(f"unknown autoreset_mode {mode!r}")
        return AutoresetMode[key]
    if hasattr(mode, "name") and mode.name in _MODE_CODE:
        return AutoresetMode[mode.name]
    raise ValueError(f"unknown autoreset_mode {mode!r}")


class DeviceVectorEnv(VectorEnvBase):
    """Base of every batched env: owns the native handle, the device and the output buffers."""

    _abi = None  # e.g. "cge_snake"

    def _init_common(self, num_envs, device, autoreset_mode, env_index0, reuse_buffers):
        if int(num_envs) <= 0:
            raise ValueError("num_envs must be positive")
        self.num_envs = int(num_envs)
        self.env_index0 = int(env_index0)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeLibraryError(
                f"{type(self).__name__} runs only on an MI355X (device 'cuda:N' under PyTorch-ROCm); got {device!r}. "
                "There is no CPU path.")
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError("no HIP device is visible to PyTorch; there is no CPU path")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self._dev_index)
        self.autoreset_mode = parse_autoreset_mode(autoreset_mode)
        self._mode_code = _MODE_CODE[self.autoreset_mode.name]
        self.metadata = dict(getattr(type(self), "metadata", {}), autoreset_mode=self.autoreset_mode)
        self._reuse = bool(reuse_buffers)
        self._lib = _native.lib()
        self._h = None
        self._bufs = {}
        self.closed = False

    # ------------------------------------------------------------------ native helpers
    def _fn(self, name):
        return getattr(self._lib, f"{self._abi}_{name}")

    def _check(self, status, what):
        if status:                                             # hot path: no lookups or string formatting on success
            _native.check(status, self._h, self._fn("last_error"), f"{self._abi}_{what}")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _out(self, key, shape, dtype):
        """Output tensor: a fresh allocation per call (gymnasium's `copy=True` contract) or, with
        reuse_buffers=True, one persistent buffer per output that the next call overwrites."""
        if self._reuse:
            t = self._bufs.get(key)
            if t is None:
                t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=self.device)
            return t
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _as_device(self, x, dtype, shape, what):
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=dtype, non_blocking=True)
        else:
            t = torch.as_tensor(np.asarray(x), device=self.device).to(dtype)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        return t.contiguous()

    def _seed_native(self, seed):
        """seed: None (streams continue), int (env i gets seed + env_index0 + i) or a per-env sequence."""
        if seed is None:
            return
        if isinstance(seed, (int, np.integer)):
            if seed < 0:
                raise ValueError("seed must be non-negative")
            self._check(self._fn("seed")(self._h, None, int(seed), self._stream()), "seed")
            return
        arr = np.asarray(seed)
        if arr.shape != (self.num_envs,) or np.any(arr < 0):
            raise ValueError(f"seed sequence must hold {self.num_envs} non-negative ints")
        t = torch.from_numpy(arr.astype(np.uint64).view(np.int64)).to(self.device)
        self._check(self._fn("seed")(self._h, t.data_ptr(), 0, self._stream()), "seed")
        self._keepalive = t

    def device_bytes(self):
        return int(self._fn("device_bytes")(self._h))

    def snapshot(self):
        """Whole-batch checkpoint as an opaque uint8 array (env types without a canonical per-env `get_state` record).
        Restores only into an env created with the same num_envs and config; synchronises the stream."""
        if not hasattr(self._lib, f"{self._abi}_snapshot_bytes"):
            raise NotImplementedError(f"{self._abi}: use get_state()/set_state()")
        import numpy as np
        buf = np.zeros(int(self._fn("snapshot_bytes")(self._h)), np.uint8)
        self._check(self._fn("snapshot_get")(self._h, buf.ctypes.data, self._stream()), "snapshot_get")
        return buf

    def restore(self, buf):
        import numpy as np
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.shape != (int(self._fn("snapshot_bytes")(self._h)),):
            raise ValueError("not a snapshot of an env of this type and size")
        self._check(self._fn("snapshot_set")(self._h, buf.ctypes.data, self._stream()), "snapshot_set")

    def close_extras(self, **kwargs):
        if getattr(self, "_h", None):
            self._fn("destroy")(self._h)
            self._h = None
