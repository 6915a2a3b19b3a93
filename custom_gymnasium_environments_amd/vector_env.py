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
        self._ep_ret = self._ep_len = None
        self.closed = False

    # ------------------------------------------------------------------ native helpers
    def _fn(self, name):
        return getattr(self._lib, f"{self._abi}_{name}")

    def _check(self, status, what):
        if status:                                             # hot path: no lookups or string formatting on success
            _native.check(status, self._h, self._fn("last_error"), f"{self._abi}_{what}")

    def _stream(self):
        # the raw handle of torch's current stream on this device (the private getter skips building a Stream object: ~2 us per call
        # on the step() path, where a kernel is 30-40 us)
        get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        return get(self._dev_index) if get is not None else torch.cuda.current_stream(self.device).cuda_stream

    def _out(self, key, shape, dtype):
        """Output tensor: a fresh allocation per call (gymnasium's `copy=True` contract) or, with
        reuse_buffers=True, one persistent buffer per output that the next call overwrites."""
        if self._reuse:
            t = self._bufs.get(key)
            if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == dtype:
                return t
            numel = int(np.prod(shape))
            if t is None or t.dtype != dtype or t.numel() < numel:           # first use, or a larger request (e.g. a longer rollout)
                t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=self.device)
                return t
            return t.view(-1)[:numel].view(shape)                            # a shorter rollout reuses the front of the buffer
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

    def last_kernel(self):
        """Name(s) of the kernel(s) the last step() / rollout() launched, as rocprofv3 prints them ("" before the first call)."""
        raw = self._fn("last_kernel")(self._h)
        return raw.decode() if raw else ""

    # ------------------------------------------------------------------ episode statistics
    def record_episode_statistics(self, enable=True):
        """What gymnasium.wrappers.vector.RecordEpisodeStatistics adds around a vector env, computed by the step kernel itself:
        with it enabled every step()'s infos carries `episode = {"r": float64[N], "l": int32[N]}` and the mask `_episode`
        (= terminated | truncated); r / l of env i are the return (float64 sum of the episode's rewards in step order) and
        the length in env steps of the episode that just ended — the numbers the reference's RLlib scripts report as
        episode_return_mean / episode_len_mean (smart_parking_env/examples/training.py:55).  Entries where `_episode` is False
        keep the env's previous episode.  Enable it before the episode starts (constructor flag or before reset())."""
        if enable:
            self._ep_ret = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
            self._ep_len = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            self._check(self._fn("episode_stats")(self._h, self._ep_ret.data_ptr(), self._ep_len.data_ptr()), "episode_stats")
        elif getattr(self, "_ep_ret", None) is not None:
            torch.cuda.current_stream(self.device).synchronize()          # no kernel may still be writing the buffers
            self._check(self._fn("episode_stats")(self._h, None, None), "episode_stats")
            self._ep_ret = self._ep_len = None
        else:
            self._ep_ret = self._ep_len = None

    def episode_statistics(self):
        """(return float64[N], length int32[N]) of each env's last finished episode (also after a rollout()), or None."""
        return None if self._ep_ret is None else (self._ep_ret, self._ep_len)

    def _episode_infos(self, infos, done):
        if self._ep_ret is not None:
            infos["episode"] = {"r": self._ep_ret, "l": self._ep_len}
            infos["_episode"] = done
        return infos

    # ------------------------------------------------------------------ terminal observations of fused SAME_STEP rollouts
    _obs_dtype = torch.float32

    def collect_final_obs(self, rows_per_env=4):
        """A SAME_STEP `rollout()` writes the RESET observation of an env that finishes at step t to slot t of its trajectory —
        what `step()` returns as `obs`; the terminal observation, which `step()` hands over as `infos["final_obs"]` (and the
        reference's own step() returns), goes to a side output once this is enabled.  The rows are compacted per segment of S
        consecutive envs (the envs one wavefront steps: `self.final_obs_segment`); a segment holds S * rows_per_env rows per rollout
        call — size it for the episodes a rollout can end (`final_obs_dropped()` says whether it was enough).  Read the rows back with
        `final_obs()`; `rows_per_env=0` disables the output again."""
        seg = self.final_obs_segment = int(self._fn("final_obs_segment")(self._h))
        if rows_per_env <= 0:
            torch.cuda.current_stream(self.device).synchronize()
            self._check(self._fn("rollout_final_obs")(self._h, None, None, 0, None), "rollout_final_obs")
            self._fin = None
            return
        nseg, cap = -(-self.num_envs // seg), int(seg * rows_per_env)
        rows = torch.empty((nseg * cap,) + tuple(self._obs_shape[1:]), dtype=self._obs_dtype, device=self.device)
        index = torch.empty(nseg * cap, dtype=torch.int64, device=self.device)
        count = torch.zeros(nseg, dtype=torch.int32, device=self.device)
        self._check(self._fn("rollout_final_obs")(self._h, rows.data_ptr(), index.data_ptr(), cap, count.data_ptr()), "rollout_final_obs")
        self._fin = (rows, index, count, cap)

    def _final_obs_begin(self):
        pass                                     # every rollout call writes the segments' counts itself

    def final_obs(self):
        """(rows [m, *obs_shape], step [m], env [m]) delivered by the last rollout(), sorted by (step, env): row j is the terminal
        observation of env[j] at step step[j] of that call.  Gathers the segments' rows on the device (boolean indexing: synchronises)."""
        rows, index, count, cap = self._fin
        nseg = count.shape[0]
        keep = (torch.arange(cap, device=self.device)[None, :] < count.clamp(max=cap)[:, None]).reshape(-1)
        idx = index[keep]
        order = torch.argsort(idx)
        idx = idx[order]
        return rows[keep][order], idx // self.num_envs, idx % self.num_envs

    def final_obs_dropped(self):
        rows, _, count, cap = self._fin
        return int((count - cap).clamp(min=0).sum().item())

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


class FlagsVectorEnv(DeviceVectorEnv):
    """Shared façade of the env types whose C ABI has the same shape: int32 actions of a fixed per-env shape, float32
    observations, both `terminated` and `truncated` reported, float64 info fields (fleet, manufacturing, hospital).
    A subclass sets `_abi`, `_obs_dim`, `_action_shape`, `INFO_FIELDS`, builds the spaces and creates the native handle."""

    _obs_dim = None
    _action_shape = ()
    INFO_FIELDS = {}

    def _finish_init(self, info_fields):
        self.info_fields = tuple(info_fields)
        self._done_ptr = 0
        self._obs_shape = (self.num_envs, self._obs_dim)

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._fn("reset")(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(), self._stream()), "reset")
        return obs, self._infos()

    def step(self, actions):
        a = self._as_device(actions, torch.int32, (self.num_envs,) + self._action_shape, "actions")
        obs = self._out("obs", self._obs_shape, torch.float32)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        trunc = self._out("truncated", (self.num_envs,), torch.bool)
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.float32) if same else None
        done = None
        if same or self._ep_ret is not None:                   # terminated | truncated, written by the step kernel itself
            done = self._out("done", (self.num_envs,), torch.bool)
            if done.data_ptr() != self._done_ptr:
                self._done_ptr = done.data_ptr()
                self._check(self._fn("done_mask")(self._h, self._done_ptr), "done_mask")
        elif self._done_ptr:
            self._done_ptr = 0
            self._check(self._fn("done_mask")(self._h, None), "done_mask")
        self._check(self._fn("step")(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                                     fin.data_ptr() if same else None, self._stream()), "step")
        infos = self._infos()
        if same:
            infos["final_obs"] = fin
            infos["_final_obs"] = done
        if self._ep_ret is not None:
            self._episode_infos(infos, done)
        return obs, rew, term, trunc, infos

    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
        """k env steps queued by one C-ABI call (actions: None -> counter-hash actions, or int32 [k, N, ...]).  Returns
        (obs, reward_sum, done_count); with per_step=True (obs, reward[k, N], flags[k, N], reward_sum, done_count) where
        flags = terminated | truncated << 1 (uint8).  obs is [k, N, obs_dim] if trajectory else the last step's."""
        k = int(k_steps)
        a = None if actions is None else self._as_device(actions, torch.int32, (k, self.num_envs) + self._action_shape, "actions")
        obs, stride = None, 0
        if want_obs:
            if trajectory:
                obs = self._out("traj", (k,) + self._obs_shape, torch.float32)
                stride = self.num_envs * self._obs_dim
            else:
                obs = self._out("obs", self._obs_shape, torch.float32)
        rs = self._out("reward_sum", (self.num_envs,), torch.float64)
        dc = self._out("done_count", (self.num_envs,), torch.int32)
        rt = tt = None
        if per_step:
            rt = self._out("reward_traj", (k, self.num_envs), torch.float32)
            tt = self._out("flags_traj", (k, self.num_envs), torch.uint8)
        self._check(self._fn("rollout")(self._h, k, a.data_ptr() if a is not None else None, int(action_seed), int(t0),
                                        obs.data_ptr() if obs is not None else None, stride,
                                        rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                        rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field):
        out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        self._check(self._fn("info")(self._h, self.INFO_FIELDS[field], out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        d = {f: self.info(f) for f in self.info_fields}
        if getattr(self, "_reference_info", False):
            d.update(self.reference_info())
        return d

    def reference_info(self):
        """The reference's own `info` dict (its keys, its derived expressions) as tensors; subclasses define it."""
        raise NotImplementedError
