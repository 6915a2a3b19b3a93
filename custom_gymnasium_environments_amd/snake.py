"""SnakeVectorEnv — batched drop-in for SnakeEnvClassic (snake_env_classic/snake_env.py:9-143)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, Discrete, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"score": 0, "snake_length": 1, "steps": 2, "direction": 3, "food_r": 4, "food_c": 5,
               "board_full": 6, "episodes": 7, "head_r": 8, "head_c": 9, "needs_reset": 10}


class SnakeVectorEnv(DeviceVectorEnv):
    """N independent SnakeEnvClassic instances stepped by one HIP kernel launch.

    Same spaces as the reference (snake_env.py:26-32): `Discrete(4)` actions (0 up, 1 right,
    2 down, 3 left), obs `Box(0, 2, (G, G), int8)` (0 empty, 1 snake, 2 food); rewards
    -10 / 0 / +10; `terminated` on wall or self collision or after `max_steps`=1000 steps;
    `truncated` is always False (snake_env.py:119).

    RNG protocol: the reference draws food positions from the process-global `random` and never
    seeds it; here env i owns the stream `random.seed(seed + env_index0 + i)` (bit-exact CPython
    MT19937), which is what one gets from the reference by running that env alone after
    `random.seed(...)`.  Auto-reset continues the stream, as `env.reset()` does.

    info_fields: names from INFO_FIELDS to return in `infos` each step (the reference returns
    `score` and `snake_length`, snake_env.py:63,117); each costs one small kernel, default none.
    """

    _abi = "cge_snake"
    _obs_dtype = torch.int8
    metadata = {"render_modes": ["rgb_array"]}

    def __init__(self, num_envs, grid_size=20, device="cuda:0", autoreset_mode="NextStep", env_index0=0,
                 max_steps=1000, reuse_buffers=False, info_fields=(), record_episode_statistics=False, render_mode=None,
                 reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        if render_mode not in (None, "rgb_array"):
            raise ValueError("render_mode must be None or 'rgb_array' (the pygame window of 'human' is out of scope)")
        self.render_mode = render_mode
        self.grid_size = int(grid_size)
        self.max_steps = int(max_steps)
        self.single_action_space = Discrete(4)
        self.single_observation_space = Box(0, 2, (self.grid_size, self.grid_size), np.int8)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        for f in info_fields:
            if f not in INFO_FIELDS:
                raise ValueError(f"unknown info field {f!r}; choose from {sorted(INFO_FIELDS)}")
        self.info_fields = tuple(info_fields)
        cfg = _native.SnakeConfig(self.grid_size, self.max_steps, self._mode_code, 0)
        h = C.c_void_p()
        st = self._lib.cge_snake_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h))
        if st == -3:
            raise ValueError(f"grid_size={grid_size} is not compiled into libcge_amd.so (supported: every size from 4 to 30)")
        _native.check(st, what="cge_snake_create")
        self._h = h
        self._obs_shape = (self.num_envs, self.grid_size, self.grid_size)
        self.record_episode_statistics(record_episode_statistics)

    # ------------------------------------------------------------------ gymnasium API
    def reset(self, *, seed=None, options=None):
        """Reset every env (or those in options['reset_mask']).  Returns (obs, infos)."""
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.int8)
        self._check(self._lib.cge_snake_reset(self._h, mask.data_ptr() if mask is not None else None,
                                              obs.data_ptr(), self._stream()), "reset")
        return obs, self._infos()

    def step(self, actions):
        a = self._as_device(actions, torch.int32, (self.num_envs,), "actions")
        obs = self._out("obs", self._obs_shape, torch.int8)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        # the reference never truncates (snake_env.py:119): one shared all-False tensor, never rewritten
        trunc = self._never_truncated()
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.int8) if same else None
        self._check(self._lib.cge_snake_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(),
                                             None, fin.data_ptr() if same else None, self._stream()), "step")
        infos = self._infos()
        if same:
            # rows of final_obs are valid where _final_obs is True (gymnasium's SAME_STEP convention)
            infos["final_obs"] = fin
            infos["_final_obs"] = term
        return obs, rew, term, trunc, self._episode_infos(infos, term)

    def _never_truncated(self):
        t = self._bufs.get("_truncated")
        if t is None:
            t = self._bufs["_truncated"] = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        return t

    # ------------------------------------------------------------------ extras
    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
        """k fused step()s in one launch (state stays in registers).  actions: None -> counter-hash
        random actions (cge_hash_action) or an int32 [k, N] tensor.  Returns (obs, reward_sum, done_count)
        with obs of shape [k, N, G, G] if trajectory else the last step's [N, G, G]; with per_step=True
        returns (obs, reward[k, N], terminated[k, N], reward_sum, done_count) — the outputs of k step() calls."""
        k = int(k_steps)
        a = None if actions is None else self._as_device(actions, torch.int32, (k, self.num_envs), "actions")
        obs = None
        stride = 0
        if want_obs:
            if trajectory:
                obs = self._out("traj", (k,) + self._obs_shape, torch.int8)
                stride = self.num_envs * self.grid_size * self.grid_size
            else:
                obs = self._out("obs", self._obs_shape, torch.int8)
        rs = self._out("reward_sum", (self.num_envs,), torch.float32)
        dc = self._out("done_count", (self.num_envs,), torch.int32)
        rt = tt = None
        if per_step:
            rt = self._out("reward_traj", (k, self.num_envs), torch.float32)
            tt = self._out("terminated_traj", (k, self.num_envs), torch.bool)
        self._check(self._lib.cge_snake_rollout(self._h, k, a.data_ptr() if a is not None else None, int(action_seed),
                                                int(t0), obs.data_ptr() if obs is not None else None, stride,
                                                rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        if per_step:
            return obs, rt, tt, rs, dc
        return obs, rs, dc

    def render_rgb(self):
        """render_mode="rgb_array" for the whole batch (snake_env.py:175-188): uint8 [N, G, G, 3] of the current states —
        empty black, snake (0, 255, 0), food (255, 0, 0)."""
        out = self._out("rgb", self._obs_shape + (3,), torch.uint8)
        self._check(self._lib.cge_snake_render_rgb(self._h, out.data_ptr(), self._stream()), "render_rgb")
        return out

    def render(self):
        """gymnasium.vector.VectorEnv.render(): a tuple with one frame per env when render_mode == "rgb_array"."""
        if self.render_mode != "rgb_array":
            return None
        return tuple(self.render_rgb())

    def info(self, field):
        out = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        self._check(self._lib.cge_snake_info(self._h, INFO_FIELDS[field], out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        d = {f: self.info(f) for f in self.info_fields}
        if self._reference_info:
            d.update(self.reference_info())
        return d

    def reference_info(self):
        """The reference's `info` (snake_env.py:62,117): {"score", "snake_length"} of the env's CURRENT state — after a SAME_STEP
        auto-reset that is the fresh episode's (the finished episode's score is what `record_episode_statistics` reports:
        return = 10 * score - 10 on a crash).  `reference_info=True` merges it into every `infos`."""
        return {"score": self.info("score"), "snake_length": self.info("snake_length")}

    def invalid_action_count(self):
        """Synchronises; number of out-of-range actions since the last call (reference: ValueError)."""
        return int(self._lib.cge_snake_error_count(self._h, self._stream()))

    def check_actions(self):
        n = self.invalid_action_count()
        if n:
            raise ValueError(f"Invalid action in {n} env-step(s)")  # snake_env.py:69-70

    def get_state(self):
        rec = int(self._lib.cge_snake_state_bytes(self._h))
        buf = np.zeros((self.num_envs, rec), np.uint8)
        self._check(self._lib.cge_snake_get_state(self._h, buf.ctypes.data, self._stream()), "get_state")
        return buf

    def set_state(self, buf):
        rec = int(self._lib.cge_snake_state_bytes(self._h))
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.shape != (self.num_envs, rec):
            raise ValueError(f"state buffer must be uint8 {(self.num_envs, rec)}")
        self._check(self._lib.cge_snake_set_state(self._h, buf.ctypes.data, self._stream()), "set_state")
