"""ManufacturingVectorEnv — batched drop-in for SmartManufacturingEnv (smart_manufacturing_env/manufacturing_env.py:69-606)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, Discrete, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"raw_material": 0, "energy_consumption": 1, "total_reward": 2, "in_system": 3, "completed": 4, "scrapped": 5,
               "product_ids": 6, "history_len": 7, "oee_availability": 8, "oee_performance": 9, "oee_quality": 10,
               "timestep": 11, "episodes": 12, "needs_reset": 13, "overflow": 14}
OBS_DIM = 73


class ManufacturingVectorEnv(DeviceVectorEnv):
    """N independent SmartManufacturingEnv instances (5 stations, 6 product types, quality checkpoints, machine
    breakdowns, supply disruptions) stepped by one HIP kernel launch.  Actions `Discrete(25)` (:303-359), obs float32
    (73,).  Both `terminated` (:555-578) and `truncated` (timestep >= 1500) are reported; auto-reset triggers on either.
    `reset(seed=s)` gives env i gymnasium's `np_random` for seed s + env_index0 + i (PCG64 from SeedSequence, :115);
    a later `reset()` continues the stream.  Bit-exact with the reference, including the NumPy pairwise-summed
    per-type quality means of the observation."""

    _abi = "cge_manufacturing"
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_steps=1500, reuse_buffers=False,
                 info_fields=()):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self.single_action_space = Discrete(25)
        self.single_observation_space = Box(0.0, 500.0, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.info_fields = tuple(info_fields)
        cfg = _native.ManufacturingConfig(int(max_steps), self._mode_code)
        h = C.c_void_p()
        _native.check(self._lib.cge_manufacturing_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)),
                      what="cge_manufacturing_create")
        self._h = h
        self._obs_shape = (self.num_envs, OBS_DIM)

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._lib.cge_manufacturing_reset(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(),
                                              self._stream()), "reset")
        return obs, self._infos()

    def step(self, actions):
        a = self._as_device(actions, torch.int32, (self.num_envs,), "actions")
        obs = self._out("obs", self._obs_shape, torch.float32)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        trunc = self._out("truncated", (self.num_envs,), torch.bool)
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.float32) if same else None
        self._check(self._lib.cge_manufacturing_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                                             fin.data_ptr() if same else None, self._stream()), "step")
        infos = self._infos()
        if same:
            infos["final_obs"] = fin
            infos["_final_obs"] = term | trunc
        return obs, rew, term, trunc, infos

    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
        """k fused steps; with per_step=True the flags trajectory holds terminated | truncated << 1 (uint8)."""
        k = int(k_steps)
        a = None if actions is None else self._as_device(actions, torch.int32, (k, self.num_envs), "actions")
        obs, stride = None, 0
        if want_obs:
            if trajectory:
                obs = self._out("traj", (k,) + self._obs_shape, torch.float32)
                stride = self.num_envs * OBS_DIM
            else:
                obs = self._out("obs", self._obs_shape, torch.float32)
        rs = self._out("reward_sum", (self.num_envs,), torch.float64)
        dc = self._out("done_count", (self.num_envs,), torch.int32)
        rt = tt = None
        if per_step:
            rt = self._out("reward_traj", (k, self.num_envs), torch.float32)
            tt = self._out("flags_traj", (k, self.num_envs), torch.uint8)
        self._check(self._lib.cge_manufacturing_rollout(self._h, k, a.data_ptr() if a is not None else None, int(action_seed), int(t0),
                                                obs.data_ptr() if obs is not None else None, stride,
                                                rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field):
        out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        self._check(self._lib.cge_manufacturing_info(self._h, INFO_FIELDS[field], out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        return {f: self.info(f) for f in self.info_fields}
