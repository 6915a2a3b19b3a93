"""ClimateVectorEnv — batched drop-in for SmartClimateEnv (smartclimate_rl-main/smartclimate/env.py:10-116)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"room_temp": 0, "outside_temp": 1, "ac_setting": 2, "energy_usage": 3, "total_reward": 4, "num_people": 5,
               "step": 6, "comfort_time": 7, "episodes": 8, "needs_reset": 9}
OBS_DIM = 9


class ClimateVectorEnv(DeviceVectorEnv):
    """N independent SmartClimateEnv instances stepped by one HIP kernel launch.

    Observation `Box((9,), float32)`: room_temp, num_people, time_of_day, outside_temp, ac_setting, 4 light
    states (:74-83).  Action: the reference's Dict space (:39-43) batched — `{"ac_temp": float32 (N,1),
    "lights": int8 (N,4)}` (a tuple `(ac_temp, lights)` is accepted too).  `reset(seed=s)` gives env i the
    private generator `np.random.default_rng(s + env_index0 + i)` (:63-65); float64 dynamics, float32 obs.
    """

    _abi = "cge_climate"
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_occupancy=8, episode_minutes=1440,
                 reuse_buffers=False, info_fields=(), record_episode_statistics=False, reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        self._last_obs = None
        lo = np.array([0.0, 0, 0.0, 10.0, 16.0, 0, 0, 0, 0]); hi = np.array([50.0, max_occupancy, 23.99, 50.0, 32.0, 1, 1, 1, 1])
        self.single_observation_space = Box(lo, hi, (OBS_DIM,), np.float32)
        self.single_action_space = {"ac_temp": Box(16.0, 32.0, (1,), np.float32), "lights": Box(0, 1, (4,), np.int8)}
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.action_space = {k: batch_space(v, self.num_envs) for k, v in self.single_action_space.items()}
        self.info_fields = tuple(info_fields)
        cfg = _native.ClimateConfig(int(max_occupancy), int(episode_minutes), self._mode_code, 0)
        h = C.c_void_p()
        _native.check(self._lib.cge_climate_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)),
                      what="cge_climate_create")
        self._h = h
        self._obs_shape = (self.num_envs, OBS_DIM)
        self.record_episode_statistics(record_episode_statistics)

    def _split(self, actions, k=None):
        ac, li = (actions["ac_temp"], actions["lights"]) if isinstance(actions, dict) else actions
        lead = (self.num_envs,) if k is None else (k, self.num_envs)
        ac = torch.as_tensor(ac) if not isinstance(ac, torch.Tensor) else ac
        ac = ac.reshape(lead).to(device=self.device, dtype=torch.float32).contiguous()
        li = self._as_device(li, torch.int8, lead + (4,), "lights")
        return ac, li

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._lib.cge_climate_reset(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(),
                                                self._stream()), "reset")
        self._last_obs = obs
        return obs, self._infos()

    def step(self, actions):
        ac, li = self._split(actions)
        obs = self._out("obs", self._obs_shape, torch.float32)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        trunc = self._bufs.get("_truncated")
        if trunc is None:
            trunc = self._bufs["_truncated"] = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.float32) if same else None
        self._check(self._lib.cge_climate_step(self._h, ac.data_ptr(), li.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(),
                                               None, fin.data_ptr() if same else None, self._stream()), "step")
        self._last_obs = obs
        infos = self._infos()
        if same:
            infos["final_obs"] = fin
            infos["_final_obs"] = term
        return obs, rew, term, trunc, self._episode_infos(infos, term)

    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
        k = int(k_steps)
        ac = li = None
        if actions is not None:
            ac, li = self._split(actions, k)
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
            tt = self._out("terminated_traj", (k, self.num_envs), torch.bool)
        self._check(self._lib.cge_climate_rollout(self._h, k, ac.data_ptr() if ac is not None else None,
                                                  li.data_ptr() if li is not None else None, int(action_seed), int(t0),
                                                  obs.data_ptr() if obs is not None else None, stride,
                                                  rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                  rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field):
        out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        self._check(self._lib.cge_climate_info(self._h, INFO_FIELDS[field], out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        d = {f: self.info(f) for f in self.info_fields}
        if self._reference_info:
            d.update(self.reference_info())
        return d

    def reference_info(self, obs=None):
        """The reference's step() `info` under ITS keys (env.py:105-110): the three terms of calculate_reward (utils.py:30-50) —
        comfort (10 / 5 / 0 / -15 |T - 22| by the room temperature's band), ac_penalty = -0.5 |ac_setting - outside_temp|,
        light_penalty = -max(0, lights_on - min(4, ceil(num_people / 2))) — and comfort_time, energy_usage, step; float64, from the
        env's CURRENT state (room / outside temperature, AC setting and occupancy from the state record, the light switches from the
        observation: `obs`, default the last one step() / reset() returned).  `reference_info=True` merges it into every `infos`."""
        obs = self._last_obs if obs is None else obs
        room, out, ac, people = self.info("room_temp"), self.info("outside_temp"), self.info("ac_setting"), self.info("num_people")
        comfort = torch.where((room >= 20) & (room <= 24), torch.full_like(room, 10.0),
                              torch.where((room >= 18) & (room <= 26), torch.full_like(room, 5.0),
                                          torch.where((room >= 16) & (room <= 28), torch.zeros_like(room), -15.0 * (room - 22.0).abs())))
        lights_on = obs[:, 5:9].to(torch.float64).sum(1)
        required = torch.clamp(torch.ceil(people / 2.0), max=4.0)
        return {"comfort": comfort, "ac_penalty": -0.5 * (ac - out).abs(), "light_penalty": -torch.clamp(lights_on - required, min=0.0),
                "comfort_time": self.info("comfort_time").to(torch.int64), "energy_usage": self.info("energy_usage"),
                "step": self.info("step").to(torch.int64)}
