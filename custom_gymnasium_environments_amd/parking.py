"""ParkingVectorEnv — batched drop-in for SmartParkingEnv (smart_parking_env/core/parking_env.py:23-433)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, Discrete, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"timestep": 0, "total_customers": 1, "rejected": 2, "satisfied": 3, "total_wait_time": 4, "queue_length": 5,
               "price_changes_this_hour": 6, "zone_occupied": 7, "price_level": 8, "episodes": 9, "needs_reset": 10}
INFO64_FIELDS = {"episode_revenue": 0, "episode_satisfaction": 1}
OBS_DIM = 13


class ParkingVectorEnv(DeviceVectorEnv):
    """N independent SmartParkingEnv instances (50 spots in 3 zones, 10-slot queue, 1440 one-minute steps)
    stepped by one HIP kernel launch.  Spaces as the reference (:53-62): `Discrete(8)` actions (0 idle, 1-3
    assign the queue head to zone A/B/C, 4 reject it, 5-7 toggle the price level of zone A/B/C), obs
    `Box(0, 1, (13,), float32)`.  Bit-exact with the reference.  The reference never seeds `random`
    (:81), so env i owns the stream `random.seed(seed + env_index0 + i)`, as SnakeVectorEnv does."""

    _abi = "cge_parking"
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_steps=1440, reuse_buffers=False,
                 info_fields=(), record_episode_statistics=False, reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        self.single_action_space = Discrete(8)
        self.single_observation_space = Box(0.0, 1.0, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.info_fields = tuple(info_fields)
        cfg = _native.ParkingConfig(int(max_steps), self._mode_code)
        h = C.c_void_p()
        _native.check(self._lib.cge_parking_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)),
                      what="cge_parking_create")
        self._h = h
        self._obs_shape = (self.num_envs, OBS_DIM)
        self.record_episode_statistics(record_episode_statistics)

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._lib.cge_parking_reset(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(),
                                                self._stream()), "reset")
        return obs, self._infos()

    def step(self, actions):
        a = self._as_device(actions, torch.int32, (self.num_envs,), "actions")
        obs = self._out("obs", self._obs_shape, torch.float32)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        trunc = self._bufs.get("_truncated")
        if trunc is None:
            trunc = self._bufs["_truncated"] = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.float32) if same else None
        self._check(self._lib.cge_parking_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), None,
                                               fin.data_ptr() if same else None, self._stream()), "step")
        infos = self._infos()
        if same:
            infos["final_obs"] = fin
            infos["_final_obs"] = term
        return obs, rew, term, trunc, self._episode_infos(infos, term)

    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
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
            tt = self._out("terminated_traj", (k, self.num_envs), torch.bool)
        self._check(self._lib.cge_parking_rollout(self._h, k, a.data_ptr() if a is not None else None, int(action_seed), int(t0),
                                                  obs.data_ptr() if obs is not None else None, stride,
                                                  rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                  rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field, index=0):
        if field in INFO64_FIELDS:
            out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
            self._check(self._lib.cge_parking_info64(self._h, INFO64_FIELDS[field], out.data_ptr(), self._stream()), "info64")
            return out
        out = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        self._check(self._lib.cge_parking_info(self._h, INFO_FIELDS[field], int(index), out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        d = {f: self.info(f) for f in self.info_fields}
        if self._reference_info:
            d.update(self.reference_info())
        return d

    def reference_info(self):
        """The reference's `info` dict under ITS keys (parking_env.py:371-399 merged with CustomerManager.get_statistics,
        customer.py:334-354), one tensor of length N per key (zone vectors [N, 3]), computed in float64 from the env's counters
        with the reference's expressions.  `reference_info=True` in the constructor merges it into every step's / reset's infos
        (a dozen small kernels per call: for callbacks that read e.g. info["total_revenue"], not for the hot loop)."""
        f64 = torch.float64
        tc, rej, sat, tw = (self.info(k) for k in ("total_customers", "rejected", "satisfied", "total_wait_time"))
        some = tc > 0                                               # customer.py:341-347: all rates are 0.0 before the first customer
        tcf = tc.to(f64)
        t = self.info("timestep")
        occ = torch.stack([self.info("zone_occupied", z) for z in range(3)], 1)
        lvl = torch.stack([self.info("price_level", z) for z in range(3)], 1).long()
        spots = torch.tensor([15.0, 20.0, 15.0], dtype=f64, device=self.device)          # config.py:6-10
        base = torch.tensor([8.0, 5.0, 3.0], dtype=f64, device=self.device)
        mult = torch.tensor([0.7, 1.0, 1.3], dtype=f64, device=self.device)              # PRICE_LEVELS, config.py:82-86
        zero = torch.zeros_like(tcf)
        return {
            "total_customers": tc,
            "rejection_rate": torch.where(some, rej.to(f64) / tcf, zero),
            "satisfaction_rate": torch.where(some, sat.to(f64) / tcf, zero),
            "avg_wait_time": torch.where(some, tw.to(f64) / torch.clamp(tcf, min=1.0), zero),
            "hour": t // 60, "minute": t % 60, "timestep": t,                              # :378-379
            "total_revenue": self.info("episode_revenue"),
            "rejections": rej,                                                              # episode_rejections: bumped with the manager's counter (:209)
            "occupancy_rate": occ.sum(1).to(f64) / 50.0,
            "queue_length": self.info("queue_length"),
            "zone_occupancy": occ.to(f64) / spots,                                          # parking_lot.py:254-266
            "zone_prices": base * mult[lvl],                                                # pricing.py:59-64, :87-94
            "price_changes_this_hour": self.info("price_changes_this_hour"),
        }
