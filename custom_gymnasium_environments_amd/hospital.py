"""HospitalVectorEnv — batched drop-in for HospitalManagementEnv (hospital_management_env/hospital_env.py:76-742)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, Discrete, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"deaths": 0, "patients_treated": 1, "total_wait_time": 2, "time": 3, "outbreak_active": 4, "mass_casualty_event": 5,
               "next_patient_id": 6, "queue0": 7, "queue1": 8, "queue2": 9, "queue3": 10, "queue4": 11, "queue5": 12,
               "occupied_beds": 13, "medicine_total": 14, "episodes": 15, "needs_reset": 16, "overflow": 17}
OBS_DIM = 243   # what _get_observation() returns (:256-321); the declared space says 295 (:157-162)


class HospitalVectorEnv(DeviceVectorEnv):
    """N independent HospitalManagementEnv instances (15 doctors, 25 nurses, 40 beds, six patient queues, equipment,
    medicine, outbreaks and mass-casualty events) stepped by one HIP kernel launch.  Actions `Discrete(35)` (:371-464), obs
    float32 (243,).  Both `terminated` (:726-742) and `truncated` (current_time >= 1440) are reported; auto-reset triggers on
    either.  The reference env never seeds `random` (:186), so `reset(seed=s)` here means `random.seed(s + env_index0 + i)`
    for env i followed by `reset()`; a later `reset()` continues the stream.  Bit-exact with the reference."""

    _abi = "cge_hospital"
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_episode_length=1440, reuse_buffers=False,
                 info_fields=()):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self.single_action_space = Discrete(35)
        self.single_observation_space = Box(0.0, 1.0, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.info_fields = tuple(info_fields)
        cfg = _native.HospitalConfig(int(max_episode_length), self._mode_code)
        h = C.c_void_p()
        _native.check(self._lib.cge_hospital_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)),
                      what="cge_hospital_create")
        self._h = h
        self._obs_shape = (self.num_envs, OBS_DIM)

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._lib.cge_hospital_reset(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(),
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
        self._check(self._lib.cge_hospital_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(),
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
        self._check(self._lib.cge_hospital_rollout(self._h, k, a.data_ptr() if a is not None else None, int(action_seed), int(t0),
                                                obs.data_ptr() if obs is not None else None, stride,
                                                rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field):
        out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        self._check(self._lib.cge_hospital_info(self._h, INFO_FIELDS[field], out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        return {f: self.info(f) for f in self.info_fields}
