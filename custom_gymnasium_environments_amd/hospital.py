"""HospitalVectorEnv — batched drop-in for HospitalManagementEnv (hospital_management_env/hospital_env.py:76-742)."""
import ctypes as C

import numpy as np

from . import _native
from ._spaces import Box, Discrete, MultiDiscrete, batch_space  # noqa: F401
from .vector_env import FlagsVectorEnv

INFO_FIELDS = {"deaths": 0, "patients_treated": 1, "total_wait_time": 2, "time": 3, "outbreak_active": 4, "mass_casualty_event": 5,
               "next_patient_id": 6, "queue0": 7, "queue1": 8, "queue2": 9, "queue3": 10, "queue4": 11, "queue5": 12,
               "occupied_beds": 13, "medicine_total": 14, "episodes": 15, "needs_reset": 16, "overflow": 17}
OBS_DIM = 243   # what _get_observation() returns (:256-321); the declared space says 295 (:157-162)


class HospitalVectorEnv(FlagsVectorEnv):
    """N independent HospitalManagementEnv instances (15 doctors, 25 nurses, 40 beds, six patient queues, equipment, medicine,
    outbreaks and mass-casualty events) stepped by one HIP kernel launch.  Actions `Discrete(35)` (:371-464), obs float32
    (243,).  Both `terminated` (:726-742) and `truncated` (current_time >= 1440) are reported; auto-reset triggers on either.
    The reference env never seeds `random` (:186), so `reset(seed=s)` here means `random.seed(s + env_index0 + i)` for env i
    followed by `reset()`; a later `reset()` continues the stream.  Bit-exact with the reference."""

    _abi = "cge_hospital"
    _obs_dim = OBS_DIM
    _action_shape = ()
    INFO_FIELDS = INFO_FIELDS
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_episode_length=1440, reuse_buffers=False,
                 info_fields=(), record_episode_statistics=False, reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        self.single_action_space = Discrete(35)
        self.single_observation_space = Box(0.0, 1.0, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        cfg = _native.HospitalConfig(int(max_episode_length), self._mode_code)
        h = C.c_void_p()
        _native.check(self._fn("create")(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)), what="cge_hospital_create")
        self._h = h
        self._finish_init(info_fields)
        self.record_episode_statistics(record_episode_statistics)

    def reference_info(self):
        """The reference's step() `info` under ITS keys (hospital_env.py:362-367): deaths, patients_treated,
        avg_wait_time = total_wait_time / max(patients_treated, 1), time.  `reference_info=True` merges it into every `infos`."""
        import torch
        treated = self.info("patients_treated")
        return {"deaths": self.info("deaths").to(torch.int64), "patients_treated": treated.to(torch.int64),
                "avg_wait_time": self.info("total_wait_time") / torch.clamp(treated, min=1.0), "time": self.info("time").to(torch.int64)}
