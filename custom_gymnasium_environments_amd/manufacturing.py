"""ManufacturingVectorEnv — batched drop-in for SmartManufacturingEnv (smart_manufacturing_env/manufacturing_env.py:69-606)."""
import ctypes as C

import numpy as np

from . import _native
from ._spaces import Box, Discrete, MultiDiscrete, batch_space  # noqa: F401
from .vector_env import FlagsVectorEnv

INFO_FIELDS = {"raw_material": 0, "energy_consumption": 1, "total_reward": 2, "in_system": 3, "completed": 4, "scrapped": 5,
               "product_ids": 6, "history_len": 7, "oee_availability": 8, "oee_performance": 9, "oee_quality": 10,
               "timestep": 11, "episodes": 12, "needs_reset": 13, "overflow": 14,
               "completed_A": 15, "completed_B": 16, "completed_C": 17, "completed_D": 18, "completed_E": 19, "completed_F": 20}
OBS_DIM = 73


class ManufacturingVectorEnv(FlagsVectorEnv):
    """N independent SmartManufacturingEnv instances (5 stations, 6 product types, quality checkpoints, machine breakdowns,
    supply disruptions) stepped by one HIP kernel launch.  Actions `Discrete(25)` (:303-359), obs float32 (73,).  Both
    `terminated` (:555-578) and `truncated` (timestep >= 1500) are reported; auto-reset triggers on either.  `reset(seed=s)`
    gives env i gymnasium's `np_random` for seed s + env_index0 + i (PCG64 from SeedSequence, :115); a later `reset()`
    continues the stream.  Bit-exact with the reference, including the NumPy pairwise-summed per-type quality means."""

    _abi = "cge_manufacturing"
    _obs_dim = OBS_DIM
    _action_shape = ()
    INFO_FIELDS = INFO_FIELDS
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_steps=1500, reuse_buffers=False,
                 info_fields=(), record_episode_statistics=False, reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        self.single_action_space = Discrete(25)
        self.single_observation_space = Box(0.0, 500.0, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        cfg = _native.ManufacturingConfig(int(max_steps), self._mode_code)
        h = C.c_void_p()
        _native.check(self._fn("create")(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)), what="cge_manufacturing_create")
        self._h = h
        self._finish_init(info_fields)
        self.record_episode_statistics(record_episode_statistics)

    def reference_info(self):
        """The reference's `info` under ITS keys (manufacturing_env.py:293-299, reset: :184-190): timestep, total_reward,
        products_completed (the per-type dict, :157,485), oee {availability, performance, quality} (:533-549), energy_consumption.
        `reference_info=True` merges it into every `infos`."""
        import torch
        return {"timestep": self.info("timestep").to(torch.int64), "total_reward": self.info("total_reward"),
                "products_completed": {k: self.info(f"completed_{k}").to(torch.int64) for k in "ABCDEF"},
                "oee": {"availability": self.info("oee_availability"), "performance": self.info("oee_performance"), "quality": self.info("oee_quality")},
                "energy_consumption": self.info("energy_consumption")}
