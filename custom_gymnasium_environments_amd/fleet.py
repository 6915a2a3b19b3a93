"""FleetVectorEnv — batched drop-in for FleetManagementEnv (fleet_management_env/fleet_env.py:101-608)."""
import ctypes as C

import numpy as np

from . import _native
from ._spaces import Box, Discrete, MultiDiscrete, batch_space  # noqa: F401
from .vector_env import FlagsVectorEnv

INFO_FIELDS = {"timestep": 0, "missed_deadlines": 1, "completed_deliveries": 2, "num_requests": 3, "weather_effect": 4,
               "total_reward": 5, "episodes": 6, "needs_reset": 7, "fuel0": 8, "fuel1": 9, "fuel2": 10}
OBS_DIM = 76   # what _get_observation() returns (:555-593); the declared space says 87 (:151-154)


class FleetVectorEnv(FlagsVectorEnv):
    """N independent FleetManagementEnv instances (3 vehicles on a 25x25 grid, 8-12 deliveries) stepped on the GPU.  Actions
    `MultiDiscrete([8]*3)` (0 stay, 1-4 move, 5 pick up, 6 drop off, 7 refuel), obs float32 (76,).  Both `terminated`
    (:537-553) and `truncated` (timestep >= 800) are reported; auto-reset triggers on either.  `reset(seed=s)` seeds env i's
    NumPy-legacy and CPython streams with s + env_index0 + i (:187-189).  Bit-exact with the reference."""

    _abi = "cge_fleet"
    _obs_dim = OBS_DIM
    _action_shape = (3,)
    INFO_FIELDS = INFO_FIELDS
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, max_timesteps=800, reuse_buffers=False,
                 info_fields=(), record_episode_statistics=False, reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        self.single_action_space = MultiDiscrete([8, 8, 8])
        self.single_observation_space = Box(-1.0, 25.0, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        cfg = _native.FleetConfig(int(max_timesteps), self._mode_code)
        h = C.c_void_p()
        _native.check(self._fn("create")(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)), what="cge_fleet_create")
        self._h = h
        self._finish_init(info_fields)
        self.record_episode_statistics(record_episode_statistics)

    def reference_info(self):
        """The reference's `_get_info()` dict under ITS keys (fleet_env.py:595-608): timestep, active_deliveries (requests not yet
        completed: a request is marked completed where completed_deliveries is counted, :405,436), completed_deliveries,
        vehicles_with_fuel (fuel > 0), weather_effect, total_reward, missed_deadlines.  Seven small info kernels: for callbacks, not
        the hot loop.  `reference_info=True` merges it into every `infos`."""
        import torch
        done, fuel = self.info("completed_deliveries"), torch.stack([self.info(f"fuel{k}") for k in range(3)], 1)
        return {"timestep": self.info("timestep").to(torch.int64), "active_deliveries": (self.info("num_requests") - done).to(torch.int64),
                "completed_deliveries": done.to(torch.int64), "vehicles_with_fuel": (fuel > 0).sum(1),
                "weather_effect": self.info("weather_effect"), "total_reward": self.info("total_reward"),
                "missed_deadlines": self.info("missed_deadlines").to(torch.int64)}
