"""TrafficVectorEnv — batched drop-in for TrafficManagementEnv (traffic_management_env/environment.py:31-384)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, MultiDiscrete, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"timestep": 0, "num_vehicles": 1, "light_phase": 2, "light_timer": 3, "vehicles_passed": 4,
               "total_waiting_time": 5, "queue_len": 6, "queue_dest": 7, "queue_wait": 8, "episodes": 9, "needs_reset": 10}
LIGHT_PHASES = ("NS_GREEN", "NS_YELLOW", "EW_GREEN", "EW_YELLOW")   # config.py:17
OBS_DIM = 130                    # default layout (9 intersections); an instance's own width is `env.obs_dim`
SUPPORTED_INTERSECTIONS = tuple(range(2, 17))   # 4, 9, 16 (what the reference's scripts build) have kernels of their own


class TrafficVectorEnv(DeviceVectorEnv):
    """N independent TrafficManagementEnv instances (default 5x5 grid, 9 controlled intersections) stepped by
    one HIP kernel launch.

    Constructor arguments as the reference's (environment.py:62-83): `grid_size=(rows, cols)`, `num_intersections` (the env
    controls NI = min(num_intersections, rows*cols) of them), `max_vehicles`, `spawn_rate`.  Any NI from 2 to 16 over any grid; what
    the reference's own scripts build — simple_test.py:71-76 ((3,3), 4, 20, 0.4), the default ((5,5), 9, 50, 0.3),
    USAGE_EXAMPLES.md:32-38 ((6,6), 16, 80, 0.5) — runs on kernels specialised for 4 / 9 / 16.  (NI = 1 raises in the reference at
    the first spawn: randint(2, 1), utils.py:181.)
    Spaces as the reference (:108-130): actions `MultiDiscrete([3]*NI)` (0 maintain, 1 switch to NS_GREEN, 2 switch to
    EW_GREEN), obs `Box(0, inf, (14*NI + 4,), float32)`; reward is the reference's cumulative
    expression (:287-311) returned as float32; terminated when timestep >= 1000; truncated always False.
    Bit-exact with the reference (integer state, float64 reward, float32 obs).  `reset(seed=s)` gives env i
    the private stream `random.seed(s + env_index0 + i)` (:145-146).
    """

    _abi = "cge_traffic"
    metadata = {"render_modes": []}

    def __init__(self, num_envs, device="cuda:0", autoreset_mode="NextStep", env_index0=0, grid_size=(5, 5),
                 num_intersections=9, max_vehicles=50, spawn_rate=0.3, max_steps=1000, reuse_buffers=False, info_fields=(), record_episode_statistics=False,
                 reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        cfg = _native.TrafficConfig()
        self._lib.cge_traffic_default_config(C.byref(cfg))
        cfg.grid_rows, cfg.grid_cols = int(grid_size[0]), int(grid_size[1])
        cfg.num_intersections = int(num_intersections)
        cfg.max_vehicles, cfg.spawn_rate, cfg.max_steps = int(max_vehicles), float(spawn_rate), int(max_steps)
        cfg.autoreset_mode = self._mode_code
        self.grid_size = (int(grid_size[0]), int(grid_size[1]))
        self.num_intersections = min(int(num_intersections), self.grid_size[0] * self.grid_size[1])       # environment.py:79
        self.max_vehicles, self.spawn_rate, self.max_steps = int(max_vehicles), float(spawn_rate), int(max_steps)
        self.obs_dim = 14 * self.num_intersections + 4                                                   # :108-121
        self.single_action_space = MultiDiscrete([3] * self.num_intersections)
        self.single_observation_space = Box(0.0, np.inf, (self.obs_dim,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.info_fields = tuple(info_fields)
        h = C.c_void_p()
        st = self._lib.cge_traffic_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h))
        if st == -3:
            raise ValueError(f"num_intersections={self.num_intersections} is not supported: the device record spreads an env over 16 "
                             f"(slot, lane) places and the reference needs at least 2 (supported: 2..16)")
        if st == -1:
            raise ValueError("TrafficVectorEnv: grid_size within 1..64 per side, 0 <= max_vehicles <= 127, max_steps <= 65535, "
                             "max_vehicles * max_steps <= 262143 and spawn_rate >= 0 are required")
        _native.check(st, what="cge_traffic_create")
        self._h = h
        self._obs_shape = (self.num_envs, self.obs_dim)
        self.record_episode_statistics(record_episode_statistics)

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._lib.cge_traffic_reset(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(),
                                                self._stream()), "reset")
        return obs, self._infos()

    def step(self, actions):
        a = self._as_device(actions, torch.int32, (self.num_envs, self.num_intersections), "actions")
        obs = self._out("obs", self._obs_shape, torch.float32)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        trunc = self._bufs.get("_truncated")
        if trunc is None:
            trunc = self._bufs["_truncated"] = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.float32) if same else None
        self._check(self._lib.cge_traffic_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), None,
                                               fin.data_ptr() if same else None, self._stream()), "step")
        infos = self._infos()
        if same:
            infos["final_obs"] = fin
            infos["_final_obs"] = term
        return obs, rew, term, trunc, self._episode_infos(infos, term)

    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
        """k fused step()s in one launch; see SnakeVectorEnv.rollout.  reward_sum is float64."""
        k = int(k_steps)
        a = None if actions is None else self._as_device(actions, torch.int32, (k, self.num_envs, self.num_intersections), "actions")
        obs, stride = None, 0
        if want_obs:
            if trajectory:
                obs = self._out("traj", (k,) + self._obs_shape, torch.float32)
                stride = self.num_envs * self.obs_dim
            else:
                obs = self._out("obs", self._obs_shape, torch.float32)
        rs = self._out("reward_sum", (self.num_envs,), torch.float64)
        dc = self._out("done_count", (self.num_envs,), torch.int32)
        rt = tt = None
        if per_step:
            rt = self._out("reward_traj", (k, self.num_envs), torch.float32)
            tt = self._out("terminated_traj", (k, self.num_envs), torch.bool)
        self._final_obs_begin()
        self._check(self._lib.cge_traffic_rollout(self._h, k, a.data_ptr() if a is not None else None, int(action_seed), int(t0),
                                                  obs.data_ptr() if obs is not None else None, stride,
                                                  rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                  rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field, index=0):
        out = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        self._check(self._lib.cge_traffic_info(self._h, INFO_FIELDS[field], int(index), out.data_ptr(), self._stream()), "info")
        return out

    def total_reward(self):
        out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        self._check(self._lib.cge_traffic_total_reward(self._h, out.data_ptr(), self._stream()), "total_reward")
        return out

    def _infos(self):
        d = {f: self.info(f) for f in self.info_fields}
        if self._reference_info:
            d.update(self.reference_info())
        return d

    def reference_info(self):
        """The reference's `_get_info()` dict under ITS keys (environment.py:365-384): timestep, num_vehicles, total_reward,
        `metrics` (calculate_traffic_metrics, utils.py:251-267) and `intersection_states` — here one dict of tensors instead of a list
        of per-intersection dicts: light_phase [N, NI] (index into LIGHT_PHASES), queue_lengths [N, NI, 4] (NORTH, EAST, SOUTH,
        WEST), vehicles_passed, total_waiting_time [N, NI].  Built from ~7 NI small info kernels: for callbacks, not the hot loop."""
        ni, f64 = self.num_intersections, torch.float64
        passed = torch.stack([self.info("vehicles_passed", k) for k in range(ni)], 1)
        wait = torch.stack([self.info("total_waiting_time", k) for k in range(ni)], 1)
        qlen = torch.stack([self.info("queue_len", q) for q in range(4 * ni)], 1).reshape(self.num_envs, ni, 4)
        phase = torch.stack([self.info("light_phase", k) for k in range(ni)], 1)
        tp, tw, tq = passed.sum(1).to(f64), wait.sum(1).to(f64), qlen.sum((1, 2)).to(f64)
        metrics = {"total_vehicles_passed": passed.sum(1), "total_waiting_time": wait.sum(1),
                   "average_waiting_time": tw / torch.clamp(tp, min=1.0), "total_queue_length": qlen.sum((1, 2)),
                   "average_queue_length": tq / ni, "throughput": tp / ni}
        return {"timestep": self.info("timestep"), "num_vehicles": self.info("num_vehicles"), "total_reward": self.total_reward(),
                "metrics": metrics,
                "intersection_states": {"light_phase": phase, "queue_lengths": qlen, "vehicles_passed": passed, "total_waiting_time": wait}}

    def get_state(self):
        rec = int(self._lib.cge_traffic_state_bytes(self._h))
        buf = np.zeros((self.num_envs, rec), np.uint8)
        self._check(self._lib.cge_traffic_get_state(self._h, buf.ctypes.data, self._stream()), "get_state")
        return buf

    def set_state(self, buf):
        rec = int(self._lib.cge_traffic_state_bytes(self._h))
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.shape != (self.num_envs, rec):
            raise ValueError(f"state buffer must be uint8 {(self.num_envs, rec)}")
        self._check(self._lib.cge_traffic_set_state(self._h, buf.ctypes.data, self._stream()), "set_state")
