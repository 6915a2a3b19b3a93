"""CryptoVectorEnv — batched drop-in for CryptoTradingEnv (crypto_trading_env/crypto_trading_env.py:224-561)."""
import ctypes as C

import numpy as np
import torch

from . import _native
from ._spaces import Box, Discrete, batch_space
from .vector_env import DeviceVectorEnv

INFO_FIELDS = {"portfolio_value": 0, "cash": 1, "holdings": 2, "current_price": 3, "market_psychology": 4,
               "regime": 5, "step": 6, "trend_strength": 7, "episodes": 8, "needs_reset": 9, "cash_kind": 10}
REGIME_NAMES = ("bull_run", "bear_market", "sideways", "crash", "recovery")   # MarketRegime, :20-25
OBS_DIM = 261   # actual length of _get_observation() (:505-561); the reference's declared space says 260 (:286)


class CryptoVectorEnv(DeviceVectorEnv):
    """N independent CryptoTradingEnv instances stepped by one HIP kernel launch.

    Spaces: obs float32 (261,) — the length the reference actually returns; actions `Discrete(5)`
    (0 hold, 1 buy 5 %, 2 buy 20 %, 3 sell 5 %, 4 sell 20 %) or `Box(-1, 1, (2,), float32)` when
    action_type="continuous".  reward = portfolio change at the pre-step price, -1 when no trade
    happened (:440-446), returned as float32; terminated when step >= 1000, portfolio <= 0 or
    portfolio >= 10x initial (:382-386); truncated always False.

    RNG protocol: `reset(seed=s)` gives env i both generator families the reference seeds
    (`random.seed(s+i)` and `np.random.seed(s+i)`, :305-307) as private per-env MT19937 streams;
    the market simulator state survives resets, as in the reference (:257).
    """

    _abi = "cge_crypto"
    metadata = {"render_modes": []}

    def __init__(self, num_envs, action_type="discrete", device="cuda:0", autoreset_mode="NextStep", env_index0=0,
                 config=None, max_steps=1000, reuse_buffers=False, info_fields=(), record_episode_statistics=False, reference_info=False):
        self._init_common(num_envs, device, autoreset_mode, env_index0, reuse_buffers)
        self._reference_info = bool(reference_info)
        if action_type not in ("discrete", "continuous"):
            raise ValueError("action_type must be 'discrete' or 'continuous'")
        self.action_type = action_type
        self.continuous = action_type == "continuous"
        cfg = _native.CryptoConfig()
        self._lib.cge_crypto_default_config(C.byref(cfg))
        for k, v in (config or {}).items():    # TradingConfig field names (:28-38)
            if not hasattr(cfg, k):
                raise ValueError(f"unknown or unsupported TradingConfig field {k!r} (history_length is fixed at 50 in this build)")
            setattr(cfg, k, v)
        cfg.max_steps = int(max_steps)
        cfg.action_type = int(self.continuous)
        cfg.autoreset_mode = self._mode_code
        self.single_action_space = Box(-1.0, 1.0, (2,), np.float32) if self.continuous else Discrete(5)
        self.single_observation_space = Box(-np.inf, np.inf, (OBS_DIM,), np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        for f in info_fields:
            if f not in INFO_FIELDS:
                raise ValueError(f"unknown info field {f!r}; choose from {sorted(INFO_FIELDS)}")
        self.info_fields = tuple(info_fields)
        h = C.c_void_p()
        _native.check(self._lib.cge_crypto_create(C.byref(cfg), self.num_envs, self._dev_index, self.env_index0, C.byref(h)),
                      what="cge_crypto_create")
        self._h = h
        self._obs_shape = (self.num_envs, OBS_DIM)
        self.record_episode_statistics(record_episode_statistics)

    def _actions(self, actions, k=None):
        shape = (self.num_envs, 2) if self.continuous else (self.num_envs,)
        if k is not None:
            shape = (k,) + shape
        return self._as_device(actions, torch.float32 if self.continuous else torch.int32, shape, "actions")

    def reset(self, *, seed=None, options=None):
        self._seed_native(seed)
        mask = None
        if options and options.get("reset_mask") is not None:
            mask = self._as_device(options["reset_mask"], torch.uint8, (self.num_envs,), "reset_mask")
        obs = self._out("obs", self._obs_shape, torch.float32)
        self._check(self._lib.cge_crypto_reset(self._h, mask.data_ptr() if mask is not None else None, obs.data_ptr(),
                                               self._stream()), "reset")
        return obs, self._infos()

    def step(self, actions):
        a = self._actions(actions)
        obs = self._out("obs", self._obs_shape, torch.float32)
        rew = self._out("reward", (self.num_envs,), torch.float32)
        term = self._out("terminated", (self.num_envs,), torch.bool)
        trunc = self._bufs.get("_truncated")
        if trunc is None:
            trunc = self._bufs["_truncated"] = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        same = self._mode_code == _native.AUTORESET_SAME_STEP
        fin = self._out("final_obs", self._obs_shape, torch.float32) if same else None
        self._check(self._lib.cge_crypto_step(self._h, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), None,
                                              fin.data_ptr() if same else None, self._stream()), "step")
        infos = self._infos()
        if same:
            infos["final_obs"] = fin
            infos["_final_obs"] = term
        return obs, rew, term, trunc, self._episode_infos(infos, term)

    def rollout(self, k_steps, actions=None, action_seed=0, t0=0, trajectory=False, want_obs=True, per_step=False):
        """k fused step()s in one launch; see SnakeVectorEnv.rollout.  reward_sum is float64."""
        k = int(k_steps)
        a = None if actions is None else self._actions(actions, k)
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
        self._check(self._lib.cge_crypto_rollout(self._h, k, a.data_ptr() if a is not None else None, int(action_seed), int(t0),
                                                 obs.data_ptr() if obs is not None else None, stride,
                                                 rt.data_ptr() if per_step else None, tt.data_ptr() if per_step else None,
                                                 rs.data_ptr(), dc.data_ptr(), self._stream()), "rollout")
        return (obs, rt, tt, rs, dc) if per_step else (obs, rs, dc)

    def info(self, field):
        out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        self._check(self._lib.cge_crypto_info(self._h, INFO_FIELDS[field], out.data_ptr(), self._stream()), "info")
        return out

    def _infos(self):
        d = {f: self.info(f) for f in self.info_fields}
        if self._reference_info:
            d.update(self.reference_info())
        return d

    def reference_info(self):
        """The reference's step() `info` dict under ITS keys (crypto_trading_env.py:390-398): portfolio_value, cash, holdings,
        current_price, market_psychology as float64 tensors of length N, and market_regime as int32 codes into REGIME_NAMES
        (the reference puts the enum's string there; `regime_names()` maps a host copy).  `trade_info` (the dict describing
        the step's trade) is not kept on the device and is not reproduced."""
        return {"portfolio_value": self.info("portfolio_value"), "cash": self.info("cash"), "holdings": self.info("holdings"),
                "current_price": self.info("current_price"), "market_regime": self.info("regime").to(torch.int32),
                "market_psychology": self.info("market_psychology")}

    @staticmethod
    def regime_names(codes):
        """MarketRegime values (:20-25) for the int codes in infos["market_regime"] (a host-side convenience)."""
        return np.asarray(REGIME_NAMES, dtype=object)[np.asarray(torch.as_tensor(codes).cpu(), dtype=np.int64)]

    def get_state(self):
        rec = int(self._lib.cge_crypto_state_bytes(self._h))
        buf = np.zeros((self.num_envs, rec), np.uint8)
        self._check(self._lib.cge_crypto_get_state(self._h, buf.ctypes.data, self._stream()), "get_state")
        return buf

    def set_state(self, buf):
        rec = int(self._lib.cge_crypto_state_bytes(self._h))
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.shape != (self.num_envs, rec):
            raise ValueError(f"state buffer must be uint8 {(self.num_envs, rec)}")
        self._check(self._lib.cge_crypto_set_state(self._h, buf.ctypes.data, self._stream()), "set_state")
