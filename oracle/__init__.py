"""CPU oracle for the env-stepping hot path — TEST INFRASTRUCTURE ONLY.

A plain-C restatement (oracle/orc_*.c) of the reference's step()/reset() dynamics, wrapped with
ctypes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / reported CPU baseline.  The product package
(custom_gymnasium_environments_amd) never imports it and has no CPU fallback.

Parity status: PINNED — every env restated here is checked against golden vectors produced by
executing the reference's own Python in the build container (tests/golden/gen/*.py); see
tests/test_oracle_*.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcge_oracle.so")


def build(force=False):
    srcs = [f for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH) for f in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a, dtype=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    if dtype is not None:
        assert a.dtype == np.dtype(dtype), (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def _declare(L):
    vp, i32, i64, u32, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    L.orc_mt_new.restype = vp
    L.orc_mt_free.argtypes = [vp]
    L.orc_mt_py_seed.argtypes = [vp, u64]
    L.orc_mt_np_seed.argtypes = [vp, u32]
    L.orc_mt_next_u32.argtypes = [vp]; L.orc_mt_next_u32.restype = u32
    L.orc_mt_random.argtypes = [vp]; L.orc_mt_random.restype = dbl
    L.orc_mt_randbelow.argtypes = [vp, u32]; L.orc_mt_randbelow.restype = u32
    L.orc_mt_randint.argtypes = [vp, i32, i32]; L.orc_mt_randint.restype = i32
    L.orc_mt_uniform.argtypes = [vp, dbl, dbl]; L.orc_mt_uniform.restype = dbl
    L.orc_mt_normal.argtypes = [vp, dbl, dbl]; L.orc_mt_normal.restype = dbl
    L.orc_mt_get.argtypes = [vp, vp, vp]
    L.orc_pcg_new.restype = vp
    L.orc_pcg_free.argtypes = [vp]
    L.orc_pcg_seed_export.argtypes = [vp, u64]
    L.orc_pcg_next64_export.argtypes = [vp]; L.orc_pcg_next64_export.restype = u64
    L.orc_pcg_random_export.argtypes = [vp]; L.orc_pcg_random_export.restype = dbl
    L.orc_pcg_uniform_export.argtypes = [vp, dbl, dbl]; L.orc_pcg_uniform_export.restype = dbl
    L.orc_pcg_integers_export.argtypes = [vp, i64, i64]; L.orc_pcg_integers_export.restype = i64
    L.orc_pcg_normal_export.argtypes = [vp, dbl, dbl]; L.orc_pcg_normal_export.restype = dbl
    L.orc_pcg_choice4_export.argtypes = [vp, vp]; L.orc_pcg_choice4_export.restype = i32
    L.orc_pcg_state_export.argtypes = [vp, vp]
    L.orc_hash_action_export.argtypes = [u64, u64, u64, u32, u32]; L.orc_hash_action_export.restype = u32

    L.orc_snake_create.argtypes = [i64, i32, i32]; L.orc_snake_create.restype = vp
    L.orc_snake_destroy.argtypes = [vp]
    L.orc_snake_seed.argtypes = [vp, vp]
    L.orc_snake_reset.argtypes = [vp, vp, vp]
    L.orc_snake_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]; L.orc_snake_step.restype = i32
    L.orc_snake_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_snake_info.argtypes = [vp, i32, vp]
    L.orc_snake_state_bytes.argtypes = [vp]; L.orc_snake_state_bytes.restype = C.c_size_t
    L.orc_snake_get_state.argtypes = [vp, vp]
    L.orc_snake_set_state.argtypes = [vp, vp]
    L.orc_snake_render_rgb.argtypes = [vp, vp]

    L.orc_crypto_create.argtypes = [i64, i32, i32]; L.orc_crypto_create.restype = vp
    L.orc_crypto_set_config.argtypes = [vp, vp]
    L.orc_crypto_destroy.argtypes = [vp]
    L.orc_crypto_seed.argtypes = [vp, vp]
    L.orc_crypto_reset.argtypes = [vp, vp, vp]
    L.orc_crypto_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]; L.orc_crypto_step.restype = i32
    L.orc_crypto_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_crypto_info.argtypes = [vp, i32, vp]
    L.orc_crypto_state_bytes.argtypes = []; L.orc_crypto_state_bytes.restype = C.c_size_t
    L.orc_crypto_get_state.argtypes = [vp, vp]
    L.orc_crypto_set_state.argtypes = [vp, vp]

    L.orc_traffic_create.argtypes = [i64, i32]; L.orc_traffic_create.restype = vp
    L.orc_traffic_destroy.argtypes = [vp]
    L.orc_traffic_seed.argtypes = [vp, vp]
    L.orc_traffic_reset.argtypes = [vp, vp, vp]
    L.orc_traffic_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_traffic_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_traffic_info.argtypes = [vp, i32, i32, vp]
    L.orc_traffic_total_reward.argtypes = [vp, vp]
    L.orc_traffic_state_bytes.argtypes = [vp]; L.orc_traffic_state_bytes.restype = C.c_size_t
    L.orc_traffic_set_layout.argtypes = [vp, i32, i32, i32, i32, dbl]; L.orc_traffic_set_layout.restype = i32
    L.orc_traffic_obs_dim.argtypes = [vp]; L.orc_traffic_obs_dim.restype = i32
    L.orc_traffic_num_intersections.argtypes = [vp]; L.orc_traffic_num_intersections.restype = i32
    L.orc_traffic_get_state.argtypes = [vp, vp]
    L.orc_traffic_set_state.argtypes = [vp, vp]

    L.orc_parking_create.argtypes = [i64, i32]; L.orc_parking_create.restype = vp
    L.orc_parking_destroy.argtypes = [vp]
    L.orc_parking_seed.argtypes = [vp, vp]
    L.orc_parking_reset.argtypes = [vp, vp, vp]
    L.orc_parking_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_parking_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_parking_info.argtypes = [vp, i32, i32, vp]
    L.orc_parking_info64.argtypes = [vp, i32, vp]

    L.orc_climate_create.argtypes = [i64, i32]; L.orc_climate_create.restype = vp
    L.orc_climate_destroy.argtypes = [vp]
    L.orc_climate_seed.argtypes = [vp, vp]
    L.orc_climate_reset.argtypes = [vp, vp, vp]
    L.orc_climate_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_climate_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_climate_info.argtypes = [vp, i32, vp]
    L.orc_climate_hash_action.argtypes = [u64, u64, u64, vp, vp]
    L.orc_climate_set_max_occupancy.argtypes = [vp, i32]

    L.orc_fleet_create.argtypes = [i64, i32]; L.orc_fleet_create.restype = vp
    L.orc_fleet_destroy.argtypes = [vp]
    L.orc_fleet_seed.argtypes = [vp, vp]
    L.orc_fleet_reset.argtypes = [vp, vp, vp]
    L.orc_fleet_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_fleet_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_fleet_info.argtypes = [vp, i32, vp]
    L.orc_hospital_create.argtypes = [i64, i32]; L.orc_hospital_create.restype = vp
    L.orc_hospital_destroy.argtypes = [vp]
    L.orc_hospital_seed.argtypes = [vp, vp]
    L.orc_hospital_reset.argtypes = [vp, vp, vp]
    L.orc_hospital_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_hospital_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_hospital_info.argtypes = [vp, i32, vp]
    L.orc_manufacturing_create.argtypes = [i64, i32]; L.orc_manufacturing_create.restype = vp
    L.orc_manufacturing_destroy.argtypes = [vp]
    L.orc_manufacturing_seed.argtypes = [vp, vp]
    L.orc_manufacturing_reset.argtypes = [vp, vp, vp]
    L.orc_manufacturing_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_manufacturing_rollout.argtypes = [vp, i32, u64, i64, i64, vp, vp, vp]
    L.orc_manufacturing_info.argtypes = [vp, i32, vp]
    for _nm in ("snake", "crypto", "traffic", "parking", "climate", "fleet", "hospital", "manufacturing"):
        getattr(L, f"orc_{_nm}_set_max_steps").argtypes = [vp, i32]
        getattr(L, f"orc_{_nm}_episode_stats").argtypes = [vp, vp, vp]


NEXT_STEP, SAME_STEP, DISABLED = 0, 1, 2


class _EpisodeStats:
    """(return float64[n], length int32[n]) of each env's LAST finished episode — what gymnasium's RecordEpisodeStatistics
    would report in infos["episode"] = {"r", "l"} at the step the episode ended (oracle/orc_epstats.h)."""

    def episode_stats(self):
        ret, ln = np.zeros(self.n, np.float64), np.zeros(self.n, np.int32)
        getattr(lib(), f"orc_{self._name}_episode_stats")(self.h, _p(ret), _p(ln))
        return ret, ln


class MT:
    """One MT19937 stream with CPython `random` and NumPy-legacy helpers (for RNG known answers)."""

    def __init__(self):
        self.h = lib().orc_mt_new()

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals are None at interpreter shutdown)
            lib().orc_mt_free(self.h)
            self.h = None

    def py_seed(self, s): lib().orc_mt_py_seed(self.h, s); return self
    def np_seed(self, s): lib().orc_mt_np_seed(self.h, s); return self
    def next_u32(self): return lib().orc_mt_next_u32(self.h)
    def random(self): return lib().orc_mt_random(self.h)
    def randbelow(self, n): return lib().orc_mt_randbelow(self.h, n)
    def randint(self, a, b): return lib().orc_mt_randint(self.h, a, b)
    def uniform(self, a, b): return lib().orc_mt_uniform(self.h, a, b)
    def normal(self, loc, scale): return lib().orc_mt_normal(self.h, loc, scale)

    def state(self):
        mt = np.zeros(624, np.uint32)
        idx = C.c_int(0)
        lib().orc_mt_get(self.h, _p(mt), C.addressof(idx))
        return mt, idx.value


class PCG:
    """One NumPy-Generator-compatible PCG64 stream (for RNG known answers)."""

    def __init__(self, seed):
        self.h = lib().orc_pcg_new()
        lib().orc_pcg_seed_export(self.h, seed)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals are None at interpreter shutdown)
            lib().orc_pcg_free(self.h)
            self.h = None

    def next64(self): return lib().orc_pcg_next64_export(self.h)
    def random(self): return lib().orc_pcg_random_export(self.h)
    def uniform(self, lo, hi): return lib().orc_pcg_uniform_export(self.h, lo, hi)
    def integers(self, lo, hi): return lib().orc_pcg_integers_export(self.h, lo, hi)
    def normal(self, loc, scale): return lib().orc_pcg_normal_export(self.h, loc, scale)

    def choice4(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        return lib().orc_pcg_choice4_export(self.h, _p(p))

    def state(self):
        out = np.zeros(4, np.uint64)
        lib().orc_pcg_state_export(self.h, _p(out))
        return (int(out[0]) << 64) | int(out[1]), (int(out[2]) << 64) | int(out[3])


def hash_action(a_seed, env, t, n, j=0):
    return lib().orc_hash_action_export(a_seed, env, t, n, j)


class SnakeOracle(_EpisodeStats):
    """Batch of independent SnakeEnvClassic restatements (oracle/orc_snake.c)."""
    _name = "snake"

    def __init__(self, n, grid=10, mode=SAME_STEP, max_steps=None):
        self.n, self.grid, self.mode = int(n), int(grid), int(mode)
        self.h = lib().orc_snake_create(self.n, self.grid, self.mode)
        if not self.h:
            raise ValueError("orc_snake_create failed")
        if max_steps is not None:
            lib().orc_snake_set_max_steps(self.h, int(max_steps))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals are None at interpreter shutdown)
            lib().orc_snake_destroy(self.h)
            self.h = None

    def seed(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.n,)
        lib().orc_snake_seed(self.h, _p(seeds))

    def reset(self, mask=None):
        obs = np.zeros((self.n, self.grid, self.grid), np.int8)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_snake_reset(self.h, _p(m), _p(obs))
        return obs

    def step(self, actions, want_final=False):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        obs = np.zeros((self.n, self.grid, self.grid), np.int8)
        rew = np.zeros(self.n, np.float32)
        te = np.zeros(self.n, np.uint8)
        tr = np.zeros(self.n, np.uint8)
        fin = np.zeros_like(obs) if want_final else None
        bad = lib().orc_snake_step(self.h, _p(a), _p(obs), _p(rew), _p(te), _p(tr), _p(fin))
        if bad:
            raise ValueError(f"Invalid action in {bad} env(s)")
        return (obs, rew, te, tr, fin) if want_final else (obs, rew, te, tr)

    def rollout(self, k, a_seed, t0=0, env0=0):
        obs = np.zeros((self.n, self.grid, self.grid), np.int8)
        rs = np.zeros(self.n, np.float32)
        dc = np.zeros(self.n, np.int32)
        lib().orc_snake_rollout(self.h, k, a_seed, t0, env0, _p(obs), _p(rs), _p(dc))
        return obs, rs, dc

    def info(self, field):
        out = np.zeros(self.n, np.int32)
        lib().orc_snake_info(self.h, field, _p(out))
        return out

    def render_rgb(self):
        out = np.zeros((self.n, self.grid, self.grid, 3), np.uint8)
        lib().orc_snake_render_rgb(self.h, _p(out))
        return out

    def get_state(self):
        rec = lib().orc_snake_state_bytes(self.h)
        buf = np.zeros((self.n, rec), np.uint8)
        lib().orc_snake_get_state(self.h, _p(buf))
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        assert buf.shape == (self.n, lib().orc_snake_state_bytes(self.h))
        lib().orc_snake_set_state(self.h, _p(buf))


CRYPTO_OBS = 261
CRYPTO_INFO = {"portfolio_value": 0, "cash": 1, "holdings": 2, "current_price": 3, "market_psychology": 4,
               "regime": 5, "step": 6, "trend_strength": 7, "episodes": 8, "needs_reset": 9, "cash_kind": 10}


class CryptoOracle(_EpisodeStats):
    """Batch of independent CryptoTradingEnv restatements (oracle/orc_crypto.c)."""
    _name = "crypto"

    # TradingConfig fields (crypto_trading_env.py:28-38) in the order orc_crypto_set_config takes them, with the reference defaults
    CONFIG_FIELDS = (("initial_balance", 10000.0), ("trading_fee_rate", 0.001), ("slippage_rate", 0.0005), ("min_price", 100.0),
                     ("max_price", 100000.0), ("volatility_base", 0.02), ("market_psychology_factor", 0.1))

    def __init__(self, n, action_type="discrete", mode=SAME_STEP, max_steps=None, config=None):
        self.n, self.mode = int(n), int(mode)
        self.continuous = action_type == "continuous"
        self.h = lib().orc_crypto_create(self.n, int(self.continuous), self.mode)
        if not self.h:
            raise ValueError("orc_crypto_create failed")
        if max_steps is not None:
            lib().orc_crypto_set_max_steps(self.h, int(max_steps))
        if config:
            unknown = set(config) - {k for k, _ in self.CONFIG_FIELDS}
            if unknown:
                raise ValueError(f"unknown TradingConfig field(s) {sorted(unknown)}")
            cfg = np.array([float(config.get(k, d)) for k, d in self.CONFIG_FIELDS], np.float64)
            lib().orc_crypto_set_config(self.h, _p(cfg))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals are None at interpreter shutdown)
            lib().orc_crypto_destroy(self.h)
            self.h = None

    def seed(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.n,)
        lib().orc_crypto_seed(self.h, _p(seeds))

    def reset(self, mask=None):
        obs = np.zeros((self.n, CRYPTO_OBS), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_crypto_reset(self.h, _p(m), _p(obs))
        return obs

    def step(self, actions, want_final=False):
        if self.continuous:
            a = np.ascontiguousarray(actions, dtype=np.float32)
            assert a.shape == (self.n, 2)
        else:
            a = np.ascontiguousarray(actions, dtype=np.int32)
            assert a.shape == (self.n,)
        obs = np.zeros((self.n, CRYPTO_OBS), np.float32)
        rew = np.zeros(self.n, np.float32)
        rew64 = np.zeros(self.n, np.float64)
        te = np.zeros(self.n, np.uint8)
        tr = np.zeros(self.n, np.uint8)
        fin = np.zeros_like(obs) if want_final else None
        bad = lib().orc_crypto_step(self.h, _p(a), _p(obs), _p(rew), _p(rew64), _p(te), _p(tr), _p(fin))
        if bad:
            raise ValueError(f"Invalid action in {bad} env(s)")
        self.last_reward64 = rew64
        return (obs, rew, te, tr, fin) if want_final else (obs, rew, te, tr)

    def rollout(self, k, a_seed, t0=0, env0=0, want_obs=True):
        obs = np.zeros((self.n, CRYPTO_OBS), np.float32) if want_obs else None
        rs = np.zeros(self.n, np.float64)
        dc = np.zeros(self.n, np.int32)
        lib().orc_crypto_rollout(self.h, k, a_seed, t0, env0, _p(obs), _p(rs), _p(dc))
        return obs, rs, dc

    def info(self, field):
        out = np.zeros(self.n, np.float64)
        lib().orc_crypto_info(self.h, CRYPTO_INFO[field] if isinstance(field, str) else field, _p(out))
        return out

    def get_state(self):
        buf = np.zeros((self.n, lib().orc_crypto_state_bytes()), np.uint8)
        lib().orc_crypto_get_state(self.h, _p(buf))
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        assert buf.shape == (self.n, lib().orc_crypto_state_bytes())
        lib().orc_crypto_set_state(self.h, _p(buf))


TRAFFIC_OBS = 130
TRAFFIC_INFO = {"timestep": 0, "num_vehicles": 1, "light_phase": 2, "light_timer": 3, "vehicles_passed": 4,
                "total_waiting_time": 5, "queue_len": 6, "queue_dest": 7, "queue_wait": 8, "episodes": 9, "needs_reset": 10}


class TrafficOracle(_EpisodeStats):
    """Batch of independent TrafficManagementEnv restatements (oracle/orc_traffic.c).  grid_size / num_intersections / max_vehicles /
    spawn_rate are the reference constructor's arguments (environment.py:62-83); num_intersections <= 16 here."""
    _name = "traffic"

    def __init__(self, n, mode=SAME_STEP, max_steps=None, grid_size=(5, 5), num_intersections=9, max_vehicles=50, spawn_rate=0.3):
        self.n, self.mode = int(n), int(mode)
        self.h = lib().orc_traffic_create(self.n, self.mode)
        if not self.h:
            raise ValueError("orc_traffic_create failed")
        if max_steps is not None:
            lib().orc_traffic_set_max_steps(self.h, int(max_steps))
        if lib().orc_traffic_set_layout(self.h, int(grid_size[0]), int(grid_size[1]), int(num_intersections), int(max_vehicles), float(spawn_rate)):
            raise ValueError("layout outside the oracle's capacity (num_intersections <= 16)")
        self.obs_dim = int(lib().orc_traffic_obs_dim(self.h))
        self.ni = int(lib().orc_traffic_num_intersections(self.h))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals are None at interpreter shutdown)
            lib().orc_traffic_destroy(self.h)
            self.h = None

    def seed(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.n,)
        lib().orc_traffic_seed(self.h, _p(seeds))

    def reset(self, mask=None):
        obs = np.zeros((self.n, self.obs_dim), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_traffic_reset(self.h, _p(m), _p(obs))
        return obs

    def step(self, actions, want_final=False):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.n, self.ni)
        obs = np.zeros((self.n, self.obs_dim), np.float32)
        rew = np.zeros(self.n, np.float32)
        rew64 = np.zeros(self.n, np.float64)
        te = np.zeros(self.n, np.uint8)
        tr = np.zeros(self.n, np.uint8)
        fin = np.zeros_like(obs) if want_final else None
        lib().orc_traffic_step(self.h, _p(a), _p(obs), _p(rew), _p(rew64), _p(te), _p(tr), _p(fin))
        self.last_reward64 = rew64
        return (obs, rew, te, tr, fin) if want_final else (obs, rew, te, tr)

    def rollout(self, k, a_seed, t0=0, env0=0, want_obs=True):
        obs = np.zeros((self.n, self.obs_dim), np.float32) if want_obs else None
        rs = np.zeros(self.n, np.float64)
        dc = np.zeros(self.n, np.int32)
        lib().orc_traffic_rollout(self.h, k, a_seed, t0, env0, _p(obs), _p(rs), _p(dc))
        return obs, rs, dc

    def info(self, field, idx=0):
        out = np.zeros(self.n, np.int32)
        lib().orc_traffic_info(self.h, TRAFFIC_INFO[field] if isinstance(field, str) else field, idx, _p(out))
        return out

    def total_reward(self):
        out = np.zeros(self.n, np.float64)
        lib().orc_traffic_total_reward(self.h, _p(out))
        return out

    def get_state(self):
        buf = np.zeros((self.n, lib().orc_traffic_state_bytes(self.h)), np.uint8)
        lib().orc_traffic_get_state(self.h, _p(buf))
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        assert buf.shape == (self.n, lib().orc_traffic_state_bytes(self.h))
        lib().orc_traffic_set_state(self.h, _p(buf))


PARKING_OBS = 13
PARKING_INFO = {"timestep": 0, "total_customers": 1, "rejected": 2, "satisfied": 3, "total_wait_time": 4, "queue_length": 5,
                "price_changes_this_hour": 6, "zone_occupied": 7, "price_level": 8, "episodes": 9, "needs_reset": 10}


class _SimpleOracle(_EpisodeStats):
    """Shared ctypes plumbing for the small discrete-action envs (int32 action per env, float32 obs)."""
    _name = None
    _obs = None
    _nact = None
    _adim = 1

    def __init__(self, n, mode=SAME_STEP, max_steps=None):
        self.n, self.mode = int(n), int(mode)
        self.h = getattr(lib(), f"orc_{self._name}_create")(self.n, self.mode)
        if not self.h:
            raise ValueError("create failed")
        if max_steps is not None:                          # the env type's time limit (episode_minutes, max_timesteps, ...)
            getattr(lib(), f"orc_{self._name}_set_max_steps")(self.h, int(max_steps))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:          # (module globals are None at interpreter shutdown)
            getattr(lib(), f"orc_{self._name}_destroy")(self.h)
            self.h = None

    def seed(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.n,)
        getattr(lib(), f"orc_{self._name}_seed")(self.h, _p(seeds))

    def reset(self, mask=None):
        obs = np.zeros((self.n, self._obs), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        getattr(lib(), f"orc_{self._name}_reset")(self.h, _p(m), _p(obs))
        return obs

    def step(self, actions, want_final=False):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == ((self.n,) if self._adim == 1 else (self.n, self._adim))
        obs = np.zeros((self.n, self._obs), np.float32)
        rew = np.zeros(self.n, np.float32); rew64 = np.zeros(self.n, np.float64)
        te = np.zeros(self.n, np.uint8); tr = np.zeros(self.n, np.uint8)
        fin = np.zeros_like(obs) if want_final else None
        getattr(lib(), f"orc_{self._name}_step")(self.h, _p(a), _p(obs), _p(rew), _p(rew64), _p(te), _p(tr), _p(fin))
        self.last_reward64 = rew64
        return (obs, rew, te, tr, fin) if want_final else (obs, rew, te, tr)

    def rollout(self, k, a_seed, t0=0, env0=0, want_obs=True):
        obs = np.zeros((self.n, self._obs), np.float32) if want_obs else None
        rs = np.zeros(self.n, np.float64); dc = np.zeros(self.n, np.int32)
        getattr(lib(), f"orc_{self._name}_rollout")(self.h, k, a_seed, t0, env0, _p(obs), _p(rs), _p(dc))
        return obs, rs, dc


class ParkingOracle(_SimpleOracle):
    """Batch of independent SmartParkingEnv restatements (oracle/orc_parking.c)."""
    _name, _obs, _nact = "parking", PARKING_OBS, 8

    def info(self, field, idx=0):
        out = np.zeros(self.n, np.int32)
        lib().orc_parking_info(self.h, PARKING_INFO[field], idx, _p(out))
        return out

    def info64(self, field):
        out = np.zeros(self.n, np.float64)
        lib().orc_parking_info64(self.h, {"episode_revenue": 0, "episode_satisfaction": 1}[field], _p(out))
        return out


CLIMATE_OBS = 9
CLIMATE_INFO = {"room_temp": 0, "outside_temp": 1, "ac_setting": 2, "energy_usage": 3, "total_reward": 4, "num_people": 5,
                "step": 6, "comfort_time": 7, "episodes": 8, "needs_reset": 9}


class ClimateOracle(_SimpleOracle):
    """Batch of independent SmartClimateEnv restatements (oracle/orc_climate.c).  max_occupancy / episode_minutes (= max_steps) are the
    reference constructor's arguments (smartclimate/env.py:16-28)."""
    _name, _obs = "climate", CLIMATE_OBS

    def __init__(self, n, mode=SAME_STEP, max_steps=None, max_occupancy=None):
        super().__init__(n, mode, max_steps)
        if max_occupancy is not None:
            lib().orc_climate_set_max_occupancy(self.h, int(max_occupancy))

    def step(self, ac_temp, lights, want_final=False):
        ac = np.ascontiguousarray(ac_temp, dtype=np.float32).reshape(self.n)
        li = np.ascontiguousarray(lights, dtype=np.int8)
        assert li.shape == (self.n, 4)
        obs = np.zeros((self.n, self._obs), np.float32)
        rew = np.zeros(self.n, np.float32); rew64 = np.zeros(self.n, np.float64)
        te = np.zeros(self.n, np.uint8); tr = np.zeros(self.n, np.uint8)
        fin = np.zeros_like(obs) if want_final else None
        lib().orc_climate_step(self.h, _p(ac), _p(li), _p(obs), _p(rew), _p(rew64), _p(te), _p(tr), _p(fin))
        self.last_reward64 = rew64
        return (obs, rew, te, tr, fin) if want_final else (obs, rew, te, tr)

    def info(self, field):
        out = np.zeros(self.n, np.float64)
        lib().orc_climate_info(self.h, CLIMATE_INFO[field], _p(out))
        return out

    @staticmethod
    def hash_action(a_seed, env, t):
        ac = np.zeros(1, np.float32); li = np.zeros(4, np.int8)
        lib().orc_climate_hash_action(a_seed, env, t, _p(ac), _p(li))
        return ac[0], li


FLEET_OBS = 76
FLEET_INFO = {"timestep": 0, "missed_deadlines": 1, "completed_deliveries": 2, "num_requests": 3, "weather_effect": 4,
              "total_reward": 5, "episodes": 6, "needs_reset": 7, "fuel0": 8, "fuel1": 9, "fuel2": 10}


class FleetOracle(_SimpleOracle):
    """Batch of independent FleetManagementEnv restatements (oracle/orc_fleet.c); actions int32 (n, 3)."""
    _name, _obs, _nact, _adim = "fleet", FLEET_OBS, 8, 3

    def info(self, field):
        out = np.zeros(self.n, np.float64)
        lib().orc_fleet_info(self.h, FLEET_INFO[field] if isinstance(field, str) else field, _p(out))
        return out


MANUFACTURING_OBS = 73
MANUFACTURING_INFO = {"raw_material": 0, "energy_consumption": 1, "total_reward": 2, "in_system": 3, "completed": 4, "scrapped": 5,
                      "product_ids": 6, "history_len": 7, "oee_availability": 8, "oee_performance": 9, "oee_quality": 10,
                      "timestep": 11, "episodes": 12, "needs_reset": 13, "overflow": 14}


class ManufacturingOracle(_SimpleOracle):
    """Batch of independent SmartManufacturingEnv restatements (oracle/orc_manufacturing.c); Discrete(25) actions."""
    _name, _obs, _nact = "manufacturing", MANUFACTURING_OBS, 25

    def info(self, field):
        out = np.zeros(self.n, np.float64)
        lib().orc_manufacturing_info(self.h, MANUFACTURING_INFO[field] if isinstance(field, str) else field, _p(out))
        return out


HOSPITAL_OBS = 243
HOSPITAL_INFO = {"deaths": 0, "patients_treated": 1, "total_wait_time": 2, "time": 3, "outbreak_active": 4, "mass_casualty_event": 5,
                 "next_patient_id": 6, "queue0": 7, "queue1": 8, "queue2": 9, "queue3": 10, "queue4": 11, "queue5": 12,
                 "occupied_beds": 13, "medicine_total": 14, "episodes": 15, "needs_reset": 16, "overflow": 17}


class HospitalOracle(_SimpleOracle):
    """Batch of independent HospitalManagementEnv restatements (oracle/orc_hospital.c); Discrete(35) actions."""
    _name, _obs, _nact = "hospital", HOSPITAL_OBS, 35

    def info(self, field):
        out = np.zeros(self.n, np.float64)
        lib().orc_hospital_info(self.h, HOSPITAL_INFO[field] if isinstance(field, str) else field, _p(out))
        return out
