"""Golden vectors for CryptoTradingEnv, produced by running the reference's own
crypto_trading_env/crypto_trading_env.py (unmodified, imported from /root/reference).

Protocol: env i is a fresh CryptoTradingEnv run alone after
`reset(seed=S+i)` (which seeds BOTH the global `random` and the NumPy legacy generator,
crypto_trading_env.py:305-307); auto-reset = `env.reset()` without a seed right after a terminal step
(streams and the MarketSimulator state continue, :257).  Actions: counter hash mod 5 (discrete) or
hashed floats in [-1,1] (continuous).
Outputs: tests/golden/crypto_discrete.npz, crypto_continuous.npz, crypto_config.npz, crypto_kat.json
crypto_config: the reference constructed with a non-default TradingConfig (every field the batched env's constructor forwards),
the way crypto_trading_env/quick_demo.py:17-24 builds it.
"""
import json
import os

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("crypto_trading_env")
import crypto_trading_env as ref  # noqa: E402  (reference code)

REGIMES = ["bull_run", "bear_market", "sideways", "crash", "recovery"]   # MarketRegime order, :20-25


def cont_action(a_seed, i, t):
    u0 = common.action_hash(a_seed, i, t, 0) >> 40          # 24 bits
    u1 = common.action_hash(a_seed, i, t, 1) >> 40
    return np.array([u0 / 2.0**23 - 1.0, u1 / 2.0**23 - 1.0], dtype=np.float32)


def run_env(kind, seed, T, a_seed, i, config=None):
    env = ref.CryptoTradingEnv(action_type=kind, config=ref.TradingConfig(**config) if config else None)
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    D = obs.shape[0]
    O = np.zeros((T, D), np.float32)
    R = np.zeros(T, np.float64)
    TE = np.zeros(T, np.uint8)
    A = np.zeros((T, 2), np.float32) if kind == "continuous" else np.zeros(T, np.int32)
    INFO = np.zeros((T, 6), np.float64)   # pv, cash, holdings, price, psych, regime index
    resets = []
    for t in range(T):
        a = cont_action(a_seed, i, t) if kind == "continuous" else common.hash_action(a_seed, i, t, 5)
        obs, rew, term, trunc, info = env.step(a)
        assert not trunc
        A[t] = a
        O[t] = obs
        R[t] = rew
        TE[t] = term
        INFO[t] = [info["portfolio_value"], info["cash"], info["holdings"], info["current_price"],
                   info["market_psychology"], REGIMES.index(info["market_regime"])]
        if term:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, INFO, resets


def make(name, kind, n_envs, T, seed0, a_seed, config=None):
    rows = [run_env(kind, seed0 + i, T, a_seed, i, config) for i in range(n_envs)]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[6]:
            ridx.append((i, t))
            robs.append(ob)
    D = rows[0][0].shape[0]
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, kind=np.array(kind), seed0=np.int64(seed0), a_seed=np.int64(a_seed),
        obs0=np.stack([r[0] for r in rows]), actions=np.stack([r[1] for r in rows]),
        obs=np.stack([r[2] for r in rows]), reward=np.stack([r[3] for r in rows]),
        terminated=np.stack([r[4] for r in rows]), info=np.stack([r[5] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, D),
        versions=np.array(json.dumps(common.versions())), **({"config": np.array(json.dumps(config))} if config else {}))
    R = np.stack([r[3] for r in rows])
    print(name, "obs dim", D, "episodes", len(ridx), "sum reward", R.sum(), os.path.getsize(out), "bytes")


def kat_c1():
    """SURVEY 8c KAT-C1: reset(seed=42), discrete, 1000 steps, actions hash(123, 0, t) mod 5."""
    env = ref.CryptoTradingEnv(action_type="discrete")
    obs, _ = env.reset(seed=42)
    h = common.RunningHash()
    h.obs(obs)
    total = 0.0
    first = dict(close=float(env.price_history[-1][3]), psych=float(env.market_sim.market_psychology))
    for t in range(1000):
        a = common.hash_action(123, 0, t, 5)
        obs, r, te, tr, info = env.step(a)
        h.step(obs, r, te, tr)
        total += r
    kat = dict(after_reset=first, sum_reward=total, final_pv=float(info["portfolio_value"]), sha256=h.hexdigest(),
               obs_dim=int(obs.shape[0]), **common.versions())
    with open(os.path.join(common.GOLDEN, "crypto_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-C1", kat)


if __name__ == "__main__":
    kat_c1()
    make("crypto_discrete", "discrete", 6, 1100, seed0=100, a_seed=123)
    make("crypto_continuous", "continuous", 3, 400, seed0=500, a_seed=77)
    # a small balance and a wide, fast market: episodes end on the 10x / <= 0 portfolio bounds and both price clips are reached
    make("crypto_config", "discrete", 4, 1100, seed0=900, a_seed=31,
         config=dict(initial_balance=2500.0, trading_fee_rate=0.002, slippage_rate=0.001, min_price=2000.0, max_price=60000.0,
                     volatility_base=0.035, market_psychology_factor=0.25))
