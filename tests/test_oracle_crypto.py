"""Pins oracle/orc_crypto.c against golden vectors produced by running the reference's own
crypto_trading_env/crypto_trading_env.py (tests/golden/gen/gen_crypto.py): float32 observations
bit-for-bit, float64 rewards and info scalars bit-for-bit, flags exact."""
import hashlib
import json

import numpy as np
import pytest

from conftest import golden


# crypto_config: the reference built with a non-default TradingConfig (gen_crypto.py; crypto_trading_env/quick_demo.py:17-24)
@pytest.mark.parametrize("name", ["crypto_discrete.npz", "crypto_continuous.npz", "crypto_config.npz"])
def test_same_step_matches_reference_bitwise(oracle, name):
    fx = golden(name)
    kind = str(fx["kind"])
    A = fx["actions"]
    n, T = A.shape[0], A.shape[1]
    config = json.loads(str(fx["config"])) if "config" in fx else None
    o = oracle.CryptoOracle(n, kind, oracle.SAME_STEP, config=config)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(int(fx["seed0"])))
    obs0 = o.reset()
    assert np.array_equal(obs0.view(np.uint32), fx["obs0"].view(np.uint32))
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    for t in range(T):
        obs, rew, te, tr, fin = o.step(A[:, t], want_final=True)
        done = te.astype(bool)
        assert np.array_equal(te, fx["terminated"][:, t]), t
        assert np.array_equal(o.last_reward64, fx["reward"][:, t]), (t, o.last_reward64, fx["reward"][:, t])
        step_obs = np.where(done[:, None], fin, obs)
        assert np.array_equal(step_obs.view(np.uint32), fx["obs"][:, t].view(np.uint32)), t
        for i in np.nonzero(done)[0]:
            assert np.array_equal(obs[i].view(np.uint32), fx["reset_obs"][reset_at[(int(i), t)]].view(np.uint32))
        live = ~done
        info = fx["info"][:, t]
        assert np.array_equal(o.info("cash")[live], info[live, 1])
        assert np.array_equal(o.info("holdings")[live], info[live, 2])
        assert np.array_equal(o.info("current_price")[live], info[live, 3])
        assert np.array_equal(o.info("regime")[live], info[live, 5])
    assert len(reset_at) == int(np.sum(fx["terminated"]))


def test_kat_c1(oracle):
    kat = golden("crypto_kat.json")
    o = oracle.CryptoOracle(1, "discrete", oracle.DISABLED)
    o.seed(np.array([42], np.uint64))
    obs = o.reset()
    assert o.info("current_price")[0] == kat["after_reset"]["close"]
    assert o.info("market_psychology")[0] == kat["after_reset"]["psych"]
    h = hashlib.sha256()
    h.update(obs.tobytes())
    total = 0.0
    for t in range(1000):
        a = oracle.hash_action(123, 0, t, 5)
        obs, rew, te, tr = o.step(np.array([a], np.int32))
        h.update(obs.tobytes()); h.update(np.float64(o.last_reward64[0]).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += float(o.last_reward64[0])
    assert obs.shape == (1, kat["obs_dim"])
    assert total == kat["sum_reward"]
    assert o.info("portfolio_value")[0] == kat["final_pv"]
    assert h.hexdigest() == kat["sha256"]


def test_numpy_reductions_restated_exactly(oracle):
    """np.mean/np.std over 14 and 20 float64 values == the oracle's pairwise sums (through the obs):
    drive the indicators with injected histories and compare with NumPy computed here."""
    rng = np.random.default_rng(0)
    n = 64
    o = oracle.CryptoOracle(n, "discrete", oracle.DISABLED)
    o.reset()
    st = o.get_state()
    off = 96 + 2 * 2496
    hist = st[:, off:].view(np.float64).reshape(n, 50, 5)
    closes = 30000 + np.cumsum(rng.normal(0, 300, (n, 50)), axis=1)
    hist[:, :, 3] = closes
    o.set_state(st)
    obs = o.reset(mask=np.zeros(n, np.uint8))          # no reset: just re-observe
    for i in range(n):
        c = closes[i]
        d = np.diff(c)
        g = np.where(d > 0, d, 0); l_ = np.where(d < 0, -d, 0)
        ag, al = np.mean(g[-14:]), np.mean(l_[-14:])
        rsi = 100.0 if al == 0 else 100 - (100 / (1 + ag / al))
        assert obs[i, 253] == np.float32(rsi / 100.0)
        sma, sd = np.mean(c[-20:]), np.std(c[-20:])
        up, lo = sma + 2 * sd, sma - 2 * sd
        assert obs[i, 257] == np.float32((c[-1] - lo) / (up - lo))
        assert obs[i, 258] == np.float32((up - lo) / sma)
        assert obs[i, 259] == np.float32((c[-1] - sma) / sma)


def test_next_step_and_rollout_consistency(oracle):
    n = 8
    a = oracle.CryptoOracle(n, "discrete", oracle.SAME_STEP)
    a.seed(np.arange(n, dtype=np.uint64) + np.uint64(9)); a.reset()
    b = oracle.CryptoOracle(n, "discrete", oracle.SAME_STEP)
    b.set_state(a.get_state())
    oa, ra, da = a.rollout(1050, 5)
    rs = np.zeros(n)
    dc = np.zeros(n, np.int32)
    for t in range(1050):
        acts = np.array([oracle.hash_action(5, i, t, 5) for i in range(n)], np.int32)
        ob, rew, te, tr = b.step(acts)
        rs += b.last_reward64
        dc += te
    assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32)) and np.array_equal(ra, rs) and np.array_equal(da, dc)
    assert dc.min() >= 1
