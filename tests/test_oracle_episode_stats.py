"""The oracle's episode statistics (oracle/orc_epstats.h) against the reference fixtures: the return it reports when an episode
ends is the float64 sum, in step order, of the rewards the REFERENCE returned during that episode; the length is its step count."""
import numpy as np
import pytest

from conftest import golden

CASES = [("SnakeOracle", "snake_g10_short.npz"), ("SnakeOracle", "snake_g10_greedy.npz"), ("CryptoOracle", "crypto_discrete.npz"),
         ("TrafficOracle", "traffic_hash.npz"), ("ParkingOracle", "parking_hash.npz"), ("ClimateOracle", "climate_hash.npz"),
         ("FleetOracle", "fleet_hash.npz"), ("ManufacturingOracle", "manufacturing_hash.npz"), ("HospitalOracle", "hospital_hash.npz")]


@pytest.mark.parametrize("oname,fixture", CASES)
def test_oracle_episode_returns_are_the_reference_reward_sums(oracle, oname, fixture):
    fx = golden(fixture)
    R = fx["reward"].astype(np.float64)
    n, T = R.shape
    T = min(T, 1600)
    if oname == "SnakeOracle":
        o = oracle.SnakeOracle(n, int(fx["grid"]), oracle.SAME_STEP, max_steps=int(fx["max_steps"]))
    elif oname == "CryptoOracle":
        o = oracle.CryptoOracle(n, str(fx["kind"]), oracle.SAME_STEP)
    else:
        o = getattr(oracle, oname)(n, oracle.SAME_STEP)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(int(fx["seed0"])))
    o.reset()
    acc, length, episodes = np.zeros(n), np.zeros(n, np.int64), 0
    for t in range(T):
        res = o.step(fx["ac_temp"][:, t], fx["lights"][:, t]) if oname == "ClimateOracle" else o.step(fx["actions"][:, t])
        for i in range(n):
            acc[i] = acc[i] + R[i, t]
        length += 1
        done = (res[2] | res[3]).astype(bool)
        if done.any():
            r, l = o.episode_stats()
            assert np.array_equal(r[done], acc[done]) and np.array_equal(l[done], length[done]), t
            episodes += int(done.sum())
            acc[done] = 0.0
            length[done] = 0
    assert episodes > 0
