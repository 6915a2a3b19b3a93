"""Pins oracle/orc_parking.c against golden vectors produced by running the reference's own
smart_parking_env/core (tests/golden/gen/gen_parking.py): float32 obs bit-for-bit, float64 rewards and
revenue/satisfaction accumulators bit-for-bit, flags and counters exact."""
import hashlib

import numpy as np
import pytest

from conftest import golden


@pytest.mark.parametrize("name", ["parking_hash.npz", "parking_busy.npz"])
def test_same_step_matches_reference_bitwise(oracle, name):
    fx = golden(name)
    A = fx["actions"]
    n, T = A.shape
    o = oracle.ParkingOracle(n, oracle.SAME_STEP)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(int(fx["seed0"])))
    assert np.array_equal(o.reset().view(np.uint32), fx["obs0"].view(np.uint32))
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    for t in range(T):
        obs, rew, te, tr, fin = o.step(A[:, t], want_final=True)
        done = te.astype(bool)
        assert np.array_equal(te, fx["terminated"][:, t]), t
        assert np.array_equal(o.last_reward64, fx["reward"][:, t]), (t, o.last_reward64, fx["reward"][:, t])
        step_obs = np.where(done[:, None], fin, obs)
        assert np.array_equal(step_obs.view(np.uint32), fx["obs"][:, t].view(np.uint32)), t
        for i in np.nonzero(done)[0]:
            assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(int(i), t)]])
        S, live = fx["stats"][:, t], ~done
        for col, f in enumerate(["total_customers", "rejected", "satisfied", "total_wait_time", "queue_length",
                                 "price_changes_this_hour", "timestep"]):
            assert np.array_equal(o.info(f)[live], S[live, col]), (t, f)
        for z in range(3):
            assert np.array_equal(o.info("zone_occupied", z)[live], S[live, 7 + z])
            assert np.array_equal(o.info("price_level", z)[live], S[live, 10 + z])
        assert np.array_equal(o.info64("episode_revenue")[live], fx["money"][live, t, 0])
        assert np.array_equal(o.info64("episode_satisfaction")[live], fx["money"][live, t, 1])


def test_kat_p1(oracle):
    kat = golden("parking_kat.json")
    o = oracle.ParkingOracle(1, oracle.SAME_STEP)
    o.seed(np.array([7], np.uint64))
    obs = o.reset()
    acts = np.random.default_rng(7).integers(0, 8, 3000)
    h = hashlib.sha256(); h.update(obs.tobytes())
    total, episodes = 0.0, 0
    for a in acts:
        obs, rew, te, tr, fin = o.step(np.array([a], np.int32), want_final=True)
        step_obs = fin if te[0] else obs
        r = float(o.last_reward64[0])
        h.update(step_obs.tobytes()); h.update(np.float64(r).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += r
        if te[0]:
            episodes += 1
            h.update(obs.tobytes())
    assert total == kat["sum_reward"] and episodes == kat["episodes"] and h.hexdigest() == kat["sha256"]
