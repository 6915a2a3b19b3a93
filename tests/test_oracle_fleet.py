"""Pins oracle/orc_fleet.c against golden vectors produced by running the reference's own fleet_env.py
(tests/golden/gen/gen_fleet.py): float32 obs bit-for-bit, rewards exact, terminated/truncated exact, fuel float64 exact."""
import hashlib

import numpy as np
import pytest

from conftest import golden


@pytest.mark.parametrize("name", ["fleet_hash.npz", "fleet_courier.npz"])
def test_same_step_matches_reference_bitwise(oracle, name):
    fx = golden(name)
    A = fx["actions"]
    n, T = A.shape[0], A.shape[1]
    o = oracle.FleetOracle(n, oracle.SAME_STEP)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(int(fx["seed0"])))
    assert np.array_equal(o.reset().view(np.uint32), fx["obs0"].view(np.uint32))
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    for t in range(T):
        obs, rew, te, tr, fin = o.step(A[:, t], want_final=True)
        done = (te | tr).astype(bool)
        assert np.array_equal(te, fx["terminated"][:, t]) and np.array_equal(tr, fx["truncated"][:, t]), t
        assert np.array_equal(o.last_reward64, fx["reward"][:, t]), (t, o.last_reward64, fx["reward"][:, t])
        step_obs = np.where(done[:, None], fin, obs)
        assert np.array_equal(step_obs.view(np.uint32), fx["obs"][:, t].view(np.uint32)), (t, np.argwhere(step_obs != fx["obs"][:, t])[:5])
        for i in np.nonzero(done)[0]:
            assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(int(i), t)]]), (i, t)
        live = ~done
        S = fx["internal"][:, t]
        assert np.array_equal(o.info("timestep")[live], S[live, 12]) and np.array_equal(o.info("missed_deadlines")[live], S[live, 13])
        assert np.array_equal(o.info("completed_deliveries")[live], S[live, 14]) and np.array_equal(o.info("num_requests")[live], S[live, 15])
        for k in range(3):
            assert np.array_equal(o.info(f"fuel{k}")[live], fx["fuel"][live, t, k])
        assert np.array_equal(o.info("weather_effect")[live], fx["fuel"][live, t, 3])
    assert len(reset_at) > n


def test_kat_f1(oracle):
    kat = golden("fleet_kat.json")
    o = oracle.FleetOracle(1, oracle.SAME_STEP)
    o.seed(np.array([7], np.uint64))
    obs = o.reset()
    acts = np.random.default_rng(7).integers(0, 8, (3000, 3))
    h = hashlib.sha256(); h.update(obs.tobytes())
    total, episodes = 0.0, 0
    for a in acts:
        obs, rew, te, tr, fin = o.step(a[None, :].astype(np.int32), want_final=True)
        done = bool(te[0] or tr[0])
        step_obs = fin if done else obs
        r = float(o.last_reward64[0])
        h.update(step_obs.tobytes()); h.update(np.float64(r).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += r
        if done:
            episodes += 1
            h.update(obs.tobytes())
    assert total == kat["sum_reward"] and episodes == kat["episodes"] and h.hexdigest() == kat["sha256"]
