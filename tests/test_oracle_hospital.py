"""Pins oracle/orc_hospital.c against golden vectors produced by running the reference's own hospital_env.py
(tests/golden/gen/gen_hospital.py): float32 obs (243) bit-for-bit, integer rewards exact, terminated/truncated exact,
counters and queue lengths exact."""
import hashlib

import numpy as np
import pytest

from conftest import golden

STATE = ["deaths", "patients_treated", "total_wait_time", "time", "outbreak_active", "mass_casualty_event", "next_patient_id",
         "queue0", "queue1", "queue2", "queue3", "queue4", "queue5", "occupied_beds", "medicine_total"]


@pytest.mark.parametrize("name", ["hospital_hash.npz", "hospital_surge.npz"])
def test_same_step_matches_reference_bitwise(oracle, name):
    fx = golden(name)
    A = fx["actions"]
    n, T = A.shape[0], A.shape[1]
    o = oracle.HospitalOracle(n, oracle.SAME_STEP)
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
        if t % 16 == 0:
            live = ~done
            for k, f in enumerate(STATE):
                assert np.array_equal(o.info(f)[live], fx["state"][live, t, k]), (t, f)
    assert len(reset_at) >= 4 and o.info("overflow").sum() == 0


def test_kat_h1(oracle):
    kat = golden("hospital_kat.json")
    o = oracle.HospitalOracle(1, oracle.SAME_STEP)
    o.seed(np.array([7], np.uint64))
    obs = o.reset()
    acts = np.random.default_rng(7).integers(0, 35, 3000)
    h = hashlib.sha256(); h.update(obs.tobytes())
    total, episodes = 0.0, 0
    for a in acts:
        obs, rew, te, tr, fin = o.step(np.array([a], np.int32), want_final=True)
        done = bool(te[0] or tr[0])
        step_obs = fin if done else obs
        r = float(o.last_reward64[0])
        h.update(step_obs.tobytes()); h.update(np.float64(r).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += r
        if done:
            episodes += 1
            h.update(obs.tobytes())
    assert total == kat["sum_reward"] and episodes == kat["episodes"] and h.hexdigest() == kat["sha256"]


def test_next_step_and_rollout_agree_with_step(oracle):
    n, K = 5, 600
    a = oracle.HospitalOracle(n, oracle.NEXT_STEP); b = oracle.HospitalOracle(n, oracle.NEXT_STEP)
    seeds = np.arange(n, dtype=np.uint64) + np.uint64(40)
    a.seed(seeds); b.seed(seeds); a.reset(); b.reset()
    rs = np.zeros(n)
    for t in range(K):
        acts = np.array([oracle.hash_action(9, i, t, 35, 0) for i in range(n)], np.int32)
        obs, rew, te, tr = a.step(acts)
        rs += a.last_reward64
    obs_b, rs_b, dc_b = b.rollout(K, 9)
    assert np.array_equal(obs.view(np.uint32), obs_b.view(np.uint32)) and np.array_equal(rs, rs_b)
