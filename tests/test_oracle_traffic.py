"""Pins oracle/orc_traffic.c (collapsed-state restatement) against golden vectors produced by running the
reference's own traffic_management_env (tests/golden/gen/gen_traffic.py): float32 obs bit-for-bit,
float64 rewards bit-for-bit, flags and every observable internal counter exact."""
import hashlib
import json

import numpy as np
import pytest

from conftest import golden


# traffic_3x3 / traffic_6x6: the reference constructed the way its own scripts do (simple_test.py:71-76, USAGE_EXAMPLES.md:32-38):
# other grid, number of intersections (obs 60 / 228 wide), vehicle cap and spawn rate
@pytest.mark.parametrize("name", ["traffic_hash.npz", "traffic_lazy.npz", "traffic_3x3.npz", "traffic_6x6.npz"])
def test_same_step_matches_reference_bitwise(oracle, name):
    fx = golden(name)
    A = fx["actions"]
    n, T, ni = A.shape[0], A.shape[1], A.shape[2]
    ctor = json.loads(str(fx["ctor"])) if "ctor" in fx else {}
    o = oracle.TrafficOracle(n, oracle.SAME_STEP, **ctor)
    assert o.ni == ni and o.obs_dim == fx["obs"].shape[2]
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
        S = fx["internal"][:, t]
        live = ~done
        for it in range(ni):
            assert np.array_equal(o.info("light_phase", it)[live], S[live, it * 8 + 0])
            assert np.array_equal(o.info("light_timer", it)[live], S[live, it * 8 + 1])
            assert np.array_equal(o.info("vehicles_passed", it)[live], S[live, it * 8 + 2])
            assert np.array_equal(o.info("total_waiting_time", it)[live], S[live, it * 8 + 3])
            for d in range(4):
                assert np.array_equal(o.info("queue_len", it * 4 + d)[live], S[live, it * 8 + 4 + d])
        assert np.array_equal(o.info("num_vehicles")[live], S[live, 8 * ni])
    assert len(reset_at) == n


def test_kat_t1(oracle):
    kat = golden("traffic_kat.json")
    o = oracle.TrafficOracle(1, oracle.DISABLED)
    o.seed(np.array([42], np.uint64))
    obs = o.reset()
    h = hashlib.sha256()
    h.update(obs.tobytes())
    total = 0.0
    for t in range(1000):
        a = np.array([[oracle.hash_action(123, 0, t, 3, j) for j in range(9)]], np.int32)
        obs, rew, te, tr = o.step(a)
        r = float(o.last_reward64[0])
        h.update(obs.tobytes()); h.update(np.float64(r).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += r
        if str(t + 1) in kat["rewards"]:
            assert r == kat["rewards"][str(t + 1)]
    assert total == kat["sum_reward"] and int(o.info("num_vehicles")[0]) == kat["num_vehicles"]
    assert h.hexdigest() == kat["sha256"]


def test_rollout_and_state_roundtrip(oracle):
    n = 16
    a = oracle.TrafficOracle(n, oracle.SAME_STEP)
    a.seed(np.arange(n, dtype=np.uint64) + np.uint64(9)); a.reset()
    a.rollout(333, 5)
    b = oracle.TrafficOracle(n, oracle.SAME_STEP)
    b.set_state(a.get_state())
    oa, ra, da = a.rollout(800, 5, t0=333)
    rs = np.zeros(n); dc = np.zeros(n, np.int32)
    for t in range(333, 1133):
        acts = np.array([[oracle.hash_action(5, i, t, 3, j) for j in range(9)] for i in range(n)], np.int32)
        ob, rew, te, tr = b.step(acts)
        rs += b.last_reward64; dc += te
    assert np.array_equal(oa, ob) and np.array_equal(ra, rs) and np.array_equal(da, dc) and dc.min() == 1
