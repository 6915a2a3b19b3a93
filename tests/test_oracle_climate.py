"""Pins oracle/orc_climate.c against golden vectors produced by running the reference's own smartclimate
package (tests/golden/gen/gen_climate.py): float32 obs bit-for-bit, float64 rewards/state bit-for-bit."""
import hashlib
import json

import numpy as np
import pytest

from conftest import golden


# climate_small: the reference constructed with max_occupancy=3, episode_minutes=300 (smartclimate/env.py:16-28)
@pytest.mark.parametrize("name", ["climate_hash.npz", "climate_small.npz"])
def test_same_step_matches_reference_bitwise(oracle, name):
    fx = golden(name)
    AC, LI = fx["ac_temp"], fx["lights"]
    n, T = AC.shape
    ctor = json.loads(str(fx["ctor"])) if "ctor" in fx else {}
    o = oracle.ClimateOracle(n, oracle.SAME_STEP, max_steps=ctor.get("episode_minutes"), max_occupancy=ctor.get("max_occupancy"))
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(int(fx["seed0"])))
    assert np.array_equal(o.reset().view(np.uint32), fx["obs0"].view(np.uint32))
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    for t in range(T):
        obs, rew, te, tr, fin = o.step(AC[:, t], LI[:, t], want_final=True)
        done = te.astype(bool)
        assert np.array_equal(te, fx["terminated"][:, t]), t
        assert np.array_equal(o.last_reward64, fx["reward"][:, t]), (t, o.last_reward64, fx["reward"][:, t])
        step_obs = np.where(done[:, None], fin, obs)
        assert np.array_equal(step_obs.view(np.uint32), fx["obs"][:, t].view(np.uint32)), t
        for i in np.nonzero(done)[0]:
            assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(int(i), t)]])
        live = ~done
        S = fx["state"][:, t]
        assert np.array_equal(o.info("room_temp")[live], S[live, 0]) and np.array_equal(o.info("outside_temp")[live], S[live, 1])
        assert np.array_equal(o.info("energy_usage")[live], S[live, 2]) and np.array_equal(o.info("comfort_time")[live], S[live, 3])
        assert np.array_equal(o.info("total_reward")[live], S[live, 4])
    assert len(reset_at) == int(fx["terminated"].sum()) >= 2 * n
    if ctor:
        assert fx["obs"][:, :, 1].max() == ctor["max_occupancy"]          # the occupancy clip is reached


def test_kat_k1(oracle):
    kat = golden("climate_kat.json")
    o = oracle.ClimateOracle(1, oracle.SAME_STEP)
    o.seed(np.array([7], np.uint64))
    obs = o.reset()
    g = np.random.default_rng(7)
    ac = g.uniform(16, 32, (3000, 1)).astype(np.float32)
    li = g.integers(0, 2, (3000, 4)).astype(np.int8)
    h = hashlib.sha256(); h.update(obs.tobytes())
    total, episodes = 0.0, 0
    for t in range(3000):
        obs, rew, te, tr, fin = o.step(ac[t], li[t][None, :], want_final=True)
        step_obs = fin if te[0] else obs
        r = float(o.last_reward64[0])
        h.update(step_obs.tobytes()); h.update(np.float64(r).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += r
        if te[0]:
            episodes += 1
            h.update(obs.tobytes())
    assert total == kat["sum_reward"] and episodes == kat["episodes"] and h.hexdigest() == kat["sha256"]


def test_hash_actions_match_generator(oracle):
    fx = golden("climate_hash.npz")
    for i, t in [(0, 1), (3, 500), (7, 2999)]:
        ac, li = oracle.ClimateOracle.hash_action(int(fx["a_seed"]), i, t)
        if t % 7:
            assert ac == fx["ac_temp"][i, t]
        assert np.array_equal(li, fx["lights"][i, t])
