"""Pins oracle/orc_snake.c against golden vectors produced by running the reference's own
snake_env_classic/snake_env.py (tests/golden/gen/gen_snake.py)."""
import hashlib

import numpy as np
import pytest

from conftest import golden


def _replay(oracle, fx, mode):
    grid = int(fx["grid"])
    A = fx["actions"]
    n, T = A.shape
    o = oracle.SnakeOracle(n, grid, mode, max_steps=int(fx["max_steps"]))
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(int(fx["seed0"])))
    obs0 = o.reset()
    assert np.array_equal(obs0, fx["obs0"])
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    return o, A, n, T, reset_at


# snake_g10_short: max_steps=9, so the time limit (snake_env.py:113-114) often fires on a step that also eats (:101-104, 228 times)
@pytest.mark.parametrize("name", ["snake_g10_hash.npz", "snake_g10_greedy.npz", "snake_g20_greedy.npz", "snake_g10_short.npz", "snake_g15_greedy.npz"])
def test_same_step_autoreset_matches_reference(oracle, name):
    fx = golden(name)
    o, A, n, T, reset_at = _replay(oracle, fx, oracle.SAME_STEP)
    O, R, TE, TR = fx["obs"], fx["reward"], fx["terminated"], fx["truncated"]
    for t in range(T):
        obs, rew, te, tr, fin = o.step(A[:, t], want_final=True)
        assert np.array_equal(te, TE[:, t]) and np.array_equal(tr, TR[:, t]), t
        assert np.array_equal(rew.astype(np.float64), R[:, t]), t
        done = te.astype(bool)
        assert np.array_equal(obs[~done], O[:, t][~done]), t
        assert np.array_equal(fin[done], O[:, t][done]), t          # terminal obs
        for i in np.nonzero(done)[0]:
            assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(int(i), t)]]), (i, t)
        # info parity: score as returned by the step (terminal envs: score before reset is gone,
        # so only live envs are compared), length after the step
        assert np.array_equal(o.info(0)[~done], fx["score"][:, t][~done])
        assert np.array_equal(o.info(1)[~done], fx["length"][:, t][~done])


@pytest.mark.parametrize("name", ["snake_g10_hash.npz", "snake_g10_short.npz"])
def test_next_step_autoreset_matches_reference(oracle, name):
    fx = golden(name)
    o, A, n, T, reset_at = _replay(oracle, fx, oracle.NEXT_STEP)
    O, R, TE = fx["obs"], fx["reward"], fx["terminated"]
    # in NEXT_STEP mode an env consumes one extra step() per episode; keep a per-env cursor
    cur = np.zeros(n, np.int64)
    pending = np.zeros(n, bool)
    last_done_t = np.zeros(n, np.int64)
    for _ in range(T):
        act = A[np.arange(n), np.minimum(cur, T - 1)]
        obs, rew, te, tr = o.step(act)
        for i in range(n):
            if cur[i] >= T:
                continue
            if pending[i]:
                assert rew[i] == 0 and te[i] == 0 and tr[i] == 0
                assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(i, int(last_done_t[i]))]])
                pending[i] = False
            else:
                t = int(cur[i])
                assert np.array_equal(obs[i], O[i, t]) and rew[i] == R[i, t] and te[i] == TE[i, t]
                if te[i]:
                    pending[i] = True
                    last_done_t[i] = t
                cur[i] += 1


def test_disabled_mode_keeps_stepping_like_reference(oracle):
    # reference semantics without any reset: a wall death leaves the state untouched and the
    # same move dies again (snake_env.py:88-90)
    o = oracle.SnakeOracle(1, 10, oracle.DISABLED)
    o.seed(np.array([0], np.uint64))
    o.reset()
    for _ in range(4):
        obs, rew, te, tr = o.step(np.array([1], np.int32))
    assert te[0] == 0
    obs5, rew, te, tr = o.step(np.array([1], np.int32))
    assert te[0] == 1 and rew[0] == -10.0
    obs6, rew, te, tr = o.step(np.array([1], np.int32))
    assert te[0] == 1 and rew[0] == -10.0 and np.array_equal(obs5, obs6)


def test_kat_s1_global_stream_protocol(oracle):
    """SURVEY.md 8c KAT-S1 / BASELINE config 1: one env, random.seed(0), 10k steps, sha256 recipe."""
    kat = golden("snake_kat.json")
    o = oracle.SnakeOracle(1, 10, oracle.SAME_STEP)
    o.seed(np.array([0], np.uint64))
    obs = o.reset()
    assert (int(o.info(4)[0]), int(o.info(5)[0])) == tuple(kat["first_food"])
    acts = np.random.default_rng(123).integers(0, 4, 10000)
    h = hashlib.sha256()
    h.update(obs.tobytes())
    total, episodes = 0.0, 0
    for t, a in enumerate(acts):
        obs, rew, te, tr, fin = o.step(np.array([a], np.int32), want_final=True)
        step_obs = fin if te[0] else obs
        h.update(step_obs.tobytes()); h.update(np.float64(rew[0]).tobytes()); h.update(bytes([int(te[0]), int(tr[0])]))
        total += float(rew[0])
        if te[0]:
            episodes += 1
            h.update(obs.tobytes())
    assert total == kat["sum_reward"] and episodes == kat["episodes"]
    assert h.hexdigest() == kat["sha256"]


def test_invalid_action_raises(oracle):
    o = oracle.SnakeOracle(2, 10, oracle.SAME_STEP)
    o.reset()
    with pytest.raises(ValueError):
        o.step(np.array([0, 4], np.int32))


def test_state_roundtrip_and_rollout(oracle):
    n = 16
    a = oracle.SnakeOracle(n, 10, oracle.SAME_STEP)
    a.seed(np.arange(n, dtype=np.uint64) + np.uint64(50))
    a.reset()
    a.rollout(137, 5)
    st = a.get_state()
    b = oracle.SnakeOracle(n, 10, oracle.SAME_STEP)
    b.set_state(st)
    oa, ra, da = a.rollout(300, 5, t0=137)
    ob, rb, db = b.rollout(300, 5, t0=137)
    assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db)
    # rollout == step-by-step with the same hash actions
    c = oracle.SnakeOracle(n, 10, oracle.SAME_STEP)
    c.seed(np.arange(n, dtype=np.uint64) + np.uint64(50))
    c.reset()
    rs = np.zeros(n, np.float32)
    for t in range(437):
        acts = np.array([oracle.hash_action(5, i, t, 4) for i in range(n)], np.int32)
        obs, rew, te, tr = c.step(acts)
        if t >= 137:
            rs += rew
    assert np.array_equal(obs, oa) and np.array_equal(rs, ra)


LUT = np.array([[0, 0, 0], [0, 255, 0], [255, 0, 0]], np.uint8)       # snake_env.py:181-186: empty, snake, food


def test_render_rgb_is_the_reference_lut(oracle):
    """snake_rgb.npz holds env.render() frames of the reference in render_mode="rgb_array" next to the observations they were
    rendered from: the frame is the 3-colour look-up of the observation, and the oracle's render does exactly that."""
    fx = golden("snake_rgb.npz")
    assert np.array_equal(LUT[fx["obs"]], fx["rgb"]) and fx["rgb"].shape[1:] == (10, 10, 3)
    assert set(np.unique(fx["obs"])) == {0, 1, 2}
    o = oracle.SnakeOracle(50, 10, oracle.SAME_STEP)
    o.seed(np.arange(50, dtype=np.uint64) + np.uint64(9))
    obs = o.reset()
    assert np.array_equal(o.render_rgb(), LUT[obs])
    obs, _, _ = o.rollout(77, 4)
    assert np.array_equal(o.render_rgb(), LUT[obs])
