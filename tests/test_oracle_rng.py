"""Pins oracle/orc_rng.h against CPython `random` and NumPy legacy RandomState executed here,
plus the literal known answers recorded in SURVEY.md section 8c."""
import random

import numpy as np


def test_known_answers_literal(oracle):
    m = oracle.MT().py_seed(0)
    assert [m.randint(0, 9) for _ in range(6)] == [6, 6, 0, 4, 8, 7]
    m = oracle.MT().py_seed(42)
    assert m.random() == 0.6394267984578837
    assert m.uniform(0.5, 2.0) == 0.5375161328340003
    m = oracle.MT().np_seed(42)
    got = [m.normal(0, 0.016) for _ in range(3)]
    assert got == [0.007947426448179723, -0.0022122288187389543, 0.01036301660961108]


def test_py_seed_state_matches_cpython(oracle):
    for seed in [0, 1, 42, 2**31, 2**32 - 1, 2**32, 2**40 + 12345, 2**63 + 5]:
        random.seed(seed)
        st = random.getstate()[1]
        mt, idx = oracle.MT().py_seed(seed).state()
        assert idx == st[-1] == 624
        assert np.array_equal(mt, np.array(st[:-1], dtype=np.uint32)), seed


def test_draw_streams_match_cpython(oracle):
    for seed in [3, 99991, 2**33 + 7]:
        random.seed(seed)
        m = oracle.MT().py_seed(seed)
        for _ in range(1500):  # crosses two regenerations
            assert m.next_u32() == random.getrandbits(32)
        for n in [2, 3, 4, 9, 10, 20, 26, 144, 1000]:
            for _ in range(200):
                assert m.randbelow(n) == random.randrange(n)
        for _ in range(500):
            assert m.random() == random.random()
            assert m.uniform(0.98, 1.0) == random.uniform(0.98, 1.0)
            assert m.randint(5, 30) == random.randint(5, 30)


def test_numpy_legacy_normal(oracle):
    for seed in [0, 7, 42, 2**32 - 1]:
        np.random.seed(seed)
        m = oracle.MT().np_seed(seed)
        ref = np.array([np.random.normal(0.0, 0.024) for _ in range(2001)])
        got = np.array([m.normal(0.0, 0.024) for _ in range(2001)])
        assert np.array_equal(ref, got)
        assert m.random() == np.random.random_sample()  # cache + stream position agree


def test_hash_action_matches_generator_definition(oracle):
    sys_path_hack = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "gen")
    import sys
    sys.path.insert(0, sys_path_hack)
    try:
        import common
    finally:
        sys.path.pop(0)
    for a_seed, env, t, n, j in [(123, 0, 0, 4, 0), (123, 1048575, 999, 4, 0), (7, 77, 12345, 3, 8), (2**63, 5, 6, 5, 0)]:
        assert oracle.hash_action(a_seed, env, t, n, j) == common.hash_action(a_seed, env, t, n, j)
    ids = np.arange(1000)
    v = common.hash_actions_np(123, ids, 17, 4)
    assert all(int(v[i]) == common.hash_action(123, int(i), 17, 4) for i in range(0, 1000, 37))
    assert 0.2 < np.mean(v == 0) < 0.3


def test_pcg64_seedsequence_and_draws_match_numpy(oracle):
    for seed in [0, 1, 7, 42, 2**31, 2**32 + 5, 2**63 + 11]:
        g = np.random.default_rng(seed)
        st = g.bit_generator.state["state"]
        p = oracle.PCG(seed)
        assert p.state() == (st["state"], st["inc"]), seed
        raw = np.random.PCG64(seed).random_raw(50)
        q = oracle.PCG(seed)
        assert [q.next64() for _ in range(50)] == [int(x) for x in raw]
        # mixed call sequence exactly as SmartClimateEnv issues it (uniform, integers, normal, choice)
        for _ in range(300):
            assert p.uniform(22.0, 26.0) == g.uniform(22.0, 26.0)
            assert p.integers(0, 9) == int(g.integers(0, 9))
            assert p.normal(25, 5) == g.normal(25, 5)
            assert [-1, 0, 1, 2][p.choice4([0.1, 0.3, 0.4, 0.2])] == int(g.choice([-1, 0, 1, 2], p=[0.1, 0.3, 0.4, 0.2]))
            assert [-2, -1, 0, 1][p.choice4([0.2, 0.4, 0.3, 0.1])] == int(g.choice([-2, -1, 0, 1], p=[0.2, 0.4, 0.3, 0.1]))
            assert p.random() == g.random()


def test_pcg64_known_answers_literal(oracle):
    p = oracle.PCG(0)                                   # default_rng(0)
    assert p.uniform(22, 26) == 24.547846749285817
    assert p.integers(0, 9) == 4
    assert p.normal(25, 5) == 28.20211325221641
    assert [-2, -1, 0, 1][p.choice4([.2, .4, .3, .1])] == -2


def test_ziggurat_normal_bulk(oracle):
    g = np.random.default_rng(99)
    ref = g.standard_normal(200000)
    p = oracle.PCG(99)
    got = np.array([p.normal(0.0, 1.0) for _ in range(200000)])
    assert np.array_equal(ref, got)
