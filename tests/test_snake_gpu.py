"""GPU parity tests for the snake hot path: every call goes through the C ABI (libcge_amd.so) via
SnakeVectorEnv and is compared bit-for-bit with (a) the golden vectors recorded from the reference
and (b) the CPU oracle on the same seeds and action streams."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cge():
    import custom_gymnasium_environments_amd as m
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    m.native_lib()
    return m


def _np(t):
    return t.cpu().numpy()


# snake_g10_short: max_steps=9 — the time limit often fires on a step that also eats (two food placements in one SameStep step)
@pytest.mark.parametrize("name", ["snake_g10_hash.npz", "snake_g10_greedy.npz", "snake_g20_greedy.npz", "snake_g10_short.npz", "snake_g15_greedy.npz"])
def test_same_step_matches_reference_fixture(cge, name):
    fx = golden(name)
    grid = int(fx["grid"])
    A = fx["actions"]
    n, T = A.shape
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", max_steps=int(fx["max_steps"]))
    obs, _ = env.reset(seed=int(fx["seed0"]))
    assert np.array_equal(_np(obs), fx["obs0"])
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    A_dev = torch.from_numpy(A).cuda()
    for t in range(T):
        obs, rew, te, tr, info = env.step(A_dev[:, t])
        obs, rew, te, tr, fin = _np(obs), _np(rew), _np(te), _np(tr), _np(info["final_obs"])
        assert np.array_equal(te, fx["terminated"][:, t].astype(bool)), t
        assert not tr.any()
        assert np.array_equal(rew.astype(np.float64), fx["reward"][:, t]), t
        assert np.array_equal(obs[~te], fx["obs"][:, t][~te]), t
        assert np.array_equal(fin[te], fx["obs"][:, t][te]), t
        for i in np.nonzero(te)[0]:
            assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(int(i), t)]]), (i, t)
        if t % 97 == 0:
            live = ~te
            assert np.array_equal(_np(env.info("score"))[live], fx["score"][:, t][live])
            assert np.array_equal(_np(env.info("snake_length"))[live], fx["length"][:, t][live])
    env.close()


@pytest.mark.parametrize("mode", ["NextStep", "SameStep", "Disabled"])
# every grid_size the library compiles (4..30): even and odd, below and above the 144-cell switch to 64-thread workgroups, grids whose
# obs row is narrower than a digit ring (4, 5), the reference scripts' 15 (test_visualization.py:17) and the largest
@pytest.mark.parametrize("grid,n", [(10, 1000), (6, 77), (16, 130), (20, 65), (15, 200), (4, 70), (5, 129), (7, 300), (13, 66), (23, 65), (30, 70)])
def test_step_matches_oracle_all_modes(cge, oracle, mode, grid, n):
    code = {"NextStep": oracle.NEXT_STEP, "SameStep": oracle.SAME_STEP, "Disabled": oracle.DISABLED}[mode]
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode=mode, env_index0=5)
    o = oracle.SnakeOracle(n, grid, code)
    seeds = np.arange(n, dtype=np.uint64) + np.uint64(5 + 900)
    o.seed(seeds)
    obs_d, _ = env.reset(seed=900)
    assert np.array_equal(_np(obs_d), o.reset())
    rng = np.random.default_rng(1)
    for t in range(300):
        a = rng.integers(0, 4, n).astype(np.int32)
        od, rd, ted, trd, _ = env.step(a)
        oo, ro, teo, tro = o.step(a)
        assert np.array_equal(_np(od), oo), t
        assert np.array_equal(_np(rd), ro) and np.array_equal(_np(ted), teo.astype(bool)) and not _np(trd).any()
    for f, k in [("score", 0), ("snake_length", 1), ("steps", 2), ("direction", 3), ("food_r", 4), ("food_c", 5),
                 ("episodes", 7), ("head_r", 8), ("head_c", 9), ("needs_reset", 10)]:
        assert np.array_equal(_np(env.info(f)), o.info(k)), f
    env.close()


def test_explicit_per_env_seeds_including_64bit(cge, oracle):
    n = 300
    seeds = np.random.default_rng(3).integers(0, 2**63, n, dtype=np.uint64)
    seeds[:4] = [0, 2**32 - 1, 2**32, 2**40 + 17]
    env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="SameStep")
    o = oracle.SnakeOracle(n, 10, oracle.SAME_STEP)
    o.seed(seeds)
    obs, _ = env.reset(seed=seeds)
    assert np.array_equal(_np(obs), o.reset())
    _, rs, dc = env.rollout(500, action_seed=11)
    oo, ro, do = o.rollout(500, 11)
    assert np.array_equal(_np(rs), ro) and np.array_equal(_np(dc), do)
    env.close()


def test_rollout_matches_oracle_and_step_path(cge, oracle):
    n, grid = 4096 + 13, 10
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", env_index0=1000)
    o = oracle.SnakeOracle(n, grid, oracle.SAME_STEP)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(1000 + 42))
    env.reset(seed=42)
    o.reset()
    # hash actions, last-obs mode, two consecutive launches (t0 continues)
    obs, rs, dc = env.rollout(600, action_seed=123, t0=0)
    oo, ro, do = o.rollout(600, 123, t0=0, env0=1000)
    assert np.array_equal(_np(obs), oo) and np.array_equal(_np(rs), ro) and np.array_equal(_np(dc), do)
    obs, rs, dc = env.rollout(400, action_seed=123, t0=600)
    oo, ro, do = o.rollout(400, 123, t0=600, env0=1000)
    assert np.array_equal(_np(obs), oo) and np.array_equal(_np(rs), ro) and np.array_equal(_np(dc), do)
    # trajectory mode with explicit actions == step-by-step on a twin env
    twin = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", env_index0=1000)
    twin.set_state(env.get_state())
    acts = torch.randint(0, 4, (50, n), dtype=torch.int32, device="cuda")
    traj, rt, tt, rs, dc = env.rollout(50, actions=acts, trajectory=True, per_step=True)
    rsum = torch.zeros(n, device="cuda")
    for t in range(50):
        ob, r, te, tr, _ = twin.step(acts[t])
        assert torch.equal(ob, traj[t]), t
        assert torch.equal(r, rt[t]) and torch.equal(te, tt[t]), t
        rsum += r
    assert torch.equal(rsum, rs)
    env.close(); twin.close()


@pytest.mark.parametrize("grid,shrink", [(6, 1), (10, 12), (10, 1), (12, 9)])
def test_state_injection_long_snakes_and_full_board(cge, oracle, grid, shrink):
    """Rare branches: food placement among long bodies (heavy rejection), tail-cell collision rule,
    and the board-full guard (reference would spin forever, snake_env.py:123).  On the 10x10 grid the lengths 100, 88, ..., 16
    also cover every storage class of the device record: hot column only (<= 29 cells), one cold column, both (>= 94)."""
    n = 64
    o = oracle.SnakeOracle(n, grid, oracle.DISABLED)
    o.seed(np.arange(n, dtype=np.uint64))
    o.reset()
    st = o.get_state()
    rec = st.shape[1]
    hdr = st[:, :32].view(np.int32)
    body = st[:, 32 + 624 * 4: 32 + 624 * 4 + grid * grid * 2].view(np.uint16)
    # boustrophedon path covering the whole board; env i gets a snake of length G*G - (i % 8) * shrink
    path = []
    for r in range(grid):
        cols = range(grid) if r % 2 == 0 else range(grid - 1, -1, -1)
        path += [r * grid + c for c in cols]
    for i in range(n):
        L = grid * grid - (i % 8) * shrink
        cells = path[:L][::-1]                      # head = last cell of the prefix
        body[i, :] = 0xFFFF
        body[i, :L] = cells
        hdr[i, 0] = L
        free = [c for c in range(grid * grid) if c not in cells]
        hdr[i, 1] = 1
        if free:
            hdr[i, 2], hdr[i, 3] = free[0] // grid, free[0] % grid
        else:
            hdr[i, 2] = hdr[i, 3] = -1
        hdr[i, 4] = L - 1
    o.set_state(st)
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="Disabled")
    env.set_state(st)
    back = env.get_state()
    assert np.array_equal(back[:, :28], st[:, :28])            # header (mt_idx representation may differ)
    assert np.array_equal(back[:, 32 + 624 * 4:], st[:, 32 + 624 * 4:])
    rng = np.random.default_rng(0)
    for t in range(60):
        a = rng.integers(0, 4, n).astype(np.int32)
        od, rd, ted, _, _ = env.step(a)
        oo, ro, teo, _ = o.step(a)
        assert np.array_equal(_np(od), oo), t
        assert np.array_equal(_np(rd), ro) and np.array_equal(_np(ted), teo.astype(bool))
    assert np.array_equal(_np(env.info("board_full")), o.info(6))
    assert np.array_equal(_np(env.info("snake_length")), o.info(1))
    back, ref = env.get_state(), o.get_state()                        # bodies after 60 steps, head first
    assert np.array_equal(back[:, 32 + 624 * 4:], ref[:, 32 + 624 * 4:]) and np.array_equal(back[:, :28], ref[:, :28])
    # the fused kernel loads / stores the same records
    obs, rs, dc = env.rollout(40, action_seed=3)
    oo, ro, do = o.rollout(40, 3)
    assert np.array_equal(_np(obs), oo) and np.array_equal(_np(rs), ro) and np.array_equal(_np(dc), do)
    assert np.array_equal(env.get_state()[:, 32 + 624 * 4:], o.get_state()[:, 32 + 624 * 4:])
    env.close()


def test_set_state_rejects_bodies_that_are_not_paths(cge):
    env = cge.SnakeVectorEnv(4, grid_size=10)
    env.reset(seed=1)
    good = env.get_state()
    for cells in ([55, 57], [55, 55], [59, 60], [50, 49, 48, 38, 39, 49]):          # gap, repeat, row wrap, self-crossing
        bad = good.copy()
        bad[1, :32].view(np.int32)[0] = len(cells)
        body = bad[1, 32 + 624 * 4:32 + 624 * 4 + 200].view(np.uint16)
        body[:] = 0xFFFF
        body[:len(cells)] = cells
        with pytest.raises(cge.NativeLibraryError):
            env.set_state(bad)
    env.close()


def test_get_state_roundtrip_through_oracle(cge, oracle):
    n = 128
    env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="SameStep")
    env.reset(seed=77)
    env.rollout(777, action_seed=5)
    o = oracle.SnakeOracle(n, 10, oracle.SAME_STEP)
    o.set_state(env.get_state())
    obs, rs, dc = env.rollout(500, action_seed=5, t0=777)
    oo, ro, do = o.rollout(500, 5, t0=777)
    assert np.array_equal(_np(obs), oo) and np.array_equal(_np(rs), ro) and np.array_equal(_np(dc), do)
    env.close()


def test_invalid_actions_are_counted_not_applied(cge):
    env = cge.SnakeVectorEnv(100, grid_size=10, autoreset_mode="SameStep")
    obs0, _ = env.reset(seed=1)
    obs0 = obs0.clone()
    a = torch.full((100,), 7, dtype=torch.int32, device="cuda")
    a[::2] = -1
    obs, rew, te, tr, _ = env.step(a)
    assert torch.equal(obs, obs0) and not rew.any() and not te.any()
    with pytest.raises(ValueError):
        env.check_actions()
    assert env.invalid_action_count() == 0
    env.close()


def test_partial_reset_mask(cge, oracle):
    n = 513
    env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="Disabled")
    o = oracle.SnakeOracle(n, 10, oracle.DISABLED)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(9))
    env.reset(seed=9); o.reset()
    env.rollout(20, action_seed=2); o.rollout(20, 2)
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    obs, _ = env.reset(options={"reset_mask": mask})
    assert np.array_equal(_np(obs), o.reset(mask))
    env.close()


def test_million_env_config_properties_and_sampled_parity(cge, oracle):
    """BASELINE config 2: SnakeEnv 10x10, 1,048,576 envs.  Full-size checks through size-independent
    properties plus bit-exact parity with the oracle on slices at both ends and the middle."""
    n, grid, T = 1 << 20, 10, 200
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", reuse_buffers=True)
    obs, _ = env.reset(seed=0)
    assert int((obs == 1).sum()) == n and int((obs == 2).sum()) == n      # one body cell, one food per env
    obs, rs, dc = env.rollout(T, action_seed=123)
    assert int((obs == 2).sum()) == n
    length = env.info("snake_length")
    assert torch.equal((obs == 1).sum(dim=(1, 2)).to(torch.int32), length)
    assert torch.equal(env.info("score") + 1, length)
    # reward accounting: every done costs -10 (no env survives 1000 steps in 200), every food +10
    assert bool((env.info("steps") < 1000).all())
    for lo in [0, n // 2 - 1000, n - 2048]:
        m = 2048
        o = oracle.SnakeOracle(m, grid, oracle.SAME_STEP)
        o.seed(np.arange(lo, lo + m, dtype=np.uint64))
        o.reset()
        oo, ro, do = o.rollout(T, 123, t0=0, env0=lo)
        assert np.array_equal(_np(obs[lo:lo + m]), oo)
        assert np.array_equal(_np(rs[lo:lo + m]), ro) and np.array_equal(_np(dc[lo:lo + m]), do)
    # sharding invariance: a second handle that owns only the upper half reproduces the same rows
    half = cge.SnakeVectorEnv(n // 2, grid_size=grid, autoreset_mode="SameStep", env_index0=n // 2)
    half.reset(seed=0)
    oh, rh, dh = half.rollout(T, action_seed=123)
    assert torch.equal(oh, obs[n // 2:]) and torch.equal(rh, rs[n // 2:]) and torch.equal(dh, dc[n // 2:])
    env.close(); half.close()


@pytest.mark.parametrize("mode", ["NextStep", "SameStep", "Disabled"])
@pytest.mark.parametrize("grid,n", [(6, 200), (8, 333), (12, 257), (16, 130), (20, 191)])
def test_rollout_trajectory_equals_stepping_the_oracle(cge, oracle, mode, grid, n):
    """The fused rollout (writer wave, incremental LDS obs rows, cooperative food placement) for every supported grid:
    each step's obs / reward / terminated of a [K, N, G, G] trajectory equals the oracle stepped with the same actions."""
    code = {"NextStep": oracle.NEXT_STEP, "SameStep": oracle.SAME_STEP, "Disabled": oracle.DISABLED}[mode]
    K = 120
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode=mode, env_index0=3)
    o = oracle.SnakeOracle(n, grid, code)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(3 + 17))
    env.reset(seed=17); o.reset()
    acts = np.random.default_rng(grid).integers(0, 4, (K, n)).astype(np.int32)
    obs, rt, tt, rs, dc = env.rollout(K, actions=torch.from_numpy(acts).cuda(), trajectory=True, per_step=True)
    obs, rt, tt = _np(obs), _np(rt), _np(tt)
    for t in range(K):
        oo, ro, teo, tro = o.step(acts[t])
        assert np.array_equal(obs[t], oo), (t, np.argwhere(obs[t] != oo)[:5])
        assert np.array_equal(rt[t], ro) and np.array_equal(tt[t], teo.astype(bool)), t
    # and once more without observations: same rewards, state carried on
    o2 = oracle.SnakeOracle(n, grid, code)
    o2.seed(np.arange(n, dtype=np.uint64) + np.uint64(3 + 17)); o2.reset()
    env.reset(seed=17)
    _, rs2, dc2 = env.rollout(K, actions=torch.from_numpy(acts).cuda(), want_obs=False)
    assert np.array_equal(_np(rs2), _np(rs)) and np.array_equal(_np(dc2), _np(dc))
    env.close()


def test_render_rgb_matches_reference_lut_and_oracle(cge, oracle):
    """rgb_array rendering (snake_env.py:175-188) for the whole batch: the device frames equal the reference's look-up table
    (pinned by tests/golden/snake_rgb.npz, see test_oracle_snake.py) applied to the device observation, and the oracle's frames."""
    lut = np.array([[0, 0, 0], [0, 255, 0], [255, 0, 0]], np.uint8)
    fx = golden("snake_rgb.npz")
    assert np.array_equal(lut[fx["obs"]], fx["rgb"])
    for grid, n in [(10, 1000), (6, 70), (20, 129)]:
        env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", render_mode="rgb_array", env_index0=3)
        o = oracle.SnakeOracle(n, grid, oracle.SAME_STEP)
        o.seed(np.arange(n, dtype=np.uint64) + np.uint64(3 + 5))
        obs, _ = env.reset(seed=5); o.reset()
        assert np.array_equal(_np(env.render_rgb()), lut[_np(obs)])
        obs, _, _ = env.rollout(150, action_seed=8); o.rollout(150, 8, env0=3)
        rgb = _np(env.render_rgb())
        assert rgb.shape == (n, grid, grid, 3) and rgb.dtype == np.uint8
        assert np.array_equal(rgb, lut[_np(obs)]) and np.array_equal(rgb, o.render_rgb())
        frames = env.render()
        assert len(frames) == n and tuple(frames[0].shape) == (grid, grid, 3)
        env.close()
    assert cge.SnakeVectorEnv(4, grid_size=10).render() is None
    with pytest.raises(ValueError):
        cge.SnakeVectorEnv(4, grid_size=10, render_mode="human")


def test_set_state_rejects_malformed_records(cge, oracle):
    """cge_snake_set_state validates every header field it packs into the bit-fields of the device record (a food cell or step
    count out of range would spill into neighbouring fields or index outside the env's LDS row)."""
    n = 8
    env = cge.SnakeVectorEnv(n, grid_size=10)
    env.reset(seed=1)
    good = env.get_state()
    env.set_state(good)                                               # round trip is accepted
    for field, value in [(2, 10), (3, -2), (2, 99), (4, 101), (4, -1), (5, 70000), (5, -3), (6, 2), (0, 0), (1, 4), (7, 625)]:
        bad = good.copy()
        bad[3].view(np.int32)[field] = value
        with pytest.raises(cge.NativeLibraryError):
            env.set_state(bad)
    env.close()


@pytest.mark.parametrize("grid", [15, 5, 9, 27])
def test_odd_and_unusual_grids_rollout_render_and_state(cge, oracle, grid):
    """Odd grids keep padded LDS rows and stream them out through a realigning copy: the fused rollout (trajectory and last-obs
    forms, ragged last wave), rgb_array and the state round trip against the oracle."""
    n = 64 * 3 + 11
    env = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", env_index0=3)
    o = oracle.SnakeOracle(n, grid, oracle.SAME_STEP)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(3 + 8))
    obs, _ = env.reset(seed=8)
    assert np.array_equal(_np(obs), o.reset())
    obs, rs, dc = env.rollout(300, action_seed=5, t0=0)
    oo, ro, do = o.rollout(300, 5, t0=0, env0=3)
    assert np.array_equal(_np(obs), oo) and np.array_equal(_np(rs), ro) and np.array_equal(_np(dc), do)
    rgb = _np(env.render_rgb())
    exp = np.zeros((n, grid, grid, 3), np.uint8)
    exp[oo == 1] = (0, 255, 0); exp[oo == 2] = (255, 0, 0)                  # snake_env.py:175-188
    assert np.array_equal(rgb, exp)
    twin = cge.SnakeVectorEnv(n, grid_size=grid, autoreset_mode="SameStep", env_index0=3)
    twin.set_state(env.get_state())
    o2 = oracle.SnakeOracle(n, grid, oracle.SAME_STEP)
    o2.set_state(env.get_state())
    acts = torch.randint(0, 4, (40, n), dtype=torch.int32, device="cuda")
    traj, rt, tt, rs, dc = env.rollout(40, actions=acts, trajectory=True, per_step=True)
    for t in range(40):
        ob, r, te, _, info = twin.step(acts[t])
        oo, ro, teo, _, fo = o2.step(_np(acts[t]), want_final=True)
        assert torch.equal(ob, traj[t]) and torch.equal(r, rt[t]) and torch.equal(te, tt[t]), t
        assert np.array_equal(_np(ob), oo) and np.array_equal(_np(info["final_obs"])[teo.astype(bool)], fo[teo.astype(bool)]), t
    env.close(); twin.close()


def test_grid_sizes_outside_4_to_30_are_refused(cge):
    for g in (3, 31, 64):
        with pytest.raises(ValueError, match="grid_size"):
            cge.SnakeVectorEnv(4, grid_size=g)
