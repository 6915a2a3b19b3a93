"""GPU parity tests for the hospital hot path (through the C ABI via HospitalVectorEnv): bit-exact obs (243 f32), integer
rewards, terminated/truncated flags and counters against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

STATE = ["deaths", "patients_treated", "total_wait_time", "time", "outbreak_active", "mass_casualty_event", "next_patient_id",
         "queue0", "queue1", "queue2", "queue3", "queue4", "queue5", "occupied_beds", "medicine_total"]


@pytest.fixture(scope="module")
def cge():
    import custom_gymnasium_environments_amd as m
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    m.native_lib()
    return m


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("name", ["hospital_hash.npz", "hospital_surge.npz"])
def test_same_step_matches_reference_fixture(cge, name):
    fx = golden(name)
    A = fx["actions"]
    n, T = A.shape[0], A.shape[1]
    env = cge.HospitalVectorEnv(n, autoreset_mode="SameStep")
    obs, _ = env.reset(seed=int(fx["seed0"]))
    assert np.array_equal(_np(obs), fx["obs0"])
    reset_at = {(int(i), int(t)): k for k, (i, t) in enumerate(fx["reset_index"])}
    A_dev = torch.from_numpy(A).cuda()
    for t in range(T):
        obs, rew, te, tr, info = env.step(A_dev[:, t])
        obs, rew, te, tr, fin = _np(obs), _np(rew), _np(te), _np(tr), _np(info["final_obs"])
        assert np.array_equal(te, fx["terminated"][:, t].astype(bool)) and np.array_equal(tr, fx["truncated"][:, t].astype(bool)), t
        assert np.array_equal(rew, fx["reward"][:, t].astype(np.float32)), (t, rew, fx["reward"][:, t])
        done = te | tr
        step_obs = np.where(done[:, None], fin, obs)
        assert np.array_equal(step_obs.view(np.uint32), fx["obs"][:, t].view(np.uint32)), (t, np.argwhere(step_obs != fx["obs"][:, t])[:8])
        for i in np.nonzero(done)[0]:
            assert np.array_equal(obs[i], fx["reset_obs"][reset_at[(int(i), t)]]), (i, t)
        if t % 97 == 5:
            live = ~done
            for k, f in enumerate(STATE):
                assert np.array_equal(_np(env.info(f))[live], fx["state"][live, t, k]), (t, f)
    assert _np(env.info("overflow")).sum() == 0
    env.close()


@pytest.mark.parametrize("mode", ["NextStep", "SameStep", "Disabled"])
def test_step_matches_oracle_all_modes(cge, oracle, mode):
    code = {"NextStep": oracle.NEXT_STEP, "SameStep": oracle.SAME_STEP, "Disabled": oracle.DISABLED}[mode]
    n, T = 300, 1600 if mode != "Disabled" else 1300
    env = cge.HospitalVectorEnv(n, autoreset_mode=mode, env_index0=1)
    o = oracle.HospitalOracle(n, code)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(1 + 30))
    od, _ = env.reset(seed=30)
    assert np.array_equal(_np(od), o.reset())
    rng = np.random.default_rng(3)
    bias = rng.integers(0, 3, n)                                   # a third each: uniform / mass-casualty heavy / staffing heavy
    for t in range(T):
        a = rng.integers(0, 35, n).astype(np.int32)
        r = rng.random(n)
        a = np.where((bias == 1) & (r < 0.5), 33, a)
        a = np.where((bias == 2) & (r < 0.5), rng.integers(0, 12, n), a).astype(np.int32)
        if t % 13 == 0:
            a[rng.random(n) < 0.02] = 40                            # out of range: no-op
        od, rd, ted, trd, _ = env.step(a)
        oo, ro, teo, tro = o.step(a)
        assert np.array_equal(_np(od).view(np.uint32), oo.view(np.uint32)), (t, np.argwhere(_np(od) != oo)[:8])
        assert np.array_equal(_np(rd), ro) and np.array_equal(_np(ted), teo.astype(bool)) and np.array_equal(_np(trd), tro.astype(bool)), t
    for f in STATE + ["episodes", "needs_reset", "overflow"]:
        assert np.array_equal(_np(env.info(f)), o.info(f)), f
    env.close()


def test_reset_mask_returns_every_row(cge, oracle):
    n = 200
    env = cge.HospitalVectorEnv(n, autoreset_mode="Disabled")
    o = oracle.HospitalOracle(n, oracle.DISABLED)
    o.seed(np.arange(n, dtype=np.uint64) + np.uint64(5)); env.reset(seed=5); o.reset()
    rng = np.random.default_rng(1)
    for t in range(120):
        a = rng.integers(0, 35, n).astype(np.int32)
        env.step(a); o.step(a)
    mask = rng.random(n) < 0.3
    od, _ = env.reset(options={"reset_mask": torch.from_numpy(mask.astype(np.uint8)).cuda()})
    oo = o.reset(mask.astype(np.uint8))
    assert np.array_equal(_np(od).view(np.uint32), oo.view(np.uint32)), np.argwhere(_np(od) != oo)[:8]
    env.close()


def test_rollout_config5_size_and_sharding(cge, oracle):
    n, T = 1 << 17, 200
    env = cge.HospitalVectorEnv(n, autoreset_mode="SameStep", reuse_buffers=True)
    env.reset(seed=0)
    obs, rs, dc = env.rollout(T, action_seed=123)
    for lo in [0, n - 1000]:
        m = 1000
        o = oracle.HospitalOracle(m, oracle.SAME_STEP)
        o.seed(np.arange(lo, lo + m, dtype=np.uint64)); o.reset()
        oo, ro, do = o.rollout(T, 123, env0=lo)
        assert np.array_equal(_np(obs[lo:lo + m]), oo) and np.array_equal(_np(rs[lo:lo + m]), ro) and np.array_equal(_np(dc[lo:lo + m]), do)
    half = cge.HospitalVectorEnv(n // 2, autoreset_mode="SameStep", env_index0=n // 2)
    half.reset(seed=0)
    oh, rh, dh = half.rollout(T, action_seed=123)
    assert torch.equal(oh, obs[n // 2:]) and torch.equal(rh, rs[n // 2:]) and torch.equal(dh, dc[n // 2:])
    env.close(); half.close()
