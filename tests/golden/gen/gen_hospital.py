"""Golden vectors for HospitalManagementEnv, produced by running the reference's own hospital_management_env/hospital_env.py
(unmodified, imported from /root/reference).

Protocol (SURVEY.md 8c KAT-H1 / 8d config 5): the env never seeds `random` (hospital_env.py:186 only seeds the unused
gymnasium generator), so env i is constructed (its __init__ already calls reset()), then `random.seed(S+i)`, then
`reset(seed=S+i)`; auto-reset = `env.reset()` after a terminal step, the stream continues.  Actions: counter hash mod 35,
or a `surge` mix that keeps the mass-casualty protocol on, rarely adds capacity and sometimes transfers / discharges, so
that queues grow into the hundreds, critical patients wait (death rolls) and the insurance-delay path runs.
Outputs: tests/golden/hospital_hash.npz, hospital_surge.npz, hospital_kat.json
"""
import json
import os
import random

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("hospital_management_env")
from hospital_env import HospitalManagementEnv, Department  # noqa: E402  (reference code)

SURGE = [33, 33, 33, 12, 13, 15, 32, 31, 31, 18, 19, 24, 25, 30, 0, 3, 6, 9, 11, 33, 33, 14, 16, 17, 20, 26, 1, 7, 34, 33, 33, 33]


def action_for(a_seed, i, t, policy):
    if policy == 0:
        return common.hash_action(a_seed, i, t, 35, 0)
    return SURGE[common.hash_action(a_seed, i, t, 32, 0)]


def scalars(env):
    return [env.deaths, env.patients_treated, env.total_wait_time, env.current_time, int(env.outbreak_active), int(env.mass_casualty_event),
            env.next_patient_id, *[len(env.patient_queues[d]) for d in Department], sum(1 for b in env.beds if b.occupied),
            sum(env.medicine_inventory.values())]


def run_env(seed, T, a_seed, i, policy):
    env = HospitalManagementEnv()
    random.seed(seed)
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    O = np.zeros((T, 243), np.float32); R = np.zeros(T, np.float64); TE = np.zeros(T, np.uint8); TR = np.zeros(T, np.uint8)
    A = np.zeros(T, np.int32); S = np.zeros((T, 15), np.int64)
    resets = []
    for t in range(T):
        a = action_for(a_seed, i, t, policy)
        obs, rew, term, trunc, info = env.step(a)
        A[t] = a; O[t] = obs; R[t] = rew; TE[t] = term; TR[t] = trunc; S[t] = scalars(env)
        if term or trunc:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, TR, S, resets


def make(name, n_envs, T, seed0, a_seed, policy):
    rows = [run_env(seed0 + i, T, a_seed, i, policy) for i in range(n_envs)]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[7]:
            ridx.append((i, t)); robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, seed0=np.int64(seed0), a_seed=np.int64(a_seed), policy=np.int64(policy),
        obs0=np.stack([r[0] for r in rows]), actions=np.stack([r[1] for r in rows]), obs=np.stack([r[2] for r in rows]),
        reward=np.stack([r[3] for r in rows]), terminated=np.stack([r[4] for r in rows]), truncated=np.stack([r[5] for r in rows]),
        state=np.stack([r[6] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, 243),
        versions=np.array(json.dumps(common.versions())))
    R = np.stack([r[3] for r in rows]); S = np.stack([r[6] for r in rows])
    print(name, "episodes", len(ridx), "sum reward", R.sum(), "max deaths", S[:, :, 0].max(), "max treated", S[:, :, 1].max(),
          "max queues", S[:, :, 7:13].max(axis=(0, 1)), "max ids", S[:, :, 6].max(), os.path.getsize(out), "bytes")


def kat_h1():
    """SURVEY 8c KAT-H1: construct, random.seed(7), reset(seed=7); actions default_rng(7).integers(0,35,3000)."""
    env = HospitalManagementEnv()
    random.seed(7)
    obs, _ = env.reset(seed=7)
    acts = np.random.default_rng(7).integers(0, 35, 3000)
    h = common.RunningHash(); h.obs(obs)
    total, episodes = 0.0, 0
    for t in range(3000):
        obs, r, te, tr, _ = env.step(int(acts[t]))
        h.step(obs, r, te, tr); total += r
        if te or tr:
            episodes += 1
            obs, _ = env.reset(); h.obs(obs)
    kat = dict(sum_reward=float(total), episodes=episodes, sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "hospital_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-H1", kat)


if __name__ == "__main__":
    kat_h1()
    make("hospital_hash", 5, 3000, seed0=1000, a_seed=123, policy=0)
    make("hospital_surge", 5, 3000, seed0=1100, a_seed=321, policy=1)
