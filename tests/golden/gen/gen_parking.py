"""Golden vectors for SmartParkingEnv, produced by running the reference's own smart_parking_env/core/*.py
(unmodified, imported from /root/reference).

Protocol (SURVEY.md 8d config 5 / KAT-P1): the env never seeds `random` (parking_env.py:81 only seeds the unused
gymnasium generator), so env i is a fresh SmartParkingEnv run alone after `random.seed(S+i)`; auto-reset =
`env.reset()` after a terminal step, stream continues.  Actions: hash mod 8, or a "sensible" policy mix that
assigns/toggles so that spots fill up, queues form and departures happen.
Outputs: tests/golden/parking_hash.npz, parking_busy.npz, parking_kat.json
"""
import json
import os
import random

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("smart_parking_env")
from core.parking_env import SmartParkingEnv  # noqa: E402  (reference code)


def stats_vec(env):
    cm = env.customer_manager
    occ = [sum(1 for s in env.parking_lot.zone_spots[z] if env.parking_lot.spots[s].is_occupied) for z in "ABC"]
    return [cm.total_customers, cm.rejected_customers, cm.satisfied_customers, cm.total_wait_time, len(env.parking_lot.queue),
            env.price_changes_this_hour, env.current_timestep, *occ, *[env.pricing_manager.price_levels[z] for z in "ABC"]]


def run_env(seed, T, a_seed, i, policy):
    random.seed(seed)
    env = SmartParkingEnv()
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    O = np.zeros((T, 13), np.float32); R = np.zeros(T, np.float64); TE = np.zeros(T, np.uint8); A = np.zeros(T, np.int32)
    S = np.zeros((T, 13), np.int64); F = np.zeros((T, 2), np.float64)
    resets = []
    for t in range(T):
        if policy == "hash":
            a = common.hash_action(a_seed, i, t, 8)
        else:
            # busy policy: mostly assign to a hashed zone when someone queues, occasionally reject/toggle/idle
            u = common.hash_action(a_seed, i, t, 100)
            if env.parking_lot.queue and u < 80:
                a = 1 + common.hash_action(a_seed, i, t, 3, 1)
            elif u < 84:
                a = 4
            elif u < 90:
                a = 5 + common.hash_action(a_seed, i, t, 3, 2)
            else:
                a = 0
        obs, rew, term, trunc, info = env.step(a)
        assert not trunc
        A[t] = a; O[t] = obs; R[t] = rew; TE[t] = term; S[t] = stats_vec(env)
        F[t] = [env.episode_revenue, env.episode_satisfaction]
        if term:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, S, F, resets


def make(name, n_envs, T, seed0, a_seed, policy):
    rows = [run_env(seed0 + i, T, a_seed, i, policy) for i in range(n_envs)]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[7]:
            ridx.append((i, t)); robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, seed0=np.int64(seed0), a_seed=np.int64(a_seed), policy=np.array(policy),
        obs0=np.stack([r[0] for r in rows]), actions=np.stack([r[1] for r in rows]), obs=np.stack([r[2] for r in rows]),
        reward=np.stack([r[3] for r in rows]), terminated=np.stack([r[4] for r in rows]), stats=np.stack([r[5] for r in rows]),
        money=np.stack([r[6] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, 13),
        versions=np.array(json.dumps(common.versions())))
    R = np.stack([r[3] for r in rows])
    print(name, "episodes", len(ridx), "sum reward", R.sum(), "max occupied", np.stack([r[5] for r in rows])[:, :, 7:10].sum(-1).max(),
          os.path.getsize(out), "bytes")


def kat_p1():
    """SURVEY 8c KAT-P1: random.seed(7); reset(seed=7); actions default_rng(7).integers(0,8,3000)."""
    random.seed(7)
    env = SmartParkingEnv()
    obs, _ = env.reset(seed=7)
    acts = np.random.default_rng(7).integers(0, 8, 3000)
    h = common.RunningHash(); h.obs(obs)
    total, episodes = 0.0, 0
    for a in acts:
        obs, r, te, tr, _ = env.step(int(a))
        h.step(obs, r, te, tr); total += r
        if te or tr:
            episodes += 1
            obs, _ = env.reset(); h.obs(obs)
    kat = dict(sum_reward=total, episodes=episodes, sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "parking_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-P1", kat)


if __name__ == "__main__":
    kat_p1()
    make("parking_hash", 8, 1500, seed0=400, a_seed=123, policy="hash")
    make("parking_busy", 8, 3000, seed0=450, a_seed=31, policy="busy")
