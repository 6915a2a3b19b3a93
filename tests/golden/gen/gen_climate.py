"""Golden vectors for SmartClimateEnv, produced by running the reference's own
smartclimate_rl-main/smartclimate/{env,utils}.py (unmodified, imported from /root/reference).

Protocol (SURVEY.md 8c KAT-K1 / 8d config 5): env i = SmartClimateEnv(seed=S+i) then reset(seed=S+i) (a private
np.random.default_rng, env.py:30,63-65); auto-reset = reset() with no seed (generator continues).  Actions from the
counter hash: ac_temp = float32(16 + 16*u24) with u24 = hash(.., j=0) >> 40, lights[k] = hash(.., n=2, j=1+k).
Outputs: tests/golden/climate_hash.npz, climate_kat.json, and climate_small.npz: the reference constructed with
max_occupancy=3, episode_minutes=300 (the constructor arguments of smartclimate/env.py:16-28 that reach the dynamics).
"""
import json
import logging
import os

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("smartclimate_rl-main")
from smartclimate.env import SmartClimateEnv  # noqa: E402  (reference code)


def hash_act(a_seed, i, t):
    u = common.action_hash(a_seed, i, t, 0) >> 40
    ac = np.float32(16.0 + 16.0 * (u / 2.0**24))
    lights = np.array([common.hash_action(a_seed, i, t, 2, 1 + k) for k in range(4)], np.int8)
    return ac, lights


def run_env(seed, T, a_seed, i, extreme, ctor=None):
    env = SmartClimateEnv(seed=seed, log_level=logging.ERROR, **(ctor or {}))
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    O = np.zeros((T, 9), np.float32); R = np.zeros(T, np.float64); TE = np.zeros(T, np.uint8)
    AC = np.zeros(T, np.float32); LI = np.zeros((T, 4), np.int8); S = np.zeros((T, 5), np.float64)
    resets = []
    for t in range(T):
        ac, lights = hash_act(a_seed, i, t)
        if extreme and t % 7 == 0:
            ac = np.float32([-5.0, 40.0, 16.0, 32.0][(t // 7) % 4])      # exercises the clip at 16/32
        obs, rew, term, trunc, info = env.step({"ac_temp": np.array([ac], np.float32), "lights": lights})
        assert not trunc
        AC[t] = ac; LI[t] = lights; O[t] = obs; R[t] = rew; TE[t] = term
        S[t] = [env.room_temp, env.outside_temp, env.energy_usage, env.comfort_time, env.total_reward]
        if term:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, AC, LI, O, R, TE, S, resets


def make(name, n_envs, T, seed0, a_seed, extreme, ctor=None):
    rows = [run_env(seed0 + i, T, a_seed, i, extreme, ctor) for i in range(n_envs)]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[7]:
            ridx.append((i, t)); robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, seed0=np.int64(seed0), a_seed=np.int64(a_seed),
        obs0=np.stack([r[0] for r in rows]), ac_temp=np.stack([r[1] for r in rows]), lights=np.stack([r[2] for r in rows]),
        obs=np.stack([r[3] for r in rows]), reward=np.stack([r[4] for r in rows]), terminated=np.stack([r[5] for r in rows]),
        state=np.stack([r[6] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, 9),
        versions=np.array(json.dumps(common.versions())), **({"ctor": np.array(json.dumps(ctor))} if ctor else {}))
    R = np.stack([r[4] for r in rows])
    print(name, "episodes", len(ridx), "sum reward", R.sum(), os.path.getsize(out), "bytes")


def kat_k1():
    """SURVEY 8c KAT-K1: SmartClimateEnv(seed=7), reset(seed=7); one default_rng(7) draws uniform(16,32,(3000,1)).astype(f32)
    then integers(0,2,(3000,4)).astype(i8)."""
    env = SmartClimateEnv(seed=7, log_level=logging.ERROR)
    obs, _ = env.reset(seed=7)
    g = np.random.default_rng(7)
    ac = g.uniform(16, 32, (3000, 1)).astype(np.float32)
    li = g.integers(0, 2, (3000, 4)).astype(np.int8)
    h = common.RunningHash(); h.obs(obs)
    total, episodes = 0.0, 0
    for t in range(3000):
        obs, r, te, tr, _ = env.step({"ac_temp": ac[t], "lights": li[t]})
        h.step(obs, r, te, tr); total += r
        if te or tr:
            episodes += 1
            obs, _ = env.reset(); h.obs(obs)
    kat = dict(sum_reward=total, episodes=episodes, sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "climate_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-K1", kat)


if __name__ == "__main__":
    logging.disable(logging.CRITICAL)
    kat_k1()
    make("climate_hash", 8, 3000, seed0=600, a_seed=123, extreme=True)
    make("climate_small", 6, 1000, seed0=700, a_seed=41, extreme=True, ctor=dict(max_occupancy=3, episode_minutes=300))
