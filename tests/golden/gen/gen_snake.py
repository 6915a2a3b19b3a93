"""Golden vectors for SnakeEnvClassic, produced by running the reference's own
snake_env_classic/snake_env.py (unmodified, imported from /root/reference).

Protocols:
  * per-env stream: `random.seed(S+i)` immediately before env i is created and run alone;
    auto-reset = `env.reset()` (no seed) right after a terminal step, stream continues.
  * KAT-S1 (config 1): random.seed(0); reset(seed=0); 10,000 actions from
    np.random.default_rng(123).integers(0,4,10000); reset() after each done.
Outputs: tests/golden/snake_g10_hash.npz, snake_g10_greedy.npz, snake_g20_greedy.npz, snake_g10_short.npz, snake_rgb.npz, snake_kat.json
"""
import json
import os
import random

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("snake_env_classic")
from snake_env import SnakeEnvClassic  # noqa: E402  (reference code)


def run_env(grid, seed, T, policy, a_seed, env_index, eps=0.1, max_steps=None):
    random.seed(seed)
    env = SnakeEnvClassic(grid_size=grid)
    if max_steps is not None:
        env.max_steps = max_steps          # the reference's own attribute (snake_env.py:47), read at :113
    obs, info = env.reset()
    obs0 = obs.copy()
    rng = np.random.default_rng([a_seed, env_index])
    O = np.zeros((T, grid, grid), np.int8)
    R = np.zeros(T, np.float64)
    TE = np.zeros(T, np.uint8)
    TR = np.zeros(T, np.uint8)
    A = np.zeros(T, np.int32)
    SC = np.zeros(T, np.int32)      # info["score"] returned by the step
    LEN = np.zeros(T, np.int32)     # len(env.snake) after the step (before auto-reset)
    resets = []
    for t in range(T):
        if policy == "hash":
            a = common.hash_action(a_seed, env_index, t, 4)
        else:
            if rng.random() < eps:
                a = int(rng.integers(0, 4))
            else:
                hr, hc = env.snake[0]
                fr, fc = env.food
                cand = []
                if fr < hr: cand.append(0)
                if fc > hc: cand.append(1)
                if fr > hr: cand.append(2)
                if fc < hc: cand.append(3)
                moves = {0: (-1, 0), 1: (0, 1), 2: (1, 0), 3: (0, -1)}
                def ok(a_):
                    if abs(a_ - env.direction) == 2:
                        a_ = env.direction
                    r, c = hr + moves[a_][0], hc + moves[a_][1]
                    return 0 <= r < grid and 0 <= c < grid and (r, c) not in env.snake
                good = [a_ for a_ in cand if ok(a_)]
                if not good:
                    good = [a_ for a_ in range(4) if ok(a_)]
                a = int(good[rng.integers(0, len(good))]) if good else int(rng.integers(0, 4))
        obs, rew, term, trunc, info = env.step(a)
        A[t] = a
        O[t] = obs
        R[t] = rew
        TE[t] = term
        TR[t] = trunc
        SC[t] = info["score"]
        LEN[t] = len(env.snake)
        if term or trunc:
            obs, info = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, TR, SC, LEN, resets


def make(name, grid, n_envs, T, policy, seed0, a_seed, eps=0.1, max_steps=None):
    obs0 = np.zeros((n_envs, grid, grid), np.int8)
    A = np.zeros((n_envs, T), np.int32)
    O = np.zeros((n_envs, T, grid, grid), np.int8)
    R = np.zeros((n_envs, T), np.float64)
    TE = np.zeros((n_envs, T), np.uint8)
    TR = np.zeros((n_envs, T), np.uint8)
    SC = np.zeros((n_envs, T), np.int32)
    LEN = np.zeros((n_envs, T), np.int32)
    ridx, robs = [], []
    for i in range(n_envs):
        o0, a, o, r, te, tr, sc, ln, resets = run_env(grid, seed0 + i, T, policy, a_seed, i, eps, max_steps)
        obs0[i], A[i], O[i], R[i], TE[i], TR[i], SC[i], LEN[i] = o0, a, o, r, te, tr, sc, ln
        for t, ob in resets:
            ridx.append((i, t))
            robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, grid=np.int32(grid), seed0=np.int64(seed0), a_seed=np.int64(a_seed),
        policy=np.array(policy), max_steps=np.int32(1000 if max_steps is None else max_steps), obs0=obs0, actions=A, obs=O, reward=R, terminated=TE,
        truncated=TR, score=SC, length=LEN,
        reset_index=np.array(ridx, np.int32).reshape(-1, 2),
        reset_obs=np.array(robs, np.int8).reshape(-1, grid, grid),
        versions=np.array(json.dumps(common.versions())))
    print(name, "episodes", len(ridx), "max len", LEN.max(), "sum reward", R.sum(),
          os.path.getsize(out), "bytes")


def kat_s1():
    random.seed(0)
    env = SnakeEnvClassic(grid_size=10)
    obs, _ = env.reset(seed=0)
    first_food = tuple(int(x) for x in env.food)
    acts = np.random.default_rng(123).integers(0, 4, 10000)
    h = common.RunningHash()
    h.obs(obs)
    total, episodes = 0.0, 0
    heads = []
    for t, a in enumerate(acts):
        obs, r, te, tr, _ = env.step(int(a))
        h.step(obs, r, te, tr)
        total += r
        if t < 8:
            heads.append([int(env.snake[0][0]), int(env.snake[0][1]), float(r), int(te)])
        if te or tr:
            episodes += 1
            obs, _ = env.reset()
            h.obs(obs)
    kat = dict(first_food=first_food, first8=heads, first8_actions=[int(a) for a in acts[:8]],
               sum_reward=total, episodes=episodes, sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "snake_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-S1", kat)


def rgb_fixture():
    """render_mode="rgb_array": env.render() (snake_env.py:175-188) after the reset and after each of 120 greedy steps of 6 envs."""
    obs_l, rgb_l = [], []
    for i in range(6):
        random.seed(9000 + i)
        env = SnakeEnvClassic(render_mode="rgb_array", grid_size=10)
        obs, _ = env.reset()
        rng = np.random.default_rng([13, i])
        obs_l.append(obs.copy()); rgb_l.append(env.render().copy())
        for t in range(120):
            hr, hc = env.snake[0]
            fr, fc = env.food
            cand = [a for a, ok in ((0, fr < hr), (1, fc > hc), (2, fr > hr), (3, fc < hc)) if ok] or [int(rng.integers(0, 4))]
            obs, _, te, tr, _ = env.step(int(cand[rng.integers(0, len(cand))]))
            if te or tr:
                obs, _ = env.reset()
            obs_l.append(obs.copy()); rgb_l.append(env.render().copy())
    out = os.path.join(common.GOLDEN, "snake_rgb.npz")
    np.savez_compressed(out, obs=np.array(obs_l, np.int8), rgb=np.array(rgb_l, np.uint8), versions=np.array(json.dumps(common.versions())))
    print("snake_rgb", len(obs_l), "frames", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    kat_s1()
    rgb_fixture()
    make("snake_g10_hash", 10, 64, 1000, "hash", seed0=0, a_seed=123)
    make("snake_g10_greedy", 10, 32, 1500, "greedy", seed0=1000, a_seed=7, eps=0.05)
    make("snake_g20_greedy", 20, 8, 1500, "greedy", seed0=5000, a_seed=9, eps=0.03)
    # short horizon: the time limit (snake_env.py:113-114) fires every 9 steps, often on a step that also eats (:101-104)
    make("snake_g10_short", 10, 64, 300, "greedy", seed0=7000, a_seed=11, eps=0.05, max_steps=9)
    # an odd grid, the one the reference's own test_visualization.py:17 builds
    make("snake_g15_greedy", 15, 8, 1200, "greedy", seed0=9000, a_seed=13, eps=0.04)
