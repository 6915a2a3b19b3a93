"""Golden vectors for TrafficManagementEnv, produced by running the reference's own
traffic_management_env/{environment,utils,config}.py (unmodified, imported from /root/reference).

Protocol: env i is a fresh TrafficManagementEnv() run alone after
`reset(seed=S+i)` (seeds the global `random`, environment.py:145-147); auto-reset = `env.reset()` with no
seed after a terminal step (stream continues).  Actions: counter hash mod 3 per (env, t, intersection).
Outputs: tests/golden/traffic_hash.npz, traffic_lazy.npz, traffic_kat.json and, for the constructor arguments the reference's own
scripts vary, traffic_3x3.npz (simple_test.py:71-76: grid (3,3), 4 intersections, 20 vehicles, spawn 0.4) and traffic_6x6.npz
(USAGE_EXAMPLES.md:32-38: grid (6,6), 16 intersections, 80 vehicles, spawn 0.5).
"""
import json
import os

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("traffic_management_env")
from environment import TrafficManagementEnv  # noqa: E402  (reference code)

PHASES = ["NS_GREEN", "NS_YELLOW", "EW_GREEN", "EW_YELLOW"]


def internal(env):
    """Collapsed observable state for bisecting: per intersection phase/timer/passed/wait, per queue len."""
    out = []
    for it in env.intersections:
        out += [PHASES.index(it.traffic_light.current_phase), it.traffic_light.phase_timer, it.vehicles_passed, it.total_waiting_time]
        out += [len(it.vehicle_queues[d]) for d in it.vehicle_queues]
    out += [len(env.vehicles), env.current_timestep]
    return out


def run_env(seed, T, a_seed, i, policy, ctor=None):
    env = TrafficManagementEnv(**(ctor or {}))
    ni = env.num_intersections
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    O = np.zeros((T, obs.shape[0]), np.float32)
    R = np.zeros(T, np.float64)
    TE = np.zeros(T, np.uint8)
    A = np.zeros((T, ni), np.int32)
    S = np.zeros((T, ni * 8 + 2), np.int32)
    resets = []
    for t in range(T):
        if policy == "hash":
            a = np.array([common.hash_action(a_seed, i, t, 3, j) for j in range(ni)], np.int32)
        else:  # mostly "maintain" so lights run on their own random timers (exercises randint(5,30) heavily)
            a = np.array([common.hash_action(a_seed, i, t, 3, j) if common.hash_action(a_seed, i, t, 10, 16 + j) == 0 else 0
                          for j in range(ni)], np.int32)
        obs, rew, term, trunc, info = env.step(a)
        assert not trunc
        A[t] = a; O[t] = obs; R[t] = rew; TE[t] = term; S[t] = internal(env)
        if term:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, S, resets


def make(name, n_envs, T, seed0, a_seed, policy, ctor=None):
    rows = [run_env(seed0 + i, T, a_seed, i, policy, ctor) for i in range(n_envs)]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[6]:
            ridx.append((i, t)); robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, seed0=np.int64(seed0), a_seed=np.int64(a_seed), policy=np.array(policy),
        obs0=np.stack([r[0] for r in rows]), actions=np.stack([r[1] for r in rows]), obs=np.stack([r[2] for r in rows]),
        reward=np.stack([r[3] for r in rows]), terminated=np.stack([r[4] for r in rows]), internal=np.stack([r[5] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, rows[0][0].shape[0]),
        versions=np.array(json.dumps(common.versions())), **({"ctor": np.array(json.dumps(ctor))} if ctor else {}))
    R = np.stack([r[3] for r in rows])
    print(name, "episodes", len(ridx), "sum reward", R.sum(), os.path.getsize(out), "bytes")


def kat_t1():
    """SURVEY 8c KAT-T1 shape: reset(seed=42), 1000 steps, hash(123, 0, t, 3, j) actions."""
    env = TrafficManagementEnv()
    obs, _ = env.reset(seed=42)
    h = common.RunningHash()
    h.obs(obs)
    total, rewards = 0.0, {}
    for t in range(1000):
        a = np.array([common.hash_action(123, 0, t, 3, j) for j in range(9)], np.int32)
        obs, r, te, tr, info = env.step(a)
        h.step(obs, r, te, tr)
        total += r
        if t + 1 in (1, 2, 6, 11, 1000):
            rewards[str(t + 1)] = float(r)
    kat = dict(rewards=rewards, sum_reward=total, num_vehicles=info["num_vehicles"],
               total_passed=int(info["metrics"]["total_vehicles_passed"]), sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "traffic_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-T1", kat)


if __name__ == "__main__":
    kat_t1()
    make("traffic_hash", 6, 1100, seed0=200, a_seed=123, policy="hash")
    make("traffic_lazy", 6, 1100, seed0=300, a_seed=55, policy="lazy")
    make("traffic_3x3", 5, 1100, seed0=400, a_seed=77, policy="lazy",
         ctor=dict(grid_size=(3, 3), num_intersections=4, max_vehicles=20, spawn_rate=0.4))       # simple_test.py:71-76
    make("traffic_6x6", 4, 1100, seed0=500, a_seed=99, policy="lazy",
         ctor=dict(grid_size=(6, 6), num_intersections=16, max_vehicles=80, spawn_rate=0.5))      # USAGE_EXAMPLES.md:32-38
