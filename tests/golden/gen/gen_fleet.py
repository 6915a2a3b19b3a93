"""Golden vectors for FleetManagementEnv, produced by running the reference's own
fleet_management_env/fleet_env.py (unmodified, imported from /root/reference).

Protocol (SURVEY.md 8c KAT-F1 / 8d config 5): env i = fresh FleetManagementEnv() run alone after reset(seed=S+i)
(seeds BOTH np.random legacy and `random`, fleet_env.py:187-189); auto-reset = reset() with no seed after
terminated or truncated.  Actions: hash mod 8 per (env, t, vehicle), or a scripted courier policy (move toward the
pickup / drop-off cell, pick up, drop off, refuel) so that pickups, deliveries, deadlines and refuels all occur.
Outputs: tests/golden/fleet_hash.npz, fleet_courier.npz, fleet_kat.json
"""
import json
import os

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("fleet_management_env")
import fleet_env as ref  # noqa: E402  (reference code)


def courier_action(env, vid, a_seed, i, t):
    v = env.vehicles[vid]
    u = common.hash_action(a_seed, i, t, 100, vid)
    if u < 6:
        return common.hash_action(a_seed, i, t, 8, 4 + vid)          # some noise, incl. invalid pick/drop/refuel
    x, y = v.position

    def toward(tx, ty):
        if x < tx: return 4
        if x > tx: return 3
        if y < ty: return 2
        if y > ty: return 1
        return 0
    if v.fuel < 15:
        fx, fy = min(env.fuel_stations, key=lambda s: abs(s[0] - x) + abs(s[1] - y))
        return 7 if (x, y) == (fx, fy) else toward(fx, fy)
    if v.assigned_delivery != -1:
        d = env.delivery_requests[v.assigned_delivery]
        return 6 if (x, y) == d.delivery_location else toward(*d.delivery_location)
    best = None
    for k, d in enumerate(env.delivery_requests):
        if d.completed or d.assigned_vehicle != -1 or not d.is_available(env.timestep):
            continue
        if d.required_vehicle is not None and d.required_vehicle != v.vehicle_type:
            continue
        if (k + vid) % 3 and best is not None:
            continue
        best = d
    if best is None:
        return 0
    return 5 if (x, y) == best.pickup_location else toward(*best.pickup_location)


def internal(env):
    out = []
    for v in env.vehicles:
        out += [v.position[0], v.position[1], v.cargo_used, v.assigned_delivery]
    out += [env.timestep, env.missed_deadlines, env.completed_deliveries, len(env.delivery_requests)]
    return out


def run_env(seed, T, a_seed, i, policy):
    env = ref.FleetManagementEnv()
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    D = obs.shape[0]
    O = np.zeros((T, D), np.float32); R = np.zeros(T, np.float64); TE = np.zeros(T, np.uint8); TR = np.zeros(T, np.uint8)
    A = np.zeros((T, 3), np.int32); S = np.zeros((T, 16), np.int64); F = np.zeros((T, 4), np.float64)
    resets = []
    for t in range(T):
        if policy == "hash":
            a = np.array([common.hash_action(a_seed, i, t, 8, j) for j in range(3)], np.int32)
        else:
            a = np.array([courier_action(env, j, a_seed, i, t) for j in range(3)], np.int32)
        obs, rew, term, trunc, info = env.step(a)
        A[t] = a; O[t] = obs; R[t] = rew; TE[t] = term; TR[t] = trunc; S[t] = internal(env)
        F[t] = [env.vehicles[0].fuel, env.vehicles[1].fuel, env.vehicles[2].fuel, env.weather_effect]
        if term or trunc:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, TR, S, F, resets


def make(name, n_envs, T, seed0, a_seed, policy):
    rows = [run_env(seed0 + i, T, a_seed, i, policy) for i in range(n_envs)]
    D = rows[0][0].shape[0]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[8]:
            ridx.append((i, t)); robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, seed0=np.int64(seed0), a_seed=np.int64(a_seed), policy=np.array(policy),
        obs0=np.stack([r[0] for r in rows]), actions=np.stack([r[1] for r in rows]), obs=np.stack([r[2] for r in rows]),
        reward=np.stack([r[3] for r in rows]), terminated=np.stack([r[4] for r in rows]), truncated=np.stack([r[5] for r in rows]),
        internal=np.stack([r[6] for r in rows]), fuel=np.stack([r[7] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, D),
        versions=np.array(json.dumps(common.versions())))
    R = np.stack([r[3] for r in rows])
    print(name, "obs dim", D, "episodes", len(ridx), "sum reward", R.sum(), "completed max",
          np.stack([r[6] for r in rows])[:, :, 14].max(), os.path.getsize(out), "bytes")


def kat_f1():
    """SURVEY 8c KAT-F1: reset(seed=7); actions default_rng(7).integers(0,8,(3000,3))."""
    env = ref.FleetManagementEnv()
    obs, _ = env.reset(seed=7)
    acts = np.random.default_rng(7).integers(0, 8, (3000, 3))
    h = common.RunningHash(); h.obs(obs)
    total, episodes = 0.0, 0
    for a in acts:
        obs, r, te, tr, _ = env.step(a)
        h.step(obs, r, te, tr); total += r
        if te or tr:
            episodes += 1
            obs, _ = env.reset(); h.obs(obs)
    kat = dict(sum_reward=float(total), episodes=episodes, sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "fleet_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-F1", kat)


if __name__ == "__main__":
    kat_f1()
    make("fleet_hash", 8, 1500, seed0=700, a_seed=123, policy="hash")
    make("fleet_courier", 8, 2500, seed0=750, a_seed=19, policy="courier")
