"""Golden vectors for SmartManufacturingEnv, produced by running the reference's own
smart_manufacturing_env/manufacturing_env.py (unmodified, imported from /root/reference).

Protocol (SURVEY.md 8c KAT-M1 / 8d config 5): env i = SmartManufacturingEnv() then reset(seed=S+i) (gymnasium's
self.np_random = Generator(PCG64(SeedSequence(S+i))), manufacturing_env.py:115); auto-reset = reset() with no seed (the
generator continues).  Actions from the counter hash (n=25); the `biased` set maps the hash onto a production-heavy
action mix so that completions, scrapping, the last-10/20 quality means and the 100-entry history mean are all exercised.
Outputs: tests/golden/manufacturing_hash.npz, manufacturing_biased.npz, manufacturing_typea.npz, manufacturing_kat.json
"""
import json
import os

import numpy as np

import common

common.use_stubs()
common.add_reference_dir("smart_manufacturing_env")
from manufacturing_env import SmartManufacturingEnv  # noqa: E402  (reference code)

# production-heavy mix for the `biased` set: 32 hash buckets -> action
BIASED = [0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 5, 6, 7, 8, 9, 10, 22, 23, 24, 24, 23, 16, 17, 18, 24, 0, 5]


# single-station products in quality mode: the most completions per episode the dynamics allow (last-10 mean with n >= 8)
TYPEA = [0, 0, 0, 0, 0, 0, 0, 0, 23, 23, 23, 23, 0, 0, 5, 1, 23, 23, 6, 11, 0, 0, 0, 0, 23, 23, 0, 0, 0, 0, 0, 23]


def action_for(a_seed, i, t, biased):
    if not biased:
        return common.hash_action(a_seed, i, t, 25, 0)
    return (BIASED if biased == 1 else TYPEA)[common.hash_action(a_seed, i, t, 32, 0)]


def scalars(env):
    return [env.raw_material_inventory, env.energy_consumption, env.total_reward, len(env.products_in_system),
            len(env.completed_products), len(env.scrapped_products), env.product_id_counter, len(env.quality_rate_history),
            env.oee_metrics['availability'], env.oee_metrics['performance'], env.oee_metrics['quality']]


def run_env(seed, T, a_seed, i, biased):
    env = SmartManufacturingEnv()
    obs, _ = env.reset(seed=seed)
    obs0 = obs.copy()
    O = np.zeros((T, 73), np.float32); R = np.zeros(T, np.float64); TE = np.zeros(T, np.uint8); TR = np.zeros(T, np.uint8)
    A = np.zeros(T, np.int32); S = np.zeros((T, 11), np.float64)
    resets = []
    for t in range(T):
        a = action_for(a_seed, i, t, biased)
        obs, rew, term, trunc, info = env.step(a)
        A[t] = a; O[t] = obs; R[t] = rew; TE[t] = term; TR[t] = trunc; S[t] = scalars(env)
        if term or trunc:
            obs, _ = env.reset()
            resets.append((t, obs.copy()))
    return obs0, A, O, R, TE, TR, S, resets


def make(name, n_envs, T, seed0, a_seed, biased):
    rows = [run_env(seed0 + i, T, a_seed, i, biased) for i in range(n_envs)]
    ridx, robs = [], []
    for i, r in enumerate(rows):
        for t, ob in r[7]:
            ridx.append((i, t)); robs.append(ob)
    out = os.path.join(common.GOLDEN, name + ".npz")
    np.savez_compressed(
        out, seed0=np.int64(seed0), a_seed=np.int64(a_seed), biased=np.int64(biased),
        obs0=np.stack([r[0] for r in rows]), actions=np.stack([r[1] for r in rows]), obs=np.stack([r[2] for r in rows]),
        reward=np.stack([r[3] for r in rows]), terminated=np.stack([r[4] for r in rows]), truncated=np.stack([r[5] for r in rows]),
        state=np.stack([r[6] for r in rows]),
        reset_index=np.array(ridx, np.int32).reshape(-1, 2), reset_obs=np.array(robs, np.float32).reshape(-1, 73),
        versions=np.array(json.dumps(common.versions())))
    R = np.stack([r[3] for r in rows]); S = np.stack([r[6] for r in rows])
    print(name, "episodes", len(ridx), "sum reward", R.sum(), "max in system", S[:, :, 3].max(), "max completed", S[:, :, 4].max(),
          "max scrapped", S[:, :, 5].max(), "max ids", S[:, :, 6].max(), "max hist", S[:, :, 7].max(), os.path.getsize(out), "bytes")


def kat_m1():
    """SURVEY 8c KAT-M1: reset(seed=7); actions default_rng(7).integers(0,25,3000)."""
    env = SmartManufacturingEnv()
    obs, _ = env.reset(seed=7)
    acts = np.random.default_rng(7).integers(0, 25, 3000)
    h = common.RunningHash(); h.obs(obs)
    total, episodes = 0.0, 0
    for t in range(3000):
        obs, r, te, tr, _ = env.step(int(acts[t]))
        h.step(obs, r, te, tr); total += r
        if te or tr:
            episodes += 1
            obs, _ = env.reset(); h.obs(obs)
    kat = dict(sum_reward=float(total), episodes=episodes, sha256=h.hexdigest(), **common.versions())
    with open(os.path.join(common.GOLDEN, "manufacturing_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("KAT-M1", kat)


if __name__ == "__main__":
    kat_m1()
    make("manufacturing_hash", 6, 3200, seed0=800, a_seed=123, biased=False)
    make("manufacturing_biased", 6, 3200, seed0=900, a_seed=321, biased=1)
    make("manufacturing_typea", 4, 1600, seed0=950, a_seed=77, biased=2)
