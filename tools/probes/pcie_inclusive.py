"""The PCIe-inclusive rate of the boundary's host-array form (cge.make_vec(..., numpy=True): NumPy actions in, NumPy obs / reward /
flags out every step — what a SyncVectorEnv consumer holds) beside the device-resident rate bench.py reports.
usage (GPU box): python tools/probes/pcie_inclusive.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import custom_gymnasium_environments_amd as cge  # noqa: E402

for env_id, n, kw in (("snake_env_classic-v0", 1 << 20, dict(grid_size=10)), ("CryptoTrading-v0", 1 << 20, {}), ("TrafficManagement-v0", 1 << 18, {})):
    env = cge.make_vec(env_id, n, numpy=True, autoreset_mode="SameStep", **kw)
    env.reset(seed=1)
    sp = env.single_action_space
    rng = np.random.default_rng(0)
    if hasattr(sp, "nvec"):
        acts = [rng.integers(0, 3, size=(n, len(sp.nvec)), dtype=np.int32) for _ in range(4)]
    else:
        acts = [rng.integers(0, int(sp.n), size=n, dtype=np.int32) for _ in range(4)]
    for t in range(5):
        env.step(acts[t % 4])
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for t in range(K):
        obs, rew, term, trunc, info = env.step(acts[t % 4])
    dt = (time.perf_counter() - t0) / K
    nbytes = obs.nbytes + rew.nbytes + term.nbytes + trunc.nbytes + acts[0].nbytes
    print(f"{env_id}: {n} envs, host arrays both ways: {dt * 1e3:.3f} ms per step = {n / dt:.3e} env-steps/s, {nbytes / 1e6:.1f} MB over PCIe per step "
          f"({nbytes / dt / 1e9:.1f} GB/s)", flush=True)
    env.close()
