"""Replays the manufacturing reference fixtures (and a 4,096-env whole-episode rollout) on the bounds-checked debug build:
    python -m custom_gymnasium_environments_amd.build --guard          # -> custom_gymnasium_environments_amd/libcge_amd_guard.so
    CGE_AMD_LIBRARY=$PWD/custom_gymnasium_environments_amd/libcge_amd_guard.so python tools/probes/mfg_guard.py
Every [row][env] table index of the kernel goes through GX(site, index, limit) there; the first index outside its table is recorded
(site, index, limit, block, lane) instead of dereferenced.  Prints the record; exit code 1 on a violation or an obs mismatch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import custom_gymnasium_environments_amd as cge  # noqa: E402
from custom_gymnasium_environments_amd import _native  # noqa: E402

lib = C.CDLL(_native.LIB_PATH)
out = (C.c_uint * 8)()
bad = False
for name in ["manufacturing_hash.npz", "manufacturing_biased.npz", "manufacturing_typea.npz"]:
    fx = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", name))
    A = fx["actions"]
    n, T = A.shape
    env = cge.ManufacturingVectorEnv(n, autoreset_mode="SameStep")
    env.reset(seed=int(fx["seed0"]))
    A_dev = torch.from_numpy(A).cuda()
    for t in range(T):
        obs, rew, te, tr, info = env.step(A_dev[:, t])
        step_obs = np.where((te | tr).cpu().numpy()[:, None], info["final_obs"].cpu().numpy(), obs.cpu().numpy())
        if not np.array_equal(step_obs.view(np.uint32), fx["obs"][:, t].view(np.uint32)):
            print(name, "first obs mismatch at step", t, flush=True)
            bad = True
            break
    torch.cuda.synchronize()
    lib.cge_manufacturing_debug_guard(out)
    print(name, "record [count, site, index, limit, block, lane]:", list(out)[:6], flush=True)
    env.close()
env = cge.ManufacturingVectorEnv(4096, autoreset_mode="SameStep")
env.reset(seed=3)
env.rollout(1600, action_seed=11)
torch.cuda.synchronize()
lib.cge_manufacturing_debug_guard(out)
print("4,096 envs x 1,600 fused steps: record", list(out)[:6])
sys.exit(1 if (bad or out[0]) else 0)
