"""Replays the manufacturing, hospital and fleet reference fixtures (and a 4,096-env whole-episode rollout of each) on the
bounds-checked debug build:
    python -m custom_gymnasium_environments_amd.build --guard          # -> custom_gymnasium_environments_amd/libcge_amd_guard.so
    CGE_AMD_LIBRARY=$PWD/custom_gymnasium_environments_amd/libcge_amd_guard.so python tools/probes/guard_run.py
There every [row][env] table index of the manufacturing kernel (GX, manufacturing.hip), every ring index of the hospital kernel, the
LDS slots of its draw ring and the work-list slots / entries of the fleet kernels (CGE_GX, cge_device.hpp) are checked; the first index
outside its table is recorded (site, index, limit, block, lane) instead of dereferenced.  Prints the records; exit code 1 on a
violation or an obs mismatch.  One pass — not a loop."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import custom_gymnasium_environments_amd as cge  # noqa: E402
from custom_gymnasium_environments_amd import _native  # noqa: E402

lib = C.CDLL(_native.LIB_PATH)
out = (C.c_uint * 8)()
bad = False
CASES = [("manufacturing", cge.ManufacturingVectorEnv, ["manufacturing_hash.npz", "manufacturing_biased.npz", "manufacturing_typea.npz"], 1600),
         ("hospital", cge.HospitalVectorEnv, ["hospital_hash.npz", "hospital_surge.npz"], 1500),
         ("fleet", cge.FleetVectorEnv, ["fleet_hash.npz", "fleet_courier.npz"], 900)]
for env_name, cls, fixtures, k_long in CASES:
    record = getattr(lib, f"cge_{env_name}_debug_guard")
    for name in fixtures:
        fx = np.load(os.path.join(ROOT, "tests", "golden", name))
        A = fx["actions"]
        n, T = A.shape[0], A.shape[1]
        env = cls(n, autoreset_mode="SameStep")
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
        record(out)
        print(name, "record [count, site, index, limit, block, lane]:", list(out)[:6], flush=True)
        bad = bad or out[0] != 0
        env.close()
    env = cls(4096, autoreset_mode="SameStep")
    env.reset(seed=3)
    env.rollout(k_long, action_seed=11)
    for _ in range(60):                                                      # and the step() entry on the same handle
        env.step(env.action_space.sample())
    torch.cuda.synchronize()
    record(out)
    print(f"{env_name}: 4,096 envs x {k_long} fused steps + 60 step() calls: record", list(out)[:6], flush=True)
    bad = bad or out[0] != 0
    env.close()
sys.exit(1 if bad else 0)
