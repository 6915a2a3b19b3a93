"""Debug probe: runs the manufacturing fixture replay on a -DCGE_MFG_GUARD build (every list/table index bounds-checked in the
kernel, the first violation recorded instead of dereferenced) and prints the record.  CGE_AMD_LIBRARY must point at that build."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import custom_gymnasium_environments_amd as cge  # noqa: E402

lib = cge.native_lib()
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "manufacturing_hash.npz"))
A = fx["actions"]
n, T = A.shape
env = cge.ManufacturingVectorEnv(n, autoreset_mode="SameStep")
env.reset(seed=int(fx["seed0"]))
A_dev = torch.from_numpy(A).cuda()
out = (C.c_uint * 8)()
bad_t = None
for t in range(T):
    obs, rew, te, tr, info = env.step(A_dev[:, t])
    torch.cuda.synchronize()
    lib.cge_manufacturing_debug_guard(out)
    if out[0] and bad_t is None:
        bad_t = t
        print("first violation at step", t, "record [count, site, index, limit, block, lane]:", list(out)[:6], flush=True)
    ok = np.array_equal(np.where((te | tr).cpu().numpy()[:, None], info["final_obs"].cpu().numpy(), obs.cpu().numpy()).view(np.uint32), fx["obs"][:, t].view(np.uint32))
    if not ok:
        print("first obs mismatch at step", t, flush=True)
        break
print("done; violations:", out[0])
