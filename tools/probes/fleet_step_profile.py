"""Per-call time of fleet's step() over one episode (GPU box): events around every call, so the every-50-steps redraw calls can be told
from the ordinary ones, plus how far the batch has drifted out of phase (envs whose timestep is not the call index).
usage: [CGE_AMD_LIBRARY=...] python tools/probes/fleet_step_profile.py [n_envs] [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import custom_gymnasium_environments_amd as cge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 800
env = cge.FleetVectorEnv(n, device="cuda:0", autoreset_mode="SameStep", reuse_buffers=True)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 8, (calls, n, 3), dtype=torch.int32, device="cuda", generator=g)
for t in range(20):
    env.step(acts[t])
env.reset(seed=0)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
in_phase = []
ev[0].record()
for t in range(calls):
    env.step(acts[t])
    ev[t + 1].record()
    if (t + 1) % 100 == 0:
        in_phase.append((t + 1, float((env.info("timestep") == (t + 1)).float().mean())))
torch.cuda.synchronize()
us = np.array([ev[t].elapsed_time(ev[t + 1]) * 1e3 for t in range(calls)])
idx = np.arange(1, calls + 1)
red = idx % 50 == 0
print(f"{n} envs, {calls} step() calls: mean {us.mean():.1f} us, median {np.median(us):.1f}, p90 {np.percentile(us, 90):.1f}")
print(f"  calls with index % 50 == 0: mean {us[red].mean():.1f} us (min {us[red].min():.1f}, max {us[red].max():.1f}); the others: mean {us[~red].mean():.1f}, median {np.median(us[~red]):.1f}")
print("  envs still in phase with the call index (share):", ", ".join(f"{t}: {s:.3f}" for t, s in in_phase))
print("  (the info() calls above sit between events: their time lands in the following step's sample)" if False else "")
for lo in range(0, calls, 100):
    seg = us[lo:lo + 100]
    print(f"  calls {lo + 1}-{lo + len(seg)}: mean {seg.mean():.1f} median {np.median(seg):.1f} max {seg.max():.1f}")
