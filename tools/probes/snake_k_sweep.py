"""Snake fused rollout: time per 1M-env step as a function of the steps per launch, one process, one box (measurement tool).
Every step's obs / reward / flag goes to its own slot of a [k, N, ...] trajectory, as in bench.py's rollout leg."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import custom_gymnasium_environments_amd as cge

n = 1 << 20
env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="SameStep", reuse_buffers=True)
env.reset(seed=0)
env.rollout(200, action_seed=1, trajectory=True, per_step=True)
t0 = 200
acts = torch.randint(0, 4, (64, n), dtype=torch.int32, device="cuda")
for rnd in range(2):
    for k in [5, 10, 20, 40, 100, 200]:
        reps = max(2, 200 // k)
        env.rollout(k, action_seed=1, t0=t0, trajectory=True, per_step=True); t0 += k
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(reps):
            env.rollout(k, action_seed=1, t0=t0, trajectory=True, per_step=True); t0 += k
        b.record(); torch.cuda.synchronize()
        print(f"round {rnd} k={k:4d}: {a.elapsed_time(b) / reps / k * 1e3:7.2f} us/step  ({env.last_kernel()})", flush=True)
    for r in range(5):
        env.step(acts[r])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(64):
        env.step(acts[r])
    b.record(); torch.cuda.synchronize()
    print(f"round {rnd} step(): {a.elapsed_time(b) / 64 * 1e3:7.2f} us/step  ({env.last_kernel()})", flush=True)
