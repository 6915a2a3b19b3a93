"""Snake fused rollout at the batch's steady state: launch-start top-up threshold (and, on a build that has it, the fill-up threshold) A/B'd INSIDE one
process — the same library and arguments differ by +-2 us per step between two processes on one box, which hides effects of this size.
Needs a -DCGE_SNAKE_TOPUP_ENV build (CGE_AMD_LIBRARY=tools/ab/libcge_tenv.so): cge_snake_rollout reads CGE_SNAKE_TOPUP / CGE_SNAKE_TOPUP2 at every call.
usage: python tools/probes/snake_topup_inprocess.py K "12" "20" "12:24" ...   (threshold or threshold:fillup)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import custom_gymnasium_environments_amd as cge

K = int(sys.argv[1])
settings = sys.argv[2:]
n = 1 << 20
env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="SameStep", reuse_buffers=True)
env.reset(seed=0)
t0 = 0


def run(steps):
    global t0
    for _ in range(steps // K):
        env.rollout(K, action_seed=1, t0=t0, trajectory=True, per_step=True)
        t0 += K


run(2000)
for rnd in range(3):
    for s in settings:
        a, _, b = s.partition(":")
        os.environ["CGE_SNAKE_TOPUP"] = a
        os.environ["CGE_SNAKE_TOPUP2"] = b or "0"
        run(800)                                   # the rings settle into this setting's steady state
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(800)
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} k={K} threshold {s:>6}: {e0.elapsed_time(e1) / 800 * 1e3:6.2f} us/step  ({env.last_kernel()})", flush=True)
