"""Snake fused rollout, one process: us per 1M-env step for K steps per launch, (a) right after a reset of the whole batch (launches 2..4 after it),
(b) at the batch's steady state (5,000 steps later), for several launch-start top-up thresholds (needs a -DCGE_SNAKE_TOPUP_ENV build for thresholds
other than 12).  usage: python tools/probes/snake_fresh_vs_steady.py "20 200" "12 0 64" """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import custom_gymnasium_environments_amd as cge

KS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20 200").split()]
TS = (sys.argv[2] if len(sys.argv) > 2 else "12").split()
n = 1 << 20
env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="SameStep", reuse_buffers=True)


def timed(K, launches, t0):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for j in range(launches):
        env.rollout(K, action_seed=1, t0=t0 + j * K, trajectory=True, per_step=True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (launches * K) * 1e3


for K in KS:
    for T in TS:
        os.environ["CGE_SNAKE_TOPUP"] = T
        fresh = []
        for rep in range(3):
            env.reset(seed=rep)
            env.rollout(K, action_seed=1, t0=0, trajectory=True, per_step=True)          # launch 1: allocation / first touch of the trajectory
            fresh.append(timed(K, max(1, 60 // K), K))
        for j in range(5000 // K):
            env.rollout(K, action_seed=1, t0=(j + 4) * K, trajectory=True, per_step=True)
        steady = [timed(K, max(3, 600 // K), 6000 + r * 1000) for r in range(3)]
        print(f"k={K:4d} threshold {T:>3}: fresh {' '.join('%.2f' % x for x in fresh)} | steady {' '.join('%.2f' % x for x in steady)} us/step", flush=True)
