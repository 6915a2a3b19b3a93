"""timing probe for the snake fused rollout: trajectory vs in-place obs, per-step outputs on/off (measurement tool)"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import custom_gymnasium_environments_amd as cge

n, K = 1 << 20, 200
env = cge.SnakeVectorEnv(n, grid_size=10, autoreset_mode="SameStep", reuse_buffers=True)
env.reset(seed=0)
for traj, per_step, want_obs in [(True, True, True), (False, True, True), (False, False, True), (True, False, True), (False, False, False)]:
    env.rollout(K, action_seed=1, trajectory=traj, per_step=per_step, want_obs=want_obs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(3):
        env.rollout(K, action_seed=1, t0=K * (r + 1), trajectory=traj, per_step=per_step, want_obs=want_obs)
    b.record(); torch.cuda.synchronize()
    print(f"trajectory={traj!s:5} per_step={per_step!s:5} obs={want_obs!s:5}: {a.elapsed_time(b) / 3 / K * 1e3:7.2f} us/step")
