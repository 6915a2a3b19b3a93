"""crypto resident rollout with and without the observation rows (how much of a step is the row stores?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import custom_gymnasium_environments_amd as cge
env = cge.CryptoVectorEnv(1 << 20, device="cuda:0", autoreset_mode="SameStep", reuse_buffers=True)
env.reset(seed=1)
for want, traj in ((True, True), (False, False), (True, False)):
    env.rollout(8, action_seed=1, trajectory=traj, want_obs=want)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.rollout(16, action_seed=2, t0=8, trajectory=traj, want_obs=want)
    b.record()
    torch.cuda.synchronize()
    print(f"want_obs={want} trajectory={traj}: {a.elapsed_time(b) * 1e3 / 16:.1f} us per 1M-env step")
