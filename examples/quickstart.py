"""Quickstart: the batched drop-in where the reference would build a SyncVectorEnv of its own envs.

    python examples/quickstart.py            (needs an MI355X; build first: python -m custom_gymnasium_environments_amd.build)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import custom_gymnasium_environments_amd as cge  # noqa: E402

N = 1 << 16
# gym.vector.SyncVectorEnv([lambda: SnakeEnvClassic(grid_size=10)] * N) becomes:
env = cge.SnakeVectorEnv(N, grid_size=10, device="cuda:0", autoreset_mode="SameStep")
obs, info = env.reset(seed=0)                                 # env i plays random.seed(0 + i); obs: int8 [N, 10, 10] on the GPU
total = torch.zeros(N, device="cuda")
for t in range(100):
    actions = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda")     # a policy network's output goes here
    obs, reward, terminated, truncated, info = env.step(actions)
    total += reward                                            # info["final_obs"] holds the terminal rows of SameStep auto-reset
print("mean return over 100 random steps:", float(total.mean()))

# the same loop fused into one launch (actions from the reproducible counter hash, or pass an int32 [K, N] tensor):
obs, reward_sum, done_count = env.rollout(1000, action_seed=1)
print("episodes finished in 1000 fused steps:", int(done_count.sum()))

# every other env type has the same surface
hosp = cge.HospitalVectorEnv(4096, autoreset_mode="NextStep")
obs, _ = hosp.reset(seed=7)                                   # float32 [4096, 243]
obs, rew, term, trunc, _ = hosp.step(torch.randint(0, 35, (4096,), dtype=torch.int32, device="cuda"))
snap = hosp.snapshot()                                        # checkpoint ...
hosp.restore(snap)                                            # ... and resume
print("hospital obs", tuple(obs.shape), "device bytes", hosp.device_bytes())
env.close(); hosp.close()
