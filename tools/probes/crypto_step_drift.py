"""Does crypto step() slow down over a run?  Per-chunk timing of 240 step() calls at 1M envs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import custom_gymnasium_environments_amd as cge
env = cge.CryptoVectorEnv(1 << 20, device="cuda:0", autoreset_mode="SameStep", reuse_buffers=True)
env.reset(seed=1)
acts = torch.randint(0, 5, (1 << 20,), dtype=torch.int32, device="cuda")
for _ in range(5):
    env.step(acts)
torch.cuda.synchronize()
for chunk in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        env.step(acts)
    b.record()
    torch.cuda.synchronize()
    print(f"steps {5 + chunk * 20:4d}..{24 + chunk * 20:4d}: {a.elapsed_time(b) * 1e3 / 20:7.1f} us per step")
