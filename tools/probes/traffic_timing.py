"""Per-phase wall-clock of traffic's rollout step: tools/build_variant.sh ttiming traffic.hip -DCGE_TRAFFIC_TIMING, then
CGE_AMD_LIBRARY=tools/ab/libcge_ttiming.so python tools/probes/traffic_timing.py.  One wave = 16 envs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import custom_gymnasium_environments_amd as cge
from custom_gymnasium_environments_amd import _native
L = ctypes.CDLL(_native.LIB_PATH)
env = cge.TrafficVectorEnv(262144, device="cuda:0")
env.reset(seed=1)
buf = (ctypes.c_ulonglong * 16)()
names = ["kernel start (record load)", "prepare: ring top-up + view (+ the previous step's stores draining)", "actions (hash) + lights", "draws + their effects",
         "process vehicles + reward", "final_obs / reset", "observation row"]
for chunk in (0, 1, 2, 16, 17):                     # steps 0-89 (vehicles still spawning every step), then 480-539 (saturated)
    if chunk == 16:
        env.rollout(390, action_seed=7, t0=90)
        torch.cuda.synchronize()
        L.cge_traffic_debug_timing(buf, 1)
    env.rollout(30, action_seed=7, t0=chunk * 30, trajectory=True)
    torch.cuda.synchronize()
    L.cge_traffic_debug_timing(buf, 1)
    n = max(1, buf[15])
    print(f"steps {chunk*30}..{chunk*30+29}: wave-steps {buf[15]}, total {sum(buf[k] for k in range(7)) * 10.0 / n / 1e3:.1f} us per wave-step")
    for k, nm in enumerate(names):
        print(f"   {nm:48s} {buf[k] * 10.0 / n / 1e3:8.2f} us")

# the step() path (one launch per step: record load, 16-word window fill and flush, LDS-staged rows)
acts = torch.randint(0, 3, (262144, 9), dtype=torch.int32, device="cuda")
L.cge_traffic_debug_timing(buf, 1)
for _ in range(30):
    env.step(acts)
torch.cuda.synchronize()
L.cge_traffic_debug_timing(buf, 1)
n = max(1, buf[15])
print(f"step() x 30: wave-steps {buf[15]}, total {sum(buf[k] for k in range(7)) * 10.0 / n / 1e3:.1f} us per wave-step (slot 0 includes the record load and the window fill)")
for k, nm in enumerate(names):
    print(f"   {nm:48s} {buf[k] * 10.0 / n / 1e3:8.2f} us")
