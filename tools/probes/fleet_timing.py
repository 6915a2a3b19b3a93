"""Per-phase wall-clock of fleet's dense kernel (build with -DCGE_FLEET_TIMING into tools/ab/libcge_timing.so).
usage: CGE_AMD_LIBRARY=tools/ab/libcge_timing.so python tools/probes/fleet_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import custom_gymnasium_environments_amd as cge
from custom_gymnasium_environments_amd import _native
L = ctypes.CDLL(_native.LIB_PATH)
env = cge.FleetVectorEnv(131072, device="cuda:0")
env.reset(seed=1)
buf = (ctypes.c_ulonglong * 16)()
names = ["entry+load", "fill L", "fill P", "traffic/weather", "final_obs pass", "do_reset", "obs pass", "flush", "store", "counters scan", "", "", "entry", "L loads", "P loads"]
for chunk in range(4):
    env.rollout(40, action_seed=7, t0=chunk * 40, trajectory=True)
    torch.cuda.synchronize()
    L.cge_fleet_debug_timing(buf, 1)
    n = max(1, buf[15])
    print(f"steps {chunk*40}..{chunk*40+39}: wave-iterations {buf[15]}")
    for k, nm in enumerate(names):
        if not nm: continue
        print(f"   {nm:18s} {buf[k] * 10.0 / n / 1e3:8.2f} us per wave iteration")
    print(f"   slowest wave {buf[10] * 10.0 / 1e3:.1f} us; wave iterations with a serial refill: {buf[11]}")
