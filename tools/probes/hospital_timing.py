"""Per-phase wall-clock of hospital's wave_step (build with -DCGE_HOSP_TIMING into tools/ab/libcge_htiming.so).
usage: CGE_AMD_LIBRARY=tools/ab/libcge_htiming.so python tools/probes/hospital_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import custom_gymnasium_environments_amd as cge
from custom_gymnasium_environments_amd import _native
L = ctypes.CDLL(_native.LIB_PATH)
env = cge.HospitalVectorEnv(131072, device="cuda:0")
env.reset(seed=1)
buf = (ctypes.c_ulonglong * 16)()
names = ["action + arrivals", "DOC/BED load", "action effects + treatments", "assign_dept x3", "fatigue + store + emit", "update_queue x3 + done",
         "NUR", "EQ", "events + misc emit", "reset", "  EQ: load", "  EQ: 10 machines", "  EQ: 15 medicines", "  top-of-step refill"]
for chunk in range(3):
    env.rollout(20, action_seed=7, t0=chunk * 20, trajectory=True)
    torch.cuda.synchronize()
    L.cge_hospital_debug_timing(buf, 1)
    n = max(1, buf[15])
    print(f"steps {chunk*20}..{chunk*20+19}: wave-steps {buf[15]}, total {sum(buf[k] for k in range(14)) * 10.0 / n / 1e3:.1f} us per wave-step")
    for k, nm in enumerate(names):
        print(f"   {nm:30s} {buf[k] * 10.0 / n / 1e3:8.2f} us")
