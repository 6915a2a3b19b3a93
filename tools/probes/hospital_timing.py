"""Per-phase wall-clock of hospital's step, one wave = 16 envs (tools/build_variant.sh htiming hospital.hip -DCGE_HOSP_TIMING).
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
names = ["action + arrivals", "prepare: next step's generator words (twists + ring units)", "action effects + treatments", "assign_dept x3", "doctor fatigue",
         "update_queue x3 + flags", "nurses", "(-)", "events + episode end", "reset", "rows: LDS image + stores", "10 machines", "15 medicines", "terminal row"]
for chunk in range(3):
    env.rollout(20, action_seed=7, t0=chunk * 20, trajectory=True)
    torch.cuda.synchronize()
    L.cge_hospital_debug_timing(buf, 1)
    n = max(1, buf[15])
    print(f"steps {chunk*20}..{chunk*20+19}: wave-steps {buf[15]}, total {sum(buf[k] for k in range(14)) * 10.0 / n / 1e3:.1f} us per wave-step")
    for k, nm in enumerate(names):
        print(f"   {nm:30s} {buf[k] * 10.0 / n / 1e3:8.2f} us")
