"""one-line digest of a bench.py JSON line (used inside gpurun command strings)"""
import json
import sys
d = json.load(open(sys.argv[1]))
r, a = d["roofline"], d["api_step"]["roofline"]
k = d["steps"]
print(sys.argv[1], "| rollout us/step", round(r["avg_launch_us"] / r["env_steps_per_launch"] * d["config"]["envs_per_gpu"], 2), "frac", round(r["frac"], 3),
      "| step us", round(a["avg_launch_us"], 2), "frac", round(a["frac"], 3), "| copy GB/s", round(r.get("peak_measured") or 0))
