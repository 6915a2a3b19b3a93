"""Copies the rocprofv3 summaries of one round from gpurun_out/prof_<tag>/ into profiles/ (tracked) and merges their per-kernel HBM
traffic into profiles/traffic.json.  usage: python tools/collect_profiles.py r03 r3_snake:snake r3_snake_k200:snake_k200 ...
(later tags do NOT overwrite traffic entries of earlier ones: list the profile of the driver's command first)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, pairs = sys.argv[1], [a.split(":") for a in sys.argv[2:]]
tj = os.path.join(ROOT, "profiles", "traffic.json")
traffic = {}
seen = set()
for tag, name in pairs:
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", f"{rnd}_{name}_summary.txt"))
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{rnd}_{name}_kernel_stats.csv"))
    with open(os.path.join(src, "traffic.json")) as f:
        for k, v in json.load(f).items():
            if k not in seen:
                v["profile"] = f"{rnd}_{name}_summary.txt"
                traffic[k] = v
                seen.add(k)
with open(tj, "w") as f:
    json.dump(traffic, f, indent=1)
print("kernels in traffic.json:", *traffic, sep="\n  ")
