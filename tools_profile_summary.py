"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  {Name:70.70s} calls={Calls:>6s} avg_ns={AverageNs:>12s} min={MinNs:>10s} max={MaxNs:>10s} pct={Percentage}".format(**row))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== pmc", os.path.relpath(f, out))
        for k, cs in agg.items():
            if "cge" not in k:
                continue
            for c, v in cs.items():
                print(f"  {k[:60]:60s} {c:24s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
