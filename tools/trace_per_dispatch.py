"""rocprofv3 --kernel-trace kernel_trace.csv -> one line per dispatch of the kernels matching a substring, in launch order: duration in us.
usage: python tools/trace_per_dispatch.py <dir> <kernel substring> [steps per launch]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
k = float(sys.argv[3]) if len(sys.argv) > 3 else None
for j, (s, e, nm) in enumerate(rows):
    d = (e - s) / 1e3
    print(f"{j:4d}  {d:10.1f} us" + (f"  {d / k:7.2f} us per step at {k:g} steps per launch" if k else "") + f"  {nm[:70]}")
