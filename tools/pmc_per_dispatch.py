"""rocprofv3 --pmc counter_collection.csv -> one line per dispatch of the kernels matching a substring: dispatch order, counters.
usage: python tools/pmc_per_dispatch.py <dir> <kernel substring>"""
import csv
import glob
import sys
from collections import OrderedDict

rows = OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in rows.values() for k in v})
print("dispatch", *names, sep="\t")
for d in sorted(rows):
    print(d, *["%.4g" % rows[d].get(k, float("nan")) for k in names], sep="\t")
