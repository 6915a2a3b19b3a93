#!/bin/bash
# usage: tools/ab_step.sh <workload> <pattern> lib1.so lib2.so ...   (GPU box, repo root)
# step()-path A/B: bench.py --path step under rocprofv3 --kernel-trace --stats once per library, two rounds (box drift shows up as
# a difference between the rounds of one library).
set -o pipefail
W=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for RND in 1 2; do
for L in "$@"; do
  T=$(basename $L .so)
  export CGE_AMD_LIBRARY=$R/$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abs_$T -- python3 $R/bench.py --workload $W --steps 60 --warmup 5 --no-cpu-baseline > $R/gpurun_out/abs_$T.log 2>&1 || { echo "$T failed"; tail -5 $R/gpurun_out/abs_$T.log; continue; }
  python3 - "$R/gpurun_out/abs_$T" "$T" "$PAT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[3] in r["Name"]: print("%-16s %-58s calls %5s avg %9.1f us min %8.1f max %9.1f"%(sys.argv[2], r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $R/gpurun_out/abs_$T
done
done
