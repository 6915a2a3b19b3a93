#!/bin/bash
# usage: tools/ab_env.sh <workload> <pattern> VAR val1 val2 ...   — like ab_kernels.sh, but varies an environment variable on the current library
set -o pipefail
W=$1; PAT=$2; VAR=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  export $VAR=$V
  T=${VAR}_$V
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$T -- python3 $R/bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ab_$T.log 2>&1 || { echo "$T failed"; tail -5 $R/gpurun_out/ab_$T.log; continue; }
  python3 - "$R/gpurun_out/ab_$T" "$T" "$PAT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[3] in r["Name"]: print("%-24s %-60s calls %5s avg %9.1f us min %8.1f max %9.1f"%(sys.argv[2], r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/ab_$T.log | tr '\n' ' '; echo
done
