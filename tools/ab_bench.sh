#!/bin/bash
# usage: tools/ab_bench.sh <workload> <steps> <warmup> lib1.so lib2.so ...  (GPU box, repo root)
# bench.py once per library, twice over; prints rollout / step() us per step and roofline fractions
W=$1; K=$2; WU=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
for RND in 1 2; do
for L in "$@"; do
  T=$(basename $L .so)
  CGE_AMD_LIBRARY=$R/$L timeout -k 10 ${ABTMO:-400} python3 $R/bench.py --workload $W --steps $K --warmup $WU --no-cpu-baseline > $R/gpurun_out/abb_$T.json 2> $R/gpurun_out/abb_$T.err || { echo "$T failed"; tail -3 $R/gpurun_out/abb_$T.err; continue; }
  python3 - "$R/gpurun_out/abb_$T.json" "$T" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; a=d["api_step"]["roofline"]
print("%-14s rollout %8.2f us/step frac %.3f (%s) | step() %8.2f us frac %.3f (%s)"%(sys.argv[2], r["avg_launch_us"]/(r["env_steps_per_launch"]/d["config"]["envs_per_gpu"]), r["frac"], r["kernel"][-28:], a["avg_launch_us"], a["frac"], a["kernel"][-28:]))
PY
done
done
