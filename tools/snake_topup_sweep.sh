#!/bin/bash
# usage: tools/snake_topup_sweep.sh <lib.so built with -DCGE_SNAKE_TOPUP_ENV> <K> <thresholds...>   (GPU box, repo root)
# snake_1m fused rollout, K steps per launch: us per step right after the reset (the contract's W = 5) and in the batch's steady state
# (bench.py: steady_state, 2,000 steps later) for every launch-start top-up threshold of the digit rings (cge_snake_rollout: dq_topup).
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$1; K=$2; shift 2
for RND in 1 2; do
for T in "$@"; do
  CGE_AMD_LIBRARY=$R/$LIB CGE_SNAKE_TOPUP=$T timeout -k 10 300 python3 $R/bench.py --steps $K --warmup 5 --repeats 5 --no-cpu-baseline --no-final-obs-leg > $R/gpurun_out/sweep_t$T.json 2> $R/gpurun_out/sweep_t$T.err || { echo "T=$T failed"; tail -3 $R/gpurun_out/sweep_t$T.err; continue; }
  python3 - "$R/gpurun_out/sweep_t$T.json" "$T" "$K" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s=d["steady_state"]
print("k=%s threshold %3s  fresh regions %s us/step | steady state %.1f us/step (events), wall %s" % (sys.argv[3], sys.argv[2], [round(x*1e3,1) for x in d["spread"]["all_ms_per_step"]], s["us_per_step"], [round(x*1e3,1) for x in s["spread"]["all_ms_per_step"]]))
PY
done
done
