#!/bin/bash
# usage: tools/bench_all.sh <out.jsonl>   (GPU box, repo root): every workload, one box, one call.  snake_1m with the driver's own
# arguments; every other single-type workload over ONE WHOLE EPISODE from its start (--episode), cpu_baseline included; the two
# heterogeneous placements with the default 200 steps.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/bench_all.jsonl}
: > $OUT
for W in snake_1m crypto_1m traffic_262k parking_131k climate_131k fleet_131k manufacturing_131k hospital_131k hetero_131k hetero_split_131k; do
  case $W in
    snake_1m) EXTRA="--steps 20 --warmup 5" ;;
    hetero*) EXTRA="--no-cpu-baseline --repeats 3" ;;
    *) EXTRA="--episode --repeats 5" ;;
  esac
  timeout -k 10 900 python3 $R/bench.py --workload $W $EXTRA >> $OUT 2>> $R/gpurun_out/bench_all.err || echo "{\"workload\": \"$W\", \"failed\": true}" >> $OUT
  echo "$W done"
done
