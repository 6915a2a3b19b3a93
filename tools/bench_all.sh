#!/bin/bash
# usage: tools/bench_all.sh <out.jsonl>   (GPU box, repo root): bench.py with its default arguments for every workload, one box, one call
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/bench_all.jsonl}
: > $OUT
for W in snake_1m crypto_1m traffic_262k parking_131k climate_131k fleet_131k manufacturing_131k hospital_131k hetero_131k hetero_split_131k; do
  EXTRA="--no-cpu-baseline"; [ $W = snake_1m ] && EXTRA=""
  [ $W = traffic_262k ] && EXTRA="$EXTRA --steps 1000"
  timeout -k 10 500 python3 $R/bench.py --workload $W $EXTRA >> $OUT 2>> $R/gpurun_out/bench_all.err || echo "{\"workload\": \"$W\", \"failed\": true}" >> $OUT
  echo "$W done"
done
