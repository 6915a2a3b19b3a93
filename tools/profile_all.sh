#!/bin/bash
# usage: tools/profile_all.sh [a|b]   (GPU box, repo root).  Part a: the two 1M-env workloads and traffic; part b: the five 131k-env ones
# (each half fits one 20-minute gpurun call).  Summaries land in gpurun_out/prof_r3_*; tools/collect_profiles.py copies them to profiles/.
set -o pipefail
cd $GRAFT_REPO_ROOT
PART=${1:-ab}
if [[ $PART == *a* ]]; then
  # the snake profile uses the DRIVER's arguments (bench.py --gpus 1 --steps 20 --warmup 5), so profiles/traffic.json describes
  # exactly the kernel the driver's bench line times; a second snake profile at the default K = 200 sits beside it
  bash tools/profile.sh r3_snake --steps 20 --warmup 5 > gpurun_out/pa_snake.log 2>&1; echo snake $?
  bash tools/profile.sh r3_snake_k200 --steps 200 --warmup 20 > gpurun_out/pa_snake_k200.log 2>&1; echo snake_k200 $?
  bash tools/profile.sh r3_crypto --workload crypto_1m --steps 40 --warmup 5 > gpurun_out/pa_crypto.log 2>&1; echo crypto $?
  bash tools/profile.sh r3_traffic --workload traffic_262k --steps 200 --warmup 5 > gpurun_out/pa_traffic.log 2>&1; echo traffic $?
fi
if [[ $PART == *b* ]]; then
  for w in parking climate fleet manufacturing hospital; do bash tools/profile.sh r3_$w --workload ${w}_131k --steps 40 --warmup 5 > gpurun_out/pa_$w.log 2>&1; echo $w $?; done
fi
