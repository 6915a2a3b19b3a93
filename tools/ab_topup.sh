#!/bin/bash
# usage: tools/ab_topup.sh T1 T2 ...  (GPU box): K sweep of the snake rollout with the launch-start ring top-up threshold forced to T
R=${GRAFT_REPO_ROOT:-$(pwd)}
for RND in 1 2; do
for T in "$@"; do
  echo "== topup $T (pass $RND)"
  CGE_SNAKE_TOPUP=$T CGE_AMD_LIBRARY=$R/tools/ab/libcge_tenv.so timeout -k 10 200 python3 $R/tools/probes/snake_k_sweep.py 2>&1 | grep "round 1" | grep -v "step()" | sed 's/(cge.*//' | tr '\n' ' '; echo
done
done
