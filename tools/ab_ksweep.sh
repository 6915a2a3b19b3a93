#!/bin/bash
# usage: tools/ab_ksweep.sh lib1.so lib2.so ...  (GPU box, repo root): tools/probes/snake_k_sweep.py once per library, twice over
R=${GRAFT_REPO_ROOT:-$(pwd)}
for RND in 1 2; do
for L in "$@"; do
  echo "== $(basename $L .so) (pass $RND)"
  CGE_AMD_LIBRARY=$R/$L timeout -k 10 200 python3 $R/tools/probes/snake_k_sweep.py 2>&1 | grep "round 1" | sed 's/(cge.*//'
done
done
