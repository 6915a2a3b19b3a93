#!/bin/bash
# usage: tools/profile.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# Writes rocprofv3 kernel stats and PMC passes (each in its own run) under gpurun_out/prof_<tag>/.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline --no-final-obs-leg --steady-steps 0 --manifest $OUT/manifest.json > $OUT/trace.log 2>&1 || exit 1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py "$@" --no-cpu-baseline --no-final-obs-leg --steady-steps 0 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed" >> $OUT/errors.log
done
python3 $R/tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
