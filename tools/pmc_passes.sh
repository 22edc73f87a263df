#!/bin/bash
# Collects PMC counters for the bench in separate passes (one rocprofv3 run per counter set).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_ATOMIC_sum" \
  "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stage-timing --no-fp32-flush --no-two-streams --no-train-iteration "$@" > $OUT/p$i.log 2>&1
  rm -f $OUT/p$i/*kernel_trace.csv $OUT/p$i/*.db
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
