#!/bin/bash
# PMC counters of a plain step loop (tools/step_loop.py) on a named workload, one rocprofv3 run per counter set, plus the
# kernel stats of the same loop.  usage: tools/pmc_steps.sh <outdir-under-gpurun_out> <workload, e.g. headline:3000000 | closeup>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; W=$2
mkdir -p $OUT
bash $GRAFT_REPO_ROOT/tools/kstats.sh tools/step_loop.py 40 $W > $OUT/kernel_stats.txt 2>&1
cp $GRAFT_REPO_ROOT/gpurun_out/ks/ks_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_ATOMIC_sum" \
  "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/step_loop.py 8 $W > $OUT/p$i.log 2>&1
  rm -f $OUT/p$i/*kernel_trace.csv $OUT/p$i/*.db
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT $OUT/traffic.json "$1 ($W)" > /dev/null 2>&1
cat $OUT/kernel_stats.txt
