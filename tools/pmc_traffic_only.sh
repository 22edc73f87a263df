#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (two rocprofv3 --pmc runs of a short bench).  usage: tools/pmc_traffic_only.sh <outdir-under-gpurun_out> [bench args...]
# (environment such as GOI_OPTIONS="bwd_order=0" is inherited: A/B the traffic of an option switch)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=2
for set in "FETCH_SIZE" "WRITE_SIZE TCC_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stage-timing --no-fp32-flush --no-two-streams --no-train-iteration --no-semantic-finetune "$@" > $OUT/p$i.log 2>&1
  rm -f $OUT/p$i/*kernel_trace.csv $OUT/p$i/*.db
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT $OUT/traffic.json "$(basename $OUT)" > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("$OUT/traffic.json"))
for k,v in d["kernels"].items():
    if "render_" in k or "reduce" in k: print(k, {a:round(b/1e6,1) for a,b in v.items() if isinstance(b,(int,float))})
PY
