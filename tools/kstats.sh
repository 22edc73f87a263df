# usage: tools/kstats.sh <python script> -- per-kernel average durations of our kernels (rocprofv3 --kernel-trace --stats)
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python $GRAFT_REPO_ROOT/$1 ${@:2} > $OUT/log.txt 2>&1
rm -f $OUT/*kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/ks_kernel_stats.csv")))
for r in rows:
    if "goi" in r["Name"]: import re; m=re.search(r"(\w+_k)\b", r["Name"]); print((m.group(1) if m else r["Name"][:40]).ljust(28), r["Calls"].rjust(5), "%10.1f us" % (float(r["AverageNs"])/1e3))
PY
