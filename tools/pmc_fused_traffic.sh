# HBM traffic of the loss kernels (FETCH_SIZE / WRITE_SIZE, separate passes) while tools/fused_loss_time.py runs
OUT=$GRAFT_REPO_ROOT/gpurun_out/fut; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
 i=$((i+1))
 timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/fused_loss_time.py > $OUT/p$i.log 2>&1
 rm -f $OUT/p$i/*kernel_trace.csv
done
python - <<PY
import csv, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for r in csv.DictReader(open("$OUT/p%d/p%d_counter_collection.csv" % (i, i))):
        m = re.search(r"(\w+_k)\b", r["Kernel_Name"])
        if m and ("codebook" in m.group(1) or "decoder" in m.group(1)):
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-24s %16s %14s %12s   (per launch; counters in KiB; FETCH_SIZE doubled: on gfx950 it tallies wide coalesced reads at half their bytes, MI355X_MICROARCH.md HBM)" % ("kernel", "fetched MB (x2)", "written MB", "sum MB"))
for k, v in sorted(agg.items()):
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024 / 1e6
    w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024 / 1e6
    print("%-24s %16.1f %14.1f %12.1f" % (k, 2 * f, w, 2 * f + w))
PY
