# usage: bash tools/gpu_g4_pmc.sh <tag> <libs under variants/ ...> -- forward blend, default (fwd_variant 1) vs the pixels x Gaussians
# builds (fwd_variant 2): kernel durations (rocprofv3 kernel stats) and instruction counters (one --pmc pass) on two workloads
cd $GRAFT_REPO_ROOT
T=$1; shift
cp goi_hyperplane_amd/lib/libgoi_raster.so /tmp/lib_keep.so
pmc() {  # $1 = workload
  O=$GRAFT_REPO_ROOT/gpurun_out/g4pmc; rm -rf $O; mkdir -p $O
  (cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -o p -- python $GRAFT_REPO_ROOT/tools/step_loop.py 8 $1 > $O/log.txt 2>&1)
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_fwd" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[r["Kernel_Name"].split("(")[0][-40:]]["VGPR"] = [float(r.get("VGPR_Count", 0) or 0)]
for k, c in agg.items():
    print("   pmc", k, {n: round(sum(v) / len(v) / (1e6 if n != "VGPR" else 1), 2) for n, v in sorted(c.items())})
PY
}
for w in headline clustered; do
  echo "== $w default: $(GOI_OPTIONS=fwd_variant=1 bash tools/kstats.sh tools/step_loop.py 30 $w 2>&1 | grep -E 'render_fwd' | tr -s ' ')"
  GOI_OPTIONS=fwd_variant=1 pmc $w
  for lib in "$@"; do
    cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
    echo "== $w $lib: $(GOI_OPTIONS=fwd_variant=2 bash tools/kstats.sh tools/step_loop.py 30 $w 2>&1 | grep -E 'render_fwd' | tr -s ' ')"
    GOI_OPTIONS=fwd_variant=2 pmc $w
    cp /tmp/lib_keep.so goi_hyperplane_amd/lib/libgoi_raster.so
  done
done 2>&1 | tee gpurun_out/${T}.txt
