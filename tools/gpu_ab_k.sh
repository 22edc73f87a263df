# usage: bash tools/gpu_ab_k.sh "<kernel name regex>" "<workloads>" <lib names under variants/ ...>
# same-box A/B of library builds: per-kernel average durations (rocprofv3) of the step loop on each workload
cd $GRAFT_REPO_ROOT
PAT=$1; WLS=$2; shift 2
cp goi_hyperplane_amd/lib/libgoi_raster.so /tmp/lib_keep.so
for lib in "$@"; do
  cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
  for w in $WLS; do
    echo "== $lib $w: $(bash tools/kstats.sh tools/step_loop.py 30 $w 2>&1 | grep -E "$PAT" | tr -s ' ' | tr '\n' ';')"
  done
done
cp /tmp/lib_keep.so goi_hyperplane_amd/lib/libgoi_raster.so
