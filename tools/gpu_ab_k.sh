# usage: bash tools/gpu_ab_k.sh "<kernel name regex>" "<workloads>" <lib names under variants/ ...>
# same-box A/B of library builds: per-kernel average durations (rocprofv3) of the step loop on each workload.
# A name may carry runtime options: "<lib>@opt=val+opt2=val2" runs variants/lib_<lib>.so with GOI_OPTIONS=opt=val,opt2=val2
cd $GRAFT_REPO_ROOT
PAT=$1; WLS=$2; shift 2
cp goi_hyperplane_amd/lib/libgoi_raster.so /tmp/lib_keep.so
for spec in "$@"; do
  lib=${spec%%@*}; opts=""
  if [ "$lib" != "$spec" ]; then opts=$(echo ${spec#*@} | tr '+' ','); fi
  cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
  for w in $WLS; do
    echo "== $spec $w: $(GOI_OPTIONS=$opts bash tools/kstats.sh tools/step_loop.py 30 $w 2>&1 | grep -E "$PAT" | tr -s ' ' | tr '\n' ';')"
  done
done
cp /tmp/lib_keep.so goi_hyperplane_amd/lib/libgoi_raster.so
