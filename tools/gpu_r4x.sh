cd $GRAFT_REPO_ROOT
cp variants/lib_b.so goi_hyperplane_amd/lib/libgoi_raster.so; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_clustered.py -m gpu -q -x 2>&1 | tail -2
for lib in a b a b; do
  cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
  echo "== lib_$lib"
  bash tools/kstats.sh tools/step_loop.py 30 clustered 2>&1 | grep -E "reduce_rows_k|reduce_big_k"
  bash tools/kstats.sh tools/step_loop.py 30 2>&1 | grep -E "reduce_big_k"
done
cp variants/lib_a.so goi_hyperplane_amd/lib/libgoi_raster.so
