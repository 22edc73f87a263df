cd $GRAFT_REPO_ROOT
cp goi_hyperplane_amd/lib/libgoi_raster.so /tmp/lib_keep.so
for lib in base pb4f; do
  cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
  for w in headline headline:3000000; do
    echo "== $lib $w two-pass: $(bash tools/kstats.sh tools/two_pass_dsh.py 20 $w 2>&1 | grep -E 'preprocess_bwd|sh_grad' | tr -s ' ' | tr '\n' ';')"
    echo "== $lib $w default : $(bash tools/kstats.sh tools/step_loop.py 20 $w 2>&1 | grep -E 'preprocess_bwd' | tr -s ' ' | tr '\n' ';')"
  done
done
cp /tmp/lib_keep.so goi_hyperplane_amd/lib/libgoi_raster.so
