cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh head a
cp variants/lib_a.so goi_hyperplane_amd/lib/libgoi_raster.so
