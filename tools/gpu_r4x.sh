cd $GRAFT_REPO_ROOT
cp variants/lib_a.so goi_hyperplane_amd/lib/libgoi_raster.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_clustered.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu_ab.sh head a
cp variants/lib_a.so goi_hyperplane_amd/lib/libgoi_raster.so
