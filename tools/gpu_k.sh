cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_operands.py tests/test_gpu_geometry_cache.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
patch -R -p1 -s < tools/build/rec.patch && python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== A"; bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_bwd_rows|render_fwd|emit_k|preprocess_fwd"
patch -p1 -s < tools/build/rec.patch && python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== B"; bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_bwd_rows|render_fwd|emit_k|preprocess_fwd"
done
