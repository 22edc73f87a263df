cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward_match or golden or trace or edge" 2>&1 | tail -2
for rep in 1 2; do
patch -R -p1 -s < tools/build/x.patch && python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== A"; bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_fwd"
patch -p1 -s < tools/build/x.patch && python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== B"; bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_fwd"
done
