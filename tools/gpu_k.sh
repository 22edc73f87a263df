cd $GRAFT_REPO_ROOT
for rep in 1 2; do
patch -R -p1 -s < tools/build/aux.patch && python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== A (before)"; bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_bwd_rows|preprocess_bwd|preprocess_fwd|emit_k|reduce_rows"
patch -p1 -s < tools/build/aux.patch && python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== B (aux)"; bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_bwd_rows|preprocess_bwd|preprocess_fwd|emit_k|reduce_rows"
done
