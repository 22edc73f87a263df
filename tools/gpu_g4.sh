# usage: bash tools/gpu_g4.sh <tag> -- fwd_variant 2 (16 pixels x 4 Gaussians forward blend): parity tests, then the forward blend kernels' durations
# on the four workloads, default vs variant 2 (rocprofv3 kernel stats of the same step loop)
cd $GRAFT_REPO_ROOT
T=${1:-g4}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pixels_x_gaussians or alternative_kernels" 2>&1 | tail -15 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for w in headline clustered closeup headline:3000000; do
  for v in 1 2; do
    echo "== $w fwd_variant=$v: $(GOI_OPTIONS=fwd_variant=$v bash tools/kstats.sh tools/step_loop.py 30 $w 2>&1 | grep -E 'render_fwd|render_bwd_rows_k' | tr -s ' ' | tr '\n' ';')"
  done
done 2>&1 | tee gpurun_out/${T}_ab.txt
