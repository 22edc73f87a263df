# usage: bash tools/gpu_pre.sh <tag> -- parity suites that pin the per-Gaussian forward, then kernel stats on the four workloads
cd $GRAFT_REPO_ROOT
T=${1:-pre}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_operands.py tests/test_gpu_fuzz.py tests/test_gpu_depth_cut.py tests/test_gpu_speculative.py tests/test_gpu_clustered.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -6 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for w in headline clustered closeup headline:3000000; do
  echo "== $w"; bash tools/kstats.sh tools/step_loop.py 40 $w 2>&1 | grep -E "preprocess_fwd_k|compact_listed|emit_k|preprocess_bwd"
done 2>&1 | tee gpurun_out/${T}_kstats.txt
