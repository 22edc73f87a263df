cd $GRAFT_REPO_ROOT
T=${1:-r5c}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_operands.py tests/test_gpu_grad_pool.py tests/test_gpu_sh_factored.py tests/test_gpu_fuzz.py tests/test_gpu_binding.py tests/test_gpu_clustered.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -8 > gpurun_out/${T}_pytest.log
for w in headline headline:3000000 closeup; do
  bash tools/kstats.sh tools/step_loop.py 40 $w > gpurun_out/${T}_kstats_${w/:/_}.txt 2>&1
done
tail -4 gpurun_out/${T}_pytest.log; grep -H "preprocess_bwd_k\|reduce_rows_k\|emit_k" gpurun_out/${T}_kstats_*.txt
