# usage: bash tools/gpu_quick.sh <tag> -- parity / speculative / fuzz / cache / config suites + per-kernel stats of the step loop
cd $GRAFT_REPO_ROOT
T=${1:-q}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_fuzz.py tests/test_gpu_geometry_cache.py tests/test_gpu_configs.py tests/test_gpu_operands.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -8 > gpurun_out/${T}_pytest.log
bash tools/kstats.sh tools/step_loop.py > gpurun_out/${T}_kstats.txt 2>&1
tail -3 gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_kstats.txt
