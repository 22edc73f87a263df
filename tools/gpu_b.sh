set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_geometry_cache.py tests/test_gpu_fuzz.py tests/test_gpu_binding.py tests/test_gpu_configs.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -15 > gpurun_out/r03_b_pytest.log
python tools/ab_variants.py bwd_order 0 1 --bwd > gpurun_out/r03_b_ab_order.log 2>&1
python bench.py --no-cpu-baseline --no-train-iteration --no-two-streams --no-semantic-finetune > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err
python tools/diag_metric_config.py > gpurun_out/r03_b_diag3m.log 2>&1
tail -5 gpurun_out/r03_b_pytest.log; tail -3 gpurun_out/r03_b_ab_order.log
