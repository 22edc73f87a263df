cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_fuzz.py tests/test_gpu_geometry_cache.py tests/test_gpu_configs.py tests/test_gpu_views_in_flight.py tests/test_gpu_train_loop.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -8 > gpurun_out/r03_d_pytest.log
bash tools/kstats.sh tools/step_loop.py > gpurun_out/r03_d_kstats.txt 2>&1
python tools/ab_variants.py bwd_order 0 1 --bwd > gpurun_out/r03_d_ab_order.log 2>&1
tail -3 gpurun_out/r03_d_pytest.log; cat gpurun_out/r03_d_kstats.txt; tail -2 gpurun_out/r03_d_ab_order.log
