cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_fuzz.py tests/test_gpu_operands.py tests/test_gpu_configs.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -8 > gpurun_out/r03_e_pytest.log
bash tools/kstats.sh tools/step_loop.py > gpurun_out/r03_e_kstats.txt 2>&1
tail -3 gpurun_out/r03_e_pytest.log; cat gpurun_out/r03_e_kstats.txt
