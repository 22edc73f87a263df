cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_clustered.py tests/test_gpu_speculative.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -12 > gpurun_out/r4h_pytest.log
bash tools/kstats.sh tools/step_loop.py 24 clustered > gpurun_out/r4h_kstats_clustered.txt 2>&1
bash tools/kstats.sh tools/step_loop.py 40 > gpurun_out/r4h_kstats.txt 2>&1
cat gpurun_out/r4h_pytest.log gpurun_out/r4h_kstats_clustered.txt gpurun_out/r4h_kstats.txt
