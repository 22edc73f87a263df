cd $GRAFT_REPO_ROOT
GOI_FUZZ_N=1200 GOI_FUZZ_SEED=9301 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "test_random_configuration" 2>&1 | tail -6 > gpurun_out/r03_soak_fuzz.log
cp gpurun_out/parity_stats.json gpurun_out/r03_soak_parity_stats.json
timeout 900 python tools/spec_soak.py > gpurun_out/r03_soak_spec.log 2>&1
tail -4 gpurun_out/r03_soak_fuzz.log; tail -6 gpurun_out/r03_soak_spec.log
