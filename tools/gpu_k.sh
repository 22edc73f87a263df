cd $GRAFT_REPO_ROOT
GOI_FUZZ_N=2500 GOI_FUZZ_SEED=77123 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "test_random_configuration" 2>&1 | grep -E "^E  .*Assertion|FAILED|passed|failed|Warning" | head -20
cp gpurun_out/parity_stats.json gpurun_out/r03_f_soak2_parity_stats.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_operands.py tests/test_gpu_speculative.py -m gpu -q 2>&1 | tail -2
