cd $GRAFT_REPO_ROOT
GOI_FUZZ_N=4000 GOI_FUZZ_SEED=424242 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "test_random_configuration" 2>&1 | grep -E "^E  .*Assertion|FAILED|passed|failed|Warning" | head -20
cp gpurun_out/parity_stats.json gpurun_out/r03_f_soak3_parity_stats.json
