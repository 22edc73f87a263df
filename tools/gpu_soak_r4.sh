cd $GRAFT_REPO_ROOT
timeout 2400 python tools/flush_soak.py 1200 9301 1800 gpurun_out/r04_flush_equivalence.json > gpurun_out/r04_flush_soak.log 2>&1
GOI_FUZZ_N=1200 GOI_FUZZ_SEED=9401 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r04_fuzz_soak.log 2>&1
cp gpurun_out/parity_stats.json gpurun_out/r04_soak_parity_stats.json 2>/dev/null
tail -5 gpurun_out/r04_fuzz_soak.log; tail -60 gpurun_out/r04_flush_soak.log | head -80
