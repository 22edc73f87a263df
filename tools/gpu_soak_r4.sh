cd $GRAFT_REPO_ROOT
timeout 2100 python tools/flush_soak.py 1200 9301 1500 gpurun_out/r04_flush_equivalence.json > gpurun_out/r04_flush_soak.log 2>&1
tail -3 gpurun_out/r04_flush_soak.log
GOI_FUZZ_N=1200 GOI_FUZZ_SEED=9401 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r04_fuzz_soak.log 2>&1
tail -2 gpurun_out/r04_fuzz_soak.log
