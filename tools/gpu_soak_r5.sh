# round-end soaks of round 5 (final build): fuzz sweep (two fresh seeds), fire-and-forget frames (the stamped read-back under load)
cd $GRAFT_REPO_ROOT
GOI_FUZZ_N=1200 GOI_FUZZ_SEED=9501 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r05_soak_fuzz.log
cp gpurun_out/parity_stats.json gpurun_out/r05_soak_parity_stats.json
GOI_FUZZ_N=1500 GOI_FUZZ_SEED=66001 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r05_soak_fuzz.log
timeout 900 python tools/spec_soak.py > gpurun_out/r05_soak_spec.log 2>&1
cat gpurun_out/r05_soak_fuzz.log; tail -6 gpurun_out/r05_soak_spec.log
