cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_clustered.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r4e_pytest.log
cp gpurun_out/parity_stats.json gpurun_out/r4e_parity_stats.json 2>/dev/null
timeout 900 python bench.py --steps 20 > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
cat gpurun_out/r4e_pytest.log; tail -5 gpurun_out/r4e_bench.err; head -c 3000 gpurun_out/r4e_bench.json
