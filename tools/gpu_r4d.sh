cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_clustered.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r4d_pytest.log
cp gpurun_out/parity_stats.json gpurun_out/r4d_parity_stats.json 2>/dev/null
bash tools/gpu_r4c.sh > gpurun_out/r4d_fwdcost.log 2>&1
cat gpurun_out/r4d_pytest.log; cat gpurun_out/r4c_fwd_masks_cost.txt
