cd $GRAFT_REPO_ROOT
T=r5b
python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_depth_cut.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${T}_pytest.log
python tools/sem_mode_host.py --profile > gpurun_out/${T}_sem_host.txt 2>&1
for pool in 1 0; do
  GOI_GRAD_POOL=$pool bash tools/kstats.sh tools/step_loop.py 40 closeup > gpurun_out/${T}_kstats_closeup_pool$pool.txt 2>&1
done
cat gpurun_out/${T}_pytest.log; head -30 gpurun_out/${T}_sem_host.txt; grep -h "preprocess_bwd_k\|reduce_rows" gpurun_out/${T}_kstats_closeup_pool*.txt
