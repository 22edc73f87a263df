cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_losses.py -m gpu -x -q 2>&1 | tail -2
GOI_SIMGRAD=2 timeout 300 python tools/fused_loss_time.py 2>&1 | tail -2
GOI_SIMGRAD=2 bash tools/kstats.sh tools/fused_loss_time.py 2>&1 | grep -E "simgrad"
