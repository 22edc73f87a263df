cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_gpu_depth_cut.py -m gpu -x -q > gpurun_out/r4k_cut.log 2>&1
python tools/ab_step.py bwd_masks 1 > gpurun_out/r4k_step_cut.txt 2>&1
GOI_DEPTH_CUT=0 python tools/ab_step.py bwd_masks 1 > gpurun_out/r4k_step_nocut.txt 2>&1
bash tools/kstats.sh tools/step_loop.py 100 > gpurun_out/r4k_kstats.txt 2>&1
tail -25 gpurun_out/r4k_cut.log; cat gpurun_out/r4k_step_cut.txt gpurun_out/r4k_step_nocut.txt gpurun_out/r4k_kstats.txt
