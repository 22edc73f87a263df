cd $GRAFT_REPO_ROOT
GOI_DEPTH_CUT=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_depth_cut.py -m gpu -x -q > gpurun_out/r4j_cut.log 2>&1
timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" > gpurun_out/r4j_pytest.log 2>&1
python tools/ab_step.py bwd_masks 1 > gpurun_out/r4j_step_cut.txt 2>&1
GOI_DEPTH_CUT=0 python tools/ab_step.py bwd_masks 1 > gpurun_out/r4j_step_nocut.txt 2>&1
bash tools/kstats.sh tools/step_loop.py 60 > gpurun_out/r4j_kstats.txt 2>&1
tail -25 gpurun_out/r4j_cut.log; tail -15 gpurun_out/r4j_pytest.log; cat gpurun_out/r4j_step_cut.txt gpurun_out/r4j_step_nocut.txt gpurun_out/r4j_kstats.txt
