# usage: bash tools/gpu_r4b.sh <tag> -- full GPU suite, per-kernel stats of the step loop, A/B of bwd_masks
cd $GRAFT_REPO_ROOT
T=${1:-r4b}
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.log
bash tools/kstats.sh tools/step_loop.py > gpurun_out/${T}_kstats.txt 2>&1
cp gpurun_out/ks/ks_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv 2>/dev/null
python tools/ab_variants.py bwd_masks 0 1 --bwd > gpurun_out/${T}_ab_stage.txt 2>&1
python tools/ab_step.py bwd_masks 0 1 > gpurun_out/${T}_ab_step.txt 2>&1
tail -5 gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_kstats.txt gpurun_out/${T}_ab_stage.txt gpurun_out/${T}_ab_step.txt
