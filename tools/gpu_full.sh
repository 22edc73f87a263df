# usage: bash tools/gpu_full.sh <tag>   -- full GPU suite + default bench + per-kernel stats, outputs in gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT
T=${1:-run}
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.log
cp gpurun_out/parity_stats.json gpurun_out/${T}_parity_stats.json 2>/dev/null
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
bash tools/kstats.sh tools/step_loop.py > gpurun_out/${T}_kstats.txt 2>&1
cp gpurun_out/ks/ks_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv 2>/dev/null
tail -4 gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_kstats.txt | head -20
