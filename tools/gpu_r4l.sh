cd $GRAFT_REPO_ROOT
T=${1:-r4l}
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/${T}_pytest_full.log 2>&1
tail -15 gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
cp gpurun_out/parity_stats.json gpurun_out/${T}_parity_stats.json 2>/dev/null
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
bash tools/kstats.sh tools/step_loop.py > gpurun_out/${T}_kstats.txt 2>&1
cp gpurun_out/ks/ks_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv 2>/dev/null
cat gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_bench.err; head -c 1500 gpurun_out/${T}_bench.json; echo; cat gpurun_out/${T}_kstats.txt
