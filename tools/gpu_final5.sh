# usage: bash tools/gpu_final5.sh <tag> -- round 5's ONE evidence set (final build): whole GPU suite, default bench, the bench at
# BASELINE's other sizes, kernel stats + PMC passes + per-dispatch timeline on headline / 3 M / close-up
cd $GRAFT_REPO_ROOT
T=${1:-r05}
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log
cp gpurun_out/parity_stats.json gpurun_out/${T}_parity_stats.json 2>/dev/null
python bench.py > gpurun_out/${T}_bench_1gpu.json 2> gpurun_out/${T}_bench.err
timeout 600 python bench.py --P 3000000 --steps 30 --no-cpu-baseline --no-clustered > gpurun_out/${T}_bench_config2_size.json 2> /dev/null
timeout 600 python bench.py --scene closeup --steps 30 --no-cpu-baseline --no-clustered > gpurun_out/${T}_bench_closeup.json 2> /dev/null
bash tools/pmc_steps.sh ${T}_pmc_headline headline > /dev/null 2>&1
bash tools/pmc_steps.sh ${T}_pmc_3m headline:3000000 > /dev/null 2>&1
bash tools/pmc_steps.sh ${T}_pmc_closeup closeup > /dev/null 2>&1
for w in headline clustered closeup headline:3000000; do echo "== $w"; bash tools/step_timeline.sh $w; done > gpurun_out/${T}_step_timeline.txt 2>&1
# the driver's own command under the profiler (kernel stats must agree with the bench line's roofline.avg_ms)
O=$GRAFT_REPO_ROOT/gpurun_out/${T}_benchprof; rm -rf $O; mkdir -p $O
(cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.txt; rm -f $O/*kernel_trace.csv $O/*/*kernel_trace.csv)
tail -3 gpurun_out/${T}_pytest.log; tail -c 600 gpurun_out/${T}_bench_1gpu.json; cat gpurun_out/${T}_pmc_headline/kernel_stats.txt
