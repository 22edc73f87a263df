# usage: bash tools/gpu_r5d.sh <tag> -- round-5 baseline: whole GPU suite, default bench, per-kernel stats on four workloads
cd $GRAFT_REPO_ROOT
T=${1:-r5d}
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
for w in headline clustered closeup headline:3000000; do
  bash tools/kstats.sh tools/step_loop.py 40 $w > gpurun_out/${T}_kstats_${w/:/_}.txt 2>&1
done
tail -4 gpurun_out/${T}_pytest.log; tail -c 3000 gpurun_out/${T}_bench.json; for w in headline clustered closeup headline_3000000; do echo "== $w"; cat gpurun_out/${T}_kstats_$w.txt; done
