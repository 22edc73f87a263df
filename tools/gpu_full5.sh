# usage: bash tools/gpu_full5.sh <tag> -- the whole GPU suite + per-kernel stats on the four workloads (round 5)
cd $GRAFT_REPO_ROOT
T=${1:-r5}
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log
for w in headline clustered closeup headline:3000000; do
  bash tools/kstats.sh tools/step_loop.py 40 $w > gpurun_out/${T}_kstats_${w/:/_}.txt 2>&1
done
tail -4 gpurun_out/${T}_pytest.log; for w in headline clustered closeup headline_3000000; do echo "== $w"; cat gpurun_out/${T}_kstats_$w.txt; done
