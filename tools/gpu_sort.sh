cd $GRAFT_REPO_ROOT
T=${1:-sort1}
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_configs.py tests/test_gpu_knn.py tests/test_gpu_clustered.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for w in headline clustered closeup headline:3000000; do
  echo "== $w"; bash tools/kstats.sh tools/step_loop.py 40 $w 2>&1 | grep -E "sweep_pass|compact_listed|emit_k|scan_"
done 2>&1 | tee gpurun_out/${T}_kstats.txt
