cd $GRAFT_REPO_ROOT
T=${1:-emit1}
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_configs.py tests/test_gpu_clustered.py tests/test_gpu_depth_cut.py tests/test_gpu_geometry_cache.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for w in headline clustered closeup headline:3000000; do
  echo "== $w"; bash tools/step_timeline.sh $w 2>&1 | grep -E "emit|tile_ranges|step span"
done 2>&1 | tee gpurun_out/${T}_kstats.txt
