# round 5, first GPU batch: new tests, measured lane utilisation, per-kernel stats at the headline / 3 M / close-up sizes
cd $GRAFT_REPO_ROOT
T=r5a
python -m pytest tests/test_gpu_blend_stats.py tests/test_gpu_train_loop.py tests/test_gpu_parity.py::test_config1_exact_shape_forward_matches_oracle tests/test_gpu_dist.py tests/test_gpu_grad_pool.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log
python tools/blend_stats.py gpurun_out/${T}_blend_stats.json > gpurun_out/${T}_blend_stats.txt 2>&1
for w in headline headline:3000000 closeup; do
  bash tools/kstats.sh tools/step_loop.py 40 $w > gpurun_out/${T}_kstats_${w/:/_}.txt 2>&1
done
cat gpurun_out/${T}_pytest.log gpurun_out/${T}_blend_stats.txt gpurun_out/${T}_kstats_*.txt
