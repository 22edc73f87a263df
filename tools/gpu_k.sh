cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward_match or golden or edge or alternative_kernels or culled or full_size or larger_scene" 2>&1 | tail -3
python tools/ab_variants.py fwd_variant 1 2>&1 | tail -1
bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "sweep_pass|compact"
