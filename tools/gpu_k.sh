cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_speculative.py tests/test_gpu_train_loop.py tests/test_gpu_geometry_cache.py tests/test_gpu_binding.py -m gpu -q 2>&1 | tail -3
