cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
python tools/ab_variants.py fwd_variant 1 2>&1 | tail -1
bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "scan_|sweep|compact"
