cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do timeout 600 python tools/ab_variants.py bwd_variant 0 2 --bwd 2>&1 | grep variant | sed 's/.*blend_fwd/blend_fwd/'; done
