cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "record_backward or alternative_kernels or forward_backward_match or golden or edge or metric_config" 2>&1 | tail -5
python tools/ab_variants.py bwd_records 0 1 --bwd 2>&1 | tail -2 | cut -c1-60,190-520
