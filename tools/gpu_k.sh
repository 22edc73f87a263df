cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "record_backward or alternative_kernels or forward_backward_match or golden or edge or metric_config or semantics_only" 2>&1 | tail -3
python tools/ab_variants.py bwd_records 1 --bwd 2>&1 | tail -1 | cut -c1-100,190-520
bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "render_bwd|reduce_rows|preprocess_bwd"
