cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward_match_oracle or golden" 2>&1 | tail -3
bash tools/kstats.sh tools/step_loop.py 2>&1 | grep -E "emit_k|reduce_rows|preprocess_fwd"
