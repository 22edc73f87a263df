cd $GRAFT_REPO_ROOT
for cfg in "-DGOI_REDUCE_INFLIGHT=32" "-DGOI_REDUCE_INFLIGHT=32 -DGOI_REDUCE_GPQ=1" "-DGOI_REDUCE_INFLIGHT=32 -DGOI_REDUCE_GPQ=3"; do
GOI_EXTRA_FLAGS="$cfg" python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
echo "== $cfg"
python tools/ab_variants.py bwd_order 1 --bwd 2>&1 | tail -1 | grep -o "'preprocess_bwd': [0-9.]*"
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_backward_match_oracle or golden or semantics_only" 2>&1 | tail -1
