cd $GRAFT_REPO_ROOT
python tools/ab_variants.py fwd_variant 1 2>&1 | tail -1 | cut -c60-320
python tools/ab_variants.py fwd_variant 1 2>&1 | tail -1 | cut -c60-320
