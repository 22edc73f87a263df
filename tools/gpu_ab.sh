cd $GRAFT_REPO_ROOT
python tools/ab_variants.py cull_variant 1 2 --bwd 2>&1 | tail -2
python tools/ab_step.py cull_variant 1 2 2>&1 | tail -2
