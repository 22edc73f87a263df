cd $GRAFT_REPO_ROOT
timeout 1800 python tools/soak_f64.py 2575 424242 2e-3 2>&1 | tail -12
