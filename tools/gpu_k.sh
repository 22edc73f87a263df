cd $GRAFT_REPO_ROOT
python tools/band_balance.py --skew 2>&1 | tail -6
