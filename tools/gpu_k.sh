cd $GRAFT_REPO_ROOT
bash tools/kstats.sh tools/factored_loop.py 2>&1 | grep -E "preprocess_bwd|reduce_rows"
