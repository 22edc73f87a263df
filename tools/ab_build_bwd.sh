#!/bin/bash
# usage: ab_build_bwd.sh "<flags A>" ... ; backward stage times (timing builds: results may be wrong)
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  GOI_EXTRA_FLAGS="$flags" python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
  echo "== flags: [$flags]"
  timeout 300 python tools/ab_variants.py bwd_variant 0 --bwd 2>&1 | tail -1 | cut -c1-12,195-500
done
