#!/bin/bash
# usage: ab_build_fwd.sh "<flags A>" "<flags B>" ... ; rebuilds with GOI_EXTRA_FLAGS and prints the forward-chain stage times of each build (on the GPU box)
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  GOI_EXTRA_FLAGS="$flags" python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
  echo "== flags: [$flags]"
  timeout 300 python tools/ab_variants.py fwd_variant 1 2>&1 | tail -1 | cut -c60-320
done
