cd $GRAFT_REPO_ROOT
# usage: bash tools/gpu_ab.sh <lib names under variants/ ...>: same-box A/B of library builds (backward stage times, two rounds)
LIBS=${@:-head a}
for rep in 1 2; do for lib in $LIBS; do
  cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
  echo "== lib_$lib $(timeout 600 python tools/ab_variants.py bwd_variant 0 --bwd 2>&1 | grep variant | sed 's/.*blend_fwd/blend_fwd/')"
done; done
