# forward: what recording the member masks costs (two builds, same box)
cd $GRAFT_REPO_ROOT
export GOI_OPTIONS=bwd_masks=0
bash tools/ab_build_fwd.sh "-DGOI_FWD_NO_MASKS" "" "-DGOI_FWD_NO_MASKS" "" > gpurun_out/r4c_fwd_masks_cost.txt 2>&1
unset GOI_OPTIONS
python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
cat gpurun_out/r4c_fwd_masks_cost.txt
