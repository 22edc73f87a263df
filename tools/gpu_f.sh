cd $GRAFT_REPO_ROOT
bash tools/kstats.sh tools/step_loop.py > gpurun_out/r03_f_kstats.txt 2>&1
GOI_EXTRA_FLAGS="-DGOI_TMP_FWD_NOACC" python -m goi_hyperplane_amd.build --force > /dev/null 2>&1
bash tools/kstats.sh tools/step_loop.py > gpurun_out/r03_f_kstats_noacc.txt 2>&1
grep -E "preprocess_fwd_k|render_fwd_k|compact" gpurun_out/r03_f_kstats.txt; grep -E "render_fwd_k" gpurun_out/r03_f_kstats_noacc.txt
