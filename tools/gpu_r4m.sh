cd $GRAFT_REPO_ROOT
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_speculative.py tests/test_gpu_depth_cut.py tests/test_gpu_geometry_cache.py tests/test_gpu_binding.py tests/test_gpu_views_in_flight.py -m gpu -x -q > gpurun_out/r4m_pytest.log 2>&1
python tools/ab_step.py bwd_masks 1 > gpurun_out/r4m_step.txt 2>&1
bash tools/kstats.sh tools/step_loop.py > gpurun_out/r4m_kstats.txt 2>&1
tail -6 gpurun_out/r4m_pytest.log; cat gpurun_out/r4m_step.txt; python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/ks/ks_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.2f%%" % (100*float(r['TotalDurationNs'])/tot))
print(tot/40/1e3)
PY
