# usage: bash tools/collect_r05.sh <gpurun_out tag> -- copies ONE evidence set (tools/gpu_final5.sh) into profiles/ under the r05_ names
T=${1:-r05}; G=gpurun_out; P=profiles
cp $G/${T}_bench_1gpu.json $P/r05_bench_1gpu.json
cp $G/${T}_bench_config2_size.json $P/r05_bench_config2_size.json
cp $G/${T}_bench_closeup.json $P/r05_bench_closeup.json
cp $G/${T}_parity_stats.json $P/r05_parity_stats.json
cp $G/${T}_step_timeline.txt $P/r05_step_timeline.txt
tail -3 $G/${T}_pytest.log > $P/r05_pytest_gpu.txt
for w in headline 3m closeup; do
  cp $G/${T}_pmc_$w/kernel_stats.csv $P/r05_kernel_stats_$w.csv
  cp $G/${T}_pmc_$w/summary.txt $P/r05_pmc_summary_$w.txt
  cp $G/${T}_pmc_$w/traffic.json $P/r05_pmc_traffic_$w.json
done
cp $G/${T}_pmc_headline/traffic.json $P/r05_pmc_traffic.json   # (what bench.py quotes as roofline.traffic: the newest r*_pmc_traffic.json)
cp $G/${T}_benchprof/b_kernel_stats.csv $P/r05_bench_kernel_stats.csv
ls -la $P | grep r05
