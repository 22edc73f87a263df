# usage: bash tools/gpu_final.sh <tag>: full GPU suite, default bench, per-kernel stats, the five PMC passes, and the traffic of bwd_order 0
cd $GRAFT_REPO_ROOT
T=${1:-final}
bash tools/gpu_full.sh $T > gpurun_out/${T}_full.log 2>&1
bash tools/pmc_passes.sh pmc_$T > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$T gpurun_out/pmc_$T/traffic.json $T > /dev/null 2>&1
GOI_OPTIONS="bwd_order=0" bash tools/pmc_traffic_only.sh pmc_${T}_order0 > gpurun_out/${T}_traffic_order0.txt 2>&1
tail -3 gpurun_out/${T}_pytest.log; head -14 gpurun_out/${T}_kstats.txt; cat gpurun_out/${T}_traffic_order0.txt
