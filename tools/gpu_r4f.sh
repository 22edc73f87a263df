cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_clustered.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r4f_pytest.log
cp gpurun_out/parity_stats.json gpurun_out/r4f_parity_stats.json 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -6 > gpurun_out/r4f_pytest2.log
timeout 900 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err
cat gpurun_out/r4f_pytest.log gpurun_out/r4f_pytest2.log; tail -3 gpurun_out/r4f_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4f_bench.json"))
print(d["value"], d["ms_per_step"], d["stages"] and {k:v["ms"] for k,v in d["stages"].items()})
print(json.dumps(d["workload_clustered"]))
PY
