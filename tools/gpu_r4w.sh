cd $GRAFT_REPO_ROOT
timeout 1800 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r4w_pytest.log 2>&1
tail -8 gpurun_out/r4w_pytest.log
timeout 900 python bench.py > gpurun_out/r4w_bench.json 2> gpurun_out/r4w_bench.err
tail -c 600 gpurun_out/r4w_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4w_bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","value_fp32_flush","value_two_views_in_flight","value_semantics_only","train_iteration","workload_clustered","parity","roofline"):
    print(k, json.dumps(d.get(k))[:600])
PY
