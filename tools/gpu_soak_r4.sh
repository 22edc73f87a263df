# round-end soaks of round 4 (final build): fuzz sweep (two seeds), fire-and-forget frames, the secondary sizes of the bench
cd $GRAFT_REPO_ROOT
GOI_FUZZ_N=1200 GOI_FUZZ_SEED=9301 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r04_soak_fuzz.log
cp gpurun_out/parity_stats.json gpurun_out/r04_soak_parity_stats.json
GOI_FUZZ_N=1500 GOI_FUZZ_SEED=55001 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r04_soak_fuzz.log
timeout 900 python tools/spec_soak.py > gpurun_out/r04_soak_spec.log 2>&1
timeout 600 python bench.py --P 3000000 --steps 30 --no-cpu-baseline --no-clustered > gpurun_out/r04_bench_config2_size.json 2> /dev/null
timeout 600 python bench.py --scene closeup --steps 30 --no-cpu-baseline --no-clustered > gpurun_out/r04_bench_closeup.json 2> /dev/null
cat gpurun_out/r04_soak_fuzz.log; tail -4 gpurun_out/r04_soak_spec.log
python - <<'PY'
import json
for f in ("r04_bench_config2_size","r04_bench_closeup"):
    try:
        b=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, b["value"], b["ms_per_step"], b["config"])
    except Exception as e: print(f, "ERR", e)
PY
