#!/bin/bash
# One parameterised launcher for the GPU box (replaces round 1-5's gpu_*.sh one-offs).  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh <tag> <step> [<step> ...]'
# Steps (each writes gpurun_out/<tag>_<step>.*; a step with arguments is quoted: "abk:PATTERN:WORKLOADS:lib1,lib2"):
#   tests[:<pytest -k expr>]     the whole -m gpu suite (or a -k selection)
#   suites:<file1,file2,...>     the named test files (tests/test_gpu_<name>.py)
#   kstats[:<workload>]          per-kernel durations of the step loop (rocprofv3 --kernel-trace --stats)
#   abk:<regex>:<wl1,wl2>:<libs> same-box A/B of library builds (variants/lib_<name>.so) by per-kernel durations; a workload's
#                                own ':' is written '%' (headline%3000000); a lib may carry options: <lib>@opt=val+opt2=val2
#   abstep:<libs>                same-box A/B of library builds by the whole step (tools/ab_step.py, option bwd_variant 0)
#   bench[:<extra args>]         python bench.py <extra args>  (',' separates arguments)
#   pmc:<workload>               kernel stats + the five --pmc passes of the step loop (tools/pmc_steps.sh)
#   timeline:<workload>          per-dispatch timeline of one step
#   pmcopt:<wl>:<opts>:<regex>   instruction counters (one --pmc pass) of the kernels matching <regex> under GOI_OPTIONS=<opts>
#   benchprof                    bench.py --steps 20 --warmup 3 under rocprofv3 --kernel-trace --stats
#   soak:<N>:<seed>              tests/test_gpu_fuzz.py with GOI_FUZZ_N / GOI_FUZZ_SEED
#   py:<script>[:args]           python tools/<script> args (',' separates arguments)
cd $GRAFT_REPO_ROOT
T=$1; shift
mkdir -p gpurun_out
for STEP in "$@"; do
  IFS=':' read -r NAME A1 A2 A3 <<< "$STEP"
  case $NAME in
    tests)
      if [ -n "$A1" ]; then python -m pytest tests -m gpu -q -k "$A1" 2>&1 | tail -15 > gpurun_out/${T}_tests.log
      else python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_tests.log; fi
      cp gpurun_out/parity_stats.json gpurun_out/${T}_parity_stats.json 2>/dev/null
      echo "== tests"; tail -4 gpurun_out/${T}_tests.log ;;
    suites)
      FILES=$(echo $A1 | tr ',' '\n' | sed 's#^#tests/test_gpu_#; s#$#.py#' | tr '\n' ' ')
      python -m pytest $FILES -m gpu -q --deselect "tests/test_gpu_parity.py::test_metric_configuration_matches_oracle[3000000--0.05]" 2>&1 | tail -15 > gpurun_out/${T}_suites.log
      echo "== suites $A1"; tail -6 gpurun_out/${T}_suites.log ;;
    kstats)
      A1=$(echo ${A1:-headline} | tr '%' ':')
      bash tools/kstats.sh tools/step_loop.py 40 ${A1:-headline} > gpurun_out/${T}_kstats_${A1:-headline}.txt 2>&1
      echo "== kstats ${A1:-headline}"; cat gpurun_out/${T}_kstats_${A1:-headline}.txt ;;
    abk)
      bash tools/gpu_ab_k.sh "$A1" "$(echo $A2 | tr ',%' ' :')" $(echo $A3 | tr ',' ' ') 2>&1 | tee gpurun_out/${T}_abk.txt ;;
    abstep)
      cp goi_hyperplane_amd/lib/libgoi_raster.so /tmp/lib_keep.so
      for rep in 1 2; do for lib in $(echo $A1 | tr ',' ' '); do
        cp variants/lib_$lib.so goi_hyperplane_amd/lib/libgoi_raster.so
        echo "== $lib: $(timeout 600 python tools/ab_step.py bwd_variant 0 2>&1 | tail -1)"
      done; done | tee gpurun_out/${T}_abstep.txt
      cp /tmp/lib_keep.so goi_hyperplane_amd/lib/libgoi_raster.so ;;
    bench)
      timeout 900 python bench.py $(echo $A1 | tr ',' ' ') > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
      echo "== bench"; python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ("value", "ms_per_step", "repeats_ms_per_step", "roofline")})
    print("orbit", r.get("workload_orbit")); print("stages", {k: v["ms"] for k, v in (r.get("stages") or {}).items()})
    print("two", r.get("two_views_in_flight")); print("train_iter", r.get("semantic_train_iteration"))
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-3000:])
PY
      ;;
    pmc) A1=$(echo $A1 | tr '%' ':'); bash tools/pmc_steps.sh ${T}_pmc_$(echo $A1 | tr ':' '_') $A1 > /dev/null 2>&1; cat gpurun_out/${T}_pmc_$(echo $A1 | tr ':' '_')/kernel_stats.txt ;;
    timeline) A1=$(echo $A1 | tr '%' ':'); F=gpurun_out/${T}_timeline_$(echo $A1 | tr ':' '_').txt; bash tools/step_timeline.sh $A1 > $F 2>&1; tail -30 $F ;;
    pmcopt)  # pmcopt:<workload>:<GOI_OPTIONS with + for ,>:<kernel regex>  -- instruction counters of an option variant (one --pmc pass)
      W=$(echo $A1 | tr '%' ':'); O=$GRAFT_REPO_ROOT/gpurun_out/${T}_pmcopt; rm -rf $O; mkdir -p $O
      (cd /tmp; export TMPDIR=/tmp; GOI_OPTIONS=$(echo $A2 | tr '+' ',') timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O -o p -- python $GRAFT_REPO_ROOT/tools/step_loop.py 8 $W > $O/log.txt 2>&1)
      python - <<PY | tee -a gpurun_out/${T}_pmcopt.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(r"$A3", r["Kernel_Name"]):
            k = r["Kernel_Name"].split("(")[0][-60:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["VGPR"] = [float(r.get("VGPR_Count", 0) or 0)]; agg[k]["AGPR"] = [float(r.get("Accum_VGPR_Count", 0) or 0)]
for k, c in agg.items():
    print("pmc [$W] [$A2]", k, {n: round(sum(v) / len(v) / (1e6 if n not in ("VGPR", "AGPR") else 1), 3) for n, v in sorted(c.items())})
PY
      ;;
    benchprof)  # the driver's own command under the profiler (kernel stats must agree with the bench line's roofline.avg_ms)
      O=$GRAFT_REPO_ROOT/gpurun_out/${T}_benchprof; rm -rf $O; mkdir -p $O
      (cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.txt; rm -f $O/*kernel_trace.csv $O/*/*kernel_trace.csv)
      python tools/kernel_stats_top.py $O 12 2>/dev/null | head -16 ;;
    soak)  # soak:<N>:<seed>  -- the fuzz sweep on a fresh seed
      GOI_FUZZ_N=${A1:-1200} GOI_FUZZ_SEED=${A2:-9601} timeout 1800 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -4 | tee -a gpurun_out/${T}_soak_fuzz.log
      cp gpurun_out/parity_stats.json gpurun_out/${T}_soak_parity_stats_${A2:-9601}.json 2>/dev/null ;;
    py) timeout 1200 python tools/$A1 $(echo $A2 | tr ',' ' ') 2>&1 | tee gpurun_out/${T}_$(basename $A1 .py).txt | tail -40 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
