# usage: tools/step_timeline.sh <workload> -- every kernel of one training step with its duration and the gap in front of it
# (rocprofv3 --kernel-trace of tools/step_loop.py, tools/trace_gaps.py on the last full step)
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $GRAFT_REPO_ROOT/tools/step_loop.py 12 $1 > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_gaps.py $(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
