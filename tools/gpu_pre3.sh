cd $GRAFT_REPO_ROOT
T=${1:-pre3}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for w in headline closeup headline:3000000; do
  echo "== $w"; bash tools/kstats.sh tools/step_loop.py 40 $w 2>&1 | grep -E "preprocess_fwd_k|compact_listed|preprocess_bwd"
  echo "== $w, two-pass dL/dSH"; bash tools/kstats.sh tools/two_pass_dsh.py 30 $w 2>&1 | grep -E "preprocess_bwd|sh_grad"
done 2>&1 | tee gpurun_out/${T}_kstats.txt
