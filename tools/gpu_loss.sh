cd $GRAFT_REPO_ROOT
T=${1:-loss}
python -m pytest tests/test_gpu_losses.py tests/test_gpu_train_loop.py -m gpu -x -q 2>&1 | tail -5
for v in 1 0; do
  echo "== GOI_DECODER_ONE_PASS=$v"; GOI_DECODER_ONE_PASS=$v python tools/fused_loss_time.py 2>&1 | tail -2
  GOI_DECODER_ONE_PASS=$v bash tools/kstats.sh tools/fused_loss_time.py 2>&1 | grep -v "^$" | head -12
done
