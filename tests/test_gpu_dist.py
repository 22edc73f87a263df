"""The multi-rank path of bench.py end to end on GPU tensors: two ranks share the one GPU of the test box over gloo
(test hook GOI_BENCH_BACKEND / GOI_BENCH_SHARE_GPU) -- view sharding, the factored dL/dSH exchange with its start-up
verification against the plain all-reduce, max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("exchange", ["factored", "allreduce"])
def test_two_ranks_on_one_gpu(exchange):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, GOI_BENCH_BACKEND="gloo", GOI_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--P", "20000", "--W", "320", "--H", "208", "--no-cpu-baseline", "--no-semantic-finetune", "--no-stage-timing",
           "--exchange", exchange]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["exchange"] == exchange
    if exchange == "factored":
        assert d["config"]["exchange_note"].startswith("verified against the plain all-reduce"), d["config"]["exchange_note"]
        assert d["config"]["allgather_bytes"] == 20000 * 3 * 4 * 2
        assert d["config"]["allreduce_bytes"] == 20000 * 27 * 4
    else:
        assert d["config"]["allreduce_bytes"] == 20000 * 75 * 4
