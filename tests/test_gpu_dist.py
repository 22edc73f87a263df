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


@pytest.mark.parametrize("exchange", ["factored", "allreduce", "visible", "allreduce_k2"])
def test_two_ranks_on_one_gpu(exchange):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, GOI_BENCH_BACKEND="gloo", GOI_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--P", "20000", "--W", "320", "--H", "208", "--no-cpu-baseline", "--no-stage-timing", "--exchange",
           exchange.split("_")[0]]
    k = 2 if exchange.endswith("_k2") else 1
    if k > 1:
        cmd += ["--views-per-exchange", "2", "--steps", "4", "--warmup", "2"]
    if exchange != "allreduce":
        cmd.append("--no-semantic-finetune")  # the "allreduce" parametrisation runs every secondary phase of the default bench
        cmd += ["--no-fp32-flush"] if exchange != "factored" else []
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["exchange"] == exchange.split("_")[0]
    # `value` has train.py's semantics (every exchange consumed inside its step); the in-flight figure stands beside it
    assert d["exchange_overlap"].startswith("waited for inside the step") and d["serialized_exchange"] is None
    assert d["exchange_in_flight"]["views_per_s"] > 0 and d["views_per_exchange"] == k
    assert "consumed before the next step" in d["config"]["exchange_semantics"]
    m = d["modelled_exchange_ms"]
    assert m["ring_ms"] >= m["direct_ms"] > 0 and abs(m["per_view_ring_ms"] * k - m["ring_ms"]) < 1e-3
    if exchange == "visible":
        assert 0 < d["config"]["rows_sent"] <= 20000
        assert d["config"]["allreduce_bytes"] == d["config"]["rows_sent"] * 75 * 4 + 20000
    if exchange == "allreduce":  # semantics-only phase, with and without the geometry cache, across two ranks
        sf = d["semantic_finetune"]
        assert sf["views_per_s"] > 0 and sf["geometry_cache"]["views_per_s"] > 0 and sf["geometry_cache"]["misses"] == 0
        assert d["value_fp32_flush"] > 0 and d["value_two_views_in_flight"] is None
    if exchange == "factored":
        assert d["config"]["exchange_note"].startswith("verified against the plain all-reduce"), d["config"]["exchange_note"]
        assert d["config"]["allgather_bytes"] == 20000 * 3 * 4 * 2
        assert d["config"]["allreduce_bytes"] == 20000 * 27 * 4
    elif exchange != "visible":
        assert d["config"]["allreduce_bytes"] == 20000 * 75 * 4


@pytest.mark.parametrize("exchange", ["factored", "allreduce"])
def test_single_rank_rccl_path_runs_on_hardware(exchange):
    """RCCL itself (torch.distributed backend "nccl"), one rank: the process group is created on the GPU, the gradient
    exchange of every step goes through real RCCL collectives (all-reduce; all-gather + all-reduce for the factored
    form), waited for inside the step for `value` and left in flight for the secondary figure -- the code path the 8-GPU
    scaling run takes, on the one GPU a test box has.  (GOI_BENCH_FORCE_DIST=1: bench.py's hook for exactly this.)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, GOI_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--P", "50000",
           "--W", "400", "--H", "304", "--no-cpu-baseline", "--exchange", exchange]
    if exchange == "factored":
        cmd += ["--no-semantic-finetune", "--no-fp32-flush"]  # "allreduce" runs the default bench's secondary phases too
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert isinstance(d["per_rank"], list) and len(d["per_rank"]) == 1 and d["per_rank"][0]["rank"] == 0
    assert d["per_rank"][0]["gpu_stage_ms_sum"] > 0 and d["per_rank"][0]["dominant_stage"] in d["stages"]
    assert d["speculation"]["overflows"] == 0


def test_direct_exchange_runs_on_rccl_in_place():
    """allreduce_gradients_direct on the RCCL backend itself (one rank: all this box has): reduce_scatter_tensor with
    `output = input[rank]` and all_gather_into_tensor back into the same span, both queued at once, blocking and in flight,
    with the gradients the rasterizer's backward leaves (views of ONE flat buffer) and with separately allocated ones.  With a
    single rank the sum is the identity: the gradients must come back bit for bit.  (What more ranks do is tested on gloo,
    tests/test_dist_cpu.py; RCCL with more than one rank needs more than one GPU.)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["GOI_ROOT"])
import torch.distributed as dist
from goi_hyperplane_amd.dist import allreduce_gradients_direct
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%s" % os.environ["GOI_PORT"])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
flat = torch.randn(8 * 1000 + 3, device=dev, generator=g)
shapes = [(1000, 3), (1000, 4), (1003,)]
params, off = [], 0
for shp in shapes:  # gradients that are views of one flat buffer, as the rasterizer's backward returns them
    n = 1
    for d_ in shp: n *= d_
    p = torch.zeros(shp, device=dev, requires_grad=True)
    p.grad = flat[off:off + n].view(shp)
    off += n
    params.append(p)
extra = torch.zeros(257, 5, device=dev, requires_grad=True)  # and one of its own
extra.grad = torch.randn(257, 5, device=dev, generator=g)
params.append(extra)
want = [p.grad.clone() for p in params]
allreduce_gradients_direct(params, dist)
torch.cuda.synchronize()
assert all(torch.equal(p.grad, w) for p, w in zip(params, want)), "blocking"
h = allreduce_gradients_direct(params, dist, async_op=True)
h.wait()
torch.cuda.synchronize()
assert all(torch.equal(p.grad, w) for p, w in zip(params, want)), "in flight"
dist.destroy_process_group()
print("OK")
'''
    env = dict(os.environ, GOI_ROOT=ROOT, GOI_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("mode", ["ring", "direct", "async"])
def test_pooled_gradient_buffers_stay_correct_across_in_place_exchanges(mode):
    """the compiled binding's gradient pool hands a buffer out again as "rows of Gaussians invisible then and
    now still hold zeros" when nothing wrote to it in place -- and c10d collectives write in place WITHOUT bumping the autograd
    version counter.  Two gloo ranks share the GPU, each renders ITS OWN camera (different visibility) for several steps with
    `p.grad = None` between them, and sums the gradients in place over the ranks; every step's reduced gradient on every rank
    must equal the sum of the two single-view gradients computed with the pool OFF (a stale row -- another rank's contribution
    of an earlier step surviving in a row this rank never rewrites -- shows up from the second step on)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["GOI_ROOT"])
import torch.distributed as dist
from goi_hyperplane_amd import _C
from goi_hyperplane_amd.dist import allreduce_gradients, allreduce_gradients_direct, allreduce_gradients_async
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_camera, make_scene
rank, mode = int(os.environ["RANK"]), os.environ["GOI_MODE"]
dist.init_process_group("gloo", rank=rank, world_size=2, init_method="tcp://127.0.0.1:%s" % os.environ["GOI_PORT"])
dev = torch.device("cuda:0")
assert _C.binding() == "compiled", _C.binding()  # (the pool lives in the compiled binding)
sc = make_scene(20000, S=16, sh_degree=3, seed=5, log_scale_mean=-3.2)
pc = GaussianSet.from_scene(sc, dev)
W, H = 320, 208
# narrow, well separated views: each rank sees a different third of the scene, and a different one every step
cams = [[TorchCamera(make_camera(W, H, fovx=0.45, yaw=0.5 * r + 0.17 * i - 0.4), dev) for i in range(4)] for r in range(2)]
gen = torch.Generator(device=dev).manual_seed(3)
gc = torch.randn((3, H, W), device=dev, generator=gen) / (W * H)
gs = torch.randn((16, H, W), device=dev, generator=gen) / (W * H)
params = list(pc.parameters())
def view_grads(cam):
    for p in params:
        p.grad = None
    out = render(cam, pc, PipelineParams(), torch.zeros(3, device=dev))
    torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))
    return out
import goi_hyperplane_amd._C as C_
ext = C_._ext()
# the references first: both ranks' single-view gradients with the pool OFF, summed locally
ext.set_grad_pool(False)
wants = []
for step in range(4):
    want = None
    for r in range(2):
        view_grads(cams[r][step])
        g = [p.grad.clone() for p in params]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    wants.append(want)
for p in params:
    p.grad = None
ext.set_grad_pool(True)
for step in range(4):
    want = wants[step]
    view_grads(cams[rank][step])
    if mode == "ring":
        allreduce_gradients(params, dist)
    elif mode == "direct":
        allreduce_gradients_direct(params, dist)
    else:
        held = [p.grad for p in params]
        h = allreduce_gradients_async(params, dist)
        for p in params:
            p.grad = None
        h.wait()
        for p, g_ in zip(params, held):
            p.grad = g_
        del held, h
    torch.cuda.synchronize()
    for p, w in zip(params, want):
        scale = float(w.abs().max()) + 1e-30
        err = float((p.grad - w).abs().max()) / scale
        assert err <= 1e-5, (step, tuple(p.shape), err)
for p in params:
    p.grad = None
hits, dirty, fresh = ext.grad_pool_stats()
assert dirty >= 2, (hits, dirty, fresh)  # the exchanged buffers came back from the pool marked as written
dist.barrier()
dist.destroy_process_group()
print("OK", rank, hits, dirty, fresh)
'''
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, GOI_ROOT=ROOT, GOI_PORT=str(port), RANK=str(r), GOI_MODE=mode, GOI_GRAD_POOL="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "OK" in so, (so[-500:], se[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("K,streams,on_device", [(2, 2, True), (4, 2, True), (5, 3, True), (3, 1, True), (2, 2, False), (4, 2, False),
                                                 (3, 1, False)])
def test_views_of_a_batch_in_flight_sum_to_the_serial_gradient(K, streams, on_device):
    """dist.backward_views: the K views of a multi-view batch alternate over several HIP streams of one GPU and the sum of their
    gradients lands in p.grad.  on_device (default): views 2, 3, ... of a stream ADD their per-Gaussian gradients to view 1's inside
    the per-Gaussian backward kernel (goi_raster_backward3, GOI_BACKWARD_ACCUMULATE) and the sums go through the activations once:
    the serial sum up to the association of fp32 additions.  on_device False: K dense sums of leaf gradients -- K = 2 on two
    streams is g0 + g1, bit-identical to two serial backward calls; with more views per stream (g0 + g2) + (g1 + g3).  The per-view
    outputs are those of the serial renders, bit for bit, either way."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from goi_hyperplane_amd.dist import backward_views
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    from goi_hyperplane_amd.scene import make_camera, make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(30000, S=16, sh_degree=3, seed=8, log_scale_mean=-3.3)
    pc = GaussianSet.from_scene(sc, dev)
    W, H = 320, 208
    cams = [TorchCamera(make_camera(W, H, fovx=0.6, yaw=0.3 * k - 0.5, pitch=0.05 * k), dev) for k in range(K)]
    gen = torch.Generator(device=dev).manual_seed(11)
    ups = [(torch.randn((3, H, W), device=dev, generator=gen) / (W * H), torch.randn((16, H, W), device=dev, generator=gen) / (W * H))
           for _ in range(K)]
    params = list(pc.parameters())
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    # the serial reference: K backward calls accumulating into p.grad
    for p in params:
        p.grad = None
    ref_out = []
    for k in range(K):
        o = render(cams[k], pc, pipe, bg)
        torch.autograd.backward((o["render"], o["semantics"]), ups[k])
        ref_out.append({key: o[key].detach().clone() for key in ("render", "semantics", "radii")})
    torch.cuda.synchronize()
    want = [p.grad.clone() for p in params]
    for rep in range(2):  # (twice: the second batch reuses the streams' pooled buffers and scratch)
        for p in params:
            p.grad = None
        outs = backward_views(cams, lambda cam: render(cam, pc, pipe, bg), lambda o, k: ((o["render"], o["semantics"]), ups[k]),
                              params, streams=streams, on_device=on_device)
        if on_device:
            from goi_hyperplane_amd import _C, rasterizer
            if _C._ext() is not None:
                assert rasterizer.last_backward_kernel() == "full_accumulate"
        torch.cuda.synchronize()
        for k in range(K):
            for key in ("render", "semantics", "radii"):
                assert torch.equal(outs[k][key], ref_out[k][key]), (rep, k, key)
        for p, w in zip(params, want):
            if (K <= 2 or streams == 1) and not on_device:
                assert torch.equal(p.grad, w), rep
            else:
                scale = float(w.abs().max())
                assert float((p.grad - w).abs().max()) <= 4e-6 * scale + 1e-12, rep
    # accumulate=True adds to what is there
    backward_views(cams[:2], lambda cam: render(cam, pc, pipe, bg), lambda o, k: ((o["render"], o["semantics"]), ups[k]), params,
                   streams=2, accumulate=True, on_device=on_device)
    torch.cuda.synchronize()
    assert all(not torch.equal(p.grad, w) for p, w in zip(params[:1], want[:1]))


@pytest.mark.gpu
def test_accumulating_backward_adds_exactly_the_visible_rows():
    """goi_raster_backward3 with GOI_BACKWARD_ACCUMULATE through the binding: the second view's call adds to the first view's
    tensors -- equal to the sum of two plain calls to an ulp of the larger term, rows of Gaussians neither view sees stay exactly
    zero, and the pooled-buffer bookkeeping does not hand the accumulated buffer out as "zero rows known" afterwards."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    from goi_hyperplane_amd.scene import make_camera, make_scene
    if _C._ext() is None:
        pytest.skip("the compiled binding is not built")
    dev = torch.device("cuda:0")
    sc = make_scene(40000, S=16, sh_degree=3, seed=9, extent=(6.0, 4.0, 1.0), log_scale_mean=-3.4)
    pc = GaussianSet.from_scene(sc, dev)
    W, H = 320, 208
    cams = [TorchCamera(make_camera(W, H, fovx=0.5, yaw=y), dev) for y in (-0.6, 0.5)]
    gen = torch.Generator(device=dev).manual_seed(12)
    ups = [(torch.randn((3, H, W), device=dev, generator=gen) / (W * H), torch.randn((16, H, W), device=dev, generator=gen) / (W * H))
           for _ in cams]
    bg, pipe = torch.zeros(3, device=dev), PipelineParams()
    params = list(pc.parameters())

    def one(cam, up, state=None):
        ctx = rasterizer.accumulate_gradients(state) if state is not None else __import__("contextlib").nullcontext()
        with ctx:
            o = render(cam, pc, pipe, bg)
            g = torch.autograd.grad((o["render"], o["semantics"]), params, up, allow_unused=True)
        return o["radii"] > 0, g
    vis0, g0 = one(cams[0], ups[0])
    vis1, g1 = one(cams[1], ups[1])
    want = [a + b for a, b in zip(g0, g1)]
    state = dict(grads=None, inputs=None, views=0)
    one(cams[0], ups[0], state)
    one(cams[1], ups[1], state)
    torch.cuda.synchronize()
    assert state["views"] == 2
    got = torch.autograd.grad([state["inputs"][i] for i in (0, 1, 3, 4, 5, 6)], params,
                              [state["grads"][j].view_as(state["inputs"][i]) for i, j in ((0, 4), (1, 6), (3, 2), (4, 3), (5, 7), (6, 8))],
                              allow_unused=True)
    unseen = ~(vis0 | vis1)
    assert int(unseen.sum()) > 1000
    for a, b in zip(got, want):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 4e-6 * scale + 1e-12
        assert float(a[unseen].abs().max()) == 0.0
    # a plain backward afterwards (pool in play) is still exact
    _v, g2 = one(cams[0], ups[0])
    for a, b in zip(g2, g0):
        assert torch.equal(a, b)
