#!/usr/bin/env python3
"""bench.py -- training views/s (+ render ms/frame) of the rasterizer hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W                         # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          # N GPUs (driver launches this)

Workload (BASELINE.json metric): 1M Gaussians, 1600x1056, RGB from SH degree 3 + 16-d semantic
feature, fp32; synthetic scene of SURVEY.md 8(d) (goi_hyperplane_amd.scene.HEADLINE).
A "step" = one training view on every rank: rasterizer forward + backward through the reference-
shaped autograd operator (render() -> torch.autograd.backward with dense upstream gradients
dL/dcolour [3,H,W], dL/dsemantics [S,H,W]; the image-space loss itself is not part of the metric),
then -- when N > 1 -- the sum all-reduce of the Gaussian gradients over RCCL.  Views are independent, so ranks shard them
(weak scaling: one view per rank per step); value = N*K / max-over-ranks time.

The JSON line also carries
  roofline     : for the dominant kernel (largest share of the timed region): achieved =
                 algorithmic bytes per launch (DESIGN.md / SURVEY.md 8(d)) / average duration
                 measured with HIP events on the launch stream inside the timed region;
  cpu_baseline : the CPU oracle ("port": oracle/goi_oracle.cpp, OpenMP) timed on rank 0 at N = 1
                 on a bounded sample of the same workload;
  stages       : per-stage average ms and algorithmic GB/s (every stage, not only the dominant).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def stage_bytes(P, V, N, T, HW, S):
    """Algorithmic (compulsory) bytes per view and stage, fp32, every array touched once
    (SURVEY.md 8(d) derivation; the per-stage split sums to B_fwd / B_bwd)."""
    C = 3
    return {
        "preprocess": 60 * P + 259 * V,
        "depth_sort": 16 * V,                 # this build's extra stage: (key,id) read + write once
        "scan": 8 * P,
        "emit": 12 * V + 12 * N,
        "tile_sort": 24 * N,
        "ranges": 8 * N + 24 * T,
        "blend_fwd": (4 + 24 + 4 * (C + S + 1)) * N + 4 * (6 + S) * HW,
        "blend_bwd": (84 + 8 * S) * N + 4 * (7 + S) * HW,
        "preprocess_bwd": 4 * (76 + S) * P + 627 * V,
    }


def survey_bytes(P, V, N, T, HW, S):
    b_fwd = 68 * P + 271 * V + (72 + 4 * (4 + S)) * N + 4 * (6 + S) * HW + 24 * T
    b_bwd = 4 * (76 + S) * P + 627 * V + (84 + 8 * S) * N + 4 * (7 + S) * HW
    return b_fwd, b_bwd


def cpu_baseline(args, n_full_per_view, gpu_view=None):
    """Times the CPU oracle (forward + backward, all host threads) on ONE view of the bench workload.  By default the
    sample is the whole workload (all P Gaussians at the full resolution: ~20 s on the GPU box's cores, nothing is
    extrapolated); --cpu-sample-P bounds it to the first P_sample Gaussians, linearly extrapolated in N.  A second,
    quarter-size sample measures how linear the oracle's time is in N (reported, never used for `value`).

    gpu_view(cam, bg, grads) -> (outputs, gradients) renders the SAME view with the same upstream gradients through the
    HIP path; when the sample is the whole view the oracle's outputs are compared with it and the result is returned as
    the second value (`parity`: BASELINE.md section 3's gate, printed with the timing it belongs to)."""
    from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene
    from oracle import oracle
    oracle.build()
    cores = os.cpu_count() or 1
    cam = make_camera(args.W, args.H, fovx=HEADLINE["fovx"])
    HW = args.W * args.H
    full = make_scene(args.P, S=args.S, sh_degree=3, seed=0, extent=HEADLINE["extent"], log_scale_mean=args.mu,
                      log_scale_std=HEADLINE["log_scale_std"])  # every array has its own RNG stream: a prefix of the
                                                                # full scene IS the smaller scene

    # dense random upstream gradients on all four outputs (colour, semantics, depth, alpha), 1/HW-scaled
    rng = np.random.default_rng(4321)
    grads = [(rng.standard_normal((c, args.H, args.W), dtype=np.float32) / HW) for c in (3, args.S, 1, 1)]
    bg = np.zeros(3, np.float32)
    kept = {}

    def timed(Ps, keep=False):
        import copy
        sc = copy.copy(full)
        for k in ("means3D", "scales", "rotations", "opacities", "shs", "semantics"):
            setattr(sc, k, np.ascontiguousarray(getattr(full, k)[:Ps]))
        o = oracle.from_scene(sc, cam, bg=bg, threads=cores)
        t0 = time.perf_counter()
        f = o.forward()
        t1 = time.perf_counter()
        g = o.backward(*grads)
        t2 = time.perf_counter()
        if keep:
            kept.update(f=f, g=g)
        return int(f.num_rendered), t1 - t0, t2 - t1

    Ps = min(args.P, args.cpu_sample_P) if args.cpu_sample_P > 0 else args.P
    n_s, fwd_s, bwd_s = timed(Ps, keep=(Ps == args.P and gpu_view is not None))
    parity = None
    if kept:
        from oracle import compare
        res, g_hip = gpu_view(cam, bg, grads)
        parity = compare.summary(compare.forward_stats(res, kept["f"]), compare.backward_stats(g_hip, kept["g"]))
        parity["what"] = (f"HIP path (default forward / lists / flush) vs the CPU oracle on the workload's canonical view "
                          f"({args.P} Gaussians, {args.W}x{args.H}, S={args.S}), dense random upstream gradients on colour, "
                          "semantics, depth and alpha; forward outside the oracle's fragile pixels, gradients relative to "
                          "each tensor's largest magnitude")
        # the SAME view through the exact-fp32 flush of the backward (bwd_variant 2), and the two flushes against each other:
        # what -- if anything -- the split-f16 products of the default flush cost in accuracy at the metric's own size
        from goi_hyperplane_amd import _lib as _l
        _l.set_option("bwd_variant", 2)
        try:
            res2, g_hip2 = gpu_view(cam, bg, grads)
        finally:
            _l.set_option("bwd_variant", 0)
        p2 = compare.summary(compare.forward_stats(res2, kept["f"]), compare.backward_stats(g_hip2, kept["g"]))
        p2["what"] = "the same view and gradients with bwd_variant 2 (per-Gaussian sums as exact-fp32 MFMA products)"
        between = {k: v["max"] for k, v in compare.backward_stats(g_hip, g_hip2).items()}
        parity["fp32_flush"] = p2
        parity["flush_equivalence_same_view"] = {
            "default_vs_fp32_flush_max_by_tensor": between,
            "default_minus_fp32_error_vs_oracle": {k: parity["grad_max_by_tensor"][k] - p2["grad_max_by_tensor"][k]
                                                   for k in parity["grad_max_by_tensor"]},
            "what": "max |g_default - g_fp32flush| / scale per gradient tensor, and the difference of the two flushes' max "
                    "errors against the oracle (negative: the default flush is the closer one on that tensor)"}
        kept.clear()
    sample_s = fwd_s + bwd_s
    scale = 1.0 if Ps == args.P else max(n_full_per_view, 1) / max(n_s, 1)
    n_q, fq, bq = timed(max(1, Ps // 4))
    # seconds per instance at the two sizes: 0 = perfectly linear in N
    lin_err = (sample_s / max(n_s, 1)) / ((fq + bq) / max(n_q, 1)) - 1.0
    how = ("the whole view, nothing extrapolated" if Ps == args.P else
           f"value extrapolated linearly in N to N={n_full_per_view}")
    return parity, {
        "value": 1.0 / (sample_s * scale), "unit": "views/s", "cores": cores, "kind": "port",
        "sample": f"oracle fwd+bwd on {'all' if Ps == args.P else 'the first'} {Ps} of {args.P} Gaussians at "
                  f"{args.W}x{args.H}, S={args.S}: N={n_s}, fwd {fwd_s:.2f} s + bwd {bwd_s:.2f} s; {how}",
        "sample_seconds": sample_s, "sample_num_rendered": n_s,
        "linearity": {"quarter_sample_P": max(1, Ps // 4), "quarter_sample_num_rendered": n_q,
                      "quarter_sample_seconds": fq + bq, "seconds_per_instance_ratio_minus_1": lin_err},
    }


def lane_utilisation(st: dict) -> dict:
    """The bench line's `blend_lane_utilisation` object from the device counters of goi_raster_blend_stats (_C.blend_stats): how
    many of a wave's 64 lanes (= the pixels of an 8x8 quadrant) do useful work per loop trip (= per (quadrant, Gaussian) pair)
    of the two blend kernels, and what a split of the wave into smaller pixel blocks, each with its own list, could reach."""
    return {
        "backward": round(st["lane_utilisation_backward"], 4), "forward": round(st["lane_utilisation_forward"], 4),
        "live_lanes_per_pair": round(st["live_lanes_per_member_pair"], 2),
        "pairs_backward": st["member_pairs"], "pairs_forward": st["forward_pairs"], "live_lanes": st["live_lanes"],
        "contributions_per_pixel": round(st["contributions_per_pixel"], 2),
        "upper_bound_if_split_into": {"8x4": round(st["lane_utilisation_8x4_blocks"], 4),
                                      "4x4": round(st["lane_utilisation_4x4_blocks"], 4),
                                      "2x2": round(st["lane_utilisation_2x2_blocks"], 4)},
        "what": "live lanes / (64 x pairs) of the metric view, counted on the device from the forward's member masks and n_contrib "
                "(csrc/blend_stats.hip); backward = over the member pairs it walks, forward = over the pairs that pass its quadrant "
                "hit test; upper_bound_if_split_into = live lanes / (block pixels x (block, Gaussian) pairs with a live lane): "
                "what a wave of 2 / 4 / 16 independent pixel blocks would reach with perfect balance between its blocks and an "
                "exact per-block hit test (tools/lane_util_sim.py models the imbalance: 0.84 / 0.77 of today's trips for 4x4 / "
                "2x2 blocks on the headline view)"}


def main():
    from goi_hyperplane_amd.scene import HEADLINE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)  # >= 50: one host hiccup is < 2 % of the timed region
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region (exactly --steps steps) is run this many times; value = the median repeat")
    ap.add_argument("--P", type=int, default=HEADLINE["P"])
    ap.add_argument("--S", type=int, default=HEADLINE["S"])
    ap.add_argument("--W", type=int, default=HEADLINE["W"])
    ap.add_argument("--H", type=int, default=HEADLINE["H"])
    ap.add_argument("--mu", type=float, default=HEADLINE["log_scale_mean"], help="log-scale mean of the generator")
    ap.add_argument("--views", type=int, default=16, help="distinct cameras cycled through")
    ap.add_argument("--grads", choices=["all", "semantics"], default="all",
                    help="which Gaussian gradients are all-reduced when --gpus > 1")
    ap.add_argument("--exchange", choices=["factored", "allreduce", "visible"], default="factored",
                    help="--gpus > 1, --grads all: 'factored' all-gathers the factors of dL/dSH (12 B per Gaussian "
                         "and view) and all-reduces the other 27 gradient floats; 'allreduce' all-reduces all 75; "
                         "'visible' all-reduces all 75 but only for Gaussians some rank saw this step")
    ap.add_argument("--collective", choices=["auto", "ring", "direct"], default="auto",
                    help="--gpus > 1: how the sum inside the exchange travels -- 'ring' = one all_reduce (whatever algorithm RCCL "
                         "picks; bound by one xGMI link if it is a ring), 'direct' = reduce-scatter + all-gather over the flat "
                         "gradient span (all seven links at once), 'auto' = whichever SURVEY.md 8(e)'s link model prices lower "
                         "for this exchange's bytes (dist.pick_exchange); the JSON line says which one ran")
    ap.add_argument("--scene", choices=["headline", "clustered", "closeup"], default="headline",
                    help="the workload `value` is measured on: 'headline' = BASELINE.json's metric configuration (SURVEY 8(d)'s "
                         "uniform box); 'clustered' = the adversarial scene with a reconstruction's statistics at the same size; "
                         "'closeup' = config 5's shape (512x512 close-up of 3 M clustered Gaussians).  The default line also "
                         "carries the clustered scene as a secondary object (--no-clustered skips it)")
    ap.add_argument("--no-clustered", action="store_true", help="skip the secondary clustered-workload object")
    ap.add_argument("--no-orbit", action="store_true", help="skip the secondary orbit-camera object (workload_orbit)")
    ap.add_argument("--depth-cut", action="store_true",
                    help="turn the opt-in speculative depth cut-off of the tile lists ON for the whole run (default: off, the "
                         "package default; the default line reports the step with it as `value_with_depth_cut`)")
    ap.add_argument("--views-per-exchange", type=int, default=1,
                    help="--gpus > 1: every rank accumulates the gradients of this many views locally before one exchange "
                         "(an optimiser step per K x N views); 1 = one exchange per view, train.py's step semantics")
    ap.add_argument("--overlap", action="store_true",
                    help="--gpus > 1: make `value` the figure with the exchange left in flight behind the next step's "
                         "render + backward (gradients one step stale: NOT train.py's semantics); by default `value` waits "
                         "for every exchange inside its step and the in-flight figure is reported beside it")
    ap.add_argument("--ply", default=None,
                    help="point_cloud.ply saved by the reference (sem_* columns) instead of the synthetic scene; "
                         "cameras stay synthetic (the data sets are not in this image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-P", type=int, default=0,
                    help="bound the CPU baseline to the first N Gaussians (0 = the whole view, ~20 s on the GPU box)")
    ap.add_argument("--forward", choices=["speculative", "exact"], default=None,
                    help="forward mode (default: the package default, speculative = no host round trip)")
    ap.add_argument("--no-fp32-flush", action="store_true", help="skip the secondary exact-fp32-flush figure")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the secondary two-views-in-flight figure")
    ap.add_argument("--no-train-iteration", action="store_true", help="skip the secondary semantic-stage iteration figure")
    ap.add_argument("--no-overlap", action="store_true", help="(default since round 3; kept for older command lines)")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--no-semantic-finetune", action="store_true",
                    help="skip the secondary semantics-only-training figure")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: GOI_BENCH_BACKEND=gloo GOI_BENCH_SHARE_GPU=1 runs several ranks on ONE GPU over gloo (exercises the
    # multi-rank code path, the factored exchange included, where only one GPU is available; not a measurement)
    test_backend = os.environ.get("GOI_BENCH_BACKEND", "")
    if os.environ.get("GOI_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU; there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("GOI_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_backend:
            dist.init_process_group(backend=test_backend)
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from goi_hyperplane_amd import _C, _lib
    from goi_hyperplane_amd import rasterizer
    if args.forward:
        rasterizer.set_forward_mode(speculative=args.forward == "speculative")
    if args.depth_cut:
        rasterizer.set_forward_mode(depth_cut=True)
    from goi_hyperplane_amd.dist import (allreduce_gradients, allreduce_gradients_async, allreduce_gradients_sh_factored,
                                         allreduce_gradients_sh_factored_async, allreduce_gradients_visible,
                                         exchange_model_ms)
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    from goi_hyperplane_amd.scene import CLOSEUP, CLUSTERED, make_camera, make_scene
    from goi_hyperplane_amd.dist import allreduce_gradients_direct, pick_exchange
    _lib.load()

    WL = {"headline": HEADLINE, "clustered": CLUSTERED, "closeup": CLOSEUP}[args.scene]
    scene_kind = "uniform" if args.scene == "headline" else "clustered"
    if args.scene != "headline":  # the named workload's own size, image and generator constants
        args.P, args.S, args.W, args.H, args.mu = WL["P"], WL["S"], WL["W"], WL["H"], WL["log_scale_mean"]
        args.no_cpu_baseline = True  # (the CPU baseline + parity object belong to the metric configuration; the clustered
                                     # workloads have their oracle parity in tests/test_gpu_clustered.py)

    if args.ply:
        pc = GaussianSet.from_ply(args.ply, dev, sh_degree=3)  # the semantic width is the file's (10 for a default
        args.P = int(pc.get_xyz.shape[0])                      # reference run), not --S
        args.S = int(pc.get_semantics.shape[1])
        args.no_cpu_baseline = True  # the bounded CPU sample is defined on the synthetic scene
        sc = None
    else:
        sc = make_scene(args.P, S=args.S, sh_degree=3, seed=0, extent=WL["extent"], log_scale_mean=args.mu,
                        log_scale_std=WL["log_scale_std"], kind=scene_kind)  # identical replica on every rank
        pc = GaussianSet.from_scene(sc, dev)
    params = [pc._xyz, pc._features, pc._semantics, pc._opacity, pc._scaling, pc._rotation]
    reduce_params = params if args.grads == "all" else [pc._semantics]
    def make_cams(spec, W_, H_, n):
        return [TorchCamera(make_camera(W_, H_, fovx=spec["fovx"], yaw=spec.get("yaw", 0.0) + 0.02 * (i - n / 2),
                                        pitch=spec.get("pitch", 0.0) + 0.01 * ((i * 7) % 5 - 2),
                                        distance=spec.get("distance", 5.0)), dev) for i in range(n)]
    cams = make_cams(WL, args.W, args.H, args.views)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    HW = args.W * args.H
    T = ((args.W + 15) // 16) * ((args.H + 15) // 16)
    inv_hw = 1.0 / HW
    stats = {"V": 0, "N": 0, "views": 0}
    # upstream gradients of a dense image-space loss (fixed, so that no loss kernels sit in the step)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    g_color = torch.randn((3, args.H, args.W), device=dev, generator=gen) * inv_hw
    g_sem = torch.randn((args.S, args.H, args.W), device=dev, generator=gen) * inv_hw

    exchange = {"mode": "visible" if (args.exchange == "visible" and args.grads == "all") else "allreduce", "note": None}
    non_sh_params = [pc._xyz, pc._semantics, pc._opacity, pc._scaling, pc._rotation]

    # Gradient exchange in flight: the collective of step k is issued after its backward and waited for after step
    # k+1's backward, so it is on the wire while the next view renders.  Every step's gradients are
    # reduced, and the last exchange is drained inside the timed region.  --no-overlap keeps it inside the step.
    overlap_default = bool(args.overlap) and not args.no_overlap
    K = max(1, int(args.views_per_exchange))
    if K > 1 and args.exchange == "factored":
        args.exchange = "allreduce"  # (the factors are per view: K views would need K factor sets; the plain sum accumulates)
        exchange["note"] = "factored exchange needs one view per exchange: --views-per-exchange > 1 uses the plain all-reduce"
    # which collective carries the sum: decided once from the bytes of one exchange (the factored form all-reduces 27 of
    # the 75 gradient floats per Gaussian; the SH factors always travel by all-gather)
    per_gauss = sum(p_.numel() // max(1, p_.shape[0]) for p_ in (non_sh_params if args.exchange == "factored" and args.grads == "all"
                                                                   else reduce_params))
    sum_bytes = 4.0 * per_gauss * args.P
    collective = {"mode": (pick_exchange(sum_bytes, world) if args.collective == "auto" else args.collective)}
    direct = collective["mode"] == "direct" and world > 1
    inflight = {"h": None}
    seen = {"vis": None}

    def drain():
        if inflight["h"] is not None:
            inflight["h"].wait()
            inflight["h"] = None

    def step(i, record=False, overlap=None):
        overlap = overlap_default if overlap is None else overlap
        cam = cams[(i * world + rank) % len(cams)]  # rank r takes views r, r+G, ... of the cycle
        first, last = (i % K) == 0, (i % K) == K - 1
        if first:  # (K > 1: autograd accumulates the following views into the same .grad tensors)
            for p in params:
                p.grad = None
            seen["vis"] = None
        out = render(cam, pc, pipe, bg)
        torch.autograd.backward((out["render"], out["semantics"]), (g_color, g_sem))
        if dist is not None and exchange["mode"] == "visible":
            v = out["radii"] > 0
            seen["vis"] = v if seen["vis"] is None else (seen["vis"] | v)
        if dist is not None and last:
            if not overlap:
                drain()
                if exchange["mode"] == "visible":
                    stats["rows_sent"] = allreduce_gradients_visible(reduce_params, seen["vis"], dist, per_gaussian=reduce_params,
                                                                     direct=direct)
                elif exchange["mode"] == "factored":
                    allreduce_gradients_sh_factored(non_sh_params, (pc._features,), pc._xyz, rasterizer.take_sh_factor(), dist,
                                                    direct=direct)
                elif direct:
                    allreduce_gradients_direct(reduce_params, dist)
                else:
                    allreduce_gradients(reduce_params, dist)
            else:
                prev = inflight["h"]
                if exchange["mode"] == "visible":  # (its host synchronisation makes "in flight" meaningless: plain form)
                    inflight["h"] = allreduce_gradients_async(reduce_params, dist, direct=direct)
                elif exchange["mode"] == "factored":
                    inflight["h"] = allreduce_gradients_sh_factored_async(non_sh_params, (pc._features,), pc._xyz,
                                                                          rasterizer.take_sh_factor(), dist, direct=direct)
                else:
                    inflight["h"] = allreduce_gradients_async(reduce_params, dist, direct=direct)
                if prev is not None:
                    prev.wait()
        if record:
            stats["radii"] = out["radii"]
        return out

    if dist is not None and args.grads == "all" and args.exchange == "factored":
        # Exchange dL/dSH as its factors -- after checking, on this machine and this scene, that it reproduces the
        # plain all-reduce (the ring adds the ranks in another order: agreement to rounding, not bit for bit).  Any
        # failure falls back to the plain all-reduce and is reported in config.exchange.
        local_ok, why = True, ""
        try:
            step(0, overlap=False)
            want = pc._features.grad.clone()
            exchange["mode"] = "factored"
            rasterizer.set_backward_mode(sh_factored=True)
            step(0, overlap=False)
            err = float((pc._features.grad - want).abs().max())
            scale = float(want.abs().max())
            if not err <= 1e-5 * scale + 1e-12:
                local_ok, why = False, f"factored dL/dSH differs from the all-reduced one by {err:.3e} (scale {scale:.3e})"
            else:
                why = f"verified against the plain all-reduce: max |diff| = {err:.2e} of {scale:.2e}"
        except Exception as e:  # noqa: BLE001 -- never lose the scaling run to the optimisation
            local_ok, why = False, f"{type(e).__name__}: {e}"
        ok = torch.tensor([1.0 if local_ok else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # the ranks switch together
        if float(ok.item()) == 1.0:
            exchange["note"] = why
        else:
            exchange["mode"], exchange["note"] = "allreduce", "factored exchange disabled: " + (why or "another rank failed")
            rasterizer.set_backward_mode(sh_factored=False)

    for i in range(args.warmup):
        step(i)
    drain()
    # The steady state of a training run: every camera has been rendered before (an epoch ago), so its frames list, per tile,
    # only what that earlier frame found worth listing (the speculative depth cut-off, _C._depth_cut_for).  Two untimed passes
    # over the cameras put the run there: the first frames of a scene are exact, the next learn, the following ones are cut.
    if _C._FWD["depth_cut"] and _C._FWD["mode"] == "speculative":
        for i in range(2 * len(cams) + 4):
            step(i)
        drain()
    # workload statistics of the views this rank will time (outside the timed region).  N is SURVEY 8(d)'s instance
    # count -- every tile of every Gaussian's 3-sigma rectangle, what the reference lists and what the algorithmic
    # bytes are charged on (cull_variant 0); N_listed is what this build actually emits, sorts and walks.
    with torch.no_grad():
        stats["N_listed"] = 0
        # (exact frames, and what they teach the capacity policy is forgotten again: the un-culled lists counted here
        # are 1.6x what the timed steps emit)
        fwd_mode = _C._FWD["mode"]
        spec_saved = {k: dict(v, pending=v["pending"]) for k, v in _C._SPEC.items()}
        _C.set_forward_mode(speculative=False)
        for i in range(min(args.steps, len(cams))):
            cam = cams[((args.warmup + i) * world + rank) % len(cams)]
            for variant in (0, 2):  # (2 = the default lists: only the tiles a Gaussian's contribution ellipse reaches)
                _lib.set_option("cull_variant", variant)
                n, *_r = _C.rasterize_gaussians(bg, pc._xyz, torch.Tensor([]), pc._semantics, pc._opacity, pc._scaling,
                                                pc._rotation, 1.0, torch.Tensor([]), cam.world_view_transform,
                                                cam.full_proj_transform, np.tan(cam.FoVx * 0.5), np.tan(cam.FoVy * 0.5),
                                                args.H, args.W, pc._features, 3, cam.camera_center, False, False)
                if variant == 0:
                    stats["N"] += n
                    stats["V"] += int((_r[4] > 0).sum())
                    stats["views"] += 1
                else:
                    stats["N_listed"] += n
                    if "lane" not in stats:  # lane utilisation of the blend kernels on this view, counted on the device
                        stats["lane"] = lane_utilisation(_C.blend_stats(args.P, args.W, args.H, n, *_r[-3:]))
                del _r
        _C._SPEC.clear()
        _C._SPEC.update(spec_saved)
        _C.set_forward_mode(speculative=fwd_mode == "speculative")
    V = stats["V"] / stats["views"]
    N = stats["N"] / stats["views"]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # Stage breakdown: an untimed pass with events around every stage (the events themselves cost
    # ~0.25 ms/step, so they stay out of the timed region) ...
    timing = not args.no_stage_timing
    stages, dominant = {}, None
    if timing:
        _lib.profile_collect()  # drop anything recorded so far
        _lib.profile_enable(True)
        counts_seen = []
        for i in range(min(args.steps, 10)):
            step(args.warmup + i)
            counts_seen.append(rasterizer.last_num_rendered())
        drain()
        barrier()
        _lib.profile_enable(False)
        stages = _lib.profile_collect()
        stats["N_emitted"] = float(np.mean([int(c_) for c_ in counts_seen])) if counts_seen else None
        stats["cut_frames_in_profile_pass"] = sum(1 for c_ in counts_seen if getattr(c_, "cut_key", None) is not None)
        del counts_seen
        dominant = max(stages, key=lambda k: stages[k][0])
        # ... and, live over the timed region, events around the dominant kernel only.
        _lib.profile_stages([dominant])
    barrier()
    # host time of every step's enqueue (a clock read per step, no synchronisation: the timed region is unchanged).
    # Informational only -- `value` is K steps over the barrier-bracketed time -- it separates a host stall
    # (max >> median; see profiles/r01_n0_bench_anomalous.json) from a run that is uniformly slow.
    # The timed region (exactly K steps between barrier + synchronize) is run `--repeats` times back to back and `value`
    # is the MEDIAN repeat: with the driver's --steps 20 one region is ~30 ms of GPU time, and a single 3 ms host hiccup
    # is 10 % of it.  Every repeat is listed in the JSON line (`repeats_ms_per_step`); the events around the dominant
    # kernel, the enqueue marks and the speculation counters below are those of the median repeat's siblings as well.
    spec0 = rasterizer.speculation_stats()
    runs = []
    for rep in range(max(1, args.repeats)):
        marks = [0.0] * (args.steps + 1)
        t0 = marks[0] = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
            marks[i + 1] = time.perf_counter()
        drain()  # the last step's exchange completes inside the timed region
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:  # max over ranks, per repeat
            t_ = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            runs.append((float(t_.item()), el, marks, t0))
        else:
            runs.append((el, el, marks, t0))
    order_ = sorted(range(len(runs)), key=lambda r_: runs[r_][0])
    elapsed_max, elapsed, marks, t0 = runs[order_[len(order_) // 2]]  # the median repeat (by the max-over-ranks time)
    repeats_ms = [round(r_[0] / args.steps * 1e3, 5) for r_ in runs]
    spec1 = rasterizer.speculation_stats()
    enq = sorted((marks[i + 1] - marks[i]) * 1e3 for i in range(args.steps))
    step_enqueue_ms = {"median": round(enq[len(enq) // 2], 4), "max": round(enq[-1], 4),
                       "drain": round((elapsed - (marks[-1] - t0)) * 1e3, 4)} if enq else None
    if timing:
        _lib.profile_enable(False)
        stages[dominant] = _lib.profile_collect()[dominant]
    own_elapsed = elapsed
    elapsed = elapsed_max
    # the same steps with the OTHER exchange semantics: `value` waits for every exchange inside its step (what train.py's
    # one-optimiser-step-per-exchange loop pays) unless --overlap was given; the figure with the exchange left in flight
    # behind the next step (one-step-stale gradients) is reported beside it, and vice versa
    serialized, in_flight_fig = None, None
    if dist is not None and world > 1:
        other = not overlap_default
        for i in range(2 * K):
            step(i, overlap=other)
        drain()
        barrier()
        z0 = time.perf_counter()
        nz = max(5, min(args.steps, 20)) // K * K or K
        for i in range(nz):
            step(args.warmup // K * K + i, overlap=other)
        drain()
        barrier()
        tz = torch.tensor([time.perf_counter() - z0], dtype=torch.float64, device=dev)
        dist.all_reduce(tz, op=dist.ReduceOp.MAX)
        fig = {"views_per_s": nz * world / float(tz.item()), "ms_per_step": float(tz.item()) / nz * 1e3, "steps": nz}
        if other:
            in_flight_fig = fig
        else:
            serialized = fig
    # one line per rank, so that a scaling run explains itself: is a rank slow on the GPU, or waiting for the wire?
    per_rank = None
    if dist is not None:
        mine = {"rank": rank, "ms_per_step": own_elapsed / args.steps * 1e3,
                "gpu_stage_ms_sum": (sum(ms / max(c, 1) for ms, c in stages.values()) if stages else None),
                "dominant_stage": dominant, "dominant_ms": (stages[dominant][0] / max(stages[dominant][1], 1)
                                                            if stages and dominant else None),
                "step_enqueue_ms_median": step_enqueue_ms["median"] if step_enqueue_ms else None}
        # this rank's own views: algorithmic bytes of a view over its step time, as a fraction of the HBM peak, and the
        # dominant kernel's roofline fraction (SURVEY.md 8(d) accounting on THIS rank's V and N)
        r_fwd, r_bwd = survey_bytes(args.P, V, N, T, HW, args.S)
        mine["hbm_frac_over_step"] = (r_fwd + r_bwd) / (mine["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if mine["dominant_ms"]:
            mine["dominant_hbm_frac"] = (stage_bytes(args.P, V, N, T, HW, args.S)[dominant] / (mine["dominant_ms"] * 1e-3)
                                         / 1e9 / HBM_PEAK_GBS)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # render ms/frame: forward only, no_grad (the GUI path, gui/main.py:556-602), outside the step timing
    with torch.no_grad():
        for i in range(2):
            render(cams[i % len(cams)], pc, pipe, bg)
        barrier()
        r0 = time.perf_counter()
        nfr = max(5, min(args.steps, 50))
        for i in range(nfr):
            render(cams[(i * world + rank) % len(cams)], pc, pipe, bg)
        barrier()
        render_ms = (time.perf_counter() - r0) / nfr * 1e3
        # (frames rendered without autograd take the exact forward by default -- their image is the product, and a viewer
        # hands it to the host every frame anyway; the same loop with the speculative forward, for comparison)
        render_mode = _C._FWD["inference"]
        rasterizer.set_forward_mode(inference_speculative=True)
        for i in range(2):
            render(cams[i % len(cams)], pc, pipe, bg)
        barrier()
        r0 = time.perf_counter()
        for i in range(nfr):
            render(cams[(i * world + rank) % len(cams)], pc, pipe, bg)
        barrier()
        render_ms_spec = (time.perf_counter() - r0) / nfr * 1e3
        rasterizer.set_forward_mode(inference_speculative=render_mode == "speculative")
        # GUI frame: render + fused semantic decode (code-book argmax + hyperplane score + mask, gui/main.py:364-386)
        from goi_hyperplane_amd.semantic import LinearSVM, SemanticModel, compute_similarity, svm_score_fn
        torch.manual_seed(0)
        mlp = SemanticModel(dim_in=args.S, dim_out=300, num_layer=1, use_bias=True, device=dev)
        lut = torch.rand(300, 256, device=dev) * 0.03
        score = svm_score_fn(LinearSVM().to(dev))
        gui_ms = None
        if args.S <= 32:
            for i in range(2):
                compute_similarity(render(cams[i % len(cams)], pc, pipe, bg)["semantics"], mlp, lut, score, 0.5)
            barrier()
            r0 = time.perf_counter()
            for i in range(nfr):
                compute_similarity(render(cams[(i * world + rank) % len(cams)], pc, pipe, bg)["semantics"], mlp, lut, score, 0.5)
            barrier()
            gui_ms = (time.perf_counter() - r0) / nfr * 1e3

    # Secondary figure: the reference's DEFAULT training configuration optimises only the semantic features
    # (arguments/__init__.py:85-90).  With every other parameter frozen the feature-gradient-only backward
    # applies (goi_raster_backward_semantics, bit-identical dL/dsemantics); not part of `value`.
    sem_only = None
    if not args.no_semantic_finetune:
        from goi_hyperplane_amd import rasterizer
        rasterizer.set_backward_mode(semantics_only=True)
        for p in params:
            p.requires_grad_(False)
            p.grad = None
        pc._semantics.requires_grad_(True)

        def sem_step(i):
            cam = cams[(i * world + rank) % len(cams)]
            pc._semantics.grad = None
            out = render(cam, pc, pipe, bg)
            torch.autograd.backward((out["semantics"],), (g_sem,))
            if dist is not None:
                allreduce_gradients([pc._semantics], dist)
        for i in range(2):
            sem_step(i)
        barrier()
        s0 = time.perf_counter()
        nss = max(5, min(args.steps, 50))
        for i in range(nss):
            sem_step(i)
        barrier()
        sem_elapsed = time.perf_counter() - s0
        if dist is not None:
            tt = torch.tensor([sem_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sem_elapsed = float(tt.item())
        sem_only = {"views_per_s": nss * world / sem_elapsed, "ms_per_step": sem_elapsed / nss * 1e3, "steps": nss,
                    "what": "only the semantic features trainable (the reference's default): forward + "
                            "feature-gradient-only backward" + (" + RCCL all-reduce of dL/dsemantics" if world > 1 else "")}
        # ... and with the opt-in geometry cache: the cameras repeat (as the training views of an epoch do) and nothing but
        # the semantic features changes, so after the first pass over the cameras every frame is the blend alone
        rasterizer.set_geometry_cache(64 << 30)
        try:
            for i in range(len(cams) + 2):
                sem_step(i)
            barrier()
            c0 = rasterizer.geometry_cache_stats()
            s0 = time.perf_counter()
            c_enq = []
            for i in range(nss):
                e0 = time.perf_counter()
                sem_step(i)
                c_enq.append(time.perf_counter() - e0)  # (a clock read per step, no synchronisation)
            barrier()
            c_elapsed = time.perf_counter() - s0
            c1 = rasterizer.geometry_cache_stats()
            # is this mode bound by the host or by the GPU?  The library's own HIP events around each kernel of a few further
            # (untimed) steps: their sum per step beside the wall clock per step above
            _lib.profile_collect()
            _lib.profile_enable(True)
            for i in range(8):
                sem_step(i)
            torch.cuda.synchronize()
            _lib.profile_enable(False)
            c_stage_sum = sum(ms_ / n_ for ms_, n_ in _lib.profile_collect().values() if n_)
        finally:
            rasterizer.set_geometry_cache(0)
        if dist is not None:
            tt = torch.tensor([c_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            c_elapsed = float(tt.item())
        sem_only["geometry_cache"] = {"views_per_s": nss * world / c_elapsed, "ms_per_step": c_elapsed / nss * 1e3,
                                      "hits": c1["hits"] - c0["hits"], "misses": c1["misses"] - c0["misses"],
                                      "bytes_per_camera": c1["bytes"] // max(1, c1["entries"]),
                                      "step_enqueue_ms_median": float(np.median(c_enq)) * 1e3,
                                      "gpu_stage_sum_ms": c_stage_sum,
                                      "bound": "gpu" if c_stage_sum >= 0.9 * c_elapsed / nss * 1e3 else "host",
                                      "what": "the same with rasterizer.set_geometry_cache on (opt-in): cameras seen before "
                                              "render with the blend alone"}
        rasterizer.set_backward_mode(semantics_only="auto")
        for p in params:
            p.requires_grad_(True)

    # Secondary figure: the same step with the EXACT-fp32 flush of the backward's per-Gaussian sums (bwd_variant 2).  The
    # default flush forms those products at the 16-bit matrix rate from two f16 planes of exactly scaled operands (all four
    # partial products, 2^-22 per factor; the moments from three bf16 planes against an exact basis); storage and accumulation
    # are fp32 either way, this is the figure with one fp32 FMA chain per output.
    fp32_flush = None
    if not args.no_fp32_flush:
        _lib.set_option("bwd_variant", 2)
        try:
            for i in range(2):
                step(i)
            drain()
            barrier()
            f0 = time.perf_counter()
            nf = max(5, min(args.steps, 30))
            for i in range(nf):
                step(args.warmup + i)
            drain()
            barrier()
            f_elapsed = time.perf_counter() - f0
        finally:
            _lib.set_option("bwd_variant", 0)
        if dist is not None:
            tt = torch.tensor([f_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            f_elapsed = float(tt.item())
        fp32_flush = {"views_per_s": nf * world / f_elapsed, "ms_per_step": f_elapsed / nf * 1e3, "steps": nf,
                      "role": "cross-check only, not a product mode: same kernel template as the default (member masks, "
                              "prologue, LDS-DMA staging included); the whole difference is the flush's 32 fp32 matrix "
                              "instructions per 8 members, which run at the vector rate and add to the vector time on gfx950. "
                              "The float64 soak (flush_equivalence) is the yardstick that says the default flush loses nothing",
                      "what": "bwd_variant 2: per-Gaussian sums of the backward on v_mfma_f32_16x16x4_f32 (exact fp32 "
                              "FMA chains) instead of the split-f16 operands of the default flush"}

    # Secondary figure: the same step with the OPT-IN speculative depth cut-off of the tile lists (frames of a camera that was
    # rendered before list, per tile, only what that earlier frame found worth listing): two untimed passes over the cameras
    # (the first frames are exact, the next learn, the following ones are cut), then the timed steps.
    with_cut = None
    if not _C._FWD["depth_cut"] and _C._FWD["mode"] == "speculative":
        rasterizer.set_forward_mode(depth_cut=True)
        try:
            for i in range(2 * len(cams) + 4):
                step(i)
            drain()
            barrier()
            sp_c0 = rasterizer.speculation_stats()
            f0 = time.perf_counter()
            nf = max(5, min(args.steps, 30))
            seen_c = []
            for i in range(nf):
                step(args.warmup + i)
                seen_c.append(rasterizer.last_num_rendered())
            drain()
            barrier()
            wc_elapsed = time.perf_counter() - f0
            sp_c1 = rasterizer.speculation_stats()
            n_cut_mean = float(np.mean([int(c_) for c_ in seen_c]))
            del seen_c
        finally:
            rasterizer.set_forward_mode(depth_cut=False)
        if dist is not None:
            tt = torch.tensor([wc_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wc_elapsed = float(tt.item())
        with_cut = {"views_per_s": nf * world / wc_elapsed, "ms_per_step": wc_elapsed / nf * 1e3, "steps": nf,
                    "N_emitted_per_view": n_cut_mean, "cut_frames": sp_c1["cut_frames"] - sp_c0["cut_frames"],
                    "cut_failures": sp_c1["cut_failures"] - sp_c0["cut_failures"]}
        for i in range(4):
            step(i)
        drain()

    # Secondary figure: one iteration of the reference's semantic stage (train.py:112-199) with this build's pieces at the
    # workload's size: render -> code-book losses (goi_codebook_fused) -> backward -> three fused Adam steps; only the semantic
    # features, the decoder and the code book train (the reference's default).  tools/train_iter_bench.py also times the
    # PyTorch pieces around the same rasterizer.
    train_iter = None
    if world == 1 and not args.no_train_iteration and sc is not None and args.S <= 16:
        from goi_hyperplane_amd.optim import FusedAdam
        from goi_hyperplane_amd.semantic import SemanticModel, fused_codebook_losses
        tpc = GaussianSet.from_scene(sc, dev)
        for p_ in tpc.parameters():
            p_.requires_grad_(False)
        tpc._semantics.requires_grad_(True)
        mlp = SemanticModel(dim_in=args.S, dim_out=300, num_layer=1, use_bias=True, device=dev)
        lut = torch.nn.Parameter(torch.rand(300, 256, device=dev) * 0.03)
        gtl = torch.randn(256, args.H, args.W, device=dev)
        opts = [FusedAdam([{"params": [tpc._semantics], "lr": 5e-3, "name": "semantics"}], lr=0.0, eps=1e-15),
                FusedAdam(mlp.parameters(), lr=0.003), FusedAdam([lut], lr=0.001)]

        def train_it(i):
            o = render(cams[i % len(cams)], tpc, pipe, bg)
            loss, _terms = fused_codebook_losses(o["semantics"], mlp, lut, gtl, 10 + i)
            loss.backward()
            for o_ in opts:
                o_.step()
                o_.zero_grad(set_to_none=True)
        for i in range(4):
            train_it(i)
        torch.cuda.synchronize(dev)
        t0_ = time.perf_counter()
        nti = 12
        for i in range(nti):
            train_it(i)
        torch.cuda.synchronize(dev)
        train_iter = {"ms_per_iteration": (time.perf_counter() - t0_) / nti * 1e3, "iterations": nti,
                      "what": "render -> fused code-book losses (300 codes, 256-d ground truth) -> backward -> 3 fused Adam "
                              "steps; semantic features, decoder and code book trainable (train.py:112-199)"}
        # ... and with the opt-in geometry cache (the training cameras repeat every epoch and only the features move: after the
        # first pass over the cameras a frame is the blend alone)
        from goi_hyperplane_amd import rasterizer as _rz
        _rz.set_geometry_cache(64 << 30)
        try:
            for i in range(len(cams) + 2):
                train_it(i)
            torch.cuda.synchronize(dev)
            t0_ = time.perf_counter()
            for i in range(nti):
                train_it(i)
            torch.cuda.synchronize(dev)
            train_iter["ms_per_iteration_geometry_cache"] = (time.perf_counter() - t0_) / nti * 1e3
        finally:
            _rz.set_geometry_cache(0)
        del tpc, mlp, lut, gtl, opts
        torch.cuda.empty_cache()

    # Secondary figure: TWO independent views in flight on two HIP streams of this GPU (a micro-batch of views whose
    # gradients are accumulated, as in a multi-view batch).  Every view does the same forward + backward as in the timed
    # region; what overlaps is one view's latency-bound small kernels (sorts, scans) and kernel tails with the other
    # view's work.  train.py's own loop (one view, optimizer step, next view) cannot do this, so `value` stays the
    # one-view-at-a-time figure.
    two_streams = None
    if world == 1 and not args.no_two_streams:
        pcs = [pc, GaussianSet.from_scene(sc, dev) if sc is not None else GaussianSet.from_ply(args.ply, dev, sh_degree=3)]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

        def step2(i):
            k = i & 1
            with torch.cuda.stream(streams[k]):
                for p_ in pcs[k].parameters():
                    p_.grad = None
                o = render(cams[i % len(cams)], pcs[k], pipe, bg)
                torch.autograd.backward((o["render"], o["semantics"]), (g_color, g_sem))
        torch.cuda.synchronize(dev)
        for i in range(24):  # each stream's allocator pool (its own backward scratch: 4 x capacity x 129 bytes, 31 GB at 3 M
            step2(i)         # Gaussians) must have stopped growing: ten views were not enough at that size
        torch.cuda.synchronize(dev)
        s0 = time.perf_counter()
        n2 = max(10, min(args.steps, 60))
        for i in range(n2):
            step2(args.warmup + i)
        torch.cuda.synchronize(dev)
        s_elapsed = time.perf_counter() - s0
        two_streams = {"views_per_s": n2 / s_elapsed, "ms_per_view": s_elapsed / n2 * 1e3, "views": n2, "streams": 2,
                       "what": "two independent views in flight on two HIP streams of one GPU (same work per view)"}
        del pcs, streams
    # ... and the same as a PRODUCT path: dist.backward_views renders the K views of a multi-view batch (config 4's "8-view batch" on
    # fewer than 8 GPUs; --views-per-exchange K) alternating over two streams and leaves the sum of their gradients in p.grad
    batch_fig = None
    if world == 1 and not args.no_two_streams:
        from goi_hyperplane_amd.dist import backward_views
        batch_fig = {"what": "dist.backward_views: the K views of a batch between two optimiser steps, forward + backward each, their "
                             "gradients summed into p.grad; `serial` = one after the other on one stream, `two_streams` = alternating "
                             "over two HIP streams (the product path); views/s"}
        for K_ in (2, 4, 8):
            row = {}
            for label, ns in (("serial", 1), ("two_streams", 2)):
                def one_batch(b0):
                    vs = [cams[(b0 * K_ + k_) % len(cams)] for k_ in range(K_)]
                    backward_views(vs, lambda cam_: render(cam_, pc, pipe, bg),
                                   lambda o_, k_: ((o_["render"], o_["semantics"]), (g_color, g_sem)), params, streams=ns)
                for b0 in range(3):
                    one_batch(b0)
                torch.cuda.synchronize(dev)
                q0 = time.perf_counter()
                nb = max(3, min(args.steps, 32) // K_)
                for b0 in range(nb):
                    one_batch(b0)
                torch.cuda.synchronize(dev)
                row[label] = round(nb * K_ / (time.perf_counter() - q0), 2)
            row["gain"] = round(row["two_streams"] / row["serial"], 4)
            batch_fig[f"K={K_}"] = row
        for p_ in params:
            p_.grad = None

    # Secondary object: the SAME scene from a capture ORBIT instead of 16 cameras within +-0.16 rad of one direction
    # (scene.make_orbit_cameras: the look-at point travels around the scene, consecutive steps are seven orbit positions apart, as
    # train.py:118-124 pops its training cameras at random).  The cycle of near-identical views is the friendliest case for the
    # gradient-buffer pool (rows that already hold zeros are not rewritten), the capacity policy (2 x the high-water count) and
    # the L2-resident per-Gaussian arrays; this object says what the step costs when consecutive views see different Gaussians.
    _hw_main = int(_C._spec_state(dev).get("high_water", 0))  # (the main workload's; the orbit and the clustered scene below raise it)
    orbit = None
    if not args.no_orbit and not args.ply:
        from goi_hyperplane_amd.scene import ORBIT, make_orbit_cameras
        ocams = [TorchCamera(c_, dev) for c_ in make_orbit_cameras(args.W, args.H, fovx=WL["fovx"])]
        ext_ = _C._ext()

        def step_o(i):
            for p_ in params:
                p_.grad = None
            o_ = render(ocams[i % len(ocams)], pc, pipe, bg)
            torch.autograd.backward((o_["render"], o_["semantics"]), (g_color, g_sem))
            return o_
        # an untimed pass over the orbit: teaches the capacity policy the orbit's counts, and collects the visibility sets
        vis_o, n_o = [], []
        for i in range(len(ocams)):
            o_ = step_o(i)
            vis_o.append(o_["radii"] > 0)
            n_o.append(rasterizer.last_num_rendered())
        torch.cuda.synchronize(dev)
        n_o = [int(c_) for c_ in n_o]
        shared, kept = [], []
        for a_, b_ in zip(vis_o, vis_o[1:] + vis_o[:1]):  # consecutive steps (a_ then b_)
            shared.append(float((a_ & b_).sum()) / max(1.0, float(b_.sum())))        # of b_'s visible Gaussians, seen by a_ as well
            kept.append(float((~a_ & ~b_).sum()) / max(1.0, float((~b_).sum())))     # of b_'s invisible rows, already zero after a_
        V_o = float(np.mean([float(v_.sum()) for v_ in vis_o]))
        del vis_o
        sp0 = rasterizer.speculation_stats()
        pool0 = ext_.grad_pool_stats() if ext_ is not None else None
        for i in range(4):
            step_o(i)
        barrier()
        o0 = time.perf_counter()
        no = max(16, min(args.steps, 48)) // 16 * 16  # whole cycles of the orbit
        for i in range(no):
            step_o(i)
        barrier()
        o_el = time.perf_counter() - o0
        sp1 = rasterizer.speculation_stats()
        pool1 = ext_.grad_pool_stats() if ext_ is not None else None
        n_last = rasterizer.last_num_rendered()
        orbit = {"views_per_s": no / o_el, "ms_per_step": o_el / no * 1e3, "steps": no, "cameras": len(ocams),
                 "relative_to_value": None,  # (filled in below: orbit views/s per GPU / value per GPU)
                 "orbit": dict(ORBIT), "V": V_o, "N_listed_per_view": float(np.mean(n_o)), "N_listed_min_max": [min(n_o), max(n_o)],
                 "visible_shared_with_previous_step": {"mean": round(float(np.mean(shared)), 4), "max": round(max(shared), 4)},
                 "gradient_pool": {"zero_rows_already_zero": {"mean": round(float(np.mean(kept)), 4), "min": round(min(kept), 4)},
                                   "buffers": (None if pool0 is None else
                                               dict(zip(("hits", "dirty", "fresh"), (int(b_ - a_) for a_, b_ in zip(pool0, pool1))))),
                                   "what": "zero_rows_already_zero: of the Gaussians a step does not see, the fraction the previous "
                                           "step did not see either (their gradient rows hold zeros and are not rewritten); buffers: "
                                           "pooled gradient buffers handed out again / found modified in place / freshly allocated "
                                           "during the timed steps"},
                 "speculation": {k: sp1[k] - sp0[k] for k in sp1}, "binning_capacity": int(getattr(n_last, "capacity", 0) or 0),
                 "what": "the same step (forward + backward, all gradients, the same dense upstream gradients) on the SAME scene with "
                         "scene.make_orbit_cameras: 16 cameras whose look-at points travel around the scene, visited seven positions "
                         "apart; per GPU, no gradient exchange"}
        del ocams
        for i in range(4):  # back to the main cameras' steady state for the objects below
            step(i)
        drain()

    # Secondary object: the ADVERSARIAL workload -- a scene with the statistics of a reconstruction (clustered density,
    # heavy-tailed anisotropic sizes, opaque foreground with lists thousands deep behind it: scene.make_clustered_scene) at the
    # headline's size and image, through the same step.  Not `value`; its oracle parity is tests/test_gpu_clustered.py.
    clustered = None
    if world == 1 and args.scene == "headline" and not args.no_clustered and not args.ply:
        from goi_hyperplane_amd.scene import make_workload
        sc2, _cam2, spec2 = make_workload("clustered")
        pc2 = GaussianSet.from_scene(sc2, dev)
        del sc2
        cams2 = make_cams(spec2, spec2["W"], spec2["H"], 16)
        gen2 = torch.Generator(device=dev).manual_seed(777)
        inv2 = 1.0 / (spec2["W"] * spec2["H"])
        gc2 = torch.randn((3, spec2["H"], spec2["W"]), device=dev, generator=gen2) * inv2
        gs2 = torch.randn((spec2["S"], spec2["H"], spec2["W"]), device=dev, generator=gen2) * inv2
        params2 = list(pc2.parameters())

        def step_c(i):
            for p_ in params2:
                p_.grad = None
            o_ = render(cams2[i % len(cams2)], pc2, pipe, bg)
            torch.autograd.backward((o_["render"], o_["semantics"]), (gc2, gs2))
            return o_
        sp0 = rasterizer.speculation_stats()
        for i in range(8):  # (the first frames of a scene are exact and teach the capacity policy)
            step_c(i)
        torch.cuda.synchronize(dev)
        n_last = rasterizer.last_num_rendered()
        cap2 = int(getattr(n_last, "capacity", 0) or 0)
        # instance counts: the reference's lists (cull_variant 0) and this build's, over four of the cameras
        nstat = {"N": 0, "N_listed": 0, "V": 0, "views": 0}
        with torch.no_grad():
            spec_saved2 = {k: dict(v, pending=v["pending"]) for k, v in _C._SPEC.items()}
            fm = _C._FWD["mode"]
            _C.set_forward_mode(speculative=False)
            for cam_ in cams2[:4]:
                for variant in (0, 2):
                    _lib.set_option("cull_variant", variant)
                    n_, *_r = _C.rasterize_gaussians(bg, pc2._xyz, torch.Tensor([]), pc2._semantics, pc2._opacity, pc2._scaling,
                                                     pc2._rotation, 1.0, torch.Tensor([]), cam_.world_view_transform,
                                                     cam_.full_proj_transform, np.tan(cam_.FoVx * 0.5), np.tan(cam_.FoVy * 0.5),
                                                     spec2["H"], spec2["W"], pc2._features, 3, cam_.camera_center, False, False)
                    if variant == 0:
                        nstat["N"] += n_
                        nstat["V"] += int((_r[4] > 0).sum())
                        nstat["views"] += 1
                    else:
                        nstat["N_listed"] += n_
                        if "lane" not in nstat:
                            nstat["lane"] = lane_utilisation(_C.blend_stats(spec2["P"], spec2["W"], spec2["H"], n_, *_r[-3:]))
                    del _r
            _C._SPEC.clear()
            _C._SPEC.update(spec_saved2)
            _C.set_forward_mode(speculative=fm == "speculative")
        st2 = {}
        if timing:
            _lib.profile_collect()
            _lib.profile_enable(True)
            for i in range(8):
                step_c(i)
            torch.cuda.synchronize(dev)
            _lib.profile_enable(False)
            st2 = {k: round(ms / max(c, 1), 4) for k, (ms, c) in _lib.profile_collect().items() if c}
        torch.cuda.synchronize(dev)
        c0 = time.perf_counter()
        nc = max(5, min(args.steps, 30))
        for i in range(nc):
            step_c(i)
        torch.cuda.synchronize(dev)
        c_el = time.perf_counter() - c0
        sp1 = rasterizer.speculation_stats()
        clustered = {"views_per_s": nc / c_el, "ms_per_step": c_el / nc * 1e3, "steps": nc,
                     "P": spec2["P"], "W": spec2["W"], "H": spec2["H"], "S": spec2["S"],
                     "V": nstat["V"] / nstat["views"], "N_per_view": nstat["N"] / nstat["views"],
                     "N_listed_per_view": nstat["N_listed"] / nstat["views"],
                     "speculation": {k: sp1[k] - sp0[k] for k in sp1}, "binning_capacity": cap2,
                     "backward_scratch_bytes": int(_lib.load().goi_raster_backward_scratch_bytes(cap2, spec2["S"])) if cap2 else None,
                     "binning_bytes": int(_lib.load().goi_raster_binning_bytes(cap2)) if cap2 else None,
                     "stages_ms": st2, "blend_lane_utilisation": nstat.get("lane"),
                     "what": "the same step (forward + backward, all gradients, dense upstream gradients) on "
                             "scene.make_clustered_scene: mixture-of-clusters positions, log-normal scales (sigma 1.1) with 2 % "
                             "needles and frame-filling blobs, bimodal opacity, opaque foreground sheets; 16 cameras"}
        del pc2, cams2, gc2, gs2, params2
        _C.release_scratch(dev)
        torch.cuda.empty_cache()

    # bytes one exchange puts on the wire per rank, and SURVEY.md 8(e)'s link model for them
    row_bytes = int(sum(p_.numel() // max(1, p_.shape[0]) for p_ in reduce_params) * 4)
    if world <= 1:
        ar_bytes, ag_bytes = 0, 0
    elif exchange["mode"] == "factored":
        ar_bytes, ag_bytes = int(sum(p_.numel() for p_ in non_sh_params) * 4), int(args.P * 3 * 4 * world)
    elif exchange["mode"] == "visible":
        ar_bytes, ag_bytes = int(stats.get("rows_sent", args.P)) * row_bytes + args.P, 0  # + the P-byte visibility mask
    else:
        ar_bytes, ag_bytes = int(sum(p_.numel() for p_ in reduce_params) * 4), 0
    modelled = None
    if world > 1:
        m = exchange_model_ms(ar_bytes, world)
        gather_ms = ag_bytes * (world - 1) / world / (min(7, world - 1) * 153e9) * 1e3  # every rank receives (G-1)/G of it
        modelled = {"ring_ms": round(m["ring"] + gather_ms, 4), "direct_ms": round(m["direct"] + gather_ms, 4),
                    "per_view_ring_ms": round((m["ring"] + gather_ms) / K, 4),
                    "what": "SURVEY.md 8(e): 7 xGMI links x 153 GB/s per GPU; a ring all-reduce is bound by one link, a direct "
                            "reduce-scatter + all-gather uses all of them; per exchange, and per view at --views-per-exchange"}
    if rank == 0:
        sb = stage_bytes(args.P, V, N, T, HW, args.S)
        b_fwd, b_bwd = survey_bytes(args.P, V, N, T, HW, args.S)
        stage_out = {}
        dominant, dom_ms = None, -1.0
        for name, (ms, calls) in stages.items():
            if calls == 0:
                continue
            avg = ms / calls
            gbs = sb[name] / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            stage_out[name] = {"ms": round(avg, 4), "launches": calls, "alg_MB": round(sb[name] / 1e6, 2),
                               "alg_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
            if avg > dom_ms:
                dominant, dom_ms = name, avg
        roofline = None
        if dominant:
            d = stage_out[dominant]
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": d["alg_GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": d["hbm_frac"], "traffic": None,
                        "bytes_per_launch": sb[dominant], "avg_ms": d["ms"]}
            # the same formula on the instances this build actually LISTS (ellipse tile lists: about half of the reference's
            # num_rendered): what the kernel walks, as opposed to what SURVEY.md 8(d) prices the reference's lists at
            n_listed = stats["N_listed"] / max(stats["views"], 1)
            sb_listed = stage_bytes(args.P, V, n_listed, T, HW, args.S)[dominant]
            roofline["bytes_per_launch_listed"] = sb_listed
            roofline["frac_listed"] = round(sb_listed / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if d["ms"] > 0 else None
            # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process);
            # only quoted for the workload they were collected on
            import glob
            pmcs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                 "r*_pmc_traffic.json")))
            pmc = pmcs[-1] if pmcs else ""  # the newest committed set (names sort by round and build letter)
            headline = (args.P, args.W, args.H, args.S, args.mu) == (HEADLINE["P"], HEADLINE["W"], HEADLINE["H"],
                                                                     HEADLINE["S"], HEADLINE["log_scale_mean"])
            if headline and pmc:
                tj = json.load(open(pmc))
                k = tj["kernels"].get(tj["stage_kernel"].get(dominant, ""))
                if k:
                    roofline["traffic"] = k["hbm_bytes_corrected"]
                    # counter bytes / this run's duration / peak: the fraction of the HBM peak the kernel actually draws
                    roofline["frac_counter"] = round(k["hbm_bytes_corrected"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    roofline["traffic_source"] = ("profiles/" + os.path.basename(pmc) + ": 2*FETCH_SIZE + WRITE_SIZE of "
                                                  + tj["stage_kernel"][dominant] + ", separate rocprofv3 --pmc passes")
        gpu_ms = sum(v["ms"] for v in stage_out.values())
        ms_per_step = elapsed / args.steps * 1e3
        # device memory of one view in flight at this workload: the workspaces are sized for the speculative frame's CAPACITY
        # (headroom x the largest count seen), not for its count -- the count is not on the host when the frame is enqueued
        _hw = _hw_main
        _cap = max(_C._MIN_CAPACITY, int(_C._FWD["headroom"] * _hw) + 4096) if _hw else 0
        _l = _lib.load()
        memory = None if not _cap else {
            "num_rendered_high_water": _hw, "binning_capacity": _cap, "headroom": _C._FWD["headroom"],
            "geometry_bytes": int(_l.goi_raster_geom_bytes(args.P)), "image_bytes": int(_l.goi_raster_image_bytes(args.W, args.H)),
            "binning_bytes": int(_l.goi_raster_binning_bytes(_cap)),
            "backward_scratch_bytes": int(_l.goi_raster_backward_scratch_bytes(_cap, args.S)),
            "backward_scratch_bytes_if_sized_by_count": int(_l.goi_raster_backward_scratch_bytes(_hw, args.S)),
            "row_scratch_layouts_this_process": dict(_C.SCRATCH_STATS),
            "what": "per view in flight; the row scratch (4 quadrant rows of 128 B + a validity byte per listed instance) is laid out "
                    "for the frame's COUNT when that has reached the host by the time the backward is enqueued (goi_raster_backward3; "
                    "a free poll, nothing waits) and for its capacity otherwise -- in this bench the backward follows the forward "
                    "at once, so most frames are sized by capacity (row_scratch_layouts_this_process counts both cases over the "
                    "whole run); with a loss between the two (tests/test_gpu_speculative.py) it is the count.  The buffer is "
                    "grow-only and shared by all frames of a (device, stream)"}
        res = {
            "metric": "training views/sec (rasterizer fwd+bwd), 1M Gaussians @1600x1056 RGB+16-d feat",
            "value": args.steps * world / elapsed, "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "repeats_ms_per_step": repeats_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.ply else "ply scene, synthetic cameras",
            "config": {"workload": f"{args.scene}: {args.P} Gaussians @{args.W}x{args.H}, SH deg 3 RGB + {args.S}-d semantic, fwd+bwd"
                                   f"{' + RCCL all-reduce(' + args.grads + ' grads)' if world > 1 else ''}",
                       "P": args.P, "V": V, "N_per_view": N, "N_listed_per_view": stats["N_listed"] / stats["views"],
                       "tiles": T, "HW": HW, "S": args.S,
                       "views_per_step": world, "parallelism": f"views sharded x{world}",
                       "allreduce_bytes": ar_bytes, "allgather_bytes": ag_bytes,
                       "rows_sent": (stats.get("rows_sent") if exchange["mode"] == "visible" else None),
                       "exchange": exchange["mode"] if world > 1 else None, "exchange_note": exchange["note"],
                       "collective": (None if world <= 1 else
                                      ("reduce_scatter + all_gather over the flat gradient span (dist.allreduce_gradients_direct)"
                                       if direct else "all_reduce (the algorithm is RCCL's choice)")),
                       "collective_chosen_by": (None if world <= 1 else
                                                ("the link model (dist.pick_exchange)" if args.collective == "auto"
                                                 else "--collective")),
                       "exchange_semantics": (None if world <= 1 else
                                              f"one sum all-reduce per {K} view(s) per rank, " +
                                              ("left in flight behind the next step (stale by one step)" if overlap_default
                                               else "consumed before the next step starts (train.py: optimiser step per exchange)"))},
            "render_ms_per_frame": render_ms,
            "render_forward_mode": render_mode,  # forward of frames rendered without autograd (package default: exact)
            "render_ms_per_frame_speculative": render_ms_spec,
            "gui_frame_ms": gui_ms,  # render + fused semantic decode (300 codes)
            "step_enqueue_ms": step_enqueue_ms,  # host-side, informational (stall detector; not used for value)
            "workload_orbit": orbit,
            "workload_clustered": clustered,
            "memory_per_view": memory,
            "blend_lane_utilisation": stats.get("lane"),
            "semantic_finetune": sem_only,
            "value_fp32_flush": None if fp32_flush is None else fp32_flush["views_per_s"],
            "value_with_depth_cut": None if with_cut is None else with_cut["views_per_s"],
            "depth_cut": {"enabled_for_value": bool(_C._FWD["depth_cut"]), "opt_in_figure": with_cut,
                          "N_emitted_per_view_in_profile_pass": stats.get("N_emitted"),
                          "what": "OPT-IN (GOI_DEPTH_CUT=1): speculative training frames of a camera that was rendered before list, "
                                  "per tile, only Gaussians up to the depth that camera's previous frame found worth listing (+1/8 + "
                                  "32 list positions); bit-identical outputs, gradients equal up to one fp32 summation order while the "
                                  "cut holds (tests/test_gpu_depth_cut.py), a redone or skipped view when it does not"},
            "semantic_train_iteration": train_iter,
            "value_two_views_in_flight": None if two_streams is None else two_streams["views_per_s"],
            "two_views_in_flight": two_streams,
            "views_in_flight_batch": batch_fig,
            "fp32_flush": fp32_flush,
            # forward mode of the timed region and what the speculation did in it (exact_frames / waits / overflows
            # should all be 0: nothing in the timed steps waited for the device)
            "exchange_overlap": (None if world <= 1 else
                                 ("in flight behind the next step's render + backward (one-step-stale gradients)"
                                  if overlap_default else "waited for inside the step (train.py's semantics)")),
            "serialized_exchange": serialized,  # same steps, exchange waited for inside each step (when value is in flight)
            "exchange_in_flight": in_flight_fig,  # same steps, exchange left in flight (when value waits: the default)
            "views_per_exchange": K,
            "modelled_exchange_ms": modelled,
            "per_rank": per_rank,
            "binding": _C.binding(),
            "forward_mode": _C._FWD["mode"],
            "speculation": {k: spec1[k] - spec0[k] for k in spec1},
            "roofline": roofline,
            "whole_view": {"alg_bytes_fwd": b_fwd, "alg_bytes_bwd": b_bwd,
                           "alg_GBps_over_step": (b_fwd + b_bwd) / (ms_per_step * 1e-3) / 1e9,
                           "hbm_frac_over_step": (b_fwd + b_bwd) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "gpu_stage_ms_sum": gpu_ms},
            "stages": stage_out,
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_view(cam_np, bg_np, grads_np):
                tcam = TorchCamera(cam_np, dev)
                for p_ in params:
                    p_.grad = None
                o_ = render(tcam, pc, pipe, torch.tensor(bg_np, device=dev))
                torch.autograd.backward((o_["render"], o_["semantics"], o_["depth"], o_["alpha"]),
                                        tuple(torch.tensor(g_, device=dev) for g_ in grads_np))
                out_np = {k: o_[k].detach().cpu().numpy() for k in ("render", "semantics", "depth", "alpha", "radii")}
                g_np = dict(means3D=pc._xyz.grad, opacity=pc._opacity.grad, semantics=pc._semantics.grad,
                            sh=pc._features.grad, scales=pc._scaling.grad, rotations=pc._rotation.grad,
                            means2D=o_["viewspace_points"].grad)
                return out_np, {k: v.detach().cpu().numpy() for k, v in g_np.items()}
            res["parity"], res["cpu_baseline"] = cpu_baseline(args, int(N), gpu_view)
            # Does the default (split-f16) flush of the backward cost accuracy against the exact-fp32 flush?  On THIS run's
            # metric view: both flushes against the oracle and against each other (parity.fp32_flush, .flush_equivalence_same_
            # view); over the fuzz sweep's random configurations: the committed soak (tools/flush_soak.py, profiles/).
            import glob
            soaks = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flush_equivalence.json")))
            soak = json.load(open(soaks[-1])) if soaks else None
            same = (res["parity"] or {}).get("flush_equivalence_same_view")
            res["flush_equivalence"] = {
                "same_view": same,
                "soak": None if soak is None else {k: soak[k] for k in soak if k in ("configurations", "seed", "float64", "criterion",
                                                                                    "equivalent", "slack")},
                "soak_source": ("profiles/" + os.path.basename(soaks[-1])) if soaks else None,
                "value_is_fp32_grade": bool(soak and soak.get("equivalent") and same is not None and res["parity"]["ok"]
                                            and res["parity"]["fp32_flush"]["ok"]),
                "what": "`value` is measured with the backward's per-Gaussian sums formed at the 16-bit matrix rate from split "
                        "operands (two f16 planes of exactly scaled values, all four partial products; three bf16 planes for the "
                        "moments), `value_fp32_flush` with one fp32 FMA chain per output.  Equivalent means: on every soaked "
                        "configuration and gradient tensor the two flushes differ by no more than two legal builds of the "
                        "reference differ from each other, or else the default flush is not further from the float64 reference "
                        "than the fp32 flush by more than that spread plus the slack (1 % of the gradient tolerance); the "
                        "exceedances without the slack are listed in both directions"}
        else:
            res["parity"], res["cpu_baseline"] = None, None
        if orbit is not None:
            orbit["relative_to_value"] = round(orbit["views_per_s"] / (res["value"] / world), 4)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
