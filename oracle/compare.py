"""Parity statistics of a HIP result against the CPU oracle's.  TEST INFRASTRUCTURE ONLY (tests/, bench.py's
cpu_baseline leg): numpy in, plain dicts out, no assertion -- the callers decide what is a failure.

Tolerances are north_star's: forward outputs within 1e-4 absolute outside the oracle's "fragile" pixels (a blend
guard of CR/forward.cu:341-357 within 1e-4 relative of flipping: two correct fp32 implementations disagree there),
gradients within 1e-3 of each tensor's scale (its largest magnitude in the oracle's result).
"""
from __future__ import annotations

import numpy as np

FWD_TOL = 1e-4
BWD_TOL = 1e-3
GRAD_NAMES = ("means3D", "opacity", "semantics", "sh", "scales", "rotations", "means2D")


def forward_stats(hip: dict, f, tol: float = FWD_TOL, f_exact=None) -> dict:
    """hip: {"render","semantics","depth","alpha","radii"} numpy arrays; f: oracle.ForwardResult.

    The oracle's flag byte (ForwardResult.fragile): bit 0 = a blend guard within 1e-4 relative of flipping, bit 1 = the pixel
    is ILL-CONDITIONED for the reference's own fp32 evaluation of the pair exponent (needles: oracle/goi_oracle.cpp header).
    Pixels with any bit set are left out of the comparison with `f` -- unless f_exact (the forward of the oracle's
    "f64power" variant: the same arithmetic with that one statement evaluated exactly) is given: then every pixel that is
    not guard-fragile in either result is compared with BOTH and must be within the tolerance of one of them (the plain
    build where the reference's fp32 evaluation is well-conditioned, the exact evaluation of the reference's formula where
    it is not: the flag is a bound on first-order effects and misses pixels whose transmittance has drifted over a list
    thousands deep); only the guard-fragile pixels are left out."""
    flags = f.fragile.reshape(-1)
    ok = flags == 0
    out = {"fragile_frac": float(1.0 - ok.mean()), "radii_equal": bool((np.asarray(hip["radii"]) == f.radii).all()),
           "guard_fragile_frac": float(((flags & 1) != 0).mean()), "ill_conditioned_frac": float(((flags & 2) != 0).mean())}
    either = None
    if f_exact is not None:
        either = ((flags & 1) == 0) & ((f_exact.fragile.reshape(-1) & 1) == 0)
        out["fragile_frac"] = float(1.0 - either.mean())  # what is compared with NEITHER result
    worst, p9999, n_over = 0.0, 0.0, 0
    for k, a, attr in (("render", f.color, "color"), ("semantics", f.semantic, "semantic"), ("depth", f.depth, "depth"),
                       ("alpha", f.alpha, "alpha")):
        h = np.asarray(hip[k], np.float32).reshape(a.shape)
        d = np.abs(h - a).reshape(a.shape[0], -1)
        ds = d[:, ok]
        if either is not None:
            ds = np.minimum(d, np.abs(h - getattr(f_exact, attr)).reshape(a.shape[0], -1))[:, either]
        st = {"max": float(ds.max()) if ds.size else 0.0,
              "p9999": float(np.quantile(ds, 0.9999)) if ds.size else 0.0,
              "n_over": int((ds > tol).sum()), "n": int(ds.size),
              "max_incl_fragile": float(d.max()) if d.size else 0.0}
        out[k] = st
        worst, p9999, n_over = max(worst, st["max"]), max(p9999, st["p9999"]), n_over + st["n_over"]
    out.update(fwd_max=worst, fwd_p9999=p9999, n_over_tol=n_over)
    return out


def backward_stats(g_hip: dict, g_orc: dict, tol: float = BWD_TOL, names=GRAD_NAMES) -> dict:
    """Per gradient tensor: max and 99.99th percentile of |hip - oracle| / scale, the number of elements over `tol`,
    and the largest error relative to max(|element|, 1e-3 scale)."""
    out = {}
    for name in names:
        a, b = g_hip.get(name), g_orc.get(name)
        if a is None or b is None:
            continue
        a = np.asarray(a)
        b = np.asarray(b).reshape(a.shape)
        scale = float(np.abs(b).max()) + 1e-20
        d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / scale
        out[name] = dict(max=float(d.max()) if d.size else 0.0,
                         p9999=float(np.quantile(d, 0.9999)) if d.size else 0.0,
                         n_over=int((d > tol).sum()), n=int(d.size), scale=scale, finite=bool(np.isfinite(a).all()),
                         elem_rel_max=float((d * scale / np.maximum(np.abs(b), 1e-3 * scale)).max()) if d.size else 0.0)
    return out


def backward_stats_arbitrated(g_hip: dict, builds: list, per_gaussian=("means3D", "scales", "rotations"), tol: float = BWD_TOL,
                              names=GRAD_NAMES) -> dict:
    """Gradient parity where the reference ITSELF is noisy (needles: its cov2D -> cov3D -> scale / rotation backward cancels
    catastrophically in fp32, and legal builds of it disagree by the size of the gradient).  builds: the oracle's gradient
    dicts, the plain build first (then e.g. the exact-exponent and FMA-contracted ones).  Per tensor, relative to the plain
    build's scale:
      * an element PASSES if it is within `tol` of ANY build;
      * for the per-Gaussian geometry tensors a Gaussian is UNDECIDED when the builds disagree with each other by more than
        `tol` / 3 on any element of its row in any of those tensors (a third: this implementation's own rounding noise on
        such a row is of the size of the builds' -- the fuzz test's "three times the builds' disagreement" criterion, row by
        row): nothing can be pinned there, so such rows are only required to be finite and not blown up (|hip| <= 10 x the
        largest build value + tol); everywhere else every element must pass.
    Returns per tensor: max error against the NEAREST build over decided elements, elements failing, undecided rows."""
    plain = builds[0]
    und = None
    for name in per_gaussian:
        b0 = np.asarray(plain[name], np.float64)
        scale = float(np.abs(b0).max()) + 1e-20
        spread = np.zeros(b0.shape[0])
        for other in builds[1:]:
            spread = np.maximum(spread, np.abs(np.asarray(other[name], np.float64).reshape(b0.shape) - b0).reshape(b0.shape[0], -1).max(axis=1) / scale)
        u = spread > tol / 3.0
        und = u if und is None else (und | u)
    out = {"undecided_rows": int(und.sum()) if und is not None else 0}
    for name in names:
        a = g_hip.get(name)
        if a is None or plain.get(name) is None:
            continue
        a = np.asarray(a, np.float64)
        b0 = np.asarray(plain[name], np.float64).reshape(a.shape)
        scale = float(np.abs(b0).max()) + 1e-20
        d = np.full(a.shape, np.inf)
        big = np.zeros(a.shape)
        for b in builds:
            bb = np.asarray(b[name], np.float64).reshape(a.shape)
            d = np.minimum(d, np.abs(a - bb) / scale)
            big = np.maximum(big, np.abs(bb))
        decided = np.ones(a.shape[0], bool) if (name not in per_gaussian or und is None) else ~und
        dd = d.reshape(a.shape[0], -1)
        dec = dd[decided]
        undec_rows = ~decided
        blown = (np.abs(a).reshape(a.shape[0], -1)[undec_rows] > 10 * big.reshape(a.shape[0], -1)[undec_rows] + tol * scale)
        out[name] = dict(max=float(dec.max()) if dec.size else 0.0, p9999=float(np.quantile(dec, 0.9999)) if dec.size else 0.0,
                         n_over=int((dec > tol).sum()), n=int(dec.size), scale=scale, finite=bool(np.isfinite(a).all()),
                         undecided_rows=int(undec_rows.sum()), undecided_blown_up=int(blown.sum()),
                         max_undecided=float(dd[undec_rows].max()) if undec_rows.any() else 0.0)
    return out


def summary(fwd: dict, bwd: dict) -> dict:
    """The compact object bench.py prints as `parity`."""
    return {"fwd_max": fwd["fwd_max"], "fwd_p9999": fwd["fwd_p9999"], "fwd_n_over_tol": fwd["n_over_tol"],
            "fragile_frac": fwd["fragile_frac"], "radii_equal": fwd["radii_equal"],
            "grad_max_by_tensor": {k: v["max"] for k, v in bwd.items()},
            "grad_p9999_by_tensor": {k: v["p9999"] for k, v in bwd.items()},
            "grad_n_over_tol": int(sum(v["n_over"] for v in bwd.values())),
            "fwd_tol": FWD_TOL, "grad_tol": BWD_TOL,
            "ok": bool(fwd["fwd_max"] < FWD_TOL and fwd["radii_equal"] and fwd["fragile_frac"] < 0.02
                       and all(v["max"] < BWD_TOL and v["finite"] for v in bwd.values()))}
