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


def forward_stats(hip: dict, f, tol: float = FWD_TOL) -> dict:
    """hip: {"render","semantics","depth","alpha","radii"} numpy arrays; f: oracle.ForwardResult."""
    ok = f.fragile.reshape(-1) == 0
    out = {"fragile_frac": float(1.0 - ok.mean()), "radii_equal": bool((np.asarray(hip["radii"]) == f.radii).all())}
    worst, p9999, n_over = 0.0, 0.0, 0
    for k, a in (("render", f.color), ("semantics", f.semantic), ("depth", f.depth), ("alpha", f.alpha)):
        d = np.abs(np.asarray(hip[k], np.float32).reshape(a.shape) - a).reshape(a.shape[0], -1)
        ds = d[:, ok]
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
