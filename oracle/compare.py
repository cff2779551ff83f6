"""Error measures of the parity tests and of bench.py's oracle leg -- TEST INFRASTRUCTURE ONLY (see oracle/mrgs_oracle.c).

Two figures per gradient tensor, as SURVEY.md section 7 asks:
  * max-norm:   max|a - ref| / max|ref|                     (BASELINE.json's "grad max-rel-err")
  * per-element relative error |a - ref| / |ref| over the elements with |ref| > floor * max|ref| (floor = 1e-3):
    median, 99th percentile and maximum of that distribution.
"""
import numpy as np

GRAD_KEYS = ("means3D", "opacity", "scales", "rotations", "sh", "features", "means2D")


def max_norm_err(a, ref):
    a = np.asarray(a, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    if ref.size == 0:
        return 0.0
    den = np.abs(ref).max()
    return float(np.abs(a - ref).max() / den) if den > 0 else float(np.abs(a).max())


def elementwise_rel(a, ref, floor=1e-3):
    """Distribution of |a - ref| / |ref| over the elements above floor * max|ref|: dict(n, median, p99, max)."""
    a = np.asarray(a, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    if ref.size == 0 or np.abs(ref).max() == 0:
        return {"n": 0, "median": 0.0, "p99": 0.0, "max": 0.0}
    m = np.abs(ref) > floor * np.abs(ref).max()
    r = np.abs(a[m] - ref[m]) / np.abs(ref[m])
    return {"n": int(m.sum()), "median": float(np.median(r)), "p99": float(np.percentile(r, 99)), "max": float(r.max())}


def grad_report(grads, truth, keys=GRAD_KEYS):
    """{tensor: {"max_norm": .., "elem": {...}}} for every key present in both dictionaries with a non-empty truth tensor."""
    out = {}
    for k in keys:
        if k in grads and k in truth and np.asarray(truth[k]).size:
            a = np.asarray(grads[k]).reshape(np.asarray(truth[k]).shape)
            out[k] = {"max_norm": max_norm_err(a, truth[k]), "elem": elementwise_rel(a, truth[k])}
    return out


def three_way(hip, fused, lit32, f64, keys=GRAD_KEYS):
    """Distances from the float64 evaluation of the reference's formulas: the HIP kernels (may be None), the fp32 oracle with the
    kernels' FMA pattern, and the literal un-fused fp32 reading.  Returns {tensor: {"hip": .., "fused32": .., "literal32": ..}}
    of max-norm errors plus the same for the 99th percentile of the per-element relative error."""
    legs = {"fused32": fused, "literal32": lit32}
    if hip is not None:
        legs["hip"] = hip
    rep = {name: grad_report(g, f64, keys) for name, g in legs.items()}
    out = {}
    for k in rep["literal32"]:
        out[k] = {name: {"max_norm": float(f"{rep[name][k]['max_norm']:.3e}"), "elem_p99": float(f"{rep[name][k]['elem']['p99']:.3e}"),
                         "elem_median": float(f"{rep[name][k]['elem']['median']:.3e}")} for name in rep if k in rep[name]}
    return out
