"""Shared helpers for the HIP-vs-oracle parity tests (test infrastructure)."""
import numpy as np

# Stated fp64 tolerance (BASELINE.json north_star: "within 1e-8 relative of the reference").
# Entry-wise criterion: |gpu - ref| <= RTOL * max(|ref|, FLOOR * ||ref||_inf of the block), i.e. relative
# 1e-8 on every entry that is not itself below 1e-6 of the block's largest entry (entries that small are
# sums of cancelling O(||.||) terms).
RTOL = 1e-8
FLOOR = 1e-6


def rel_err(a, ref):
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.abs(ref), FLOOR * np.abs(ref).max() if ref.size else 0.0)
    scale = np.where(scale == 0, 1.0, scale)
    return float((np.abs(a - ref) / scale).max()) if ref.size else 0.0


def assert_parity(gpu, ref, what=""):
    v, d, h, cnt, st = gpu
    ov, od, oh, ocnt, ost = ref
    assert np.array_equal(st, ost), (what, st, ost)
    assert np.array_equal(cnt, ocnt), (what, "pixel counters", cnt, ocnt)
    ev = float(np.max(np.abs(v - ov) / np.abs(ov)))
    assert ev <= RTOL, (what, "value", ev)
    errs = {"v": ev}
    if d is not None:
        for t in range(len(v)):
            e = rel_err(d[t], od[t]); errs["d"] = max(errs.get("d", 0), e)
            assert e <= RTOL, (what, "gradient", t, e)
    if h is not None:
        for t in range(len(v)):
            assert np.array_equal(h[t], h[t].T), (what, "Hessian not exactly symmetric")
            e = rel_err(h[t], oh[t]); errs["h"] = max(errs.get("h", 0), e)
            assert e <= RTOL, (what, "hessian", t, e)
    return errs
