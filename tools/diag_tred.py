"""Diagnostic (gpurun, CELESTE_MI355X_LIB = a -DOPTIM_DEBUG_T build): the tridiagonal form and Q'g the device computes
against a numpy emulation of the same Householder reduction."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from celeste_jl_amd import cabi
import tr_reference as R
NF = 41


def tred(H, g):
    a = H.copy(); v = g.copy(); td = np.zeros(NF); ev = np.zeros(NF); hv = np.zeros(NF); ln = np.arange(NF)
    for I in range(NF - 1, 0, -1):
        l = I - 1; act = ln <= l; td[I] = a[I, I]; x = np.where(act, a[:, I], 0.0)
        if l == 0: ev[I] = x[0]; continue
        f = x[l]; hoff = np.sum(np.where(ln < l, x * x, 0))
        if hoff == 0: ev[I] = f; continue
        h = hoff + f * f; gg = -np.sqrt(h) if f >= 0 else np.sqrt(h); h -= f * gg
        u = np.where(ln == l, f - gg, x); ev[I] = gg; hv[I] = h
        vu = np.sum(u * v); p = np.where(act, (a[:, :l + 1] @ u[:l + 1]) / h, 0); v = v - (vu / h) * u
        q = p - (np.sum(p * u) * 0.5 / h) * u
        for k in range(l + 1): a[:, k] -= u * q[k] + q * u[k]
    td[0] = a[0, 0]
    return td, ev, hv, v


lib = cabi.load_library()
lib.celeste_debug_T.argtypes = [C.c_int32, C.POINTER(C.c_double)]
lib.celeste_debug_T(0, None)
probs = R.random_problems(np.random.default_rng(11))
H = np.stack([p[1] for p in probs]); g = np.stack([p[2] for p in probs]); d = np.array([p[3] for p in probs])
cabi.tr_solve_batch(H, g, d, solver=2)
out = np.zeros((len(probs), 4, NF))
lib.celeste_debug_T(len(probs), out.ctypes.data_as(C.POINTER(C.c_double)))
for k, (name, Hk, gk, dk) in enumerate(probs):
    td, ev, hv, v = tred(Hk, gk)
    sc = np.abs(Hk).max()
    print("%-32s |td - emul| %.1e  |te| %.1e  |hv| %.1e (rel)  |Q'g| %.1e   (scale of H %.1e)" % (
        name, np.abs(out[k, 0] - td).max() / sc, np.abs(out[k, 1] - ev).max() / sc,
        (np.abs(out[k, 3] - hv) / np.maximum(np.abs(hv), 1e-300)).max(), np.abs(out[k, 2] - v).max() / np.abs(gk).max(), sc))
    if k == 0:
        print("   first entries td dev", out[k, 0][-4:], "emul", td[-4:]); print("   te dev", out[k, 1][-4:], "emul", ev[-4:])
