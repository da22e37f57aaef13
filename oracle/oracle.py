"""ctypes wrapper of oracle/libceleste_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The problem marshalling (celeste_problem_t) is shared with the product's C ABI so that the
oracle and the HIP engine see byte-identical inputs.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import celeste_jl_amd  # noqa: E402,F401
from celeste_jl_amd import cabi  # noqa: E402

LIB_PATH = os.path.join(_HERE, "libceleste_oracle.so")


class OptCfg(C.Structure):
    """ElboConfig defaults (ElboMaximize.jl:43-49, 95-108)"""
    _fields_ = [("loc_width", C.c_double), ("loc_scale", C.c_double), ("max_iters", C.c_int32),
                ("include_kl", C.c_int32), ("xtol_abs", C.c_double), ("ftol_rel", C.c_double), ("gtol", C.c_double),
                ("initial_delta", C.c_double), ("delta_hat", C.c_double), ("tr_secular_iters", C.c_int32),
                ("reserved", C.c_int32)]

    def __init__(self, loc_width=1e-4, loc_scale=1.0, max_iters=50, include_kl=True, xtol_abs=1e-7, ftol_rel=1e-6,
                 gtol=1e-8, initial_delta=1.0, delta_hat=1e9, tr_secular_iters=0):
        super().__init__(loc_width, loc_scale, max_iters, int(include_kl), xtol_abs, ftol_rel, gtol, initial_delta,
                         delta_hat, tr_secular_iters, 0)
P = 44
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                             for f in ("celeste_oracle.c", "celeste_optim_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp, ip, lp = cabi.c_double_p, cabi.c_int32_p, cabi.c_int64_p
        L.celeste_oracle_elbo.argtypes = [C.POINTER(cabi.ProblemT), dp, C.c_int32, C.c_uint32, dp, dp, dp, lp, lp]
        L.celeste_oracle_elbo_batch.argtypes = [C.POINTER(cabi.ProblemT), dp, C.c_int32, ip, C.c_uint32, dp, dp, dp,
                                                lp, ip, C.c_int32]
        L.celeste_reduced_elbo_batch.argtypes = L.celeste_oracle_elbo_batch.argtypes
        L.celeste_oracle_elbo_multi.argtypes = [C.POINTER(cabi.ProblemT), dp, C.c_int32, ip, C.c_uint32,
                                                C.POINTER(C.c_double), dp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.celeste_oracle_get_bvn_cov.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.celeste_oracle_get_bvn_cov.restype = None
        L.celeste_oracle_source_brightness.argtypes = [dp, dp, dp]
        L.celeste_oracle_source_brightness.restype = None
        L.celeste_oracle_galaxy_prototypes.argtypes = [dp, dp]
        L.celeste_oracle_galaxy_prototypes.restype = None
        L.celeste_oracle_spline_coefs.argtypes = [dp, dp]
        L.celeste_oracle_spline_coefs.restype = None
        L.celeste_oracle_spline_value.argtypes = [dp, C.c_double, C.c_double]
        L.celeste_oracle_spline_value.restype = C.c_double
        L.celeste_oracle_subtract_kl.argtypes = [C.POINTER(cabi.PriorT), dp, dp, dp, dp]
        L.celeste_oracle_subtract_kl.restype = None
        L.celeste_oracle_categorical_kl.argtypes = [dp, dp, C.c_int]
        L.celeste_oracle_categorical_kl.restype = C.c_double
        L.celeste_oracle_gaussian_kl.argtypes = [C.c_double] * 4
        L.celeste_oracle_gaussian_kl.restype = C.c_double
        L.celeste_oracle_psf_at_point.argtypes = [dp, C.c_int, C.c_double, C.c_double]
        L.celeste_oracle_psf_at_point.restype = C.c_double
        L.celeste_oracle_jacobi_eig.argtypes = [C.c_int, dp, dp, dp]
        L.celeste_oracle_jacobi_eig.restype = None
        L.celeste_oracle_solve_tr.argtypes = [C.c_int, dp, dp, C.c_double, dp, ip]
        L.celeste_oracle_solve_tr.restype = C.c_double
        L.celeste_oracle_constraints_roundtrip.argtypes = [dp, C.c_double, C.c_double, dp, dp, dp]
        L.celeste_oracle_constraints_roundtrip.restype = None
        L.celeste_oracle_propagate.argtypes = [dp, dp, C.c_double, C.c_double, dp, dp, dp, dp]
        L.celeste_oracle_propagate.restype = None
        L.celeste_oracle_maximize.argtypes = [C.POINTER(cabi.ProblemT), dp, C.c_int32, C.POINTER(OptCfg), dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(cabi.c_double_p)


def elbo_batch(problem: "cabi.Problem", vp, targets, flags=cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL,
               n_threads: int = 0):
    """Returns (v[n], d[n,44], h[n,44,44], counters[n,2], status[n]) from the CPU restatement."""
    L = lib()
    vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(problem.n_sources, P))
    tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
    n = tg.size
    v = np.zeros(n); d = np.zeros((n, P)); h = np.zeros((n, P, P))
    cnt = np.zeros((n, 2), dtype=np.int64); status = np.zeros(n, dtype=np.int32)
    if n_threads <= 0:
        n_threads = int(os.environ.get("CELESTE_ORACLE_THREADS", "0")) or os.cpu_count() or 1
    n_threads = max(1, min(n_threads, n))     # (one target per thread at most: 256 threads for 40 targets only oversubscribe the host)
    L.celeste_oracle_elbo_batch(C.byref(problem.c), _dp(vp), n, tg.ctypes.data_as(cabi.c_int32_p), flags, _dp(v),
                                _dp(d), _dp(h), cnt.ctypes.data_as(cabi.c_int64_p),
                                status.ctypes.data_as(cabi.c_int32_p), n_threads)
    # h is column-major per target and symmetric (upper triangle mirrored)
    return v, d, h.transpose(0, 2, 1).copy(), cnt, status


def reduced_elbo_batch(problem: "cabi.Problem", vp, targets, flags=cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL,
                       n_threads: int = 0):
    """celeste_reduced.c: the reduced-variable algorithm on the CPU (baseline timing); same returns as elbo_batch"""
    L = lib()
    vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(problem.n_sources, P))
    tg = np.ascontiguousarray(np.asarray(targets, dtype=np.int32))
    n = tg.size
    v = np.zeros(n); d = np.zeros((n, P)); h = np.zeros((n, P, P))
    cnt = np.zeros((n, 2), dtype=np.int64); status = np.zeros(n, dtype=np.int32)
    if n_threads <= 0:
        n_threads = int(os.environ.get("CELESTE_ORACLE_THREADS", "0")) or os.cpu_count() or 1
    n_threads = max(1, min(n_threads, n))     # (one target per thread at most: 256 threads for 40 targets only oversubscribe the host)
    L.celeste_reduced_elbo_batch(C.byref(problem.c), _dp(vp), n, tg.ctypes.data_as(cabi.c_int32_p), flags, _dp(v),
                                 _dp(d), _dp(h), cnt.ctypes.data_as(cabi.c_int64_p),
                                 status.ctypes.data_as(cabi.c_int32_p), n_threads)
    return v, d, h.transpose(0, 2, 1).copy(), cnt, status


def elbo_multi(problem, vp, active, flags=cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL):
    """elbo() with the active sources `active` (ElboArgs.active_sources, in that order).
    Returns (v, d [Sa, 44], h [44 Sa, 44 Sa], counters[2], status)."""
    vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(problem.n_sources, P))
    act = np.ascontiguousarray(np.asarray(active, dtype=np.int32))
    sa = act.size
    v = C.c_double(); d = np.zeros(sa * P); h = np.zeros((sa * P, sa * P))
    na = C.c_int64(); ni = C.c_int64()
    st = lib().celeste_oracle_elbo_multi(C.byref(problem.c), _dp(vp), sa, act.ctypes.data_as(cabi.c_int32_p), flags,
                                         C.byref(v), _dp(d), _dp(h), C.byref(na), C.byref(ni))
    return v.value, d.reshape(sa, P), h.T.copy(), np.array([na.value, ni.value]), int(st)


def elbo_one(problem, vp, target, flags=cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL):
    v, d, h, cnt, st = elbo_batch(problem, vp, [target], flags, n_threads=1)
    return v[0], d[0], h[0], cnt[0], int(st[0])


def get_bvn_cov(ab, angle, scale):
    out = np.zeros(4)
    lib().celeste_oracle_get_bvn_cov(ab, angle, scale, _dp(out))
    return out.reshape(2, 2).T


def source_brightness(vs):
    vs = np.ascontiguousarray(vs, dtype=np.float64)
    El = np.zeros(10); Ell = np.zeros(10)
    lib().celeste_oracle_source_brightness(_dp(vs), _dp(El), _dp(Ell))
    return El.reshape(2, 5).T, Ell.reshape(2, 5).T  # [b, i]


def galaxy_prototypes():
    eta = np.zeros(16); nu = np.zeros(16)
    lib().celeste_oracle_galaxy_prototypes(_dp(eta), _dp(nu))
    return eta.reshape(2, 8), nu.reshape(2, 8)


def spline_coefs(stamp):
    st = np.ascontiguousarray(np.asarray(stamp, dtype=np.float64).T).reshape(-1)
    out = np.zeros(53 * 53)
    lib().celeste_oracle_spline_coefs(_dp(st), _dp(out))
    return out.reshape(53, 53).T.copy()


def spline_value(coef_hw, x, y):
    c = np.ascontiguousarray(np.asarray(coef_hw, dtype=np.float64).T).reshape(-1)
    return lib().celeste_oracle_spline_value(_dp(c), float(x), float(y))


def subtract_kl(vs, prior=None):
    vs = np.ascontiguousarray(vs, dtype=np.float64)
    v = np.zeros(1); d = np.zeros(P); h = np.zeros(P * P)
    pr = C.byref(cabi.prior_struct(prior)) if prior is not None else None
    lib().celeste_oracle_subtract_kl(pr, _dp(vs), _dp(v), _dp(d), _dp(h))
    return v[0], d, h.reshape(P, P).T.copy()


def categorical_kl(p1, p2):
    p1 = np.ascontiguousarray(p1, dtype=np.float64); p2 = np.ascontiguousarray(p2, dtype=np.float64)
    return lib().celeste_oracle_categorical_kl(_dp(p1), _dp(p2), p1.size)


def gaussian_kl(mu1, var1, mu2, var2):
    return lib().celeste_oracle_gaussian_kl(mu1, var1, mu2, var2)


def psf_at_point(psf, row, col):
    psf = np.ascontiguousarray(psf, dtype=np.float64)
    return lib().celeste_oracle_psf_at_point(_dp(psf), psf.shape[0], float(row), float(col))


def jacobi_eig(A):
    A = np.ascontiguousarray(A, dtype=np.float64); n = A.shape[0]
    w = np.zeros(n); V = np.zeros(n * n)
    lib().celeste_oracle_jacobi_eig(n, _dp(np.ascontiguousarray(A.T)), _dp(w), _dp(V))
    return w, V.reshape(n, n).T.copy()


def solve_tr(g, H, delta):
    g = np.ascontiguousarray(g, dtype=np.float64); H = np.ascontiguousarray(H, dtype=np.float64); n = g.size
    s = np.zeros(n); interior = C.c_int32(0)
    m = lib().celeste_oracle_solve_tr(n, _dp(g), _dp(np.ascontiguousarray(H.T)), float(delta), _dp(s),
                                      C.cast(C.byref(interior), cabi.c_int32_p))
    return s, m, bool(interior.value)


def constraints_roundtrip(vs, loc_width=1e-4, loc_scale=1.0):
    vs = np.ascontiguousarray(vs, dtype=np.float64)
    x = np.zeros(41); out = np.zeros(P); J = np.zeros(P * 41)
    lib().celeste_oracle_constraints_roundtrip(_dp(vs), loc_width, loc_scale, _dp(x), _dp(out), _dp(J))
    return x, out, J.reshape(41, P).T.copy()  # J[a, i]


def propagate(x, vs0, d, h, loc_width=1e-4, loc_scale=1.0):
    x = np.ascontiguousarray(x, dtype=np.float64); vs0 = np.ascontiguousarray(vs0, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64); h = np.ascontiguousarray(np.asarray(h, dtype=np.float64).T)
    gf = np.zeros(41); Hf = np.zeros(41 * 41)
    lib().celeste_oracle_propagate(_dp(x), _dp(vs0), loc_width, loc_scale, _dp(d), _dp(h), _dp(gf), _dp(Hf))
    return gf, Hf.reshape(41, 41).T.copy()


def maximize(problem, vp, target, cfg=None, pos_center=None):
    """maximize! for one target (neighbours frozen); returns (vp_new, iterations, f_evals, elbo, status).
    pos_center: centre of the position box (default: the current position)"""
    cfg = cfg or OptCfg()
    vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(problem.n_sources, P)).copy()
    stats = np.zeros(3)
    L = lib()
    L.celeste_oracle_maximize_at.argtypes = [C.POINTER(cabi.ProblemT), C.POINTER(C.c_double), C.c_int32, C.POINTER(OptCfg),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]
    pc = None if pos_center is None else np.ascontiguousarray(pos_center, dtype=np.float64)
    st = L.celeste_oracle_maximize_at(C.byref(problem.c), _dp(vp), int(target), C.byref(cfg),
                                      _dp(pc) if pc is not None else None, _dp(stats))
    return vp, int(stats[0]), int(stats[1]), float(stats[2]), int(st)


# ---- per-function hooks (tests/test_oracle_micro.py) -----------------------------------------------------------------
def micro_bvn(mean, tau, weight, x, J, ratio, angle, radius, nu_bar):
    """One BVN component through eval_bvn_pdf! / get_bvn_derivs! / GalaxySigmaDerivs / transform_bvn_derivs!.
    Returns a dict of the intermediate arrays (column-major reshaped to the reference's index order)."""
    L = lib()
    L.celeste_oracle_micro_bvn.restype = None
    out = np.zeros(87)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (mean, tau, x, np.asarray(J, dtype=np.float64).T)]
    L.celeste_oracle_micro_bvn(_dp(a[0]), _dp(a[1]), C.c_double(weight), _dp(a[2]), _dp(a[3]), C.c_double(ratio),
                               C.c_double(angle), C.c_double(radius), C.c_double(nu_bar), _dp(out))
    k = [0]

    def take(n, shape=None):
        v = out[k[0]:k[0] + n].copy(); k[0] += n
        return v if shape is None else v.reshape(shape, order="F")
    return dict(f_pre=take(1)[0], py=take(2), x_d=take(2), sig_d=take(3), xx_h=take(4, (2, 2)), xsig_h=take(6, (2, 3)),
                sigsig_h=take(9, (3, 3)), j=take(9, (3, 3)), t=take(27, (3, 3, 3)), u_d=take(2), uu_h=take(4, (2, 2)),
                s_d=take(3), ss_h=take(9, (3, 3)), us_h=take(6, (2, 3)))


def micro_brightness(vs):
    """SourceBrightness SensitiveFloats: {("E_l_a" | "E_ll_a", b, i): (v, d[10], h[10, 10])}, b = 0..4, i = 0..1"""
    L = lib()
    L.celeste_oracle_micro_brightness.restype = None
    out = np.zeros(2 * 2 * 5 * 111)
    L.celeste_oracle_micro_brightness(_dp(np.ascontiguousarray(vs, dtype=np.float64)), _dp(out))
    res = {}
    for w, name in enumerate(("E_l_a", "E_ll_a")):
        for i in range(2):
            for b in range(5):
                o = out[((w * 2 + i) * 5 + b) * 111:][:111]
                res[(name, b, i)] = (o[0], o[1:11].copy(), o[11:].reshape(10, 10).T.copy())
    return res


def micro_pixel(problem, vp, s, n, h, w):
    """One pixel (1-based h, w) of image n with source s active and alone: the SensitiveFloats fs0m, fs1m, E_G_s,
    var_G_s and elbo_log_term as (v, d, h) triples."""
    L = lib()
    vp = np.ascontiguousarray(np.asarray(vp, dtype=np.float64).reshape(problem.n_sources, P))
    out = np.zeros(7 + 43 + 3 * (1 + P + P * P))
    L.celeste_oracle_micro_pixel.argtypes = [C.POINTER(cabi.ProblemT), C.POINTER(C.c_double), C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    st = L.celeste_oracle_micro_pixel(C.byref(problem.c), _dp(vp), s, n, h, w, _dp(out))
    assert st == 0
    k = [0]

    def sf(p):
        v = out[k[0]]; d = out[k[0] + 1:k[0] + 1 + p].copy(); hh = out[k[0] + 1 + p:k[0] + 1 + p + p * p].reshape(p, p).T.copy()
        k[0] += 1 + p + p * p
        return v, d, hh
    return dict(fs0m=sf(2), fs1m=sf(6), E_G_s=sf(P), var_G_s=sf(P), elbo_log_term=sf(P))
