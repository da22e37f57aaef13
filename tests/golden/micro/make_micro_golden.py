#!/usr/bin/env python3
"""Per-function micro-goldens for the oracle, computed WITHOUT the oracle: 50-digit mpmath evaluations of the model's
definitions (tests/mp_model.py) and numerical differentiation (mp.diff) of them -- no hand-written derivative formula
of the reference, the oracle or the kernels enters.  SURVEY.md 8(c) rows a9, a11-a17, a21, a25, a27, a31 and the
whole-ELBO value (7.3 step 1).  Output: tests/golden/micro/micro_golden.json (numbers as 20-significant-digit strings).

Runs in ~15 minutes (the 50-digit ELBO of three scenes dominates); tests/test_oracle_micro.py only reads the JSON.
usage: python tests/golden/micro/make_micro_golden.py [--skip-elbo]"""
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpmath import mp, mpf

import mp_model as M
from celeste_jl_amd import synthetic

mp.dps = 50
S = lambda x: mp.nstr(x, 20)


def partial(f, theta, i, j=None):
    """d f / d theta_i, or d2 f / d theta_i d theta_j, of f(list of mpf) at theta (numerical, 50+ digits)"""
    if j is None:
        return mp.diff(lambda t: f(theta[:i] + [t] + theta[i + 1:]), theta[i])
    if i == j:
        return mp.diff(lambda t: f(theta[:i] + [t] + theta[i + 1:]), theta[i], 2)
    i, j = min(i, j), max(i, j)
    return mp.diff(lambda a, b: f(theta[:i] + [a] + theta[i + 1:j] + [b] + theta[j + 1:]), (theta[i], theta[j]), (1, 1))


# ---- A: the bivariate-normal chain (a11-a17; case "psf_k*" is test/test_psf.jl:74-143's data-free sub-case) ------------
def bvn_case(mean, tau, weight, x, J, ratio, angle, radius, nu):
    Fm = M.F
    mean, tau, x = [Fm(v) for v in mean], [Fm(v) for v in tau], [Fm(v) for v in x]
    J = [[Fm(J[a][b]) for b in range(2)] for a in range(2)]
    shape0 = [Fm(ratio), Fm(angle), Fm(radius)]
    nu = Fm(nu)

    def cov(shape):
        x11, x12, x22 = M.get_bvn_cov(shape[0], shape[1], shape[2])
        return tau[0] + nu * x11, tau[1] + nu * x12, tau[2] + nu * x22

    s0 = list(cov(shape0))
    out = {"inputs": dict(mean=[float(v) for v in mean], tau=[float(v) for v in tau], weight=weight, x=[float(v) for v in x],
                          J=[[float(v) for v in r] for r in J], ratio=ratio, angle=angle, radius=radius, nu=float(nu))}
    out["f_pre"] = S(Fm(weight) * mp.exp(M.bvn_logpdf(x[0], x[1], mean[0], mean[1], *s0)))
    # get_bvn_derivs!: derivatives of the log density with respect to x and to (Sigma11, Sigma12, Sigma22)
    lx = lambda v: M.bvn_logpdf(v[0], v[1], mean[0], mean[1], v[2], v[3], v[4])
    v0 = x + s0
    out["x_d"] = [S(partial(lx, v0, a)) for a in range(2)]
    out["sig_d"] = [S(partial(lx, v0, 2 + a)) for a in range(3)]
    out["xx_h"] = [[S(partial(lx, v0, a, b)) for b in range(2)] for a in range(2)]
    out["xsig_h"] = [[S(partial(lx, v0, a, 2 + b)) for b in range(3)] for a in range(2)]
    out["sigsig_h"] = [[S(partial(lx, v0, 2 + a, 2 + b)) for b in range(3)] for a in range(3)]
    # GalaxySigmaDerivs: j[sig, shape] = d (nu Xi_sig) / d shape, t[sig, s1, s2]; shape order (axis_ratio, angle, radius)
    xi = lambda sh, k: nu * M.get_bvn_cov(sh[0], sh[1], sh[2])[k]
    out["j"] = [[S(partial(lambda sh, k=k: xi(sh, k), shape0, a)) for a in range(3)] for k in range(3)]
    out["t"] = [[[S(partial(lambda sh, k=k: xi(sh, k), shape0, a, b)) for b in range(3)] for a in range(3)] for k in range(3)]
    # transform_bvn_derivs!: the same log density as a function of (u, shape): mean(u) = mean + J (u - u0)
    def lu(v):
        m1 = mean[0] + J[0][0] * v[0] + J[0][1] * v[1]
        m2 = mean[1] + J[1][0] * v[0] + J[1][1] * v[1]
        return M.bvn_logpdf(x[0], x[1], m1, m2, *cov(v[2:5]))
    u0 = [mpf(0), mpf(0)] + shape0
    out["u_d"] = [S(partial(lu, u0, a)) for a in range(2)]
    out["s_d"] = [S(partial(lu, u0, 2 + a)) for a in range(3)]
    out["uu_h"] = [[S(partial(lu, u0, a, b)) for b in range(2)] for a in range(2)]
    out["ss_h"] = [[S(partial(lu, u0, 2 + a, 2 + b)) for b in range(3)] for a in range(3)]
    out["us_h"] = [[S(partial(lu, u0, a, 2 + b)) for b in range(3)] for a in range(2)]
    return out


def main():
    skip_elbo = "--skip-elbo" in sys.argv
    prior = synthetic.load_prior()
    G = {}
    G["bvn"] = {
        "psf_k1": bvn_case([0.1, 0.2], [0, 0, 0], 1.0, [1.0, 2.0], [[1, 0], [0, 1]], 0.8, math.pi / 4, math.sqrt(2.0), 1.0),
        "psf_k2": bvn_case([0.1, 0.2], [0, 0, 0], 1.0, [1.0, 2.0], [[1, 0], [0, 1]], 0.8, math.pi / 4, math.sqrt(4.0), 1.0),
        "galaxy_component": bvn_case([10.3, 12.1], [1.5, 0.2, 1.2], 0.37, [11.0, 13.0], [[0.9, 0.2], [-0.1, 1.1]],
                                     0.6, 0.7, 3.1, 0.45),
    }
    # the value test_psf.jl:121-135 asserts equal to get_psf_at_point at x = (1, 2)
    G["psf_pixel_value"] = S(sum((mpf(1) / 2 + mpf(k) / 10) * mpf(G["bvn"]["psf_k%d" % k]["f_pre"]) for k in (1, 2)))

    # ---- B: SourceBrightness (a9): E_l_a[b, i], E_ll_a[b, i] with gradient / Hessian in the 10 brightness parameters
    fg = synthetic.make_sample_dataset("galaxy")
    vs = [M.F(v) for v in fg.vp[0]]
    G["brightness"] = {"vs": [float(v) for v in vs], "cases": []}
    for i in range(2):
        ids = [6 + i, 8 + i] + [10 + 4 * i + c for c in range(4)] + [18 + 4 * i + c for c in range(4)]   # bids order
        for b in (1, 3, 5):
            for which in (0, 1):
                f = lambda th, which=which, b=b, i=i: M.brightness(th, i, b)[which]
                G["brightness"]["cases"].append(dict(
                    i=i, b=b, moment=("E_l_a", "E_ll_a")[which], v=S(f(vs)),
                    d=[S(partial(f, vs, p)) for p in ids],
                    h=[[S(partial(f, vs, p, q)) for q in ids] for p in ids]))

    # ---- C: one pixel (a21 star density, a20 galaxy density, a25 calculate_G_s!, a27 add_elbo_log_term!) --------------
    n, h, w = 2, 9, 10
    img, p = fg.images[n], fg.patches[0][n]
    coef = M.spline_coefs(p.stamp)
    psf = [[M.F(x) for x in comp] for comp in img.psf]
    sky = M.F(img.sky[h - 1, w - 1])

    def dens(th):
        m1, m2 = M.patch_position(p, th[0:2])
        return (M.star_density(coef, m1, m2, mpf(h), mpf(w)),
                M.galaxy_density(psf, m1, m2, th[2], th[3], th[4], th[5], mpf(h), mpf(w)))
    f0 = lambda th: dens(th)[0]
    f1 = lambda th: dens(th)[1]
    E_s = lambda th: M.source_moments(th, *dens(th), img.b)[0]
    V_s = lambda th: M.source_moments(th, *dens(th), img.b)[1]

    def logterm(th):
        E, V = M.source_moments(th, *dens(th), img.b)
        E = E + sky
        return mp.log(E) - V / (2 * E * E)
    rng = np.random.default_rng(0)
    pairs = sorted({tuple(sorted(rng.integers(0, 28, 2))) for _ in range(60)} | {(k, k) for k in (0, 2, 3, 5, 6, 9, 12, 20, 26, 27)})
    pairs = [(int(a), int(b)) for a, b in pairs]
    G["pixel"] = {"scene": "sample galaxy (synthetic.make_sample_dataset('galaxy')), source 0, image %d, pixel (%d, %d)" % (n, h, w),
                  "n": n, "h": h, "w": w, "pairs": pairs,
                  "fs0m": dict(v=S(f0(vs)), d=[S(partial(f0, vs, a)) for a in range(2)],
                               h=[[S(partial(f0, vs, a, b)) for b in range(2)] for a in range(2)]),
                  "fs1m": dict(v=S(f1(vs)), d=[S(partial(f1, vs, a)) for a in range(6)],
                               h=[[S(partial(f1, vs, a, b)) for b in range(6)] for a in range(6)])}
    for name, f in (("E_G_s", E_s), ("var_G_s", V_s), ("elbo_log_term", logterm)):
        G["pixel"][name] = dict(v=S(f(vs)), d=[S(partial(f, vs, a)) for a in range(28)],
                                h_pairs=[S(partial(f, vs, a, b)) for a, b in pairs])
        print(name, "done", flush=True)
    # star density where the spline index leaves the stamp (clamped cell, |h - m| > 25) and on the y < 0 branch
    G["star_density"] = []
    for (hh, ww) in ((9, 10), (36, 12), (38, 40)):
        m1, m2 = M.patch_position(p, vs[0:2])
        G["star_density"].append(dict(h=hh, w=ww, v=S(M.star_density(coef, m1, m2, mpf(hh), mpf(ww))),
                                      spline=S(M.spline_value(coef, hh - m1 + 26, ww - m2 + 26))))

    # ---- D: subtract_kl (a31) -----------------------------------------------------------------------------------------
    kl = lambda th: M.neg_kl(th, prior)
    kpairs = [(5, 5), (6, 6), (6, 26), (7, 27), (8, 8), (10, 11), (10, 28), (14, 36), (18, 18), (19, 29), (22, 40),
              (26, 26), (26, 28), (27, 37), (28, 28), (28, 29), (36, 36), (12, 26), (20, 26), (9, 27)]
    G["kl"] = dict(vs=[float(v) for v in vs], v=S(kl(vs)), d=[S(partial(kl, vs, a)) for a in range(44)], pairs=kpairs,
                   h_pairs=[S(partial(kl, vs, a, b)) for a, b in kpairs])
    print("kl done", flush=True)

    # ---- E: the whole elbo() at 50 digits ------------------------------------------------------------------------------
    path = os.path.join(HERE, "micro_golden.json")
    if skip_elbo and os.path.exists(path):
        G["elbo_value"] = json.load(open(path)).get("elbo_value", {})
    else:
        G["elbo_value"] = {}
        for kind in ("star", "galaxy"):
            f = synthetic.make_sample_dataset(kind)
            G["elbo_value"]["sample_" + kind] = dict(target=0, v=S(M.elbo_value(f.images, f.patches, f.neighbors, f.vp, 0, prior)))
            print(kind, G["elbo_value"]["sample_" + kind], flush=True)
        # varying sky plane / per-row calibration / per-patch stamps, two overlapping sources (trap A2 in 50 digits)
        f = synthetic.make_field(24, 26, 2, seed=8, variable=True, margin=8)
        assert len(f.neighbors[0]) == 1
        G["elbo_value"]["variable_24x26_2src_seed8"] = dict(target=0, v=S(M.elbo_value(f.images, f.patches, f.neighbors, f.vp, 0, prior)))
        print(G["elbo_value"], flush=True)
    json.dump(G, open(path, "w"), indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
