"""The Celeste ELBO *value* in 50-digit arithmetic (mpmath) -- a third, literal restatement of the model.

Test infrastructure: written straight from the reference's formulas with
1-based loops like the Julia, value only, no derivative code.  It serves two purposes:
  * `elbo_value` pins the *arithmetic* of the C oracle (oracle/celeste_oracle.c): fp64 must agree with the 50-digit
    value to ~1e-13 on the sample scenes (tests/test_oracle_micro.py);
  * `mp.diff` of the pieces gives gradients / Hessians that owe nothing to the hand-written derivative code of the
    reference, the oracle or the kernels: the per-function micro-goldens (tests/golden/micro/make_micro_golden.py).

Reference formulas (paths relative to /root/reference/src):
  get_bvn_cov                    BivariateNormals.jl:29-43
  BvnComponent / eval_bvn_pdf!   BivariateNormals.jl:143-222
  galaxy prototypes              model/light_source_model.jl:45-75
  load_bvn_mixtures!             model/fsm_util.jl:111-169   (mean = xiBar + m_pos, cov = tauBar + nuBar XiXi)
  star_light_density!            model/fsm_util.jl:221-248   (softpluslikeinv of the spline of the conditioned stamp)
  ImagePatch ctor                model/imaged_sources.jl:97-107 (max(., 0), + 1e-6, normalise, softpluslike, spline)
  SourceBrightness               deterministic_vi/source_brightness.jl:46-50, 123-127
  calculate_G_s!, add_pixel_term!, add_elbo_log_term!   deterministic_vi/elbo_objective.jl:17-392
  subtract_kl                    deterministic_vi/elbo_kl.jl:94-154
Third-party arithmetic restated from its published algorithm: Interpolations.jl BSpline(Cubic(Line())), OnGrid()
(prefilter rows [1/6 2/3 1/6], boundary rows c0 - 2 c1 + c2 = 0, cell index clamped to [1, 50]).
"""
import numpy as np
from mpmath import mp, mpf

mp.dps = 50

DEV_AMP = ["4.26347652e-2", "2.40127183e-1", "6.85907632e-1", "1.51937350", "2.83627243", "4.46467501", "5.72440830",
           "5.60989349"]
DEV_VAR = ["2.23759216e-4", "1.00220099e-3", "4.18731126e-3", "1.69432589e-2", "6.84850479e-2", "2.87207080e-1",
           "1.33320254", "8.40215071"]
EXP_AMP = ["2.34853813e-3", "3.07995260e-2", "2.23364214e-1", "1.17949102", "4.33873750", "5.99820770"]
EXP_VAR = ["1.20078965e-3", "8.84526493e-3", "3.91463084e-2", "1.39976817e-1", "4.60962500e-1", "1.50159566"]


def F(x):
    """an fp64 input, exactly"""
    return mpf(float(x))


def galaxy_prototypes():
    """(eta, nu)[i][j]; the literals are the reference's Float64 literals (light_source_model.jl:46-72)"""
    dev_amp = [F(float(a)) for a in DEV_AMP]; dev_var = [F(float(a)) for a in DEV_VAR]
    exp_amp = [F(float(a)) for a in EXP_AMP]; exp_var = [F(float(a)) for a in EXP_VAR]
    er = (F(1.078031), F(0.928896))
    eta = [[a / sum(dev_amp) for a in dev_amp], [a / sum(exp_amp) for a in exp_amp]]
    nu = [[v / er[0] ** 2 for v in dev_var], [v / er[1] ** 2 for v in exp_var]]
    return eta, nu


def get_bvn_cov(ab, angle, scale):
    cp, sp = mp.cos(angle), mp.sin(angle)
    ab_term = ab * ab - 1
    s2 = scale * scale
    off = -s2 * cp * sp * ab_term
    return s2 * (1 + ab_term * sp * sp), off, s2 * (1 + ab_term * cp * cp)     # Sigma11, Sigma12, Sigma22


def bvn_logpdf(x1, x2, m1, m2, s11, s12, s22):
    """log N((x1, x2); (m1, m2), [[s11, s12], [s12, s22]])"""
    det = s11 * s22 - s12 * s12
    d1, d2 = x1 - m1, x2 - m2
    q = (s22 * d1 * d1 - 2 * s12 * d1 * d2 + s11 * d2 * d2) / det
    return -q / 2 - mp.log(det) / 2 - mp.log(2 * mp.pi)


def galaxy_density(psf, m1, m2, dev, ratio, angle, radius, h, w):
    """sum over (type i, prototype j, psf component k) of theta_i eta_ij alpha_k N((h, w); xi_k + m, tau_k + nu_ij Xi)"""
    eta, nu = galaxy_prototypes()
    x11, x12, x22 = get_bvn_cov(ratio, angle, radius)
    out = mpf(0)
    for i in range(2):
        th = dev if i == 0 else 1 - dev
        for j in range(8 if i == 0 else 6):
            for a, xi1, xi2, t11, t12, t22 in psf:
                lp = bvn_logpdf(h, w, xi1 + m1, xi2 + m2, t11 + nu[i][j] * x11, t12 + nu[i][j] * x12, t22 + nu[i][j] * x22)
                out += th * a * eta[i][j] * mp.exp(lp)
    return out


# ---- star density: conditioned stamp -> natural bicubic spline -> softpluslikeinv -------------------------------------
def _prefilter_1d(d):
    """n samples -> n + 2 B-spline coefficients: (c[q-1] + 4 c[q] + c[q+1]) / 6 = d[q], c0 - 2 c1 + c2 = 0 at both ends"""
    n = len(d)
    m = n + 2
    A = mp.zeros(m, m); r = mp.zeros(m, 1)
    A[0, 0], A[0, 1], A[0, 2] = 1, -2, 1
    for q in range(1, n + 1):
        A[q, q - 1], A[q, q], A[q, q + 1] = mpf(1) / 6, mpf(2) / 3, mpf(1) / 6
        r[q] = d[q - 1]
    A[m - 1, m - 3], A[m - 1, m - 2], A[m - 1, m - 1] = 1, -2, 1
    c = mp.lu_solve(A, r)
    return [c[q] for q in range(m)]


def spline_coefs(stamp):
    """stamp[h][w] raw psfmap output (51 x 51 floats) -> coef[h][w] (53 x 53 mpf)"""
    n = 51
    g = [[max(F(stamp[h][w]), mpf(0)) + mpf("1e-6") for w in range(n)] for h in range(n)]
    tot = sum(sum(row) for row in g)
    soft = lambda x: 1000 * x - 1 if 1000 * x > 1 else mp.log(1000 * x)
    g = [[soft(v / tot) for v in row] for row in g]
    tmp = [_prefilter_1d([g[h][w] for h in range(n)]) for w in range(n)]            # tmp[w][h'] along h
    coef = [[None] * (n + 2) for _ in range(n + 2)]
    for hp in range(n + 2):
        line = _prefilter_1d([tmp[w][hp] for w in range(n)])
        for wp in range(n + 2):
            coef[hp][wp] = line[wp]
    return coef


def _bw(f):
    o = 1 - f
    return (o ** 3 / 6, mpf(2) / 3 - f * f + f ** 3 / 2, mpf(2) / 3 - o * o + o ** 3 / 2, f ** 3 / 6)


def spline_value(coef, x, y):
    """itp[x, y], 1-based coordinates on the 51-grid; the cell is fixed by the fp64 value of the argument"""
    ix = min(max(int(mp.floor(x)), 1), 50)
    iy = min(max(int(mp.floor(y)), 1), 50)
    wx, wy = _bw(x - ix), _bw(y - iy)
    s = mpf(0)
    for a in range(4):
        for b in range(4):
            s += coef[ix - 1 + a][iy - 1 + b] * wx[a] * wy[b]
    return s


def star_density(coef, m1, m2, h, w):
    y = spline_value(coef, h - m1 + 26, w - m2 + 26)
    return mp.exp(y) / 1000 if y < 0 else (y + 1) / 1000      # softpluslikeinv (fsm_util.jl:222)


# ---- brightness moments, pixel term, KL -----------------------------------------------------------------------------
def brightness(vs, i, b):
    """(E[l_b | a = i], E[l_b^2 | a = i]), b = 1..5 (the reference's band index)"""
    r, v = vs[6 + i], vs[8 + i]
    cm, cv = vs[10 + 4 * i:14 + 4 * i], vs[18 + 4 * i:22 + 4 * i]
    l = r + v / 2; ll = 2 * r + 2 * v
    if b >= 4: l += cm[2] + cv[2] / 2; ll += 2 * cm[2] + 2 * cv[2]
    if b >= 5: l += cm[3] + cv[3] / 2; ll += 2 * cm[3] + 2 * cv[3]
    if b <= 2: l += -cm[1] + cv[1] / 2; ll += -2 * cm[1] + 2 * cv[1]
    if b <= 1: l += -cm[0] + cv[0] / 2; ll += -2 * cm[0] + 2 * cv[0]
    return mp.exp(l), mp.exp(ll)


def source_moments(vs, f0, f1, b):
    """E_G_s.v, var_G_s.v (calculate_G_s!, elbo_objective.jl:62-65, 204)"""
    E = mpf(0); E2 = mpf(0)
    for i, f in enumerate((f0, f1)):
        El, Ell = brightness(vs, i, b)
        E += vs[26 + i] * El * f
        E2 += vs[26 + i] * Ell * f * f
    return E, E2 - E * E


def pixel_term(x, iota, log_iota_f32, E, V):
    """add_elbo_log_term! + the linear and lgamma terms (elbo_objective.jl:288-292, 383-391)"""
    return x * (log_iota_f32 + mp.log(E) - V / (2 * E * E)) - iota * E - mp.loggamma(x + 1)


def neg_kl(vs, prior):
    """subtract_kl (elbo_kl.jl:94-154); prior: the dict of celeste.jl_amd/prior_tables.json"""
    out = mpf(0)
    a = vs[26:28]
    for i in range(2):
        out -= a[i] * (mp.log(a[i]) - mp.log(F(prior["is_star"][i])))
        k = vs[28 + 8 * i:36 + 8 * i]
        for d in range(8):
            out -= a[i] * k[d] * (mp.log(k[d]) - mp.log(F(prior["k"][i][d])))
        mu2, var2 = F(prior["flux_mean"][i]), F(prior["flux_var"][i])
        r, v = vs[6 + i], vs[8 + i]
        out -= a[i] * (mp.log(var2) - mp.log(v) + (v + (r - mu2) ** 2) / var2 - 1) / 2
        cm, cv = vs[10 + 4 * i:14 + 4 * i], vs[18 + 4 * i:22 + 4 * i]
        for d in range(8):
            S2 = mp.matrix(4, 4)
            for r_ in range(4):
                for c_ in range(4):
                    S2[r_, c_] = F(prior["color_cov"][i][d][r_ + 4 * c_])
            inv = S2 ** -1
            diff = mp.matrix([F(prior["color_mean"][i][d][c_]) - cm[c_] for c_ in range(4)])
            tr = sum(inv[c_, c_] * cv[c_] for c_ in range(4))
            quad = (diff.T * inv * diff)[0]
            kl = (tr - 4 + quad + mp.log(mp.det(S2)) - sum(mp.log(cv[c_]) for c_ in range(4))) / 2
            out -= a[i] * k[d] * kl
    x = vs[5]
    out -= (mp.log(2 * mp.pi) + mp.log(F(prior["gal_radius_px_var"])) +
            (x - F(prior["gal_radius_px_mean"])) ** 2 / F(prior["gal_radius_px_var"])) / 2
    return out


# ---- the whole elbo() for one active source (Sa = 1, neighbours value-only) -------------------------------------------
def patch_position(p, pos):
    """linear_world_to_pix (wcs_utils.jl:14-18)"""
    J = p.wcs_jacobian
    d0, d1 = pos[0] - F(p.world_center[0]), pos[1] - F(p.world_center[1])
    return (F(J[0, 0]) * d0 + F(J[0, 1]) * d1 + F(p.pixel_center[0]),
            F(J[1, 0]) * d0 + F(J[1, 1]) * d1 + F(p.pixel_center[1]))


def elbo_value(images, patches, neighbors, vp, target, prior, include_kl=True, coef_cache=None):
    """elbo(ea, vp).v for ElboArgs(images, patches[[target; neighbours], :], [1]) -- elbo_objective.jl:400-492.
    1-based loops over the active source's patch; every source of the local list that covers the pixel
    (1 <= h2 <= H2, 1 <= w2 < W2 and its bitmap) contributes."""
    coef_cache = {} if coef_cache is None else coef_cache
    src = [target] + list(neighbors[target])
    vps = {s: [F(x) for x in vp[s]] for s in src}
    total = mpf(0)
    for n, img in enumerate(images):
        pa = patches[target][n]
        H2, W2 = pa.active_pixel_bitmap.shape
        psf = [[F(x) for x in comp] for comp in img.psf]
        pos_m, coefs = {}, {}
        for s in src:
            p = patches[s][n]
            pos_m[s] = patch_position(p, vps[s][0:2])
            key = p.stamp.tobytes()
            if key not in coef_cache:
                coef_cache[key] = spline_coefs(p.stamp)
            coefs[s] = coef_cache[key]
        for w2 in range(1, W2 + 1):
            for h2 in range(1, H2 + 1):
                if not pa.active_pixel_bitmap[h2 - 1, w2 - 1]:
                    continue
                h, w = h2 + pa.bitmap_offset[0], w2 + pa.bitmap_offset[1]
                x32 = img.pixels[h - 1, w - 1]
                if np.isnan(x32):
                    continue
                E = F(img.sky[h - 1, w - 1]); V = mpf(0)
                for s in src:
                    p = patches[s][n]
                    ph2, pw2 = h - p.bitmap_offset[0], w - p.bitmap_offset[1]
                    PH2, PW2 = p.active_pixel_bitmap.shape
                    if not (1 <= ph2 <= PH2 and 1 <= pw2 < PW2) or not p.active_pixel_bitmap[ph2 - 1, pw2 - 1]:
                        continue
                    vs = vps[s]
                    m1, m2 = pos_m[s]
                    f0 = star_density(coefs[s], m1, m2, mpf(h), mpf(w))
                    f1 = galaxy_density(psf, m1, m2, vs[2], vs[3], vs[4], vs[5], mpf(h), mpf(w))
                    Es, Vs = source_moments(vs, f0, f1, img.b)
                    E += Es; V += Vs
                iota32 = img.nelec_per_nmgy[h - 1]
                log_iota = F(np.float32(float(mp.log(F(iota32)))))      # log evaluated on a Float32 (elbo_objective.jl:292)
                total += pixel_term(F(x32), F(iota32), log_iota, E, V)
    if include_kl:
        total += neg_kl(vps[target], prior)
    return total
