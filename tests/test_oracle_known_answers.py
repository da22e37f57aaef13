"""Pin the CPU oracle on every data-free known answer the reference's own tests and constants provide
(SURVEY.md Appendix C): closed forms taken from test/test_elbo.jl:45-61, test/test_psf.jl:121-135,
test/test_kl.jl:30-72, src/model/light_source_model.jl:46-72, src/DeterministicVI.jl:39-53."""
import math

import numpy as np
import pytest


def test_get_bvn_cov_closed_form(oracle):
    """test_elbo.jl:45-61"""
    ab, ang, sc = 0.7, math.pi / 5, 2.0
    S = oracle.get_bvn_cov(ab, ang, sc)
    m11 = sc ** 2 * (1 + (ab ** 2 - 1) * math.sin(ang) ** 2)
    m12 = sc ** 2 * (1 - ab ** 2) * math.cos(ang) * math.sin(ang)
    m22 = sc ** 2 * (1 + (ab ** 2 - 1) * math.cos(ang) ** 2)
    assert S[0, 0] == pytest.approx(m11, rel=1e-15)
    assert S[0, 1] == pytest.approx(m12, rel=1e-15) and S[1, 0] == S[0, 1]
    assert S[1, 1] == pytest.approx(m22, rel=1e-15)
    assert (S[0, 0], S[0, 1], S[1, 1]) == pytest.approx(
        (3.2951973342624465, 0.9700776466210568, 2.664802665737554), rel=1e-15)


def test_psf_gmm_value_at_point(oracle):
    """initialize_psf_params(2, for_test=true) (PSF.jl:217-224) at x = (1, 2) (test_psf.jl:121-135)"""
    K = 2
    psf = np.zeros((K, 6))
    for k in range(1, K + 1):
        S = oracle.get_bvn_cov(0.8, math.pi / 4, math.sqrt(2 * k))
        psf[k - 1] = [1 / K + k / 10, 0.1, 0.2, S[0, 0], S[0, 1], S[1, 1]]
    assert oracle.psf_at_point(psf, 1.0, 2.0) == pytest.approx(0.04087874823898267, rel=1e-14)
    assert oracle.psf_at_point(psf[:1], 1.0, 2.0) == pytest.approx(0.020483011485692804, rel=1e-14)


def test_source_brightness_generic_init(oracle):
    """E_l_a / E_ll_a for generic_init_source (source_brightness.jl:46-50,123-127)"""
    from celeste_jl_amd import generic_init_source
    vs = generic_init_source([0., 0.])
    El, Ell = oracle.source_brightness(vs)
    exp_l = [2.021110636890052, 2.011030305534672, 2.0010002500416717, 2.011030305534672, 2.021110636890052]
    exp_ll = [4.171577915003053, 4.088975137881753, 4.008008005336001, 4.088975137881753, 4.171577915003053]
    for i in range(2):
        assert El[:, i] == pytest.approx(exp_l, rel=1e-14)
        assert Ell[:, i] == pytest.approx(exp_ll, rel=1e-14)


def test_galaxy_prototypes(oracle):
    """light_source_model.jl:46-72"""
    eta, nu = oracle.galaxy_prototypes()
    assert eta[0].sum() == pytest.approx(1.0, rel=1e-15) and eta[1].sum() == pytest.approx(1.0, rel=1e-15)
    assert (eta[0, 0], eta[0, 7], nu[0, 0], nu[0, 7]) == pytest.approx(
        (0.002018376897592453, 0.2655785569606069, 0.0001925388760938517, 7.229828041918766), rel=1e-14)
    assert (eta[1, 0], eta[1, 5], nu[1, 0], nu[1, 5]) == pytest.approx(
        (0.00019948597671798522, 0.5094906939372705, 0.001391658736895312, 1.7402787570021794), rel=1e-14)
    assert eta[1, 6] == 0 and eta[1, 7] == 0  # type 2 has 6 components in 8 slots (fsm_util.jl:155)


def test_kl_closed_forms(oracle):
    """test_kl.jl:39-72"""
    assert oracle.categorical_kl(np.array([1, 2, 3, 4]) / 10, np.array([5, 6, 2, 1]) / 14) == \
        pytest.approx(0.6319632645245866, rel=1e-14)
    assert oracle.gaussian_kl(0.5, 2.0, 0.8, 1.8) == pytest.approx(0.027875297726642323, rel=1e-13)
    assert oracle.categorical_kl(np.array([0.5, 0.5]), np.array([0.95, 0.05])) == \
        pytest.approx(0.8303656034108253, rel=1e-14)


def test_diagmvn_kl_against_monte_carlo_free_identity(oracle):
    """diagmvn_mvn_kl (elbo_kl.jl:73-84): for a diagonal Sigma2 it reduces to a sum of gaussian_kl's."""
    import json, os
    from celeste_jl_amd.synthetic import load_prior
    from celeste_jl_amd import generic_init_source
    prior = load_prior()
    prior2 = json.loads(json.dumps(prior))
    for i in range(2):
        for d in range(8):
            c = np.diag(np.diag(np.asarray(prior["color_cov"][i][d]).reshape(4, 4)))
            prior2["color_cov"][i][d] = c.reshape(-1).tolist()
    vs = generic_init_source([1.0, 2.0])
    v, _, _ = oracle.subtract_kl(vs, prior2)
    # direct evaluation with univariate closed forms
    a = vs[26:28]
    kl = sum(a[i] * (math.log(a[i]) - math.log(prior["is_star"][i])) for i in range(2))
    for i in range(2):
        k = vs[28 + 8 * i:36 + 8 * i]
        kl += a[i] * sum(k[d] * (math.log(k[d]) - math.log(prior["k"][i][d])) for d in range(8))
        kl += a[i] * oracle.gaussian_kl(vs[6 + i], vs[8 + i], prior["flux_mean"][i], prior["flux_var"][i])
        for d in range(8):
            s = 0.0
            for c in range(4):
                s += oracle.gaussian_kl(vs[10 + 4 * i + c], vs[18 + 4 * i + c], prior2["color_mean"][i][d][c],
                                        prior2["color_cov"][i][d][5 * c])
            kl += a[i] * k[d] * s
    x = vs[5]
    logp = -0.5 * (math.log(2 * math.pi) + math.log(prior["gal_radius_px_var"]) +
                   (x - prior["gal_radius_px_mean"]) ** 2 / prior["gal_radius_px_var"])
    assert v == pytest.approx(-kl + logp, rel=1e-13)


def test_prior_tables_match_the_decoded_fixture(oracle):
    """cfg/{star,gal}_prior.jld decoded by tools/decode_priors.py"""
    import json, os
    here = os.path.dirname(os.path.abspath(__file__))
    prior = json.load(open(os.path.join(here, "golden", "priors.json")))
    assert prior["k"][0][:2] == pytest.approx([0.13654885, 0.12174624], abs=1e-8)
    assert prior["k"][1][:2] == pytest.approx([0.15035988, 0.13804955], abs=1e-8)
    assert prior["color_mean"][0][0] == pytest.approx([1.44260481, 0.59732183, 0.23383178, 0.12274786], abs=1e-8)
    assert prior["color_mean"][1][0] == pytest.approx([0.16258367, 0.91724579, 0.41012941, 0.34113808], abs=1e-8)
    for i in range(2):
        assert sum(prior["k"][i]) == pytest.approx(1.0, abs=1e-12)
    # default prior (NULL) == explicit prior
    from celeste_jl_amd import generic_init_source
    vs = generic_init_source([0.3, 0.4]); vs[26:28] = [0.3, 0.7]
    v0, d0, h0 = oracle.subtract_kl(vs)
    v1, d1, h1 = oracle.subtract_kl(vs, prior)
    assert v0 == v1 and np.array_equal(d0, d1) and np.array_equal(h0, h1)


def test_spline_is_the_natural_bicubic_interpolant(oracle):
    """Interpolations.jl BSpline(Cubic(Line())), OnGrid() == natural bicubic spline; cell clamp = extrapolation
    of the end polynomial.  Cross-checked with scipy's CubicSpline(bc_type='natural')."""
    from scipy.interpolate import CubicSpline
    from celeste_jl_amd import synthetic
    stamp = synthetic.band_psf(1); stamp = synthetic.render_psf(stamp)
    coef = oracle.spline_coefs(stamp)
    g = np.maximum(stamp, 0) + 1e-6; g = g / g.sum()
    g = np.where(1000 * g > 1, 1000 * g - 1, np.log(1000 * g))
    grid = np.arange(1, 52)
    rng = np.random.default_rng(0)
    pts = np.vstack([rng.uniform(1, 51, (40, 2)), rng.uniform(-3, 55, (20, 2))])
    for x, y in pts:
        # separable natural spline: first along h for every column, then along w
        col = np.array([CubicSpline(grid, g[:, j], bc_type="natural", extrapolate=True)(x) for j in range(51)])
        ref = CubicSpline(grid, col, bc_type="natural", extrapolate=True)(y)
        got = oracle.spline_value(coef, x, y)
        assert got == pytest.approx(float(ref), rel=1e-10, abs=1e-10)
    # interpolation property on the grid
    for (i, j) in [(1, 1), (26, 26), (51, 51), (7, 40)]:
        assert oracle.spline_value(coef, i, j) == pytest.approx(g[i - 1, j - 1], rel=1e-12, abs=1e-12)


def test_kl_closed_forms_against_monte_carlo(oracle):
    """test_kl.jl:19-28: each closed form agrees with a sample average of log q - log p within 4 standard errors;
    the whole subtract_kl (with the prior's full 4 x 4 colour covariances) is checked the same way"""
    from celeste_jl_amd.synthetic import load_prior
    from celeste_jl_amd import generic_init_source
    rng = np.random.default_rng(12)
    n = 400_000

    def check(exact, samples):
        se = samples.std() / math.sqrt(len(samples))
        assert abs(samples.mean() - exact) <= 4 * se, (exact, samples.mean(), se)

    # categorical
    p1, p2 = np.array([1, 2, 3, 4]) / 10, np.array([5, 6, 2, 1]) / 14
    idx = rng.choice(4, size=n, p=p1)
    check(oracle.categorical_kl(p1, p2), np.log(p1[idx]) - np.log(p2[idx]))
    # univariate normal
    x = rng.normal(0.5, math.sqrt(2.0), size=n)
    lq = -0.5 * (np.log(2 * np.pi * 2.0) + (x - 0.5) ** 2 / 2.0)
    lp = -0.5 * (np.log(2 * np.pi * 1.8) + (x - 0.8) ** 2 / 1.8)
    check(oracle.gaussian_kl(0.5, 2.0, 0.8, 1.8), lq - lp)
    # subtract_kl of a whole source = -E_q[log q - log p] + log p(radius): sample (a, k, r, c) from q
    prior = load_prior()
    vs = generic_init_source([3.0, 4.0])
    rng2 = np.random.default_rng(13)
    vs[26:28] = [0.3, 0.7]
    for i in range(2):
        vs[28 + 8 * i:36 + 8 * i] = rng2.dirichlet(np.ones(8) * 3)
        vs[6 + i] = 1.0 + i; vs[8 + i] = 0.4 + 0.2 * i
        vs[10 + 4 * i:14 + 4 * i] = rng2.normal(size=4) * 0.5
        vs[18 + 4 * i:22 + 4 * i] = 0.05 + 0.3 * rng2.random(4)
    exact, _, _ = oracle.subtract_kl(vs, prior)
    a = rng.choice(2, size=n, p=vs[26:28])
    total = np.log(vs[26 + a]) - np.log(np.asarray(prior["is_star"])[a])
    for i in range(2):
        m = a == i
        cnt = int(m.sum())
        kq, kp = vs[28 + 8 * i:36 + 8 * i], np.asarray(prior["k"][i])
        kd = rng.choice(8, size=cnt, p=kq)
        t = np.log(kq[kd]) - np.log(kp[kd])
        r = rng.normal(vs[6 + i], math.sqrt(vs[8 + i]), size=cnt)
        t += (-0.5 * (np.log(2 * np.pi * vs[8 + i]) + (r - vs[6 + i]) ** 2 / vs[8 + i])
              + 0.5 * (np.log(2 * np.pi * prior["flux_var"][i]) + (r - prior["flux_mean"][i]) ** 2 / prior["flux_var"][i]))
        mu, lam = vs[10 + 4 * i:14 + 4 * i], vs[18 + 4 * i:22 + 4 * i]
        c = mu + rng.normal(size=(cnt, 4)) * np.sqrt(lam)
        lqc = -0.5 * (np.log(2 * np.pi * lam).sum() + (((c - mu) ** 2) / lam).sum(axis=1))
        lpc = np.zeros(cnt)
        for d in range(8):
            sel = kd == d
            S = np.asarray(prior["color_cov"][i][d]).reshape(4, 4)
            dm = c[sel] - np.asarray(prior["color_mean"][i][d])
            lpc[sel] = -0.5 * (4 * math.log(2 * math.pi) + np.linalg.slogdet(S)[1] + np.einsum("ni,ij,nj->n", dm, np.linalg.inv(S), dm))
        total[m] += t + lqc - lpc
    x = vs[5]
    logp = -0.5 * (math.log(2 * math.pi) + math.log(prior["gal_radius_px_var"]) + (x - prior["gal_radius_px_mean"]) ** 2 / prior["gal_radius_px_var"])
    check(-(exact - logp), total)
