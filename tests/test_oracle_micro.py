"""The oracle against per-function micro-goldens that were computed without it (tests/golden/micro/micro_golden.json:
50-digit mpmath evaluations of the model definitions and numerical differentiation of them -- see
tests/golden/micro/make_micro_golden.py and tests/mp_model.py).  SURVEY.md 8(c): rows a9, a11-a17, a21, a25, a27, a31,
the data-free sub-case of test/test_psf.jl:74-143, and the whole elbo() value at 50 digits (7.3 step 1)."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "micro", "micro_golden.json")))
A = lambda x: np.array(x, dtype=object).astype(np.float64) if not isinstance(x, str) else float(x)


def close(got, want, rtol=1e-12, what=""):
    got = np.asarray(got, dtype=np.float64); want = np.asarray(A(want), dtype=np.float64)
    scale = np.maximum(np.abs(want), 1e-3 * np.abs(want).max()) if want.size else 1.0   # (entries that cancel to ~0 are judged against the block)
    err = np.abs(got - want) / np.where(scale == 0, 1.0, scale)
    assert err.max() <= rtol, (what, float(err.max()))


@pytest.mark.parametrize("name", list(G["bvn"]))
def test_bvn_chain(oracle, name):
    """eval_bvn_pdf!, get_bvn_derivs!, GalaxySigmaDerivs, transform_bvn_derivs! (BivariateNormals.jl:143-572)"""
    c = G["bvn"][name]; i = c["inputs"]
    r = oracle.micro_bvn(i["mean"], i["tau"], i["weight"], i["x"], i["J"], i["ratio"], i["angle"], i["radius"], i["nu"])
    close(r["f_pre"], c["f_pre"], 1e-13, "f_pre")
    close(r["x_d"], c["x_d"], what="bvn_x_d"); close(r["sig_d"], c["sig_d"], what="bvn_sig_d")
    close(r["xx_h"], c["xx_h"], what="bvn_xx_h"); close(r["xsig_h"], c["xsig_h"], what="bvn_xsig_h")
    close(r["sigsig_h"], c["sigsig_h"], what="bvn_sigsig_h")
    close(r["j"], c["j"], what="sig_sf.j")
    t = np.array(A(c["t"]))                      # json: t[sig][s1][s2]; hook: t[sig, s1, s2]
    close(r["t"], t, what="sig_sf.t")
    close(r["u_d"], c["u_d"], what="bvn_u_d"); close(r["s_d"], c["s_d"], what="bvn_s_d")
    close(r["uu_h"], c["uu_h"], what="bvn_uu_h"); close(r["ss_h"], c["ss_h"], what="bvn_ss_h")
    close(r["us_h"], c["us_h"], what="bvn_us_h")


def test_psf_pixel_value_and_its_derivatives(oracle):
    """test/test_psf.jl:74-143 without data: evaluate_psf_pixel_fit! at initialize_psf_params(2, for_test=true),
    x = (1, 2) -- value == get_psf_at_point, gradient / Hessian == differentiation of the value.  The pixel value and
    its derivatives in (mu, axis_ratio, angle, radius, weight) are assembled exactly as PSF.jl:399-470 does, from the
    oracle's BVN chain, and compared with the golden assembled the same way from the 50-digit derivatives."""
    assert float(G["psf_pixel_value"]) == pytest.approx(0.04087874823898267, rel=1e-15)

    def assemble(get):
        val, grads = 0.0, []
        for k in (1, 2):
            c = get(k)
            w = 0.5 + k / 10
            pdf = c["f_pre"]
            dlog = np.concatenate([c["u_d"], c["s_d"]])
            hlog = np.zeros((5, 5)); hlog[:2, :2] = c["uu_h"]; hlog[:2, 2:] = c["us_h"]; hlog[2:, :2] = np.transpose(c["us_h"])
            hlog[2:, 2:] = c["ss_h"]
            val += w * pdf
            grads.append((w * pdf * dlog, pdf, w * pdf * (hlog + np.outer(dlog, dlog)), pdf * dlog))
        return val, grads
    val_o, g_o = assemble(lambda k: oracle.micro_bvn([0.1, 0.2], [0, 0, 0], 1.0, [1.0, 2.0], np.eye(2), 0.8, math.pi / 4,
                                                     math.sqrt(2 * k), 1.0))

    def golden(k):
        c = G["bvn"]["psf_k%d" % k]
        return {key: np.array(A(c[key])) if key != "f_pre" else float(c[key]) for key in ("f_pre", "u_d", "s_d", "uu_h", "us_h", "ss_h")}
    val_g, g_g = assemble(golden)
    assert val_o == pytest.approx(val_g, rel=1e-14) and val_o == pytest.approx(float(G["psf_pixel_value"]), rel=1e-14)
    for a, b in zip(g_o, g_g):
        for x, y in zip(a, b):
            close(x, y, 1e-12, "evaluate_psf_pixel_fit! derivative")
    # and the oracle's get_psf_at_point restatement
    psf = [[0.5 + k / 10, 0.1, 0.2] + [float(v) for v in np.array(oracle.get_bvn_cov(0.8, math.pi / 4, math.sqrt(2 * k))).reshape(-1)[[0, 1, 3]]]
           for k in (1, 2)]
    assert oracle.psf_at_point(np.array(psf), 1.0, 2.0) == pytest.approx(float(G["psf_pixel_value"]), rel=1e-14)


def test_source_brightness(oracle):
    """SourceBrightness (source_brightness.jl:27-202): every moment's value, gradient and Hessian"""
    sb = oracle.micro_brightness(np.array(G["brightness"]["vs"]))
    for c in G["brightness"]["cases"]:
        v, d, h = sb[(c["moment"], c["b"] - 1, c["i"])]
        close(v, c["v"], 1e-14, "value"); close(d, c["d"], what="gradient"); close(h, c["h"], what="Hessian")


@pytest.fixture(scope="module")
def galaxy_pixel(oracle):
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("galaxy")
    assert np.array_equal(f.vp[0], np.array(G["kl"]["vs"]))
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    p = G["pixel"]
    return oracle.micro_pixel(pb, f.vp, 0, p["n"], p["h"], p["w"]), f, pb


def test_star_and_galaxy_densities_at_a_pixel(galaxy_pixel):
    """star_light_density! (fsm_util.jl:221-248) and populate_gal_fsm! (:194-346) with their 2- and 6-parameter
    derivatives"""
    r, _, _ = galaxy_pixel
    for name in ("fs0m", "fs1m"):
        v, d, h = r[name]
        c = G["pixel"][name]
        close(v, c["v"], 1e-13, name); close(d, c["d"], 1e-11, name + ".d"); close(h, c["h"], 1e-10, name + ".h")


@pytest.mark.parametrize("name", ["E_G_s", "var_G_s", "elbo_log_term"])
def test_pixel_moments_and_log_term(galaxy_pixel, name):
    """calculate_G_s! (elbo_objective.jl:17-233) and add_elbo_log_term! (:274-327): value, the 28 likelihood gradient
    entries and a sample of Hessian entries (the k block has no likelihood derivative)"""
    r, _, _ = galaxy_pixel
    v, d, h = r[name]
    c = G["pixel"][name]
    close(v, c["v"], 1e-13, name)
    close(d[:28], c["d"], 1e-11, name + ".d")
    assert np.all(d[28:] == 0) and np.all(h[28:, :] == 0)
    got = np.array([h[a, b] for a, b in G["pixel"]["pairs"]])
    want = np.array([float(x) for x in c["h_pairs"]])
    scale = np.maximum(np.abs(want), 1e-9 * np.abs(h).max())
    assert (np.abs(got - want) / scale).max() <= 1e-10, name
    # combine_sfs_hessian! fills the full matrix: symmetric to rounding only (SURVEY.md trap A22)
    assert np.abs(h - h.T).max() <= 1e-14 * np.abs(h).max()


def test_star_density_inside_clamped_and_log_branch(oracle, galaxy_pixel):
    """the spline index h - m + 26 inside the stamp, outside it (cell clamped to [1, 50]) and where the spline value is
    negative (softpluslikeinv's exponential branch)"""
    _, f, pb = galaxy_pixel
    from celeste_jl_amd import cabi
    p = f.patches[0][G["pixel"]["n"]]
    coef = oracle.spline_coefs(p.stamp)
    m = p.wcs_jacobian @ (f.vp[0][0:2] - p.world_center) + p.pixel_center
    seen_neg = False
    for c in G["star_density"]:
        y = oracle.spline_value(coef, c["h"] - m[0] + 26, c["w"] - m[1] + 26)
        close(y, c["spline"], 1e-12, "spline value")
        dens = 1e-3 * math.exp(y) if y < 0 else 1e-3 * (y + 1)
        close(dens, c["v"], 1e-10, "star density")   # exp(y) turns the spline's absolute error (fp64 prefilter, |y| ~ 15) into a relative one
        seen_neg |= y < 0
    assert seen_neg and any(abs(c["h"] - m[0]) > 25 or abs(c["w"] - m[1]) > 25 for c in G["star_density"])


def test_subtract_kl(oracle):
    """subtract_kl (elbo_kl.jl:94-154): value, all 44 gradient entries, a sample of Hessian entries"""
    from celeste_jl_amd.synthetic import load_prior
    k = G["kl"]
    v, d, h = oracle.subtract_kl(np.array(k["vs"]), load_prior())
    close(v, k["v"], 1e-13, "KL value"); close(d, k["d"], 1e-11, "KL gradient")
    got = np.array([h[a, b] for a, b in k["pairs"]]); want = np.array([float(x) for x in k["h_pairs"]])
    assert (np.abs(got - want) / np.maximum(np.abs(want), 1e-9 * np.abs(h).max())).max() <= 1e-10


@pytest.mark.parametrize("scene", list(G["elbo_value"]))
def test_elbo_value_against_50_digit_evaluation(oracle, scene):
    """elbo() of the sample star / galaxy scenes and of a two-source field with a varying sky
    plane, per-row calibration and per-patch stamps -- fp64 oracle within 1e-13 of the 50-digit value"""
    from celeste_jl_amd import synthetic, cabi
    if scene.startswith("sample_"):
        f = synthetic.make_sample_dataset(scene[len("sample_"):])
    else:
        f = synthetic.make_field(24, 26, 2, seed=8, variable=True, margin=8)
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    t = G["elbo_value"][scene]["target"]
    v, _, _, _, st = oracle.elbo_one(pb, f.vp, t)
    want = float(G["elbo_value"][scene]["v"])
    assert st == 0 and abs(v - want) <= 1e-13 * abs(want), (scene, v, want)


def test_mp_model_is_alive_on_a_few_pixels(oracle):
    """the 50-digit model itself (not only its stored outputs): a 3 x 4 pixel patch of the star scene, evaluated now"""
    import mp_model as M
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.model import get_sky_patches, neighbor_map
    f = synthetic.make_sample_dataset("star")
    patches = get_sky_patches(f.images[:1], f.catalog, radius_override_pix=1.6)
    nbrs = neighbor_map(patches)
    pb = cabi.Problem(f.images[:1], patches, nbrs)
    v, _, _, cnt, st = oracle.elbo_one(pb, f.vp, 0)
    mv = M.elbo_value(f.images[:1], patches, nbrs, f.vp, 0, synthetic.load_prior())
    assert st == 0 and cnt[0] > 0 and abs(v - float(mv)) <= 1e-13 * abs(float(mv))
