"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle on identical inputs."""
import os

import numpy as np
import pytest

from parity_util import assert_parity, rel_err

# CELESTE_FUZZ_SEEDS=N: every seeded fuzz test with N seeds instead of its default handful (a long run on a GPU box)
FUZZ_SEEDS = int(os.environ.get("CELESTE_FUZZ_SEEDS", "0"))
pytestmark = pytest.mark.gpu

ALL = 1 | 2 | 4


def _ctx(field, **kw):
    import celeste_jl_amd as cel
    return cel.FieldContext(field.images, field.patches, field.neighbors, **kw)


@pytest.mark.parametrize("kind", ["star", "galaxy", "two_body", "three_body"])
def test_sample_datasets(oracle, kind):
    """config 1 analogues (test/SampleData.jl:161-236)"""
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset(kind)
    ctx = _ctx(f)
    tg = list(range(len(f.catalog)))
    errs = assert_parity(ctx.eval_batch(f.vp, tg, ALL), oracle.elbo_batch(ctx.problem, f.vp, tg, ALL), kind)
    print(kind, errs)


@pytest.mark.parametrize("flags", [0, 4, 1, 1 | 4, 2, ALL])
def test_flags(oracle, flags):
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("two_body")
    ctx = _ctx(f)
    g = ctx.eval_batch(f.vp, [0, 1], flags)
    r = oracle.elbo_batch(ctx.problem, f.vp, [0, 1], flags)
    assert_parity(g, r, "flags=%d" % flags)


def test_small_field_with_neighbors(oracle):
    """mixed star/galaxy field, overlapping patches, 0.5 % NaN pixels (config 3b in miniature)"""
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(160, 200, 40, seed=11, nan_fraction=0.005)
    assert sum(len(n) for n in f.neighbors) > 0
    ctx = _ctx(f)
    tg = list(range(len(f.catalog)))
    errs = assert_parity(ctx.eval_batch(f.vp, tg, ALL), oracle.elbo_batch(ctx.problem, f.vp, tg, ALL), "field")
    print(errs)


def test_split_variant_record_sum(oracle):
    """CELESTE_FLAG_SPLIT: per-pixel records to HBM + streaming per-patch sum gives the
    same SensitiveFloat as the fused kernel; partial tiles, NaN pixels, neighbours, a subset of targets"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_field(160, 200, 40, seed=11, nan_fraction=0.005)
    ctx = _ctx(f)
    for tg in (list(range(len(f.catalog))), [7, 3, 21]):
        ref = oracle.elbo_batch(ctx.problem, f.vp, tg, ALL)
        g = ctx.eval_batch(f.vp, tg, ALL | cabi.FLAG_SPLIT)
        errs = assert_parity(g, ref, "split")
        fused = ctx.eval_batch(f.vp, tg, ALL)
        assert np.max(np.abs(g[0] - fused[0]) / np.abs(fused[0])) < 1e-12
        print("split", errs)
    with pytest.raises(cabi.CelesteError):
        ctx.eval_batch(f.vp, [0], 1 | cabi.FLAG_SPLIT)   # needs HESS


def test_explicit_bitmaps(oracle):
    """test_elbo.jl:64-130 manipulates active_pixel_bitmap by hand; neighbours' bitmaps gate their light"""
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("two_body")
    for n in range(5):
        f.patches[1][n].active_pixel_bitmap[:] = False
    f.patches[1][4].active_pixel_bitmap[9:11, 9:11] = True
    f.patches[0][2].active_pixel_bitmap[3:7, 5] = False
    ctx = _ctx(f)
    assert_parity(ctx.eval_batch(f.vp, [0, 1], ALL), oracle.elbo_batch(ctx.problem, f.vp, [0, 1], ALL), "bitmaps")


def test_gal_frac_dev_edge_values(oracle):
    """gal_frac_dev exactly 0 or 1 (outside catalog_init_source's clamp but legal for elbo()): the kernel's
    per-profile partial-sum shortcut is replaced by explicit accumulation"""
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(128, 128, 12, seed=5)
    vp = f.vp.copy()
    vp[0::3, 2] = 0.0
    vp[1::3, 2] = 1.0
    vp[2::3, 2] = 1e-9
    ctx = _ctx(f)
    tg = list(range(12))
    errs = assert_parity(ctx.eval_batch(vp, tg, ALL), oracle.elbo_batch(ctx.problem, vp, tg, ALL), "dev edge")
    print(errs)


def test_fp32_component_loop_within_1e4(oracle):
    """BASELINE config 5 precision mode: float galaxy component loop, everything else fp64; compared with the
    fp64 oracle at 1e-4 relative on v and on ||.||inf-scaled d (and h)"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_field(200, 240, 40, seed=4, nan_fraction=0.002)
    ctx = _ctx(f)
    tg = list(range(40))
    for flags in (1 | 4 | cabi.FLAG_FP32, ALL | cabi.FLAG_FP32):
        v, d, h, cnt, st = ctx.eval_batch(f.vp, tg, flags)
        ov, od, oh, ocnt, ost = oracle.elbo_batch(ctx.problem, f.vp, tg, flags & 7)
        assert (st == 0).all() and np.array_equal(cnt, ocnt)
        ev = np.max(np.abs(v - ov) / np.abs(ov))
        ed = max(np.abs(d[t] - od[t]).max() / np.abs(od[t]).max() for t in tg)
        assert ev <= 1e-4 and ed <= 1e-4, (ev, ed)
        if h is not None:
            eh = max(np.abs(h[t] - oh[t]).max() / np.abs(oh[t]).max() for t in tg)
            assert eh <= 1e-4, eh
            print("fp32 errs", ev, ed, eh)


def _affine_variable_psf_field(seed=7):
    """A field whose patches carry a non-identity affine WCS Jacobian, shifted world/pixel centres and a
    different PSF stamp per patch (variable PSF map): exercises u_d = -J' x_d, uu_h = J' xx_h J
    (BivariateNormals.jl:424-447) and the per-patch spline (imaged_sources.jl:97-107)."""
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.model import render_psf, make_psf
    f = synthetic.make_field(160, 200, 16, seed=seed)
    rng = np.random.default_rng(seed)
    J = np.array([[0.92, 0.11], [-0.07, 1.05]])
    Jinv = np.linalg.inv(J)
    for s, row in enumerate(f.patches):
        for n, p in enumerate(row):
            # world = Jinv (pix - pc) + wc, so that pix = J (world - wc) + pc keeps the source where it was
            p.wcs_jacobian = J.copy()
            p.world_center = np.array([3.0 + 0.1 * s, -2.0 + 0.05 * n])
            sig = 1.2 + 0.5 * rng.random()
            p.stamp = render_psf(make_psf((0.7, 0.3), [(0.05, -0.1), (0.0, 0.2)],
                                          [np.eye(2) * sig ** 2, np.array([[6.0, 0.8], [0.8, 5.0]])]))
            p.psf = make_psf((0.75, 0.25), [(0.1, -0.05), (-0.2, 0.1)],
                             [np.array([[sig ** 2, 0.2], [0.2, 1.3 * sig ** 2]]), np.array([[6.5, -0.7], [-0.7, 5.5]])])
        # re-express the source position in the new world coordinates
        p0 = row[0]
        pix = f.vp[s, 0:2].copy()
        f.vp[s, 0:2] = Jinv @ (pix - p0.pixel_center) + p0.world_center
        for p in row[1:]:
            # every patch of a source must map the same world position to the same pixel position
            p.world_center = f.vp[s, 0:2] - Jinv @ (pix - p.pixel_center)
    return f


def test_affine_wcs_and_variable_psf(oracle):
    f = _affine_variable_psf_field()
    ctx = _ctx(f)
    assert ctx.problem.c.n_stamps == 16 * 5
    tg = list(range(16))
    errs = assert_parity(ctx.eval_batch(f.vp, tg, ALL), oracle.elbo_batch(ctx.problem, f.vp, tg, ALL), "affine")
    print("affine/variable psf", errs)


def test_device_spline_prefilter_equals_the_host_function():
    """spline_prefilter_kernel (celeste_ctx_create conditions and prefilters EVERY stamp of the problem on the device) against
    celeste_spline_prefilter (one stamp, on the host -- the function test_host_logic pins to the oracle and to scipy): the 80
    stamps of the variable-PSF field, some of them doctored (negative pixels, which the constructor clamps; a dead row; a
    single hot pixel).  Same operations in the same order: equal to the last bit wherever the two logarithms agree."""
    from celeste_jl_amd import cabi
    f = _affine_variable_psf_field()
    rng = np.random.default_rng(11)
    for s in range(0, 16, 3):
        st = f.patches[s][s % 5].stamp
        st[rng.integers(0, 51, 40), rng.integers(0, 51, 40)] *= -1.0
        st[7, :] = 0.0
    f.patches[1][0].stamp[:] = 0.0
    f.patches[1][0].stamp[25, 25] = 1.0
    ctx = _ctx(f)
    K = ctx.problem.c.n_stamps
    assert K == 16 * 5
    worst, equal = 0.0, 0
    for k in range(K):
        host = cabi.spline_prefilter(ctx.problem.stamps[k].reshape(51, 51).T)
        dev = ctx.spline_coefficients(k)
        assert np.isfinite(dev).all()
        worst = max(worst, float(np.max(np.abs(dev - host)) / np.max(np.abs(host))))
        equal += int(np.array_equal(dev, host))
    print("device prefilter vs host: worst relative difference %.1e, %d of %d stamps bit-identical" % (worst, equal, K))
    assert worst <= 1e-14
    with pytest.raises(cabi.CelesteError):
        ctx.spline_coefficients(K)


def test_multifield_overlapping_images(oracle):
    """BASELINE config 5 in miniature: 2 x 2 overlapping fields (20 images); a source only has patches in the
    images it overlaps.  fp64 parity, then the fp32 component loop at the stated 1e-4 tolerance."""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_multifield((2, 2), 192, 192, 0.10, 60, seed=5)
    n_img = [sum(p.active_pixel_bitmap.size > 0 for p in row) for row in f.patches]
    assert max(n_img) >= 10 and min(n_img) >= 5      # sources in overlap regions see 2 or 4 fields
    ctx = _ctx(f)
    tg = list(range(len(f.catalog)))
    ref = oracle.elbo_batch(ctx.problem, f.vp, tg, ALL)
    errs = assert_parity(ctx.eval_batch(f.vp, tg, ALL), ref, "multifield")
    print("multifield", errs, "images per source", min(n_img), max(n_img))
    v, d, h, cnt, st = ctx.eval_batch(f.vp, tg, 1 | 4 | cabi.FLAG_FP32)
    assert np.max(np.abs(v - ref[0]) / np.abs(ref[0])) <= 1e-4
    assert max(np.abs(d[t] - ref[1][t]).max() / np.abs(ref[1][t]).max() for t in tg) <= 1e-4


def test_sparse_patch_list_abi(oracle):
    """the sparse patch list of celeste_problem_t (many-image problems) gives bit-identical results to the dense
    table, is checked for order / range, and matches the oracle"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    fd = synthetic.make_multifield((2, 3), 160, 160, 0.10, 70, seed=11)
    fs = synthetic.make_multifield((2, 3), 160, 160, 0.10, 70, seed=11, sparse=True)
    cd, cs = _ctx(fd), _ctx(fs)
    assert cs.problem.sparse and cs.problem.c.n_patch_entries < 0.5 * 70 * 30
    tg = list(range(70))
    for flags in (ALL, ALL | cabi.FLAG_SPLIT, 5 | cabi.FLAG_FP32):
        a, b = cd.eval_batch(fd.vp, tg, flags), cs.eval_batch(fs.vp, tg, flags)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert_parity(cs.eval_batch(fs.vp, tg, ALL), oracle.elbo_batch(cs.problem, fs.vp, tg, ALL), "sparse list")
    assert cd.work_stats(tg) == cs.work_stats(tg)
    assert np.array_equal(cd.maximize_batch(fd.vp, tg[:20])[0], cs.maximize_batch(fs.vp, tg[:20])[0])
    # unsorted / duplicate / out-of-range entries are rejected
    pr = cs.problem
    lib = cabi.load_library()
    import ctypes as C
    for mutate in ("swap", "dup", "range"):
        src, img = pr.patch_source.copy(), pr.patch_image.copy()
        if mutate == "swap":
            src[[0, -1]] = src[[-1, 0]]; img[[0, -1]] = img[[-1, 0]]
        elif mutate == "dup":
            src[1], img[1] = src[0], img[0]
        else:
            img[3] = 30
        c = cabi.ProblemT.from_buffer_copy(pr.c)
        c.patch_source = src.ctypes.data_as(cabi.c_int32_p); c.patch_image = img.ctypes.data_as(cabi.c_int32_p)
        h = C.c_void_p()
        assert lib.celeste_ctx_create(C.byref(c), 0, C.byref(h)) == cabi.ERR_INVALID_ARG


def test_expected_image_renderer():
    """celeste_render_expected vs an independent numpy/torch rendering (write_celeste_expectation.jl:112-156)"""
    from celeste_jl_amd import synthetic
    from joint_objective import expected_planes
    f = synthetic.make_field(160, 200, 30, seed=11, nan_fraction=0.004)
    ctx = _ctx(f)
    ref = expected_planes(f.images, f.patches, f.vp)
    for n in (0, 2, 4):
        got = ctx.render_expected(f.vp, n)
        assert got.shape == ref[n].shape
        assert np.abs(got - ref[n]).max() <= 1e-12 * np.abs(ref[n]).max()
        assert (got > 0).sum() > 1000


def test_random_parameters_inside_the_constraint_boxes(oracle):
    """stress: variational parameters drawn uniformly from the optimiser's constraint boxes
    (ElboMaximize.jl:63-93), i.e. everything maximize! can ever hand to elbo()"""
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(200, 240, 40, seed=4)
    rng = np.random.default_rng(123)
    S = len(f.catalog)
    lo = np.array([0, 0, 1e-2, 1e-2, -10, .1] + [-1] * 2 + [1e-4] * 2 + [-10] * 8 + [1e-4] * 8)
    hi = np.array([0, 0, .99, .99, 10, 70] + [10] * 2 + [.1] * 2 + [10] * 8 + [1.0] * 8)
    ctx = _ctx(f)
    worst = {}
    for trial in range(3):
        vp = f.vp.copy()
        u = rng.random((S, 26))
        vp[:, 2:26] = (lo + u * (hi - lo))[:, 2:]
        vp[:, 0:2] += rng.uniform(-1.5, 1.5, (S, 2))
        vp[:, 5] = np.exp(rng.uniform(np.log(0.1), np.log(70.0), S))          # radius: log-uniform
        vp[:, 10:18] = rng.uniform(-3, 3, (S, 8))                              # colours: keep fluxes finite
        a = rng.uniform(0.005, 0.995, S); vp[:, 26] = a; vp[:, 27] = 1 - a
        for i in range(2):
            k = rng.dirichlet(np.ones(8), S) * (1 - 0.01) + 0.01 / 8
            vp[:, 28 + 8 * i:36 + 8 * i] = k
        tg = list(range(S))
        g = ctx.eval_batch(vp, tg, ALL)
        r = oracle.elbo_batch(ctx.problem, vp, tg, ALL)
        errs = assert_parity(g, r, "random trial %d" % trial)
        for k2, v2 in errs.items():
            worst[k2] = max(worst.get(k2, 0), v2)
    print("random-parameter parity", worst)


def test_multiple_active_sources(oracle):
    """ElboArgs with Sa > 1 (test_elbo.jl:64-130): value on the union of the active patches, gradient columns,
    diagonal and cross Hessian blocks; swap invariance; a third, inactive source contributes value-only"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    for kind, acts in (("two_body", ([0, 1], [1, 0])), ("three_body", ([0, 2], [2, 1, 0], [1]))):
        f = synthetic.make_sample_dataset(kind)
        S = len(f.catalog)
        for act in acts:
            ea = cel.ElboArgs(f.images, f.patches, act)
            sf = cel.elbo(ea, f.vp)
            pb = cabi.Problem(f.images, f.patches, [[s for s in range(S) if s != a] if a in act else [] for a in range(S)])
            for flags in (ALL, 1 | 4, 0):
                ov, od, oh, ocnt, ost = oracle.elbo_multi(pb, f.vp, act, flags)
                assert ost == 0
                if len(act) == 1:
                    break
                v, d, h, cnt = ea._ctx.eval_multi(f.vp, act, flags)
                assert abs(v - ov) <= 1e-8 * abs(ov) and np.array_equal(cnt, ocnt), (kind, act, flags, v, ov, cnt, ocnt)
                if flags & 3:
                    assert max(rel_err(d[:, k], od[k]) for k in range(len(act))) <= 1e-8
                if flags & 2:
                    assert np.array_equal(h, h.T)
                    e = rel_err(h, oh)
                    assert e <= 1e-8, (kind, act, e)
                    print(kind, act, "multi-active errs", abs(v - ov) / abs(ov), e, "cross block max", np.abs(h[:44, 44:88]).max())
                    if kind == "two_body":
                        assert np.abs(h[:44, 44:88]).max() > 0
        # swap invariance (test_elbo.jl:107-129)
        if kind == "two_body":
            a01 = cel.elbo(cel.ElboArgs(f.images, f.patches, [0, 1]), f.vp)
            a10 = cel.elbo(cel.ElboArgs(f.images, f.patches, [1, 0]), f.vp)
            assert a01.v == pytest.approx(a10.v, rel=1e-13)
            assert np.allclose(a01.d[:, 0], a10.d[:, 1], rtol=1e-11) and np.allclose(a01.h[:44, 44:], a10.h[44:, :44], rtol=1e-9, atol=1e-9 * np.abs(a01.h).max())
            one = cel.elbo(cel.ElboArgs(f.images, f.patches, [0]), f.vp)
            assert np.allclose(one.h, a01.h[:44, :44], rtol=1e-9, atol=1e-10 * np.abs(one.h).max())


@pytest.mark.parametrize("K", [1, 3, 4])
def test_psf_K_other_than_two(oracle, K):
    """ElboArgs.psf_K (elbo_args.jl:197) other than the SDSS default: 14 K galaxy components per source"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.model import get_sky_patches, neighbor_map, ConstantPSFMap, render_psf
    f = synthetic.make_sample_dataset("two_body")
    rng = np.random.default_rng(K)
    for img in f.images:
        w = rng.dirichlet(np.ones(K) * 4)
        comps = []
        for k in range(K):
            s = 1.1 + 0.9 * k + 0.2 * rng.random()
            comps.append([w[k], 0.3 * rng.normal(), 0.3 * rng.normal(), s * s, 0.1 * s * s * rng.normal(), s * s * (1 + 0.2 * rng.random())])
        img.psf = np.array(comps)
        img.psfmap = ConstantPSFMap(render_psf(img.psf, (51, 51)))
    patches = get_sky_patches(f.images, f.catalog)
    nbrs = neighbor_map(patches)
    ctx = cel.FieldContext(f.images, patches, nbrs, psf_K=K)
    errs = assert_parity(ctx.eval_batch(f.vp, [0, 1], ALL), oracle.elbo_batch(ctx.problem, f.vp, [0, 1], ALL), "psf_K=%d" % K)
    print("psf_K", K, errs)


def test_duplicate_and_unordered_targets():
    """the batch's work list is built per target *slot*: a source listed twice, or in any order, gets the same numbers"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_field(200, 260, 70, seed=19, margin=20)
    ctx = _ctx(f)
    base = ctx.eval_batch(f.vp, list(range(70)), ALL)
    tg = [5, 5, 69, 0, 33, 5, 12, 69] + list(range(69, -1, -1)) * 5      # 358 slots: several count / fill blocks
    for flags in (ALL, 5, ALL | cabi.FLAG_FP32):
        ref = base if flags == ALL else ctx.eval_batch(f.vp, list(range(70)), flags)
        got = ctx.eval_batch(f.vp, tg, flags)
        for x, y in zip(got, ref):
            assert (x is None and y is None) or np.array_equal(x, y[tg])


def test_edge_cases(oracle):
    """empty target list; a target without a single active pixel (ELBO = -KL, derivatives of the KL only); a source
    whose box misses every image (clamp_box, imaged_sources.jl:10-14); the last-column rule on a 1-column patch"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("three_body")
    ctx = _ctx(f)
    v, d, h, cnt, st = ctx.eval_batch(f.vp, [], ALL)
    assert v.size == 0 and st.size == 0
    # no active pixel at all for source 1
    for n in range(5):
        f.patches[1][n].active_pixel_bitmap[:] = False
    # source 2: one-column patches (contributes nothing: 1 <= w2 < W2 is empty), still visited
    for n in range(5):
        p = f.patches[2][n]
        keep = p.active_pixel_bitmap[:, :1].copy()
        (h0, h1), (w0, w1) = p.box
        p.box = ((h0, h1), (w0, w0))
        p.active_pixel_bitmap = keep
    ctx = _ctx(f)
    g = ctx.eval_batch(f.vp, [0, 1, 2], ALL)
    r = oracle.elbo_batch(ctx.problem, f.vp, [0, 1, 2], ALL)
    assert_parity(g, r, "edge")
    assert g[3][1, 0] == 0 and g[3][2, 0] == 0 and g[3][0, 0] > 0
    kl_only = ctx.eval_batch(f.vp, [1], ALL)[0][0]
    from torch_value_model import neg_kl, load_prior
    import torch
    assert kl_only == pytest.approx(float(neg_kl(torch.tensor(f.vp[1]), load_prior())), rel=1e-12)
    # a source far outside every image: empty boxes everywhere
    from celeste_jl_amd.model import get_sky_patches, neighbor_map
    from celeste_jl_amd.params import catalog_init_source
    ce = synthetic.sample_ce([4000.0, -900.0], True)
    cat = list(f.catalog) + [ce]
    patches = get_sky_patches(f.images, cat)
    assert all(p.active_pixel_bitmap.size == 0 for p in patches[3])
    vp = np.vstack([f.vp, catalog_init_source(ce)])
    ctx = cel.FieldContext(f.images, patches, neighbor_map(patches))
    assert_parity(ctx.eval_batch(vp, [3, 0], ALL), oracle.elbo_batch(ctx.problem, vp, [3, 0], ALL), "outside")
    # the same in a multi-field problem (sparse visit lists: sources appear in a subset of the images)
    f = synthetic.make_multifield(grid=(1, 2), H=96, W=96, n_sources=10, seed=9)
    cat = list(f.catalog) + [ce]
    patches = get_sky_patches(f.images, cat)
    vp = np.vstack([f.vp, catalog_init_source(ce)])
    ctx = cel.FieldContext(f.images, patches, neighbor_map(patches))
    tg = [10, 0, 4, 9]
    assert_parity(ctx.eval_batch(vp, tg, ALL), oracle.elbo_batch(ctx.problem, vp, tg, ALL), "outside, multi-field")


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 16))
def test_randomised_small_fields(oracle, seed):
    """fuzz: random image size, source count, NaN fraction, explicit bitmaps, target subset / order, flag set, and
    either evaluation path (fused / split / visit lists forced by giving one source no patch in one image)"""
    _randomised_field(oracle, seed, 1000, (60, 140), (1, 14))


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 4))
def test_randomised_medium_fields(oracle, seed):
    """the same fuzz with 33 .. 90 sources: batches beyond the 32 targets of the one-launch path (eval_fused_kernel), i.e. the
    work list + pixel_kernel + lift chain with groups of one chunk, on crowded scenes"""
    _randomised_field(oracle, seed, 3000, (120, 240), (33, 91))


def _report_unexpected_statuses(cel, f, ctx, psf_K, tg, flags, g, what):
    """every scene of the fuzz is finite in the oracle: a non-zero status is a device-side fault.  Say what came back, whether
    the same context repeats it and whether a fresh context does, then fail (one such event was seen in ~10 000 runs)"""
    lines = ["%s: statuses %s" % (what, [(k, tg[k], int(s_)) for k, s_ in enumerate(g[4]) if s_ != 0])]
    for k in [k for k in range(len(tg)) if g[4][k] != 0][:4]:
        lines.append("  target %d: v %r counters %s non-finite d %s non-finite h %s" % (
            tg[k], g[0][k], g[3][k].tolist(), None if g[1] is None else np.argwhere(~np.isfinite(g[1][k])).ravel().tolist(),
            None if g[2] is None else np.argwhere(~np.isfinite(g[2][k]))[:12].tolist()))
    for name, c2 in (("same context again", ctx), ("fresh context", cel.FieldContext(f.images, f.patches, f.neighbors, psf_K=psf_K))):
        g2 = c2.eval_batch(f.vp, tg, flags, raise_on_error=False)
        lines.append("  %s: statuses %s" % (name, [(k, tg[k], int(s_)) for k, s_ in enumerate(g2[4]) if s_ != 0]))
    # is it the call's INPUT as the library sees it?  the same table from a fresh copy, the oracle on the same scene, and what
    # the HIP runtime believes about the table's address
    vp2 = np.array(f.vp, copy=True)
    g3 = ctx.eval_batch(vp2, tg, flags, raise_on_error=False)
    lines.append("  same context, the table copied to a new array: statuses %s" % [(k, tg[k], int(s_)) for k, s_ in enumerate(g3[4]) if s_ != 0])
    lines.append("  table finite: %s; address %#x, %d bytes" % (bool(np.isfinite(f.vp).all()), f.vp.ctypes.data, f.vp.nbytes))
    try:
        import ctypes as C
        path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0]
        hip = C.CDLL(path)

        class Attr(C.Structure):
            _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                        ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]
        for nm, arr in (("table", f.vp), ("copy", vp2)):
            for off in (0, arr.nbytes - 1):
                a = Attr()
                rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(arr.ctypes.data + off))
                lines.append("  hipPointerGetAttributes(%s + %d): rc %d type %d device %d devicePointer %s hostPointer %s flags %d" % (
                    nm, off, rc, a.type, a.device, a.devicePointer and hex(a.devicePointer), a.hostPointer and hex(a.hostPointer), a.allocationFlags))
        hip.hipGetLastError()
    except Exception as e:   # noqa: BLE001
        lines.append("  (pointer attributes not read: %r)" % (e,))
    try:
        from oracle import oracle as _orc
        r_ = _orc.elbo_batch(ctx.problem, f.vp, tg, flags)
        lines.append("  oracle on the same scene: non-finite values %d, statuses %s" % (int((~np.isfinite(r_[0])).sum()), np.asarray(r_[4]).tolist() if len(r_) > 4 else None))
    except Exception as e:   # noqa: BLE001
        lines.append("  (oracle not run: %r)" % (e,))
    try:   # the scene itself, for a comparison with the same seed built elsewhere
        os.makedirs("gpurun_out/flaky", exist_ok=True)
        np.savez_compressed("gpurun_out/flaky/scene_%s_%d.npz" % (what.split()[1], os.getpid()), vp=f.vp,
                            pixels=np.stack([im.pixels for im in f.images]), sky=np.stack([im.sky for im in f.images]),
                            iota=np.stack([im.nelec_per_nmgy for im in f.images]),
                            boxes=np.array([[[p.box[0][0], p.box[0][1], p.box[1][0], p.box[1][1]] for p in row] for row in f.patches]),
                            psf=np.array([[np.asarray(p.psf).ravel() for p in row] for row in f.patches]),
                            stamps=np.array([np.asarray(p.stamp) for p in f.patches[tg[[k for k in range(len(tg)) if g[4][k] != 0][0]]]]),
                            bad=np.array([tg[k] for k in range(len(tg)) if g[4][k] != 0]))
    except Exception as e:   # noqa: BLE001
        lines.append("  (scene not saved: %r)" % (e,))
    msg = "\n".join(lines)
    try:
        os.makedirs("gpurun_out/flaky", exist_ok=True)
        with open("gpurun_out/flaky/%s_%d.txt" % (what.replace(" ", "_").replace("(", "").replace(")", ""), os.getpid()), "w") as fh:
            fh.write(msg + "\n")
    except OSError:
        pass
    raise AssertionError(msg)


_FUZZ_HISTORY = []     # what the last few fuzz tests of this process did, and where their small arrays lived


def _check_scene_constants(f, seed):
    """make_field's calibration rows and sky planes are constants: a changed entry is a write from outside.  Seen in round 6:
    one word decremented and one zeroed in ~1 scene of 800 under `pytest -n 8` -- the HIP runtime writing into a stream object
    that hipStreamDestroy had already freed (celeste_abi.hip, stream_retire: streams are recycled since).  Says where, what
    lived at that address before, and -- with tools/heapwho preloaded -- who allocated and freed the block."""
    import gc
    lines = []
    for b, im in enumerate(f.images):
        for name, a in (("nelec_per_nmgy", im.nelec_per_nmgy), ("sky", im.sky)):
            flat = a.reshape(-1)
            ref = np.median(flat)
            bad = np.flatnonzero(flat != ref)
            if bad.size:
                lo, hi = a.ctypes.data, a.ctypes.data + a.nbytes
                lines.append("seed %d image %d %s (%d bytes at %#x): entries %s are %s instead of %r (words %s)" % (
                    seed, b, name, a.nbytes, lo, bad[:8].tolist(), flat[bad[:8]].tolist(), float(ref),
                    [hex(int(x)) for x in flat.view(np.uint32)[bad[:8]]]))
                for rec in _FUZZ_HISTORY:
                    for nm, (addr, nb) in rec["arrays"].items():
                        if addr < hi and addr + nb > lo:
                            lines.append("    overlaps %s of the test %s (%d bytes at %#x)" % (nm, rec["what"], nb, addr))
    if lines:
        lines.append("    gc counts %s; earlier tests of this process: %s" % (gc.get_count(), [r["what"] for r in _FUZZ_HISTORY]))
        try:    # tools/heapwho preloaded: the earlier owners of the corrupted block
            import ctypes as C
            hw = C.CDLL(None)
            hw.heapwho_dump.argtypes = [C.c_void_p, C.c_char_p]
            os.makedirs("gpurun_out/flaky", exist_ok=True)
            for im in f.images[:1]:
                hw.heapwho_dump(C.c_void_p(im.nelec_per_nmgy.ctypes.data), ("gpurun_out/flaky/heapwho_%d.txt" % os.getpid()).encode())
        except (AttributeError, OSError):
            pass
        try:
            os.makedirs("gpurun_out/flaky", exist_ok=True)
            with open("gpurun_out/flaky/corrupt_%d.txt" % os.getpid(), "a") as fh:
                fh.write("\n".join(lines) + "\n")
        except OSError:
            pass
        raise AssertionError("\n".join(lines))


def _randomised_field(oracle, seed, seed0, size_range, s_range):
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    rng = np.random.default_rng(seed0 + seed)
    H, W = int(rng.integers(*size_range)), int(rng.integers(*size_range))
    S = int(rng.integers(*s_range))
    f = synthetic.make_field(H, W, S, seed=seed0 + 1000 + seed, nan_fraction=float(rng.choice([0.0, 0.01, 0.05])), margin=int(rng.integers(3, 27)))
    _check_scene_constants(f, seed)
    for s_ in range(S):
        if rng.random() < 0.3:      # punch holes into some patches' bitmaps
            p = f.patches[s_][int(rng.integers(5))]
            if p.active_pixel_bitmap.size:
                p.active_pixel_bitmap &= rng.random(p.active_pixel_bitmap.shape) > 0.2
    if S > 2 and rng.random() < 0.5:  # one source loses its patch in one image: visit lists become ragged
        p = f.patches[int(rng.integers(S))][int(rng.integers(5))]
        (h0, h1), (w0, w1) = p.box
        p.box = ((h0, h0 - 1), (w0, w0 - 1))
        p.active_pixel_bitmap = np.zeros((0, 0), dtype=bool)
    psf_K = 2
    if rng.random() < 0.35:     # per-patch PSF mixtures with 1 or 3 components, off-centre, anisotropic
        from celeste_jl_amd.model import render_psf
        psf_K = int(rng.choice([1, 3]))
        for row in f.patches:
            for p in row:
                w = rng.dirichlet(np.ones(psf_K) * 4)
                p.psf = np.array([[w[k], 0.2 * rng.normal(), 0.2 * rng.normal(), (1.1 + 0.8 * k) ** 2, 0.15 * rng.normal(),
                                   (1.2 + 0.8 * k) ** 2] for k in range(psf_K)])
                p.stamp = render_psf(p.psf)
    if rng.random() < 0.35:     # affine WCS: every patch of a source maps its world position to the same pixel
        Jm = np.array([[1.0 + 0.1 * rng.normal(), 0.1 * rng.normal()], [0.1 * rng.normal(), 1.0 + 0.1 * rng.normal()]])
        Jinv = np.linalg.inv(Jm)
        for s_, row in enumerate(f.patches):
            pix = f.vp[s_, 0:2].copy()
            world = rng.normal(size=2) * 5
            f.vp[s_, 0:2] = world
            for p in row:
                p.wcs_jacobian = Jm.copy()
                p.world_center = world - Jinv @ (pix - p.pixel_center)
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors, psf_K=psf_K)
    tg = rng.permutation(S)[:int(rng.integers(max(1, s_range[0] - 1), S + 1))].tolist()
    flags = int(rng.choice([0, 4, 1, 5, 3, 7, 7, 7]))
    if rng.random() < 0.25:
        flags_dev = flags | cabi.FLAG_FP32
        g32 = ctx.eval_batch(f.vp, tg, flags_dev, raise_on_error=False)
        if g32[4].any():
            _report_unexpected_statuses(cel, f, ctx, psf_K, tg, flags_dev, g32, "fuzz %d fp32 (seed0 %d)" % (seed, seed0))
        r32 = oracle.elbo_batch(ctx.problem, f.vp, tg, flags)
        assert np.max(np.abs(g32[0] - r32[0]) / np.abs(r32[0])) <= 1e-4
        if flags & 3:
            assert max(np.abs(g32[1][k] - r32[1][k]).max() / np.abs(r32[1][k]).max() for k in range(len(tg))) <= 1e-4
    g = ctx.eval_batch(f.vp, tg, flags, raise_on_error=False)
    if g[4].any():
        _report_unexpected_statuses(cel, f, ctx, psf_K, tg, flags, g, "fuzz %d (seed0 %d)" % (seed, seed0))
    r = oracle.elbo_batch(ctx.problem, f.vp, tg, flags)
    errs = assert_parity(g, r, "fuzz %d" % seed)
    if flags & 2:
        assert_parity(ctx.eval_batch(f.vp, tg, flags | cabi.FLAG_SPLIT), r, "fuzz %d split" % seed)
    print("fuzz", seed, (H, W, S), "targets", len(tg), "flags", flags, "psf_K", psf_K, errs)
    arrays = {}
    for nm, tup in (("device", g), ("oracle", r)):
        for k, a in enumerate(tup):
            if a is not None and a.nbytes <= 4096:
                arrays["%s[%d]" % (nm, k)] = (a.ctypes.data, a.nbytes)
    for b, im in enumerate(f.images):
        arrays["iota%d" % b] = (im.nelec_per_nmgy.ctypes.data, im.nelec_per_nmgy.nbytes)
    _FUZZ_HISTORY.append({"what": "seed %d (%d x %d, %d sources, %d targets, flags %d, psf_K %d)" % (seed, H, W, S, len(tg), flags, psf_K),
                          "arrays": arrays})
    del _FUZZ_HISTORY[:-4]


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 10))
def test_single_precision_mode_counts_and_masks_like_the_fp64_path(seed):
    """CELESTE_FLAG_FP32 runs its own pixel loop (two pixels per lane, pixel_iter_px2): on fields with NaN pixels, punched
    bitmaps, a missing patch, one-column patches, tiny patches (chunks with fewer than 64 / 128 pixels) and psf_K 1 / 3 the
    pixel COUNTERS and statuses must equal the fp64 path's exactly, values / gradients / Hessians agree at the mode's 1e-4,
    and gradient-only equals the gradient of the Hessian mode"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    rng = np.random.default_rng(7000 + seed)
    H, W = int(rng.integers(50, 150)), int(rng.integers(50, 150))
    S = int(rng.integers(2, 12))
    f = synthetic.make_field(H, W, S, seed=7100 + seed, nan_fraction=float(rng.choice([0.0, 0.02, 0.08])),
                             margin=min(int(rng.integers(3, 27)), min(H, W) // 2 - 1))   # (positions are drawn in [margin, size - margin])
    for s_ in range(S):
        if rng.random() < 0.4:
            p = f.patches[s_][int(rng.integers(5))]
            if p.active_pixel_bitmap.size:
                p.active_pixel_bitmap &= rng.random(p.active_pixel_bitmap.shape) > 0.25
    if S > 2:                      # one source loses a patch (ragged visit lists), another gets one-column patches
        p = f.patches[int(rng.integers(S))][int(rng.integers(5))]
        (h0, h1), (w0, w1) = p.box
        p.box = ((h0, h0 - 1), (w0, w0 - 1))
        p.active_pixel_bitmap = np.zeros((0, 0), dtype=bool)
        q = int(rng.integers(S))
        for n in range(5):
            p = f.patches[q][n]
            if p.active_pixel_bitmap.size:
                (h0, h1), (w0, w1) = p.box
                keep = p.active_pixel_bitmap[:, :1].copy()
                p.box = ((h0, h1), (w0, w0))
                p.active_pixel_bitmap = keep
    if seed % 3 == 0:              # tiny patches: every chunk is a partial one
        for row in f.patches:
            for p in row:
                if p.active_pixel_bitmap.shape[0] > 9 and p.active_pixel_bitmap.shape[1] > 9:
                    (h0, h1), (w0, w1) = p.box
                    p.box = ((h0, h0 + 6), (w0, w0 + 7))
                    p.active_pixel_bitmap = p.active_pixel_bitmap[:7, :8].copy()
    psf_K = 2
    if seed % 4 == 1:
        from celeste_jl_amd.model import render_psf
        psf_K = int(rng.choice([1, 3]))
        for row in f.patches:
            for p in row:
                w = rng.dirichlet(np.ones(psf_K) * 4)
                p.psf = np.array([[w[k], 0.2 * rng.normal(), 0.2 * rng.normal(), (1.1 + 0.8 * k) ** 2, 0.15 * rng.normal(),
                                   (1.2 + 0.8 * k) ** 2] for k in range(psf_K)])
                p.stamp = render_psf(p.psf)
    from celeste_jl_amd.model import neighbor_map
    ctx = cel.FieldContext(f.images, f.patches, neighbor_map(f.patches), psf_K=psf_K)
    tg = rng.permutation(S).tolist()
    v64, d64, h64, c64, s64 = ctx.eval_batch(f.vp, tg, ALL)
    v32, d32, h32, c32, s32 = ctx.eval_batch(f.vp, tg, ALL | cabi.FLAG_FP32)
    assert np.array_equal(c64, c32) and np.array_equal(s64, s32) and (s64 == 0).all()
    assert c64[:, 0].sum() > 0
    ev = float(np.max(np.abs(v32 - v64) / np.abs(v64)))
    ed = max(np.abs(d32[k] - d64[k]).max() / np.abs(d64[k]).max() for k in range(len(tg)))
    eh = max(np.abs(h32[k] - h64[k]).max() / np.abs(h64[k]).max() for k in range(len(tg)))
    assert max(ev, ed, eh) <= 1e-4, (ev, ed, eh)
    assert all(np.array_equal(h32[k], h32[k].T) for k in range(len(tg)))
    vg, dg, _, cg, _ = ctx.eval_batch(f.vp, tg, 1 | 4 | cabi.FLAG_FP32)
    assert np.array_equal(cg, c64)
    assert float(np.max(np.abs(vg - v64) / np.abs(v64))) <= 1e-4
    assert max(np.abs(dg[k] - d64[k]).max() / np.abs(d64[k]).max() for k in range(len(tg))) <= 1e-4
    # a target's fp32 result does not depend on the size of the batch it is in: beyond 768 targets the neighbours' light is
    # rendered by one wavefront per item instead of two (celeste_abi.hip VALUE_WIDE_MAX) -- same per-pixel arithmetic
    # (value_pixels_f2), so the SAME targets repeated into a batch of > 768 give the same bits, masks and counters included
    reps = 800 // len(tg) + 1
    big = tg * reps
    assert len(big) > 768
    vb, db, hb, cb, sb = ctx.eval_batch(f.vp, big, ALL | cabi.FLAG_FP32)
    n = len(tg)
    for r in (0, reps // 2, reps - 1):
        sl = slice(r * n, (r + 1) * n)
        assert np.array_equal(vb[sl], v32) and np.array_equal(db[sl], d32) and np.array_equal(hb[sl], h32)
        assert np.array_equal(cb[sl], c32) and np.array_equal(sb[sl], s32)
    print("fp32 fuzz", seed, (H, W, S), "psf_K", psf_K, "errors", ev, ed, eh)


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 6))
def test_randomised_multi_active(oracle, seed):
    """fuzz of celeste_elbo_eval_multi: random crowded scene with NaNs and punched bitmaps, random active subset and
    order (Sa = 2..4), every other source a value-only neighbour"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    rng = np.random.default_rng(3000 + seed)
    S = int(rng.integers(3, 8))
    f = synthetic.make_field(int(rng.integers(50, 80)), int(rng.integers(50, 80)), S, seed=4000 + seed,
                             nan_fraction=float(rng.choice([0.0, 0.02])), margin=int(rng.integers(8, 20)))
    for s_ in range(S):
        if rng.random() < 0.4:
            p = f.patches[s_][int(rng.integers(5))]
            p.active_pixel_bitmap &= rng.random(p.active_pixel_bitmap.shape) > 0.15
    act = rng.permutation(S)[:int(rng.integers(2, min(S, 4) + 1))].tolist()
    nbrs = [[s for s in range(S) if s != a] for a in range(S)]
    ctx = cel.FieldContext(f.images, f.patches, nbrs)
    flags = int(rng.choice([7, 7, 5, 4]))
    v, d, h, cnt = ctx.eval_multi(f.vp, act, flags)
    ov, od, oh, ocnt, ost = oracle.elbo_multi(ctx.problem, f.vp, act, flags)
    assert ost == 0 and np.array_equal(cnt, ocnt), (cnt, ocnt)
    assert abs(v - ov) <= 1e-8 * abs(ov)
    if flags & 3:
        assert max(rel_err(d[:, k], od[k]) for k in range(len(act))) <= 1e-8
    if flags & 2:
        assert np.array_equal(h, h.T) and rel_err(h, oh) <= 1e-8
    print("multi fuzz", seed, "S", S, "active", act, "flags", flags, abs(v - ov) / abs(ov), rel_err(h, oh) if flags & 2 else None)


def test_multi_active_and_renderer_with_visit_lists(oracle):
    """the kernels off the hot path on a many-image problem, where the per-(source, image) tables are indexed by visit
    and a source's neighbour may be missing from an image: several active sources (cross terms between sources that
    share only some of their images) against the CPU restatement, the expected-image renderer against numpy"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    from joint_objective import expected_planes
    f = synthetic.make_multifield((2, 2), 128, 128, 0.10, 24, seed=13)
    S = len(f.catalog)
    n_img = np.array([sum(p.active_pixel_bitmap.size > 0 for p in row) for row in f.patches])
    assert n_img.min() < n_img.max() and n_img.sum() < 0.75 * S * len(f.images)   # visit lists, not the dense mode
    # active sets: two sources in different numbers of images that are neighbours, plus a third one
    pairs = [(a, b) for a in range(S) for b in f.neighbors[a] if n_img[a] != n_img[b]]
    assert pairs, "the scene has neighbours that do not share all their images"
    a, b = pairs[0]
    third = next(s for s in range(S) if s not in (a, b))
    nbrs = [[s for s in range(S) if s != t] for t in range(S)]
    ctx = cel.FieldContext(f.images, f.patches, nbrs)
    for act in ([a, b], [b, third, a]):
        v, d, h, cnt = ctx.eval_multi(f.vp, act, ALL)
        ov, od, oh, ocnt, ost = oracle.elbo_multi(ctx.problem, f.vp, act, ALL)
        assert ost == 0 and np.array_equal(cnt, ocnt)
        assert abs(v - ov) <= 1e-8 * abs(ov) and np.array_equal(h, h.T) and rel_err(h, oh) <= 1e-8
        assert max(rel_err(d[:, k], od[k]) for k in range(len(act))) <= 1e-8
    ref = expected_planes(f.images, f.patches, f.vp)
    for n in (0, 7, len(f.images) - 1):
        got = ctx.render_expected(f.vp, n)
        assert np.abs(got - ref[n]).max() <= 1e-12 * max(np.abs(ref[n]).max(), 1e-300)


def test_invalid_arguments_are_refused():
    """status codes instead of the reference's assertion failures (include/celeste_mi355x.h)"""
    import ctypes as C
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    lib = cabi.load_library()
    f = synthetic.make_sample_dataset("two_body")

    def create(mutate, **kw):
        pb = cabi.Problem(f.images, f.patches, f.neighbors, **kw)
        mutate(pb)
        h = C.c_void_p()
        st = lib.celeste_ctx_create(C.byref(pb.c), 0, C.byref(h))
        if st == 0:
            lib.celeste_ctx_destroy(h)
        return st

    assert create(lambda pb: None) == cabi.OK
    assert create(lambda pb: setattr(pb.c, "psf_K", 5)) == cabi.ERR_INVALID_ARG           # 14 K components must fit
    assert create(lambda pb: setattr(pb.c, "n_sources", 0)) == cabi.ERR_INVALID_ARG

    def patch_outside(pb):
        pb.c.patches[0].off_h = 15      # 15 + H2 > H = 20
    assert create(patch_outside) == cabi.ERR_INVALID_ARG

    def bad_stamp(pb):
        pb.c.patches[1].stamp = 99
    assert create(bad_stamp) == cabi.ERR_INVALID_ARG

    def bad_band(pb):
        pb.c.images[2].band = 6
    assert create(bad_band) == cabi.ERR_INVALID_ARG
    assert lib.celeste_ctx_create(C.byref(cabi.Problem(f.images, f.patches, f.neighbors).c), 64, C.byref(C.c_void_p())) == cabi.ERR_INVALID_ARG

    ctx = _ctx(f)
    with pytest.raises(cabi.CelesteError):
        ctx.eval_batch(f.vp, [2], ALL)            # target out of range
    with pytest.raises(cabi.CelesteError):
        ctx.eval_batch(f.vp, [-1], ALL)
    with pytest.raises(cabi.CelesteError):
        ctx.eval_multi(f.vp, [0, 0], ALL)         # duplicate active source
    with pytest.raises(cabi.CelesteError):
        ctx.maximize_batch(f.vp, [0], cel.ElboConfig(loc_width=-1.0))
    v = np.zeros(1)
    assert lib.celeste_elbo_eval_batch(ctx.handle, None, 1, None, 7, v.ctypes.data_as(cabi.c_double_p), None, None, None,
                                       None) == cabi.ERR_INVALID_ARG
    # the context survives all of the above
    assert ctx.eval_batch(f.vp, [0, 1], ALL)[4].tolist() == [0, 0]


def test_batch_equals_singles():
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(128, 128, 12, seed=5)
    ctx = _ctx(f)
    tg = list(range(12))
    v, d, h, cnt, st = ctx.eval_batch(f.vp, tg, ALL)
    for t in tg:
        v1, d1, h1, c1, s1 = ctx.eval_batch(f.vp, [t], ALL)
        assert v1[0] == v[t] and np.array_equal(d1[0], d[t]) and np.array_equal(h1[0], h[t])


def test_nonfinite_input_is_reported():
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("two_body")
    ctx = _ctx(f)
    vp = f.vp.copy(); vp[1, 7] = np.nan
    with pytest.raises(AssertionError):
        ctx.eval_batch(vp, [0], ALL)
    _, _, _, _, st = ctx.eval_batch(vp, [0], ALL, raise_on_error=False)
    assert st[0] == 2


def test_mirror_api_reads_like_the_reference(oracle):
    """elbo(ea, vp) / elbo_likelihood(ea, vp) with ElboArgs(images, patches, [active])"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("two_body")
    ea = cel.ElboArgs(f.images, f.patches, [0])
    sf = cel.elbo(ea, f.vp)
    lik = cel.elbo_likelihood(ea, f.vp)
    ov, od, oh, ocnt, _ = oracle.elbo_batch(ea._ctx.problem, f.vp, [0], ALL)
    assert abs(sf.v - ov[0]) <= 1e-8 * abs(ov[0])
    assert sf.has_hessian and sf.d.shape == (44,) and sf.h.shape == (44, 44)
    assert sf.active_pixel_counter == ocnt[0, 0] and sf.inactive_pixel_counter == ocnt[0, 1]
    # the k parameters only enter through the KL term
    assert np.all(lik.d[28:] == 0) and np.any(sf.d[28:] != 0)


def test_psf_raster(oracle, lib):
    """PSF.get_psf_at_point (PSF.jl:150-161) on the default -25:25 grid"""
    import ctypes as C
    from celeste_jl_amd import synthetic, cabi
    psf = np.ascontiguousarray(synthetic.band_psf(2))
    rows = np.arange(-25.0, 26.0); cols = np.arange(-25.0, 26.0)
    out = np.zeros(51 * 51)
    dp = cabi.c_double_p
    cabi.check(lib.celeste_psf_raster(0, psf.ctypes.data_as(dp), 2, rows.ctypes.data_as(dp), 51,
                                      cols.ctypes.data_as(dp), 51, out.ctypes.data_as(dp)))
    out = out.reshape(51, 51).T
    ref = np.array([[oracle.psf_at_point(psf, r, c) for c in cols] for r in rows])
    assert np.abs(out - ref).max() <= 1e-12 * ref.max()


def test_contexts_release_their_memory():
    """create / use / destroy many contexts (evaluation, split variant, optimiser, renderer, multi-active): device
    memory returns to where it started"""
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("free device memory is a property of the whole GPU: other test processes allocate beside this one (run without -n)")
    import gc
    import torch
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_field(160, 200, 30, seed=11)
    nbrs_all = [[s for s in range(30) if s != a] for a in range(30)]

    def cycle(i):
        ctx = cel.FieldContext(f.images, f.patches, f.neighbors if i % 2 else nbrs_all)
        tg = list(range(30))
        ctx.eval_batch(f.vp, tg, ALL)
        ctx.eval_batch(f.vp, tg, ALL | cabi.FLAG_SPLIT)
        ctx.maximize_batch(f.vp, tg[:10], cel.ElboConfig(max_iters=3))
        ctx.render_expected(f.vp, 2)
        if i % 2 == 0:
            ctx.eval_multi(f.vp, [0, 1, 2], ALL)
        ctx.close()
    cycle(0); cycle(1)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        cycle(i)
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    print("free before / after 40 context cycles: %d / %d MiB" % (free0 >> 20, free1 >> 20))
    assert free0 - free1 <= 64 << 20


def test_two_threads_two_contexts(oracle):
    """the library is thread-compatible: concurrent calls on different contexts (the reference evaluates from many
    threads, each with its own scratch, ElboMaximize.jl:146-155) give the single-threaded results"""
    from concurrent.futures import ThreadPoolExecutor
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    fields = [synthetic.make_field(120, 150, 20, seed=50 + k) for k in range(2)]
    ctxs = [_ctx(f) for f in fields]
    ref = [c.eval_batch(f.vp, list(range(20)), ALL) for c, f in zip(ctxs, fields)]
    refm = [c.maximize_batch(f.vp, list(range(20)), cel.ElboConfig(max_iters=6)) for c, f in zip(ctxs, fields)]

    def work(k):
        out = []
        for _ in range(10):
            out.append(ctxs[k].eval_batch(fields[k].vp, list(range(20)), ALL))
            out.append(ctxs[k].maximize_batch(fields[k].vp, list(range(20)), cel.ElboConfig(max_iters=6)))
        return out
    with ThreadPoolExecutor(2) as ex:
        res = list(ex.map(work, range(2)))
    for k in range(2):
        for i, r in enumerate(res[k]):
            expect = ref[k] if i % 2 == 0 else refm[k]
            for a, b in zip(r, expect):
                assert np.array_equal(a, b)
