"""Host-side input preparation: boxes, patches, neighbours, initialisers, spline prefilter, sharding."""
import math

import numpy as np
import pytest


def test_boxes_overlap_known_answers():
    """test/test_imaged_sources.jl:7-12"""
    from celeste_jl_amd.model import boxes_overlap
    assert boxes_overlap(((1, 10), (1, 10)), ((5, 15), (5, 15)))
    assert not boxes_overlap(((1, 10), (1, 10)), ((11, 15), (5, 15)))
    assert not boxes_overlap(((1, 10), (1, 10)), ((5, 15), (11, 15)))
    assert boxes_overlap(((1, 10), (1, 10)), ((10, 15), (10, 15)))
    assert not boxes_overlap(((1, 0), (1, 10)), ((1, 10), (1, 10)))  # empty range (off-image patch)


def test_sdss_background_known_answers():
    """test/test_sdssio.jl:12-40 (data-free known answers of the bilinear sky)"""
    from celeste_jl_amd.model import SDSSBackground
    small = np.array([[1., 2., 3., 4.], [5., 6., 7., 8.], [9., 10., 11., 12.]])
    sky = SDSSBackground(small, [0.1, 2.5], [0.5, 2.5, 4.], np.ones(2))
    assert sky.shape == (2, 3)
    expect = {(1, 1): 1.0, (2, 1): 7.0, (1, 2): 2.5, (2, 2): 8.5, (1, 3): 4.0, (2, 3): 10.0}
    for ij, v in expect.items():
        assert sky[ij] == pytest.approx(v, rel=1e-6)
    plane = sky.materialize()
    assert plane.dtype == np.float32 and plane[1, 1] == pytest.approx(8.5)
    oob = SDSSBackground(small, [-5.0, 4.0], [-4.0, 5.0], np.ones(2))
    assert (oob[1, 1], oob[1, 2], oob[2, 1], oob[2, 2]) == (1.0, 4.0, 9.0, 12.0)
    cal = SDSSBackground(small, [0.1, 2.5], [0.5, 2.5, 4.], [2.0, 0.5])
    assert cal[2, 2] == pytest.approx(4.25)


def test_sdss_psf_map_polynomial_weights():
    """SDSSIO.jl:262-299 restated as an explicit loop; a map with only the constant coefficient is position independent"""
    from celeste_jl_amd.model import SDSSPSFMap
    rng = np.random.default_rng(5)
    nk, nr, nc = 3, 51, 51
    rrows = rng.normal(size=(nr * nc, nk)); cmat = rng.normal(size=(4, 5, nk))
    m = SDSSPSFMap(rrows, nr, nc, cmat)
    x, y = 812.3, 1403.9
    stamp = np.zeros(nr * nc)
    for k in range(nk):
        w = 0.0
        for j in range(5):
            for i in range(4):
                w += cmat[i, j, k] * (0.001 * (x - 1.0)) ** i * (0.001 * (y - 1.0)) ** j
        stamp += w * rrows[:, k]
    got = m(x, y)
    assert got.shape == (nr, nc)
    assert np.allclose(got, stamp.reshape(nc, nr).T, rtol=1e-13, atol=1e-13)
    c0 = np.zeros((4, 5, nk)); c0[0, 0] = [1.0, 0.5, -0.25]
    const = SDSSPSFMap(rrows, nr, nc, c0)
    assert np.allclose(const(1, 1), const(2000, 1400))
    assert np.allclose(const(1, 1).T.ravel(), rrows @ c0[0, 0])


def test_clamp_box_and_rounding():
    from celeste_jl_amd.model import clamp_box, julia_round
    assert clamp_box(((-3, 7), (20, 40)), (20, 23)) == ((1, 7), (20, 23))
    assert clamp_box(((25, 30), (1, 5)), (20, 23)) == ((21, 20), (1, 5))  # empty but legal
    assert [julia_round(x) for x in (0.5, 1.5, 2.5, -0.5, 2.4999, 2.5001)] == [0, 2, 2, 0, 2, 3]


def test_catalog_init_source_and_generic():
    """DeterministicVI.jl:39-91"""
    from celeste_jl_amd import catalog_init_source, generic_init_source, ids
    from celeste_jl_amd.synthetic import sample_ce
    g = generic_init_source([3.0, 4.0])
    assert list(g[ids.pos]) == [3.0, 4.0] and g[ids.gal_radius_px] == 1.0 and np.all(g[ids.k] == 1 / 8)
    assert g.shape == (44,) and np.all(g[ids.color_var] == 1e-2) and np.all(g[ids.flux_loc] == math.log(2.0))
    star = catalog_init_source(sample_ce([10.1, 12.2], True))
    assert list(star[ids.is_star]) == [0.8, 0.2] and star[ids.gal_radius_px] == 0.2 and star[ids.gal_axis_ratio] == .8
    gal = catalog_init_source(sample_ce([8.5, 9.6], False))
    assert list(gal[ids.is_star]) == [0.2, 0.8] and gal[ids.gal_radius_px] == 4.0 and gal[ids.gal_frac_dev] == 0.1
    fl = sample_ce([0, 0], True).star_fluxes
    assert star[ids.flux_loc[0]] == pytest.approx(math.log(fl[2]))
    assert star[ids.color_mean[:, 0]] == pytest.approx([math.log(fl[c + 1] / fl[c]) for c in range(4)])


def test_patches_and_neighbors_match_brute_force():
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.model import find_neighbors, neighbor_map, choose_patch_radius
    f = synthetic.make_field(200, 240, 30, seed=4)
    nm = neighbor_map(f.patches)
    for s in range(30):
        assert nm[s] == find_neighbors(f.patches, s)
        for n, p in enumerate(f.patches[s]):
            r = choose_patch_radius(f.catalog[s], f.images[n], width_scale=1.2)
            assert 0 < r <= 25
            (h0, h1), (w0, w1) = p.box
            assert p.bitmap_offset == (h0 - 1, w0 - 1)
            assert p.active_pixel_bitmap.shape == (h1 - h0 + 1, w1 - w0 + 1)
            assert p.pixel_center[0] == (h0 + h1) / 2 and p.pixel_center[1] == (w0 + w1) / 2
    assert nm == f.neighbors


def test_spline_prefilter_matches_oracle_and_interpolates(lib, oracle):
    """product prefilter (Thomas algorithm) vs oracle (dense elimination): independent implementations"""
    from celeste_jl_amd import synthetic, cabi
    rng = np.random.default_rng(1)
    for stamp in (synthetic.render_psf(synthetic.band_psf(0)), rng.uniform(-0.1, 1.0, (51, 51))):
        a = cabi.spline_prefilter(stamp)
        b = oracle.spline_coefs(stamp)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
        # natural boundary rows: c0 - 2 c1 + c2 = 0 along both axes
        assert np.abs(a[0] - 2 * a[1] + a[2]).max() < 1e-12 and np.abs(a[:, -1] - 2 * a[:, -2] + a[:, -3]).max() < 1e-12


def test_render_psf_equals_get_psf_at_point(oracle):
    """psf_model.jl:61-75 and PSF.jl:150-161 are the same raster"""
    from celeste_jl_amd import synthetic
    psf = synthetic.band_psf(3)
    st = synthetic.render_psf(psf)
    for (i, j) in [(26, 26), (1, 1), (30, 20), (51, 2)]:
        assert st[i - 1, j - 1] == pytest.approx(oracle.psf_at_point(psf, i - 26.0, j - 26.0), rel=1e-13)


def test_problem_marshalling_layout():
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("two_body")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    assert pb.c.n_images == 5 and pb.c.n_sources == 2 and pb.c.n_stamps == 4  # bands 3 and 5 share a PSF
    im = pb.c.images[2]
    assert (im.H, im.W, im.band) == (20, 23, 3)
    # column-major: element [h, w] at h + H * w
    assert im.pixels[3 + 20 * 7] == f.images[2].pixels[3, 7]
    p = pb.c.patches[1 * 5 + 2]
    assert (p.off_h, p.off_w) == f.patches[1][2].bitmap_offset and not p.bitmap  # bitmap == !isnan -> NULL
    assert list(p.wcs_jacobian) == [1.0, 0.0, 0.0, 1.0]
    assert list(pb.nbr_off) == [0, 1, 2] and list(pb.nbr_idx[:2]) == [1, 0]


def test_synthetic_field_is_deterministic_and_poisson_like():
    from celeste_jl_amd import synthetic
    a = synthetic.make_field(96, 128, 6, seed=9)
    b = synthetic.make_field(96, 128, 6, seed=9)
    assert all(np.array_equal(x.pixels, y.pixels) for x, y in zip(a.images, b.images))
    assert np.array_equal(a.vp, b.vp)
    img = a.images[2]
    assert img.pixels.dtype == np.float32 and np.all(img.pixels >= 0) and np.all(img.pixels == np.round(img.pixels))
    # far from any source the counts are Poisson(sky * iota)
    assert abs(np.median(img.pixels) - 0.60 * 820) < 30


def test_sharding_is_balanced_and_complete():
    """load_balance_across_threads analogue (ParallelRun.jl:49-56)"""
    from celeste_jl_amd.partition import shard_targets, estimate_time, load_balance
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(200, 240, 40, seed=4)
    costs = [estimate_time(row) for row in f.patches]
    for world in (1, 2, 3, 8):
        shards = shard_targets(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(40))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)
    assert load_balance(2, [3, 1, 1, 1]) == [3.0, 3.0]


def test_cyclades_covers_all_sources_without_conflicts():
    """test/test_partition.jl:56-92"""
    from celeste_jl_amd.partition import partition_cyclades_dynamic
    rng = np.random.default_rng(0)
    target_sources = list(range(6, 22))
    neighbor_map = {s: [] for s in target_sources}
    for a, b in [(6, 7), (7, 8), (10, 15), (11, 12), (12, 13), (13, 14), (16, 21), (18, 19)]:
        neighbor_map[a].append(b); neighbor_map[b].append(a)
    for it in range(20):
        batches = partition_cyclades_dynamic(target_sources, neighbor_map, batch_size=4, rng=rng)
        assert sorted(i for b in batches for c in b for i in c) == list(range(16))
        for b in batches:
            for x in range(len(b)):
                for y in range(x + 1, len(b)):
                    for i in b[x]:
                        for j in b[y]:
                            assert target_sources[j] not in neighbor_map[target_sources[i]]


def test_color_classes_are_independent_sets():
    from celeste_jl_amd.partition import color_classes
    rng = np.random.default_rng(3)
    n = 200
    nbrs = {s: set() for s in range(n)}
    for _ in range(500):
        a, b = rng.integers(n, size=2)
        if a != b:
            nbrs[int(a)].add(int(b)); nbrs[int(b)].add(int(a))
    targets = list(range(n))
    classes = color_classes(targets, {s: sorted(v) for s, v in nbrs.items()})
    assert sorted(i for c in classes for i in c) == targets
    assert len(classes) <= max(len(v) for v in nbrs.values()) + 1
    for c in classes:
        cs = set(c)
        assert all(not (nbrs[s] & cs) for s in c)


def test_bounding_box_and_bad_sky():
    """dataset.jl:1-13; ParallelRun.jl:437-460"""
    from celeste_jl_amd import BoundingBox, synthetic
    from celeste_jl_amd.infer import bad_sky
    with pytest.raises(AssertionError):
        BoundingBox(1.0, 1.0, 0.0, 2.0)
    b = BoundingBox(10.0, 20.0, 5.0, 9.0)
    assert b.contains([15.0, 6.0]) and not b.contains([10.0, 6.0]) and not b.contains([15.0, 9.5])
    f = synthetic.make_sample_dataset("three_body")
    assert bad_sky(f.catalog[1], f.images)                 # a bright star fills the 101 x 101 box: median well above the sky
    ce = synthetic.sample_ce([30.0, 185.0], True)          # an empty corner of the 112 x 238 scene
    assert not bad_sky(ce, f.images)                       # Poisson noise around the claimed sky: median is not 5 e- above
    img = next(im for im in f.images if im.b == 4)
    img.pixels += np.float32(12.0)                          # 12 extra photo-electrons everywhere
    assert bad_sky(ce, f.images)
    assert not bad_sky(ce, [im for im in f.images if im.b != 4])


def test_variational_parameters_to_catalog_row():
    """AccuracyBenchmark.jl:150-162, 325-387: the catalog row of catalog_init_source(ce) gives the entry back"""
    import math
    from celeste_jl_amd import synthetic, catalog_init_source
    from celeste_jl_amd.catalog import (variational_parameters_to_row, celeste_to_rows, canonical_angle, color_from_fluxes,
                                        fluxes_from_colors, COLUMNS)
    from celeste_jl_amd.infer import OptimizedSource
    assert canonical_angle(190.0) == 10.0 and canonical_angle(-10.0) == 170.0 and canonical_angle(45.0) == 45.0
    assert color_from_fluxes(2.0, 4.0) == pytest.approx(math.log(2.0)) and color_from_fluxes(0.0, 1.0) is None
    fl = fluxes_from_colors(10.0, [0.1, 0.2, 0.3, 0.4])
    assert fl[2] == 10.0 and math.log(fl[1] / fl[0]) == pytest.approx(0.1) and math.log(fl[4] / fl[3]) == pytest.approx(0.4)
    for is_star in (True, False):
        ce = synthetic.sample_ce([10.1, 12.2], is_star)
        row = variational_parameters_to_row(catalog_init_source(ce))
        assert list(row) == COLUMNS
        f = ce.star_fluxes if is_star else ce.gal_fluxes
        assert (row["ra"], row["dec"]) == (10.1, 12.2) and row["is_star"] == (0.8 if is_star else 0.2)
        assert row["flux_r_nmgy"] == pytest.approx(f[2], rel=1e-12)
        for k, name in enumerate(("color_ug", "color_gr", "color_ri", "color_iz")):
            assert row[name] == pytest.approx(math.log(f[k + 1] / f[k]), rel=1e-12)
        if not is_star:
            assert row["gal_axis_ratio"] == 0.7 and row["gal_radius_px"] == pytest.approx(4.0 * math.sqrt(0.7))
            assert row["gal_angle_deg"] == pytest.approx(45.0) and row["gal_frac_dev"] == 0.1
    vs = catalog_init_source(synthetic.sample_ce([1.0, 2.0], True))
    res = [OptimizedSource(1.0, 2.0, vs, False), OptimizedSource(1.0, 2.0, vs, True)]
    assert len(celeste_to_rows(res)) == 1


def test_scoring_against_truth():
    """AccuracyBenchmark.jl:140-148 (asinh magnitudes round trip), :801-804, :813-931"""
    import math
    from celeste_jl_amd import synthetic, catalog_init_source
    from celeste_jl_amd.catalog import (flux_to_mag, mag_to_flux, degrees_to_diff, catalog_entry_to_row, get_error_row,
                                        variational_parameters_to_row, score_predictions, is_good_row)
    for band in range(1, 6):
        for flux in (0.05, 1.0, 250.0):
            assert mag_to_flux(flux_to_mag(flux, band), band) == pytest.approx(flux, rel=1e-9)
    assert flux_to_mag(1.0, 3) == pytest.approx(22.5, abs=0.02)          # 1 nanomaggy ~ 22.5 mag (asinh softening)
    assert flux_to_mag(100.0, 3) == pytest.approx(17.5, abs=1e-5)
    assert degrees_to_diff(10.0, 170.0) == 20.0 and degrees_to_diff(45.0, 225.0) == 0.0
    truth, pred = [], []
    for k, is_star in enumerate((True, False, True, False)):
        ce = synthetic.sample_ce([10.0 + k, 20.0], is_star)
        if not is_star:
            ce.gal_frac_dev = 0.99; ce.gal_axis_ratio = 0.4
        truth.append(catalog_entry_to_row(ce))
        pred.append(variational_parameters_to_row(catalog_init_source(ce)))
    e = get_error_row(truth[0], pred[0])
    assert e["missed_stars"] == 0.0 and e["missed_galaxies"] is None and e["position"] == 0.0 and e["flux_r_mag"] < 1e-9
    sc = score_predictions(truth, pred)
    assert sc["flux_r_nmgy"]["N"] == 4 and sc["flux_r_nmgy"]["first"] < 1e-9 and sc["color_gr"]["first"] < 1e-9
    assert sc["missed_stars"] == {"N": 2, "first": 0.0} and sc["gal_angle_deg"]["N"] == 2   # star rows carry gal_frac_dev = 0.1: filtered
    wrong = dict(pred[1]); wrong["is_star"] = 0.9; wrong["flux_r_nmgy"] *= 2
    sc2 = score_predictions(truth, [pred[0], wrong, pred[2], pred[3]])
    assert sc2["missed_galaxies"]["first"] == 0.5 and sc2["flux_r_mag"]["first"] == pytest.approx(2.5 * math.log10(2) / 4, rel=1e-3)
    mixed = dict(truth[1]); mixed["gal_frac_dev"] = 0.5
    assert not is_good_row(mixed, get_error_row(mixed, pred[1]), "gal_radius_px")


def test_fit_raw_psf_recovers_a_two_component_mixture():
    """PSF.fit_raw_psf_for_celeste (PSF.jl:635-673): fitting the rendered stamp of a known mixture reproduces it"""
    from celeste_jl_amd.model import make_psf, render_psf
    from celeste_jl_amd.psf import fit_raw_psf_for_celeste, trim_psf
    true = make_psf([0.75, 0.25], [[0.05, -0.03], [-0.1, 0.08]],
                    [np.array([[1.6, 0.15], [0.15, 1.4]]), np.array([[7.5, -0.6], [-0.6, 6.0]])])
    stamp = render_psf(true, (51, 51))
    psf, params = fit_raw_psf_for_celeste(stamp, 2)
    assert np.abs(render_psf(psf, (51, 51)) - stamp).max() <= 1e-7 * stamp.max()
    order = np.argsort(psf[:, 3])
    assert np.allclose(psf[order], true[np.argsort(true[:, 3])], rtol=2e-4, atol=2e-4)
    assert np.all((params[:, 2] >= 0.1) & (params[:, 2] <= 1.0) & (params[:, 5] >= 0.05))
    t = trim_psf(stamp)
    assert t.shape[0] == t.shape[1] and t.shape[0] % 2 == 1 and t.shape[0] < 51
    assert np.abs(t).sum() >= 0.999 * np.abs(stamp).sum() and t[t.shape[0] // 2, t.shape[1] // 2] == stamp[25, 25]


def test_reference_known_answers_of_the_catalog_helpers():
    """the literal known answers of test/test_accuracy_benchmarks.jl:47-97 (flux <-> asinh magnitudes from sdsspy's
    nmgy2lups, colours, angle canonicalisation, one variational-parameter vector -> catalog row)"""
    import math
    from celeste_jl_amd import generic_init_source, ids
    from celeste_jl_amd.catalog import (flux_to_mag, mag_to_flux, color_from_fluxes, fluxes_from_colors, canonical_angle,
                                        degrees_to_diff, variational_parameters_to_row)
    assert flux_to_mag(15.0, 1) == pytest.approx(19.559677, abs=1e-5)
    assert flux_to_mag(15.0, 3) == pytest.approx(19.559702, abs=1e-5)
    assert mag_to_flux(19.559677, 1) == pytest.approx(15.0, abs=1e-5)
    assert mag_to_flux(19.559702, 3) == pytest.approx(15.0, abs=1e-5)
    assert color_from_fluxes(15.0, 20.0) == pytest.approx(math.log(20 / 15))
    assert color_from_fluxes(15.0, 0.0) is None
    fl = fluxes_from_colors(10.0, [-1.0, 0.0, 1.0, 2.0])
    assert np.allclose(fl, [math.e * 10, 10.0, 10.0, math.e * 10, math.exp(3.0) * 10], rtol=1e-12)
    assert canonical_angle(95.0) == 95.0 and canonical_angle(195.0) == 15.0 and canonical_angle(-20.0) == 160.0
    assert degrees_to_diff(20.0, -30.0) == pytest.approx(50.0) and degrees_to_diff(-10.0, 190.0) == pytest.approx(20.0)
    vs = generic_init_source([1.0, 2.0])
    vs[ids.gal_axis_ratio] = 0.5
    vs[ids.gal_radius_px] = 10.0
    vs[ids.gal_angle] = -math.pi / 4
    vs[ids.flux_loc[1]] = math.log(20.0)
    vs[ids.is_star[0]] = 0.01
    vs[ids.is_star[1]] = 0.99
    row = variational_parameters_to_row(vs)
    assert row["gal_radius_px"] == pytest.approx(10 * math.sqrt(0.5))
    assert row["gal_angle_deg"] == pytest.approx(135.0)
    assert row["flux_r_nmgy"] == pytest.approx(20.0)


def test_parallel_image_generation_is_deterministic():
    """gen_images with worker processes (many-image problems): image n is sampled from PCG64([seed, n]), so the pixels
    do not depend on the number of workers"""
    from celeste_jl_amd import synthetic
    a = synthetic.make_multifield((1, 2), 90, 90, 0.10, 12, seed=8, sparse=True, workers=2)
    b = synthetic.make_multifield((1, 2), 90, 90, 0.10, 12, seed=8, sparse=True, workers=3)
    assert all(np.array_equal(x.pixels, y.pixels) for x, y in zip(a.images, b.images))
    assert all(x.pixels.dtype == np.float32 and np.isfinite(x.pixels).all() for x in a.images)
    assert np.array_equal(a.vp, b.vp) and a.neighbors == b.neighbors


def test_patch_table_is_get_sky_patches_without_the_objects():
    """model.patch_table / cabi.problem_from_table (the fast host path of infer_box) marshal byte for byte the
    celeste_problem_t that get_sky_patches + neighbor_map + cabi.Problem build: dense field with NaN pixels and sources
    off the image, and a many-image problem through the sparse patch list"""
    import ctypes as C
    from celeste_jl_amd import model, cabi, synthetic
    from celeste_jl_amd.partition import estimate_time

    def same(pa, pb, n):
        a = np.frombuffer(C.string_at(C.addressof(pa.c_patches), n * C.sizeof(cabi.PatchT)), dtype=cabi.PATCH_DTYPE)
        b = np.frombuffer(C.string_at(C.addressof(pb.c_patches), n * C.sizeof(cabi.PatchT)), dtype=cabi.PATCH_DTYPE)
        for f in cabi.PATCH_DTYPE.names:
            if f != "psf":      # a pointer
                assert np.array_equal(a[f], b[f]), f
        assert np.array_equal(pa.stamps, pb.stamps) and np.array_equal(pa.nbr_off, pb.nbr_off)
        assert np.array_equal(pa.nbr_idx, pb.nbr_idx) and pa.c.n_stamps == pb.c.n_stamps
    assert cabi.PATCH_DTYPE.itemsize == C.sizeof(cabi.PatchT)
    f = synthetic.make_field(150, 170, 25, seed=5, nan_fraction=0.02, margin=4)
    cat = list(f.catalog) + [synthetic.sample_ce([900.0, -40.0], False), synthetic.sample_ce([-3.0, 80.0], True)]
    patches = model.get_sky_patches(f.images, cat)
    tab = model.patch_table(f.images, cat)
    assert tab.dense and tab.neighbors() == model.neighbor_map(patches)
    assert np.array_equal(tab.costs(), [estimate_time(r) for r in patches])
    same(cabi.problem_from_table(f.images, tab, tab.neighbors(), marshal_images=False),
         cabi.Problem(f.images, patches, model.neighbor_map(patches), marshal_images=False), len(cat) * len(f.images))
    m = synthetic.make_multifield((2, 2), 96, 96, 0.10, 30, seed=3, sparse=True)
    tab = model.patch_table(m.images, m.catalog, sparse=True)
    po = cabi.Problem(m.images, m.patches, m.neighbors, marshal_images=False)
    pf = cabi.problem_from_table(m.images, tab, tab.neighbors(), marshal_images=False)
    assert not tab.dense and pf.sparse and tab.neighbors() == [list(r) for r in m.neighbors]
    assert np.array_equal(pf.patch_source, po.patch_source) and np.array_equal(pf.patch_image, po.patch_image)
    same(pf, po, len(tab.source))
    assert np.array_equal(tab.costs(), [estimate_time(r) for r in m.patches])
    # a fixed radius (the tests' radius_override_pix) goes through the same code
    t2 = model.patch_table(f.images, f.catalog, radius_override_pix=6.0)
    p2 = model.get_sky_patches(f.images, f.catalog, radius_override_pix=6.0)
    assert all(p2[s][n].box == ((b[0], b[1]), (b[2], b[3])) for (s, n, b) in zip(t2.source, t2.image, t2.box))


def test_init_source_table_equals_the_per_entry_functions():
    """params.init_source_table (the table of setup_vecs in one array) against catalog_init_source / generic_init_source"""
    import numpy as np
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.params import catalog_init_source, generic_init_source, init_source_table
    f = synthetic.make_field(120, 130, 25, seed=4)
    cat = f.catalog
    ref = np.stack([catalog_init_source(ce) for ce in cat])
    assert np.array_equal(ref, init_source_table(cat))
    tg = [t for t in range(len(cat)) if t % 3 != 1]
    for t in tg:
        ref[t] = generic_init_source(cat[t].pos)
    assert np.array_equal(ref, init_source_table(cat, tg))
    assert init_source_table([], []).shape == (0, 44)
    # a NaN flux / shape propagates as Julia's max / min propagate it (DeterministicVI.jl:68-69: log(max(0.1, NaN)) is NaN; Python's
    # built-in max(0.1, nan) would give 0.1), identically in both routes
    import copy
    bad = copy.deepcopy(cat[:4])
    bad[0].star_fluxes = np.array(bad[0].star_fluxes, dtype=float); bad[0].star_fluxes[2] = np.nan
    bad[1].gal_radius_px = float("nan")
    bad[2].gal_fluxes = np.array(bad[2].gal_fluxes, dtype=float); bad[2].gal_fluxes[1] = np.nan
    bad[1].is_star = False
    one = np.stack([catalog_init_source(ce) for ce in bad])
    tab = init_source_table(bad)
    assert np.array_equal(one, tab, equal_nan=True)
    from celeste_jl_amd.params import ids
    assert np.isnan(one[0, ids.flux_loc[0]]) and np.isnan(one[1, ids.gal_radius_px]) and np.isfinite(one[3]).all()


def test_bad_sky_flags_equal_the_per_entry_check():
    """infer.bad_sky_flags (all boxes gathered and sorted at once) against bad_sky (ParallelRun.jl:437-460), with clipped
    boxes, NaN pixels and both outcomes present"""
    import numpy as np
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.infer import bad_sky, bad_sky_flags
    f = synthetic.make_field(260, 300, 60, seed=9, margin=2)
    img = next(im for im in f.images if im.b == 4)
    img.pixels[40:60, 100:130] = np.nan
    img.pixels[150:, :] += 30
    ref = [bad_sky(ce, f.images) for ce in f.catalog]
    assert 0 < sum(ref) < len(ref)
    assert bad_sky_flags(f.catalog, f.images, force_torch=True) == ref
    assert bad_sky_flags(f.catalog[:3], f.images) == ref[:3]
    assert bad_sky_flags([], f.images) == []
