"""Property tests mirrored from test/test_elbo.jl, run on the CPU oracle (value-only evaluations)."""
import numpy as np
import pytest

ALL = 7


def _val(oracle, pb, vp, t=0, flags=0):
    return oracle.elbo_one(pb, vp, t, flags)[0]


def test_star_truth_is_most_likely(oracle):
    """test_elbo.jl:132-170 with true_star_init (SampleData.jl:239-249)"""
    import math
    from celeste_jl_amd import synthetic, cabi, ids
    f = synthetic.make_sample_dataset("star", perturb=False)
    vp = f.vp.copy()
    vp[0, ids.is_star] = [1.0 - 1e-4, 1e-4]
    vp[0, ids.flux_scale] = 1e-4
    vp[0, ids.flux_loc] = math.log(synthetic.SAMPLE_STAR_FLUXES[2]) - 0.5 * 1e-4
    vp[0, ids.color_var] = 1e-4
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    best = _val(oracle, pb, vp)
    for bad_a in (.3, .5, .9):
        v = vp.copy(); v[0, ids.is_star] = [1 - bad_a, bad_a]
        assert best > _val(oracle, pb, v)
    for h2 in range(-2, 3):
        for w2 in range(-2, 3):
            if h2 or w2:
                v = vp.copy(); v[0, ids.pos] += [h2 * .5, w2 * .5]
                assert best > _val(oracle, pb, v)
    for delta in (.7, .9, 1.1, 1.3):
        v = vp.copy(); v[0, ids.flux_loc] += math.log(delta)
        assert best > _val(oracle, pb, v)
    for b in range(4):
        for delta in (-.3, .3):
            v = vp.copy(); v[0, ids.color_mean[b, 0]] += delta
            assert best > _val(oracle, pb, v)


def test_galaxy_truth_is_most_likely(oracle):
    """test_elbo.jl:173-220"""
    import math
    from celeste_jl_amd import synthetic, cabi, ids
    f = synthetic.make_sample_dataset("galaxy", perturb=False)
    vp = f.vp.copy(); vp[0, ids.is_star] = [0.01, .99]
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    best = _val(oracle, pb, vp)
    for bad_a in (.3, .5, .9):
        v = vp.copy(); v[0, ids.is_star] = [1 - bad_a, bad_a]
        assert best > _val(oracle, pb, v)
    for h2 in range(-2, 3):
        for w2 in range(-2, 3):
            if h2 or w2:
                v = vp.copy(); v[0, ids.pos] += [h2 * .5, w2 * .5]
                assert best > _val(oracle, pb, v)
    for bad_scale in (.8, 1.2):
        v = vp.copy(); v[0, ids.flux_loc] += 2 * math.log(bad_scale)
        assert best > _val(oracle, pb, v)
    for name in ("gal_axis_ratio", "gal_angle", "gal_radius_px"):
        for bad_scale in (.8, 1.2):
            v = vp.copy(); v[0, getattr(ids, name)] *= bad_scale
            assert best > _val(oracle, pb, v)
    for b in range(4):
        for delta in (-.3, .3):
            v = vp.copy(); v[0, ids.color_mean[b, 1]] += delta
            assert best > _val(oracle, pb, v)


def test_inactive_source_semantics(oracle):
    """test_elbo.jl:64-130: a neighbour with no active pixels contributes nothing; with a few active pixels it
    changes the value only where both cover; derivatives belong to the active source only."""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("two_body")
    for n in range(5):
        p = f.patches[0][n]
        assert p.bitmap_offset == (0, 0) and p.active_pixel_bitmap.shape == f.images[n].pixels.shape
        assert p.active_pixel_bitmap.all()
    pb_both = cabi.Problem(f.images, f.patches, f.neighbors)
    v_both, d_both, h_both, c_both, _ = oracle.elbo_one(pb_both, f.vp, 0, 3)
    for n in range(5):
        f.patches[1][n].active_pixel_bitmap[:] = False
    pb_off = cabi.Problem(f.images, f.patches, f.neighbors)
    v_off, d_off, h_off, c_off, _ = oracle.elbo_one(pb_off, f.vp, 0, 3)
    pb_alone = cabi.Problem(f.images, [f.patches[0]], [[]])
    v_alone, d_alone, h_alone, c_alone, _ = oracle.elbo_one(pb_alone, f.vp[:1], 0, 3)
    assert v_off == v_alone and np.array_equal(d_off, d_alone) and np.array_equal(h_off, h_alone)
    assert c_off[1] == 0 and c_both[1] > 0 and c_both[0] == c_off[0]
    assert v_both != v_off
    f.patches[1][4].active_pixel_bitmap[9:11, 9:11] = True
    pb_few = cabi.Problem(f.images, f.patches, f.neighbors)
    _, _, _, c_few, _ = oracle.elbo_one(pb_few, f.vp, 0, 3)
    assert c_few[1] == 4


def test_last_patch_column_contributes_nothing(oracle):
    """elbo_objective.jl:349: the counters count (pixel, source) pairs with w2 < W2"""
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("star")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    _, _, _, cnt, _ = oracle.elbo_one(pb, f.vp, 0, 0)
    expect = sum(p.active_pixel_bitmap.shape[0] * (p.active_pixel_bitmap.shape[1] - 1) for p in f.patches[0])
    assert cnt[0] == expect and cnt[1] == 0


def test_flags_are_consistent(oracle):
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("two_body")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    v7, d7, h7, _, _ = oracle.elbo_one(pb, f.vp, 0, 7)
    v3, d3, h3, _, _ = oracle.elbo_one(pb, f.vp, 0, 3)
    v0 = _val(oracle, pb, f.vp, 0, 0)
    v4 = _val(oracle, pb, f.vp, 0, 4)
    kl, kd, kh = oracle.subtract_kl(f.vp[0])
    assert v0 == v3 and v4 == v7
    assert v7 == pytest.approx(v3 + kl, rel=1e-15)
    assert np.allclose(d7, d3 + kd, rtol=1e-14, atol=0) and np.allclose(h7, h3 + kh, rtol=1e-13, atol=1e-9)
    assert np.all(d3[28:] == 0)  # k enters only through the KL term


def test_nonfinite_input_status(oracle):
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("two_body")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    vp = f.vp.copy(); vp[1, 3] = np.inf
    assert oracle.elbo_one(pb, vp, 0)[4] == cabi.ERR_NONFINITE_INPUT


# ---- several active sources (ElboArgs.active_sources with Sa > 1) ----------------------------------------------

def _two_body():
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("two_body")
    # every other source is a neighbour (ElboArgs: all S local sources contribute to any pixel they cover)
    nb = [[1], [0]]
    return f, cabi.Problem(f.images, f.patches, nb)


def test_multi_active_matches_autograd(oracle):
    """Sa = 2: value, both gradient columns and the full 88 x 88 Hessian (diagonal and cross blocks) against
    torch.autograd of the independently written joint objective"""
    from joint_objective import joint_value_grad_hess
    f, pb = _two_body()
    v, d, h, cnt, st = oracle.elbo_multi(pb, f.vp, [0, 1])
    assert st == 0
    tv, td, th = joint_value_grad_hess(f.images, f.patches, f.vp, [0, 1])
    assert abs(v - tv) <= 1e-12 * abs(tv)
    assert np.abs(d - td).max() <= 1e-10 * np.abs(td).max()
    assert np.abs(h - th).max() <= 1e-10 * np.abs(th).max()
    assert np.abs(h[:44, 44:]).max() > 0 and np.allclose(h, h.T, rtol=1e-12, atol=1e-9 * np.abs(h).max())


def test_multi_active_swap_invariance_and_single_source_blocks(oracle):
    """test_elbo.jl:107-129: swapping the two active sources permutes d and h; the diagonal blocks and gradient
    columns are those of the single-active-source evaluations; an inactive second source has no derivatives"""
    f, pb = _two_body()
    v01, d01, h01, c01, _ = oracle.elbo_multi(pb, f.vp, [0, 1])
    v10, d10, h10, c10, _ = oracle.elbo_multi(pb, f.vp, [1, 0])
    assert abs(v01 - v10) <= 1e-13 * abs(v01) and np.array_equal(c01, c10)
    assert np.allclose(d01[0], d10[1], rtol=1e-12, atol=0) and np.allclose(d01[1], d10[0], rtol=1e-12, atol=0)
    sc = np.abs(h01).max()
    assert np.abs(h01[:44, :44] - h10[44:, 44:]).max() <= 1e-12 * sc
    assert np.abs(h01[:44, 44:] - h10[44:, :44]).max() <= 1e-12 * sc
    for k, t in enumerate((0, 1)):
        v1, d1, h1, _, _ = oracle.elbo_one(pb, f.vp, t)
        assert np.allclose(d1, d01[k], rtol=1e-11, atol=1e-9 * np.abs(d1).max())
        assert np.abs(h1 - h01[44 * k:44 * (k + 1), 44 * k:44 * (k + 1)]).max() <= 1e-11 * np.abs(h1).max()
    # the joint value visits the union of the two patches once
    from joint_objective import joint_objective
    from torch_value_model import neg_kl, load_prior
    import torch
    kl = sum(float(neg_kl(torch.tensor(f.vp[t]), load_prior())) for t in (0, 1))
    assert v01 == pytest.approx(joint_objective(f.images, f.patches, f.vp, {0, 1}) + kl, rel=1e-12)


def test_sparse_patch_list_equals_dense_table(oracle):
    """celeste_problem_t's sparse patch list (model.PatchRow rows) describes the same problem as the dense table:
    same patches, neighbours and, through the CPU restatement, the same numbers"""
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.model import PatchRow
    fd = synthetic.make_multifield((2, 2), 128, 128, 0.10, 30, seed=5)
    fs = synthetic.make_multifield((2, 2), 128, 128, 0.10, 30, seed=5, sparse=True)
    assert all(isinstance(r, PatchRow) for r in fs.patches) and fs.neighbors == fd.neighbors
    assert np.array_equal(fs.vp, fd.vp)
    for rd, rs in zip(fd.patches, fs.patches):
        assert len(rs) == len(rd) == 20
        for n in range(20):
            assert rs[n].active_pixel_bitmap.shape == rd[n].active_pixel_bitmap.shape or rd[n].active_pixel_bitmap.size == 0
            if rd[n].active_pixel_bitmap.size:
                assert rs[n].box == rd[n].box and rs[n].bitmap_offset == rd[n].bitmap_offset
    pd, ps = cabi.Problem(fd.images, fd.patches, fd.neighbors), cabi.Problem(fs.images, fs.patches, fs.neighbors)
    assert ps.sparse and not pd.sparse
    assert 0 < ps.c.n_patch_entries < 0.6 * 30 * 20 and pd.c.n_patch_entries == 0
    tg = list(range(30))
    a, b = oracle.elbo_batch(pd, fd.vp, tg, 7), oracle.elbo_batch(ps, fs.vp, tg, 7)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.array_equal(oracle.reduced_elbo_batch(pd, fd.vp, tg, 7)[2], oracle.reduced_elbo_batch(ps, fs.vp, tg, 7)[2])
