"""-m gpu: maximize! on the device (celeste_maximize_batch) against the CPU restatement of the same algorithm
and against the reference's recovery tolerances (test/test_optimization.jl)."""
import os

import numpy as np
import pytest

from test_oracle_optimizer import _verify_sample_galaxy

# CELESTE_FUZZ_SEEDS=N: every seeded fuzz test with N seeds instead of its default handful (a long run on a GPU box)
FUZZ_SEEDS = int(os.environ.get("CELESTE_FUZZ_SEEDS", "0"))
pytestmark = pytest.mark.gpu


def _ctx(f):
    import celeste_jl_amd as cel
    return cel.FieldContext(f.images, f.patches, f.neighbors)


def test_galaxy_optimization_matches_reference_tolerances_and_oracle(oracle):
    """test_optimization.jl:54-59"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("galaxy")
    ctx = _ctx(f)
    cfg = cel.ElboConfig(loc_width=3.0)
    vp, its, evals, elbo, st = ctx.maximize_batch(f.vp, [0], cfg, include_kl=False)
    assert st[0] == 0 and evals[0] == its[0] + 1
    _verify_sample_galaxy(vp[0], [8.5, 9.6])
    ovp, oit, oev, oelbo, ost = oracle.maximize(ctx.problem, f.vp, 0, oracle.OptCfg(loc_width=3.0, include_kl=False))
    print("gpu", its[0], elbo[0], "oracle", oit, oelbo)
    assert abs(elbo[0] - oelbo) <= 1e-6 * abs(oelbo)
    # same deterministic algorithm on both sides: the optima agree far inside the reference's tolerances
    assert np.abs(vp[0, :28] - ovp[0, :28]).max() <= 1e-3


def test_full_elbo_optimization(oracle):
    """test_optimization.jl:62-68 (KL on, loc_width 1.0, x_tol 0)"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("galaxy")
    ea = cel.ElboArgs(f.images, f.patches, [0])
    vp = f.vp.copy()
    evals, best, vp = cel.maximize(ea, vp, cel.ElboConfig(loc_width=1.0, xtol_abs=0.0))
    _verify_sample_galaxy(vp[0], [8.5, 9.6])
    assert best > cel.elbo(ea, f.vp).v
    assert cel.elbo(ea, vp).v == pytest.approx(best, rel=1e-10)


def test_only_the_active_source_moves():
    """test_optimization.jl:36-51"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_sample_dataset("three_body")
    ea = cel.ElboArgs(f.images, f.patches, [1], include_kl=False)
    vp = f.vp.copy()
    cel.maximize(ea, vp, cel.ElboConfig(loc_width=1.0))
    assert not np.array_equal(vp[1], f.vp[1])
    assert np.array_equal(vp[0], f.vp[0]) and np.array_equal(vp[2], f.vp[2])


def test_batch_with_frozen_neighbours_matches_oracle(oracle):
    """every target of a crowded field at once; each one sees its neighbours at their input values
    (ParallelRun.process_source, ParallelRun.jl:468-498) exactly as the per-target CPU optimiser does"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(160, 200, 24, seed=11)
    ctx = _ctx(f)
    tg = list(range(24))
    cfg = cel.ElboConfig(max_iters=12)
    vp, its, evals, elbo, st = ctx.maximize_batch(f.vp, tg, cfg)
    assert (st == 0).all()
    v0 = ctx.eval_batch(f.vp, tg, 4)[0]
    assert np.all(elbo > v0)
    worst = 0.0
    for t in (0, 5, 11, 17, 23):
        ovp, oit, oev, oelbo, ost = oracle.maximize(ctx.problem, f.vp, t, oracle.OptCfg(max_iters=12))
        assert ost == 0
        assert abs(elbo[t] - oelbo) <= 1e-7 * abs(oelbo), (t, elbo[t], oelbo, its[t], oit)
        worst = max(worst, np.abs(vp[t] - ovp[t]).max())
    print("max |vp_gpu - vp_oracle| after 12 iterations:", worst)
    assert worst <= 1e-4
    # sources that were not targets keep their parameters
    vp2, *_ = ctx.maximize_batch(f.vp, [3], cfg)
    assert np.array_equal(np.delete(vp2, 3, axis=0), np.delete(f.vp, 3, axis=0))


def test_tridiagonal_solver_agrees_with_eigen_solver(monkeypatch):
    """the default trust-region solve (Householder tridiagonalisation + O(n) secular solves) takes the same steps
    as the full eigen-decomposition Optim.jl's NewtonTrustRegion uses (CELESTE_TR_SOLVER=eig)"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(200, 240, 60, seed=4)
    ctx = _ctx(f)
    tg = list(range(60))
    for iters in (6, 50):
        cfg = cel.ElboConfig(max_iters=iters)
        monkeypatch.delenv("CELESTE_TR_SOLVER", raising=False)
        vp_t, its_t, ev_t, el_t, st_t = ctx.maximize_batch(f.vp, tg, cfg)
        monkeypatch.setenv("CELESTE_TR_SOLVER", "eig")
        vp_e, its_e, ev_e, el_e, st_e = ctx.maximize_batch(f.vp, tg, cfg)
        monkeypatch.delenv("CELESTE_TR_SOLVER", raising=False)
        assert (st_t == 0).all() and (st_e == 0).all()
        rel = np.abs(el_t - el_e) / np.abs(el_e)
        print("iters", iters, "max rel elbo diff", rel.max(), "iteration count differs for", int((its_t != its_e).sum()),
              "max |dvp|", np.abs(vp_t - vp_e).max())
        if iters == 6:
            assert np.array_equal(its_t, its_e)
            assert rel.max() <= 1e-9 and np.abs(vp_t - vp_e).max() <= 1e-6
        else:
            # converged optima.  Hard-case steps run along the lowest eigenvector, which neither solver determines
            # inside a cluster of rounding-level eigenvalues (saturated parameters), so a few sources may end
            # elsewhere; the bulk must agree and neither solver may be systematically better
            assert np.median(rel) <= 1e-9 and np.quantile(rel, 0.8) <= 1e-6, (np.median(rel), np.quantile(rel, 0.8))
            assert abs(el_t.sum() - el_e.sum()) <= 2e-3 * abs(el_e.sum())


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 6))
def test_randomised_optimiser_against_cpu(oracle, seed):
    """fuzz: random crowded scenes, random targets, random iteration budget and box width; device vs CPU restatement"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    rng = np.random.default_rng(5000 + seed)
    S = int(rng.integers(2, 9))
    f = synthetic.make_field(int(rng.integers(60, 100)), int(rng.integers(60, 100)), S, seed=6000 + seed,
                             nan_fraction=float(rng.choice([0.0, 0.01])), margin=int(rng.integers(10, 25)))
    ctx = _ctx(f)
    tg = rng.permutation(S)[:int(rng.integers(1, min(S, 4) + 1))].tolist()
    iters = int(rng.choice([3, 8, 20]))
    lw = float(rng.choice([1e-4, 1.0]))
    vp, its, evals, elbo, st = ctx.maximize_batch(f.vp, tg, cel.ElboConfig(max_iters=iters, loc_width=lw))
    assert (st == 0).all()
    for k, t in enumerate(tg):
        ovp, oit, oev, oelbo, ost = oracle.maximize(ctx.problem, f.vp, t, oracle.OptCfg(max_iters=iters, loc_width=lw))
        assert ost == 0
        if iters <= 8:   # same trajectory, step by step
            assert its[k] == oit and evals[k] == oev, (t, its[k], oit)
            assert abs(elbo[k] - oelbo) <= 1e-9 * abs(oelbo), (t, elbo[k], oelbo)
            assert np.abs(vp[t] - ovp[t]).max() <= 1e-6, (t, np.abs(vp[t] - ovp[t]).max())
        else:
            # long runs may part ways at a borderline accept / hard-case decision (rounding-level differences between
            # the two eigen-solvers are amplified along flat directions) and meet again at the optimum -- as far as the
            # stopping rule takes either of them: f_tol = 1e-6 relative (ElboMaximize.jl:228-242), and a run that ends on
            # its iteration budget is not there yet.  (1e-7 held for the first handful of seeds; 4 of 2000 seeds end 6e-7
            # apart: profiles/r07_extended_fuzz.txt)
            assert abs(int(its[k]) - oit) <= 3 and abs(elbo[k] - oelbo) <= 2e-6 * abs(oelbo), (t, its[k], oit, elbo[k], oelbo)
    print("optimiser fuzz", seed, "S", S, "targets", tg, "iters", iters, "loc_width", lw, "ok")


def test_determinism_and_batch_invariance():
    """test/outofdate.jl (stale in the reference): two runs give bit-identical results, and a target's optimum does
    not depend on which other targets share its batch (single inference: neighbours frozen)"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(200, 240, 60, seed=4)
    ctx = _ctx(f)
    cfg = cel.ElboConfig(max_iters=15)
    a = ctx.maximize_batch(f.vp, list(range(60)), cfg)
    b = ctx.maximize_batch(f.vp, list(range(60)), cfg)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    sub = [41, 3, 17, 29]
    c = ctx.maximize_batch(f.vp, sub, cfg)
    assert np.array_equal(c[0][sub], a[0][sub]) and np.array_equal(c[3], a[3][sub]) and np.array_equal(c[1], a[1][sub])


def test_joint_objective_helper_matches_single_active_elbo(oracle):
    """the multi-active score (test_infer.jl:9-29) reduces to elbo_likelihood for one active source"""
    from celeste_jl_amd import synthetic, cabi
    from joint_objective import joint_objective
    f = synthetic.make_sample_dataset("two_body")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    for t in (0, 1):
        assert joint_objective(f.images, f.patches, f.vp, {t}) == pytest.approx(oracle.elbo_one(pb, f.vp, t, 0)[0], rel=1e-12)


def test_joint_beats_single_on_overlapping_sources():
    """test_infer.jl:49-70: joint (Cyclades, 3 sweeps) objective > single-source objective"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.infer import one_node_single_infer, one_node_joint_infer
    from joint_objective import joint_objective
    f = synthetic.make_field(72, 72, 7, seed=23, margin=14)
    assert sum(len(n) for n in f.neighbors) >= 6
    ctx = _ctx(f)
    tg = list(range(len(f.catalog)))
    cfg = cel.ElboConfig(loc_width=1.0)
    vs_single = one_node_single_infer(ctx, f.catalog, tg, cfg)
    vs_joint = one_node_joint_infer(ctx, f.catalog, tg, f.neighbors, cfg, batch_size=4)
    assert np.all(np.isfinite(vs_single)) and np.all(np.isfinite(vs_joint))
    s_single = joint_objective(f.images, f.patches, vs_single, set(tg))
    s_joint = joint_objective(f.images, f.patches, vs_joint, set(tg))
    print("single", s_single, "joint", s_joint)
    assert s_joint > s_single
    # the colouring schedule (one launch per colour class) is another conflict-free sweep order; being closer to a
    # Jacobi sweep it needs more sweeps on a crowded scene, but it climbs monotonically towards the same objective
    s_col = [joint_objective(f.images, f.patches,
                             one_node_joint_infer(ctx, f.catalog, tg, f.neighbors, cfg, n_iters=k, schedule="coloring"), set(tg))
             for k in (1, 3, 6)]
    print("joint, coloring schedule after 1 / 3 / 6 sweeps", s_col)
    assert s_col[0] < s_col[1] < s_col[2] and abs(s_col[2] - s_joint) <= 2e-3 * abs(s_joint)


def test_infer_box_targets_inside_the_box_only():
    """ParallelRun.infer_box (:610-672): entries strictly inside the box are optimised, the others only lend light"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.infer import one_node_single_infer
    f = synthetic.make_field(120, 140, 12, seed=31, margin=12)
    box = cel.BoundingBox(20.0, 90.0, 15.0, 110.0)
    inside = [i for i, ce in enumerate(f.catalog) if 20.0 < ce.pos[0] < 90.0 and 15.0 < ce.pos[1] < 110.0]
    assert 0 < len(inside) < len(f.catalog)
    for method in ("single_vi", "joint_vi"):
        res = cel.infer_box(f.images, box, f.catalog, method=method, cfg=cel.ElboConfig(max_iters=10))
        assert len(res) == len(inside)
        for r, i in zip(res, inside):
            assert (r.init_ra, r.init_dec) == (f.catalog[i].pos[0], f.catalog[i].pos[1])
            assert r.vs.shape == (44,) and np.all(np.isfinite(r.vs)) and r.is_sky_bad is False
            assert abs(r.vs[0] - r.init_ra) <= 1e-4 + 1e-12      # position box of width loc_width around the catalog position
    ctx = _ctx(f)
    ref = one_node_single_infer(ctx, f.catalog, inside, cel.ElboConfig(max_iters=10))
    got = cel.infer_box(f.images, box, f.catalog, method="single_vi", cfg=cel.ElboConfig(max_iters=10))
    assert np.array_equal(np.stack([r.vs for r in got]), ref)
    assert cel.infer_box(f.images, cel.BoundingBox(-10.0, -5.0, 0.0, 1.0), f.catalog) == []


def test_infer_box_over_overlapping_fields():
    """infer_box on a 2 x 2 grid of overlapping fields (20 images; sparse patch rows): every source is optimised
    against all the images that cover it; the same numbers as with the dense patch table"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.infer import one_node_joint_infer
    from celeste_jl_amd.model import get_sky_patches, neighbor_map, PatchRow
    f = synthetic.make_multifield((2, 2), 150, 150, 0.10, 36, seed=21)
    box = cel.BoundingBox(0.0, 400.0, 0.0, 400.0)
    cfg = cel.ElboConfig(max_iters=30)
    res = cel.infer_box(f.images, box, f.catalog, method="joint_vi", cfg=cfg, n_iters=2)
    assert len(res) == 36 and all(np.all(np.isfinite(r.vs)) for r in res)
    dense = get_sky_patches(f.images, f.catalog)
    assert not isinstance(dense[0], PatchRow)
    nb = neighbor_map(dense)
    ctx = cel.FieldContext(f.images, dense, nb)
    ref = one_node_joint_infer(ctx, f.catalog, list(range(36)), nb, cfg, n_iters=2)
    assert np.array_equal(np.stack([r.vs for r in res]), ref)
    # the bright sources come back as the right type
    from celeste_jl_amd.params import ids
    flux = [ce.star_fluxes[2] if ce.is_star else ce.gal_fluxes[2] for ce in f.catalog]
    bright = np.argsort(flux)[-10:]
    ok = sum((res[i].vs[ids.is_star[0]] > 0.5) == f.catalog[i].is_star for i in bright)
    assert ok >= 7, ok


def test_end_to_end_recovers_the_synthetic_truth():
    """images drawn from a catalog -> infer_box (joint VI from generic_init_source) -> catalog rows: the brighter
    sources come back with the right type, r flux and colours (the idea of AccuracyBenchmark.score_predictions)"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.catalog import celeste_to_rows
    f = synthetic.make_field(300, 340, 40, seed=77, margin=26)
    box = cel.BoundingBox(0.0, 300.0, 0.0, 340.0)
    res = cel.infer_box(f.images, box, f.catalog, method="joint_vi")
    rows = celeste_to_rows(res)
    assert len(res) == 40 and len(rows) >= 34
    truth = {(ce.pos[0], ce.pos[1]): ce for ce in f.catalog}
    n_bright = n_type = 0
    flux_err, col_err = [], []
    for r, row in zip([x for x in res if not x.is_sky_bad], rows):
        ce = truth[(r.init_ra, r.init_dec)]
        fl = ce.star_fluxes if ce.is_star else ce.gal_fluxes
        if fl[2] < 2.0:        # faint: the posterior is broad, nothing to assert
            continue
        n_bright += 1
        n_type += int((row["is_star"] > 0.5) == ce.is_star)
        flux_err.append(abs(row["flux_r_nmgy"] / fl[2] - 1.0))
        col_err.append(abs(row["color_gr"] - np.log(fl[2] / fl[1])))
        assert abs(row["ra"] - ce.pos[0]) <= 1e-4 + 1e-9
    print("bright sources %d, type right %d, median |flux err| %.3f, median |g-r err| %.3f"
          % (n_bright, n_type, np.median(flux_err), np.median(col_err)))
    assert n_bright >= 8 and n_type >= 0.8 * n_bright
    assert np.median(flux_err) <= 0.10 and np.median(col_err) <= 0.15
    # the reference's score table (AccuracyBenchmark.score_predictions) over every source, faint ones included
    from celeste_jl_amd.catalog import catalog_entry_to_row, score_predictions
    good = [x for x in res if not x.is_sky_bad]
    scores = score_predictions([catalog_entry_to_row(truth[(r.init_ra, r.init_dec)]) for r in good], rows)
    print({k: (v["N"], round(v["first"], 3)) for k, v in scores.items()})
    assert scores["position"]["first"] <= 1.5e-4        # within the diagonal of the 1e-4 position box
    assert scores["flux_r_mag"]["first"] <= 0.25
    assert scores["missed_stars"]["first"] <= 0.35 and scores["missed_galaxies"]["first"] <= 0.35


def test_single_infer_neighbours_sit_at_catalog_init(oracle):
    """one_node_single_infer == per-target maximize! with init_sources([1], cat_local) (DeterministicVI.jl:94-103)"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, catalog_init_source, generic_init_source
    from celeste_jl_amd.infer import one_node_single_infer
    f = synthetic.make_field(72, 72, 7, seed=23, margin=14)
    ctx = _ctx(f)
    cfg = cel.ElboConfig(max_iters=6)
    out = one_node_single_infer(ctx, f.catalog, list(range(7)), cfg)
    base = np.stack([catalog_init_source(ce) for ce in f.catalog])
    for t in (0, 3, 6):
        vp = base.copy(); vp[t] = generic_init_source(f.catalog[t].pos)
        ovp, *_ = oracle.maximize(ctx.problem, vp, t, oracle.OptCfg(max_iters=6))
        assert np.abs(out[t] - ovp[t]).max() <= 1e-6
