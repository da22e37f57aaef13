"""-m gpu: BASELINE.json configurations at full size.

configs[1] (512x512x5, 100 stars): full parity against the oracle.
configs[2] (2048x1489x5, 2000 star+galaxy sources): oracle parity on every source plus
size-independent properties (order invariance, sharding invariance, exact symmetry, counters).
configs[4] scaled to one GPU's share: overlapping fields, fp32 component loop at 1e-4."""
import numpy as np
import pytest

from parity_util import assert_parity

pytestmark = pytest.mark.gpu
ALL = 7


@pytest.fixture(scope="module")
def field3():
    from celeste_jl_amd import synthetic
    return synthetic.make_field(2048, 1489, 2000, seed=3)


@pytest.fixture(scope="module")
def ctx3(field3):
    import celeste_jl_amd as cel
    return cel.FieldContext(field3.images, field3.patches, field3.neighbors)


def test_config2_stars_full_parity(oracle):
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(512, 512, 100, seed=2, stars_only=True)
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    tg = list(range(100))
    errs = assert_parity(ctx.eval_batch(f.vp, tg, ALL), oracle.elbo_batch(ctx.problem, f.vp, tg, ALL), "config2")
    print("config2", errs)


def test_config3_oracle_parity_on_every_source(oracle, field3, ctx3):
    """BASELINE configs[2] at full size: all 2000 sources against the dense CPU restatement (value, 44 gradient
    entries, 44 x 44 Hessian, pixel counters, status), fused and split paths"""
    from celeste_jl_amd import cabi
    tg = list(range(2000))
    r = oracle.elbo_batch(ctx3.problem, field3.vp, tg, ALL)
    errs = assert_parity(ctx3.eval_batch(field3.vp, tg, ALL), r, "config3")
    print("config3, all 2000 sources", errs)
    errs = assert_parity(ctx3.eval_batch(field3.vp, tg, ALL | cabi.FLAG_SPLIT), r, "config3 split")
    print("config3 split variant", errs)


def test_config3_size_independent_properties(field3, ctx3):
    S = 2000
    tg = np.arange(S)
    v, d, h, cnt, st = ctx3.eval_batch(field3.vp, tg, ALL)
    assert (st == 0).all() and np.isfinite(v).all() and np.isfinite(d).all() and np.isfinite(h).all()
    # exact symmetry of every Hessian
    assert np.array_equal(h, h.transpose(0, 2, 1))
    # counters: active pairs = sum_n H2 (W2 - 1) when nothing is masked (elbo_objective.jl:349)
    expect = np.array([sum(p.active_pixel_bitmap.shape[0] * (p.active_pixel_bitmap.shape[1] - 1) for p in row)
                       for row in field3.patches])
    assert np.array_equal(cnt[:, 0], expect)
    assert (cnt[field3_no_neighbors(field3), 1] == 0).all()
    # order invariance: a permuted batch gives bit-identical per-target results
    perm = np.random.default_rng(0).permutation(S)
    v2, d2, h2, cnt2, _ = ctx3.eval_batch(field3.vp, perm, ALL)
    assert np.array_equal(v2, v[perm]) and np.array_equal(d2, d[perm]) and np.array_equal(h2, h[perm])
    # sharding invariance: two cost-balanced shards == one sweep
    from celeste_jl_amd.partition import shard_targets, estimate_time
    costs = [estimate_time(row) for row in field3.patches]
    for shard in shard_targets(costs, 2):
        vs, ds, hs, _, _ = ctx3.eval_batch(field3.vp, shard, ALL)
        assert np.array_equal(vs, v[shard]) and np.array_equal(ds, d[shard]) and np.array_equal(hs, h[shard])
    # value-only and gradient-only evaluations agree with the full one
    v0, _, _, _, _ = ctx3.eval_batch(field3.vp, tg, 4)
    assert np.max(np.abs(v0 - v) / np.abs(v)) <= 1e-13
    # the k parameters have zero likelihood derivatives
    vl, dl, hl, _, _ = ctx3.eval_batch(field3.vp, tg[:64], 3)
    assert np.all(dl[:, 28:] == 0) and np.all(hl[:, 28:, :] == 0)


def test_config3_with_one_psf_stamp_per_patch_oracle_parity_on_a_sample(oracle):
    """configs[2]'s field as production Celeste would see it (bench.py's `variable_psf` sub-record): an SDSSPSFMap evaluated at
    every source (SDSSIO.jl:239-299) -- ~8 800 stamps, one spline per (source, band), imaged_sources.jl:97-107 -- a varying
    sky plane and per-row calibration.  Every 8th source against the dense CPU restatement; all 2000: finite, exactly symmetric,
    order-invariant; and the constant-PSF field's numbers are NOT reproduced (the stamps matter)."""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(2048, 1489, 2000, seed=3, variable=True)
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    assert ctx.problem.c.n_stamps > 8000
    tg = np.arange(2000)
    v, d, h, cnt, st = ctx.eval_batch(f.vp, tg, ALL)
    assert (st == 0).all() and np.isfinite(v).all() and np.isfinite(d).all() and np.isfinite(h).all()
    assert np.array_equal(h, h.transpose(0, 2, 1))
    sample = tg[::8]
    r = oracle.elbo_batch(ctx.problem, f.vp, sample, ALL)
    errs = assert_parity((v[sample], d[sample], h[sample], cnt[sample], st[sample]), r, "config3, one stamp per patch")
    print("config3 with SDSSPSFMap, 250 of 2000 sources", errs)
    perm = np.random.default_rng(1).permutation(2000)
    v2, d2, h2, _, _ = ctx.eval_batch(f.vp, perm, ALL)
    assert np.array_equal(v2, v[perm]) and np.array_equal(d2, d[perm]) and np.array_equal(h2, h[perm])
    ctx.close()


def field3_no_neighbors(f):
    return np.array([len(n) == 0 for n in f.neighbors])


def test_config3_neighbor_parameters_matter(field3, ctx3):
    """perturbing a neighbour's parameters changes the target's ELBO but not its own-parameter structure"""
    t = int(np.argmax([len(n) for n in field3.neighbors]))
    nb = field3.neighbors[t][0]
    v, d, h, _, _ = ctx3.eval_batch(field3.vp, [t], ALL)
    vp2 = field3.vp.copy(); vp2[nb, 6:8] += 0.3
    v2, d2, h2, _, _ = ctx3.eval_batch(vp2, [t], ALL)
    assert v2[0] != v[0]
    iso = np.flatnonzero(field3_no_neighbors(field3))[:5]
    a = ctx3.eval_batch(field3.vp, iso, ALL); b = ctx3.eval_batch(vp2, iso, ALL)
    keep = [i for i, s in enumerate(iso) if s != nb]
    assert np.array_equal(a[0][keep], b[0][keep])


def test_config5_overlapping_fields_fp32(oracle):
    """BASELINE configs[4] scaled to one GPU's share: a 2 x 4 grid of overlapping fields (40 images), 1500 sources each
    seen by 5-20 images, fp32 component loop against the fp64 oracle at 1e-4 (sampled) and against the fp64 device
    path (all sources); sharding invariance across 8 cost-balanced shards"""
    import time
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.partition import shard_targets, estimate_time
    t0 = time.time()
    f = synthetic.make_multifield(grid=(2, 4), H=400, W=400, overlap=0.10, n_sources=1500, seed=5)
    S, N = len(f.catalog), len(f.images)
    seen = np.array([sum(p.active_pixel_bitmap.size > 0 for p in row) for row in f.patches])
    assert N == 40 and seen.min() >= 5 and seen.max() >= 10
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    t1 = time.time()
    tg = np.arange(S)
    fl32 = ALL | cabi.FLAG_FP32
    v64, d64, h64, cnt64, st64 = ctx.eval_batch(f.vp, tg, ALL)
    t2 = time.time()
    v32, d32, h32, cnt32, st32 = ctx.eval_batch(f.vp, tg, fl32)
    t3 = time.time()
    print("config5/8: build %.1f s, ctx %.1f s, fp64 sweep %.3f s, fp32 sweep %.3f s (host API incl. D2H of %d Hessians)"
          % (t1 - t0, t2 - t1 - (t3 - t2), t2 - t1, t3 - t2, S))
    assert (st64 == 0).all() and (st32 == 0).all() and np.array_equal(cnt32, cnt64)
    ev = np.max(np.abs(v32 - v64) / np.abs(v64))
    ed = max(np.abs(d32[t] - d64[t]).max() / np.abs(d64[t]).max() for t in tg)
    eh = max(np.abs(h32[t] - h64[t]).max() / np.abs(h64[t]).max() for t in tg)
    print("fp32 vs fp64 device:", ev, ed, eh)
    assert ev <= 1e-4 and ed <= 1e-4 and eh <= 1e-4
    sample = sorted(set(list(range(0, S, 60)) + list(np.argsort(-seen)[:6])))
    ov, od, oh, ocnt, ost = oracle.elbo_batch(ctx.problem, f.vp, sample, ALL)
    assert np.array_equal(cnt32[sample], ocnt)
    assert np.max(np.abs(v32[sample] - ov) / np.abs(ov)) <= 1e-4
    assert max(np.abs(d32[t] - od[k]).max() / np.abs(od[k]).max() for k, t in enumerate(sample)) <= 1e-4
    errs = assert_parity((v64[sample], d64[sample], h64[sample], cnt64[sample], st64[sample]), (ov, od, oh, ocnt, ost), "config5 fp64")
    print("config5 fp64 sample of %d" % len(sample), errs)
    costs = [estimate_time(row) for row in f.patches]
    for k, shard in enumerate(shard_targets(costs, 8)):
        # a rank's shard == the full sweep, bit for bit (same flags: the same kernel instantiation)
        vs, ds, hs, _, _ = ctx.eval_batch(f.vp, shard, ALL | cabi.FLAG_FP32)
        assert np.array_equal(vs, v32[shard]) and np.array_equal(ds, d32[shard]) and np.array_equal(hs, h32[shard])
        if k == 0:   # the gradient-only instantiation rounds its single-precision pixel terms differently: 1e-6, not bits
            vg, dg, _, _, _ = ctx.eval_batch(f.vp, shard, 1 | 4 | cabi.FLAG_FP32)
            assert np.max(np.abs(vg - v32[shard]) / np.abs(v32[shard])) <= 1e-6
            assert np.max(np.abs(dg - d32[shard]).max(axis=1) / np.abs(d32[shard]).max(axis=1)) <= 1e-5


def test_config5_full_size(oracle):
    """BASELINE configs[4] at full size in one context (16 fields = 80 images of 2048 x 1489, 30 k sources, sparse
    patch list): fp32 component loop within 1e-4 of the fp64 device path on every source and of the CPU oracle on a
    sample, fp64 within 1e-8 of the oracle, one rank's shard bit-identical to the full sweep (tests/config5_full.py)"""
    import config5_full
    out = config5_full.run(log=lambda *a, **k: None)
    print({k: out[k] for k in ("generate_s", "ctx_create_s", "patch_entries", "pixel_visits", "fp64", "fp32", "fp32_grad",
                               "shard0_fp32", "fp32_vs_fp64_device", "vs_oracle", "device_memory_GB")})
    assert out["n_images"] == 80 and out["images_per_source"][2] == 20


def test_config3_optimiser_against_cpu_restatement(oracle, field3, ctx3):
    """maximize! to convergence for every source on the device; 24 of them re-optimised by the CPU restatement of the
    same algorithm (different eigen-solver): same optimum"""
    from concurrent.futures import ThreadPoolExecutor
    import celeste_jl_amd as cel
    tg = np.arange(2000)
    vp, its, evals, elbo, st = ctx3.maximize_batch(field3.vp, tg, cel.ElboConfig())
    assert (st == 0).all() and np.isfinite(vp).all()
    v0 = ctx3.eval_batch(field3.vp, tg, 4)[0]
    assert np.all(elbo >= v0) and np.mean(elbo - v0) > 1e3
    # the reported optimum is the ELBO at the returned parameters (neighbours at their input values)
    chk = np.array([ctx3.eval_batch(np.where(np.arange(2000)[:, None] == t, vp, field3.vp), [t], 4)[0][0] for t in (0, 777, 1999)])
    assert np.max(np.abs(chk - elbo[[0, 777, 1999]]) / np.abs(chk)) <= 1e-12
    sample = list(range(13, 2000, 83))

    def cpu(t):
        return oracle.maximize(ctx3.problem, field3.vp, t, oracle.OptCfg())
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(cpu, sample))
    rel = np.array([abs(elbo[t] - r[3]) / abs(r[3]) for t, r in zip(sample, res)])
    same_iters = sum(int(its[t] == r[1]) for t, r in zip(sample, res))
    print("optimiser, %d sampled sources: median |dELBO|/|ELBO| %.1e, max %.1e, identical iteration counts %d"
          % (len(sample), np.median(rel), rel.max(), same_iters))
    assert np.median(rel) <= 1e-9 and np.quantile(rel, 0.9) <= 1e-6 and rel.max() <= 1e-3
