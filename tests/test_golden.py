"""Committed golden vectors (tests/golden/make_golden.py): the oracle must keep reproducing them (CPU), and the HIP
path must reproduce them through the C ABI without the oracle present (-m gpu)."""
import numpy as np
import pytest

import golden_util as gu
from parity_util import RTOL, rel_err


@pytest.mark.parametrize("name", list(gu.CASES))
def test_fixture_inputs_are_reproducible(name):
    """the seeded generator still produces the committed inputs (host logic regression)"""
    z = np.load(gu.path(name))
    f = gu.build_case(name)
    a = gu.field_to_arrays(f)
    assert np.array_equal(a["pixels"], z["pixels"], equal_nan=True)
    assert np.array_equal(a["vp"], z["vp"]) and np.array_equal(a["pos"], z["pos"])


@pytest.mark.parametrize("name", list(gu.CASES))
def test_oracle_reproduces_golden(oracle, name):
    from celeste_jl_amd import cabi
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    tg = list(range(len(f.catalog)))
    for flags in (7, 3, 0):
        v, d, h, cnt, st = oracle.elbo_batch(pb, f.vp, tg, flags, n_threads=1)
        assert (st == 0).all() and np.array_equal(cnt, z["cnt"])
        assert np.abs(v - z["v%d" % flags]).max() <= 1e-13 * np.abs(v).max()
        if flags:
            assert np.abs(d - z["d%d" % flags]).max() <= 1e-12 * np.abs(d).max()
            assert np.abs(h - z["h%d" % flags]).max() <= 1e-12 * np.abs(h).max()


@pytest.mark.parametrize("name", list(gu.CASES))
def test_oracle_reproduces_multi_active_and_optimiser_golden(oracle, name):
    from celeste_jl_amd import cabi
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    S = len(f.catalog)
    if "multi_h" in z and S <= 3:
        full = cabi.Problem(f.images, f.patches, [[s for s in range(S) if s != a] for a in range(S)])
        v, d, h, cnt, st = oracle.elbo_multi(full, f.vp, list(range(S)), 7)
        assert st == 0 and np.array_equal(cnt, z["multi_cnt"])
        assert abs(v - z["multi_v"]) <= 1e-13 * abs(v) and np.abs(h - z["multi_h"]).max() <= 1e-12 * np.abs(h).max()
    if S <= 2:
        pb = cabi.Problem(f.images, f.patches, f.neighbors)
        ovp, oit, oev, oelbo, ost = oracle.maximize(pb, f.vp, 0, oracle.OptCfg(max_iters=12))
        assert ost == 0 and oit == int(z["opt_iters"])
        assert np.abs(ovp[0] - z["opt_vs"]).max() <= 1e-9 and abs(oelbo - z["opt_elbo"]) <= 1e-10 * abs(oelbo)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(gu.CASES))
def test_hip_reproduces_multi_active_and_optimiser_golden(name):
    import celeste_jl_amd as cel
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    S = len(f.catalog)
    if "multi_h" in z:
        ctx = cel.FieldContext(f.images, f.patches, [[s for s in range(S) if s != a] for a in range(S)])
        v, d, h, cnt = ctx.eval_multi(f.vp, list(range(S)), 7)
        assert np.array_equal(cnt, z["multi_cnt"]) and abs(v - z["multi_v"]) <= RTOL * abs(v)
        assert rel_err(d.T, z["multi_d"]) <= RTOL and rel_err(h, z["multi_h"]) <= RTOL
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    vp, its, evals, elbo, st = ctx.maximize_batch(f.vp, [0], cel.ElboConfig(max_iters=12))
    assert st[0] == 0 and its[0] == int(z["opt_iters"])
    assert abs(elbo[0] - z["opt_elbo"]) <= 1e-8 * abs(elbo[0]) and np.abs(vp[0] - z["opt_vs"]).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(gu.CASES))
def test_hip_reproduces_golden(name):
    import celeste_jl_amd as cel
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    tg = list(range(len(f.catalog)))
    for flags in (7, 3, 0):
        v, d, h, cnt, st = ctx.eval_batch(f.vp, tg, flags)
        assert (st == 0).all() and np.array_equal(cnt, z["cnt"])
        assert np.max(np.abs(v - z["v%d" % flags]) / np.abs(z["v%d" % flags])) <= RTOL
        if flags:
            for t in tg:
                assert rel_err(d[t], z["d%d" % flags][t]) <= RTOL
            if flags & 2:
                for t in tg:
                    assert rel_err(h[t], z["h%d" % flags][t]) <= RTOL
