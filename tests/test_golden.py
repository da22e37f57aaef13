"""Committed golden vectors (tools/make_golden.py): the oracle must keep reproducing them (CPU), and the HIP
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
