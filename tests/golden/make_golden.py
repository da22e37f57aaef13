#!/usr/bin/env python3
"""Generate tests/golden/*.npz: seeded synthetic inputs and the CPU oracle's outputs for them.

The reference itself cannot run here (Julia absent) and holds no numeric ELBO goldens,
so these vectors pin *our* oracle (regression) and let the GPU tests run without the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as gu
from celeste_jl_amd import cabi
from oracle import oracle

# usage: make_golden.py [case ...]   (default: every case of golden_util.CASES)
for name in (sys.argv[1:] or gu.CASES):
    f = gu.build_case(name)
    arrs = gu.field_to_arrays(f)
    f2 = gu.arrays_to_field(arrs)  # the fixture must be self-contained
    pb = cabi.Problem(f2.images, f2.patches, f2.neighbors)
    tg = list(range(len(f2.catalog)))
    out = {}
    for flags in (7, 3, 0):
        v, d, h, cnt, st = oracle.elbo_batch(pb, f2.vp, tg, flags, n_threads=1)
        assert (st == 0).all()
        out["v%d" % flags] = v
        if flags:
            out["d%d" % flags] = d; out["h%d" % flags] = h
        out["cnt"] = cnt
    S = len(f2.catalog)
    if S >= 2:   # elbo() with all sources active (Sa = S): P x Sa gradient, (P Sa)^2 Hessian
        full = cabi.Problem(f2.images, f2.patches, [[s for s in range(S) if s != a] for a in range(S)])
        mv, md, mh, mcnt, mst = oracle.elbo_multi(full, f2.vp, list(range(S)), 7)
        assert mst == 0
        out.update(multi_v=np.array(mv), multi_d=md, multi_h=mh, multi_cnt=mcnt)
    # maximize! of source 0 (neighbours frozen), 12 Newton iterations, KL on
    ovp, oit, oev, oelbo, ost = oracle.maximize(pb, f2.vp, 0, oracle.OptCfg(max_iters=12))
    assert ost == 0
    out.update(opt_vs=ovp[0], opt_elbo=np.array(oelbo), opt_iters=np.array(oit))
    np.savez_compressed(gu.path(name), **arrs, **out)
    print(name, os.path.getsize(gu.path(name)), "bytes", out["v7"][:2])
