#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | sed "s|^|$1: |"
import sys; sys.path.insert(0, "tests")
import numpy as np, golden_util as gu, celeste_jl_amd as cel
for name in gu.CASES:
    z = np.load(gu.path(name)); f = gu.arrays_to_field(z)
    if "opt_iters" not in z: continue
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    for mi in (4, 8, 10, 11, 12):
        vp, its, evals, elbo, st = ctx.maximize_batch(f.vp, [0], cel.ElboConfig(max_iters=mi))
        print(name, "max_iters", mi, "iters", its[0], "evals", evals[0], "elbo %.12g" % elbo[0], "| golden(12): iters", int(z["opt_iters"]), "elbo %.12g" % float(z["opt_elbo"]), "max|dvp| %.2e" % np.abs(vp[0] - z["opt_vs"]).max())
PY
}
one product
for f in tools/variants/lib_*.so; do CELESTE_MI355X_LIB=$PWD/$f one $(basename $f); done
