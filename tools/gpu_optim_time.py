"""Time celeste_maximize_batch on the bench field (run through gpurun)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
tg = np.arange(S, dtype=np.int32)
for max_iters in (5, 50):
    cfg = cel.ElboConfig(max_iters=max_iters)
    for rep in range(2):
        t0 = time.time()
        vp, its, evals, elbo, st = ctx.maximize_batch(fld.vp, tg, cfg)
        dt = time.time() - t0
    v0 = ctx.eval_batch(fld.vp, tg, 4)[0]
    print("max_iters %d: %.3f s, %.0f optimised sources/s, mean iters %.1f, evals %d, status!=0: %d, improved %d/%d, "
          "mean dELBO %.1f" % (max_iters, dt, S / dt, its.mean(), evals.sum(), (st != 0).sum(), (elbo > v0).sum(), S,
                               (elbo - v0).mean()))
    import ctypes as C
    from celeste_jl_amd import cabi
    st5 = (C.c_uint64 * 5)()
    cabi.load_library().celeste_optim_stats(1, st5)
    print("  TR sub-problems (2 reps): interior %d boundary %d hard %d | secular iterations mean %.1f max %d"
          % (st5[0], st5[1], st5[2], st5[3] / max(st5[1], 1), st5[4]))
    print("  iteration histogram:", np.bincount(its)[:60].tolist())
