import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bench
import celeste_jl_amd as cel
fld = bench.build_field(2048, 1489, 2000, 3)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
rng = np.random.default_rng(3)
for n in (300, 500, 750, 1000, 1250, 1500, 2000):
    tg = np.sort(rng.choice(2000, n, replace=False)).astype(np.int32)
    res = {}
    for mode in ("0", "1"):
        os.environ["CELESTE_OPT_FUSED"] = mode
        ctx.maximize_batch(fld.vp, tg, cel.ElboConfig())
        t0 = time.perf_counter()
        for _ in range(3): ctx.maximize_batch(fld.vp, tg, cel.ElboConfig())
        res[mode] = (time.perf_counter() - t0) / 3
    print("%5d targets: chained %.2f ms, fused %.2f ms" % (n, res["0"] * 1e3, res["1"] * 1e3))
