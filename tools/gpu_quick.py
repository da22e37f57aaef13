"""Quick GPU sanity + timing (run through gpurun)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic
from oracle import oracle

f = synthetic.make_field(512, 512, 100, seed=2)
ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
tg = list(range(100))
ctx.enable_timing(True)
for it in range(3):
    t0 = time.time(); g = ctx.eval_batch(f.vp, tg, 7); t1 = time.time()
    print("gpu batch wall %.3f ms, kernels %s" % ((t1 - t0) * 1e3, ctx.last_kernel_ms()))
t0 = time.time(); r = oracle.elbo_batch(ctx.problem, f.vp, tg, 7); t1 = time.time()
print("oracle %.3f s" % (t1 - t0))
for name, a, b in (("v", g[0], r[0]), ("d", g[1], r[1]), ("h", g[2], r[2])):
    print(name, "max rel-to-block err", float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
print("counters equal", np.array_equal(g[3], r[3]), g[3][:3].tolist(), "status", g[4][:5], r[4][:5])
print(ctx.work_stats(tg))
