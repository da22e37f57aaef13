"""Kernel times on a config-5-like problem (overlapping fields; run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi

grid = tuple(int(x) for x in os.environ.get("GRID", "2,4").split(","))
n_src = int(os.environ.get("NSRC", "1500"))
f = synthetic.make_multifield(grid=grid, H=400, W=400, overlap=0.10, n_sources=n_src, seed=5)
S, N = len(f.catalog), len(f.images)
ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
tg = np.arange(S, dtype=np.int32)
st = ctx.work_stats(tg)
ctx.enable_timing(True)
for flags, name in ((7, "fp64"), (7 | cabi.FLAG_FP32, "fp32"), (5 | cabi.FLAG_FP32, "fp32 grad-only")):
    ms = []
    for it in range(6):
        g = ctx.eval_batch(f.vp, tg, flags)
        ms.append(ctx.last_kernel_ms())
    ms = np.array(ms)[2:].mean(axis=0)
    print("%s: S=%d N=%d visits %d | prep %.3f pixel %.3f lift %.3f ms -> %.2f M sources/s (kernels only)"
          % (name, S, N, st["active_pixel_visits"], ms[0], ms[1], ms[2], S / ms.sum() / 1e3))
