"""Where a value_kernel workgroup spends its time (gpurun; CELESTE_MI355X_LIB = a -DVALUE_TIMING build)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd import cabi
fld = bench.build_field(2048, 1489, 2000, 3)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
lib = cabi.load_library()
tg = np.arange(len(fld.catalog), dtype=np.int32)
for n in (2000, 250):
    ctx.eval_batch(fld.vp, tg[:n], 7)
    out = (C.c_uint64 * 8)(); lib.celeste_value_clocks(1, out)
    for _ in range(5):
        ctx.eval_batch(fld.vp, tg[:n], 7)
    lib.celeste_value_clocks(1, out)
    c = np.array(out[:], dtype=float); k = max(c[7], 1)
    print("%d targets: %d active items per launch, %.2f iterations of 64 pixels each; cycles per item: descriptor + patches %.0f, "
          "LDS staging %.0f, pixel loop %.0f (%.1f / %.1f / %.1f us at 2.4 GHz)"
          % (n, k / 5, c[6] / k, c[0] / k, c[1] / k, c[2] / k, c[0] / k / 2400, c[1] / k / 2400, c[2] / k / 2400))
