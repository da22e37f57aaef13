"""Where a lift_kernel workgroup spends its time (gpurun; CELESTE_MI355X_LIB = a -DLIFT_TIMING build; the debug symbol is
kept by the export map only when its name starts with celeste_)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd import cabi
names = ["parameters / init", "pass 1: records, brightness", "pass 2: Jacobians", "pass 3: gradient + Hessian", "KL value / join", "assembly + stores"]
which = sys.argv[1] if len(sys.argv) > 1 else "3"
if which == "3":
    fld = bench.build_field(2048, 1489, 2000, 3); flags = 7
else:
    fld = bench.build_multifield((2, 2), 2048, 1489, 7500, 5); flags = 7 | cabi.FLAG_FP32
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
lib = cabi.load_library()
tg = np.arange(len(fld.catalog), dtype=np.int32)
ctx.eval_batch(fld.vp, tg, flags)
out = (C.c_uint64 * 16)(); lib.celeste_lift_clocks(1, out)
for _ in range(3):
    ctx.eval_batch(fld.vp, tg, flags)
lib.celeste_lift_clocks(1, out)
c = np.array(out[:], dtype=float); k = max(c[15], 1)
print("config %s: %d targets, cycles per workgroup (thread 0; 100 MHz wall... shader clock64) by section:" % (which, len(tg)))
for i, n in enumerate(names):
    print("  %-32s %8.0f" % (n, c[i] / k))
print("  total %.0f" % (c[:6].sum() / k))
