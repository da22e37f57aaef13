"""Does a one-member group's host-pointer sweep time depend on WHEN its context (its copy stream) was created?  (gpurun)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.group import FieldGroup

fld = bench.build_field(2048, 1489, 2000, 3)
tg = list(range(len(fld.catalog)))


def t10(f):
    f(); f(); f()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), float(np.median(ts))


keep = []
for i in range(8):
    if i % 2 == 0:
        c = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
        print("object %d: FieldContext      min %.3f ms  median %.3f ms" % ((i,) + t10(lambda: c.eval_batch(fld.vp, tg))), flush=True)
        keep.append(c)
    else:
        g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=[0])
        print("object %d: group of one      min %.3f ms  median %.3f ms" % ((i,) + t10(lambda: g.eval_batch(fld.vp, tg))), flush=True)
        keep.append(g)
print("again, in creation order:")
for i, o in enumerate(keep):
    print("object %d: min %.3f ms  median %.3f ms" % ((i,) + t10(lambda: o.eval_batch(fld.vp, tg))), flush=True)
