"""Time one_node_single_infer / one_node_joint_infer on the bench field (run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.infer import one_node_single_infer, one_node_joint_infer

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
tg = list(range(S))
for rep in range(2):
    t0 = time.time(); vs_single = one_node_single_infer(ctx, fld.catalog, tg); t1 = time.time()
    vs_joint = one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors); t2 = time.time()
for bs in (400, 2000):
    t3 = time.time(); one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors, batch_size=bs); t4 = time.time()
    print("joint infer, batch_size %d: %.3f s" % (bs, t4 - t3))
t5 = time.time(); vs_col = one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors, schedule="coloring"); t6 = time.time()
from celeste_jl_amd.partition import color_classes
cls = color_classes(tg, {t: list(fld.neighbors[t]) for t in tg})
print("joint infer, coloring: %.3f s (%.0f sources/s); %d colours, class sizes %s" % (t6 - t5, S / (t6 - t5), len(cls), [len(c) for c in cls]))
from celeste_jl_amd.partition import partition_cyclades_dynamic
for bs in (400, 2000):
    b = partition_cyclades_dynamic(tg, {t: list(fld.neighbors[t]) for t in tg}, batch_size=bs, rng=np.random.default_rng(42))
    print("batch_size", bs, "batches", len(b), "layers per sweep", sum(max(len(c) for c in comps) for comps in b),
          "largest component", max(max(len(c) for c in comps) for comps in b))
print("single infer: %.3f s (%.0f sources/s); joint infer (3 sweeps of Cyclades batches): %.3f s (%.0f sources/s)"
      % (t1 - t0, S / (t1 - t0), t2 - t1, S / (t2 - t1)))
# how much the joint schedule moves sources with neighbours
nb = np.array([len(n) > 0 for n in fld.neighbors])
d = np.abs(vs_joint - vs_single).max(axis=1)
print("sources with neighbours: %d; max |joint - single| per source: median %.2e (with neighbours) vs %.2e (without)"
      % (nb.sum(), np.median(d[nb]), np.median(d[~nb])))

# end to end, host side included: infer_box = patches + neighbours + context + inference + result rows
from celeste_jl_amd.infer import infer_box, BoundingBox
for method in ("single_vi", "joint_vi"):
    t7 = time.time()
    res = infer_box(fld.images, BoundingBox(0, 2049, 0, 1490), fld.catalog, method=method)
    t8 = time.time()
    print("infer_box(%s), %d sources, host side included: %.2f s" % (method, len(res), t8 - t7))
import time as _t
from celeste_jl_amd import model, cabi
t0 = _t.time(); tab = model.patch_table(fld.images, fld.catalog); t1 = _t.time(); nb = tab.neighbors(); t2 = _t.time()
pb = cabi.problem_from_table(fld.images, tab, nb); t3 = _t.time()
c2 = cel.FieldContext(fld.images, None, nb, problem=pb); t4 = _t.time()
print("  of which: patch table %.2f s, neighbours %.3f s, marshalling (image planes to column-major) %.2f s, context creation %.2f s"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
