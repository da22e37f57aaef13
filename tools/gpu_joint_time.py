"""Wall time of one_node_joint_infer on the bench field (2000 sources, Cyclades batches of 400, 3 sweeps) and of the same
schedule through the chained driver; run through gpurun."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.infer import one_node_joint_infer, one_node_single_infer

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
tg = list(range(S))
for mode in ("1", "0"):
    os.environ["CELESTE_OPT_FUSED"] = mode
    for rep in range(2):
        t0 = time.time(); vs = one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors); t1 = time.time()
    print("one_node_joint_infer, Cyclades batches of 400, 3 sweeps, %s driver: %.3f s (%.0f sources/s)"
          % ("fused" if mode == "1" else "chained", t1 - t0, S / (t1 - t0)))
os.environ.pop("CELESTE_OPT_FUSED")
for rep in range(2):
    t0 = time.time(); one_node_single_infer(ctx, fld.catalog, tg); t1 = time.time()
print("one_node_single_infer (default driver): %.3f s (%.0f sources/s)" % (t1 - t0, S / (t1 - t0)))
t0 = time.time(); one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors, schedule="coloring"); t1 = time.time()
print("one_node_joint_infer, colouring schedule: %.3f s" % (t1 - t0))
