"""Time single / joint inference on a many-field problem (run through gpurun).  GRID=2,2 NSRC=7500 by default:
4 overlapping SDSS-size fields (20 images), sparse patch rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic
from celeste_jl_amd.infer import one_node_single_infer, one_node_joint_infer

grid = tuple(int(x) for x in os.environ.get("GRID", "2,2").split(","))
n_src = int(os.environ.get("NSRC", "7500"))
f = synthetic.make_multifield(grid=grid, H=2048, W=1489, overlap=0.10, n_sources=n_src, seed=5, sparse=True,
                              workers=min(16, len(os.sched_getaffinity(0))))
S = len(f.catalog)
ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
tg = list(range(S))
for rep in range(2):
    t0 = time.time(); one_node_single_infer(ctx, f.catalog, tg); t1 = time.time()
    one_node_joint_infer(ctx, f.catalog, tg, f.neighbors, schedule="coloring"); t2 = time.time()
print("%d fields, %d images, %d sources: single infer %.3f s (%.0f sources/s); joint infer (3 sweeps, colouring) %.3f s "
      "(%.0f sources/s)" % (grid[0] * grid[1], len(f.images), S, t1 - t0, S / (t1 - t0), t2 - t1, S / (t2 - t1)))
for mode in ("1", "0"):
    os.environ["CELESTE_JOINT_DATAFLOW"] = mode
    for rep in range(2):
        t3 = time.time(); one_node_joint_infer(ctx, f.catalog, tg, f.neighbors); t4 = time.time()
    print("joint infer, the reference's Cyclades schedule (batches of 400, 3 sweeps), %s: %.3f s (%.0f sources/s)"
          % ("one dataflow launch" if mode == "1" else "layer by layer", t4 - t3, S / (t4 - t3)))
from celeste_jl_amd.infer import joint_layers, default_infer_config
from celeste_jl_amd.params import catalog_init_source, generic_init_source
vp = np.stack([catalog_init_source(ce) for ce in f.catalog])
for t in tg:
    vp[t] = generic_init_source(f.catalog[t].pos)
for schedule in ("cyclades", "coloring"):
    layers = joint_layers(tg, f.neighbors, schedule=schedule)
    centers = [vp[l, 0:2].copy() for l in layers]
    for mode in ("1", "0"):
        os.environ["CELESTE_JOINT_DATAFLOW"] = mode
        t5 = time.time(); out = ctx.joint_infer(vp, layers, default_infer_config(), pos_centers=centers); t6 = time.time()
        print("celeste_joint_infer, %s, %d layers (largest %d), dataflow=%s: %.3f s; evaluations %d (mean %.1f per entry, max %d)"
              % (schedule, len(layers), max(map(len, layers)), mode, t6 - t5, out[2].sum(), out[2].mean(), out[2].max()))
