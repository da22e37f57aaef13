"""Per-source contexts on a shared image handle (process_source, ParallelRun.jl:468-488): what creating, using once and
destroying one costs (celeste_ctx_create_on + one celeste_elbo_eval + celeste_ctx_destroy), C calls timed from Python."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi
f = synthetic.make_field(1024, 900, 300, seed=3)
iset = cabi.ImageSet(f.images)
FL = cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL
tm = {"marshal": [], "create": [], "first_eval": [], "second_eval": [], "destroy": []}
for t in range(200):
    loc = [t] + [int(x) for x in f.neighbors[t]]
    t0 = time.perf_counter()
    pb = cabi.Problem(f.images, [f.patches[s] for s in loc], [list(range(1, len(loc)))] + [[] for _ in loc[1:]], marshal_images=False)
    t1 = time.perf_counter()
    ctx = cel.FieldContext(f.images, None, None, image_set=iset, problem=pb)
    t2 = time.perf_counter()
    v = np.ascontiguousarray(f.vp[loc])
    ctx.eval_batch(v, [0], FL, pinned=False)
    t3 = time.perf_counter()
    ctx.eval_batch(v, [0], FL, pinned=False)
    t4 = time.perf_counter()
    ctx.close()
    t5 = time.perf_counter()
    if t >= 20:
        for k, d in zip(tm, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            tm[k].append(d * 1e6)
print("per-source context, median us over %d sources: " % len(tm["create"]) + ", ".join("%s %.0f" % (k, np.median(v)) for k, v in tm.items()),
      "| mean sources per context %.1f" % np.mean([1 + len(f.neighbors[t]) for t in range(20, 200)]))
