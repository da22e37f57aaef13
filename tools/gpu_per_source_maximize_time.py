"""process_source as the reference runs it (ParallelRun.jl:468-488): a context for the source and its neighbours on the shared
image handle, maximize! of the one source (Newton trust region, <= 50 iterations), destroy.  Times the C calls from Python."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi
f = synthetic.make_field(1024, 900, 300, seed=3)
iset = cabi.ImageSet(f.images)
tm = {"create": [], "maximize": [], "second_maximize": [], "destroy": []}
its = []
for t in range(120):
    loc = [t] + [int(x) for x in f.neighbors[t]]
    pb = cabi.Problem(f.images, [f.patches[s] for s in loc], [list(range(1, len(loc)))] + [[] for _ in loc[1:]], marshal_images=False)
    t1 = time.perf_counter()
    ctx = cel.FieldContext(f.images, None, None, image_set=iset, problem=pb)
    t2 = time.perf_counter()
    v = np.ascontiguousarray(f.vp[loc])
    r = ctx.maximize_batch(v, [0], cel.ElboConfig(), raise_on_error=False)
    t3 = time.perf_counter()
    r = ctx.maximize_batch(v, [0], cel.ElboConfig(), raise_on_error=False)
    t4 = time.perf_counter()
    ctx.close()
    t5 = time.perf_counter()
    if t >= 20:
        for k, d in zip(tm, (t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            tm[k].append(d * 1e6)
        its.append(int(r[1][0]))
print("per-source maximize!, median us over %d sources: " % len(its) + ", ".join("%s %.0f" % (k, np.median(v)) for k, v in tm.items()),
      "| Newton iterations: median %d, mean %.1f -> %.0f us per iteration of the second call" % (np.median(its), np.mean(its), np.median(tm["second_maximize"]) / max(np.median(its), 1)))
