"""celeste_elbo_eval through host pointers on per-source contexts (the literal drop-in of ElboMaximize.jl:166 inside
process_source): latency per call, one target, value + gradient + Hessian + KL.  A/B by environment: CELESTE_SMALL_ZERO_COPY_IN=0
(the table goes up with a copy command) against the default (the kernels read a small table from the page-locked block)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi
f = synthetic.make_field(512, 512, 120, seed=3)
iset = cabi.ImageSet(f.images)
FL = cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL
lat = []; ref = None
for t in range(40):
    loc = [t] + [int(x) for x in f.neighbors[t]]
    ctx = cel.FieldContext(f.images, [f.patches[s] for s in loc], [list(range(1, len(loc)))] + [[] for _ in loc[1:]], image_set=iset)
    v = np.ascontiguousarray(f.vp[loc])
    for rep in range(30):
        t1 = time.perf_counter(); g = ctx.eval_batch(v, [0], FL, pinned=False); dt = time.perf_counter() - t1
        if rep >= 5: lat.append(dt)
    if t == 7: ref = (g[0].copy(), g[1].copy(), g[2].copy())
    ctx.close()
lat = np.sort(np.array(lat)) * 1e6
import hashlib
print(os.environ.get("CELESTE_SMALL_ZERO_COPY_IN", "default"), "median %.1f us  p10 %.1f  p90 %.1f  (%d calls)  result hash %s" % (
    np.median(lat), lat[len(lat) // 10], lat[9 * len(lat) // 10], len(lat), hashlib.sha1(b"".join(a.tobytes() for a in ref)).hexdigest()[:12]))
