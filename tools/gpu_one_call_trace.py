"""A handful of one-target celeste_elbo_eval calls on a per-source context, for `rocprofv3 --hip-trace --kernel-trace
--memory-copy-trace`: where a 71 us call spends its time (API calls, copy, kernels, the wait)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi
f = synthetic.make_field(512, 512, 120, seed=3)
iset = cabi.ImageSet(f.images)
FL = cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL
t = 7
loc = [t] + [int(x) for x in f.neighbors[t]]
ctx = cel.FieldContext(f.images, [f.patches[s] for s in loc], [list(range(1, len(loc)))] + [[] for _ in loc[1:]], image_set=iset)
v = np.ascontiguousarray(f.vp[loc])
for rep in range(30):
    ctx.eval_batch(v, [0], FL, pinned=False)
    time.sleep(0.002)
print("neighbours", len(loc) - 1)
