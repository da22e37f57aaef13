"""Diagnostic (gpurun): the joint-inference parity test's layers one by one -- device vs CPU restatement of maximize!
from the SAME input table, for max_iters = 1 .. 6, to see where a difference first appears."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic
from celeste_jl_amd.infer import joint_infer_sweeps
from celeste_jl_amd.params import catalog_init_source, generic_init_source
import oracle

f = synthetic.make_field(110, 120, 12, seed=19, margin=30)
ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
targets = [0, 1, 2, 4, 5, 7, 8, 9, 11]
vp0 = np.stack([catalog_init_source(ce) for ce in f.catalog])
for t in targets:
    vp0[t] = generic_init_source(f.catalog[t].pos)
layer_no = [0]


def layer(vp, lay, pc):
    layer_no[0] += 1
    res = None
    for mi in (1, 2, 3, 4, 5, 6):
        new, its, ev, elbo, st = ctx.maximize_batch(vp, lay, cel.ElboConfig(max_iters=mi), pos_centers=pc)
        rows = []
        for t, c in zip(lay, pc):
            r = oracle.maximize(ctx.problem, vp, t, oracle.OptCfg(max_iters=mi), pos_center=c)
            rows.append((r[0][t], r[1], r[3]))
        d = [np.abs(new[t] - rows[k][0]).max() for k, t in enumerate(lay)]
        print("layer %2d max_iters %d targets %s: max |dev - cpu| per target %s | iters dev %s cpu %s | elbo diff %s"
              % (layer_no[0], mi, list(lay), ["%.1e" % x for x in d], list(its), [r[1] for r in rows],
                 ["%.1e" % abs(elbo[k] - rows[k][2]) for k in range(len(lay))]))
        res = np.stack([r[0] for r in rows])
    return res   # continue from the CPU's result, so every layer starts from identical inputs


joint_infer_sweeps(layer, vp0.copy(), targets, f.neighbors, batch_size=5, n_iters=2, rng=np.random.default_rng(3))
