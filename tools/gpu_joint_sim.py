"""Joint inference on the bench field: per-entry Newton iteration counts of the reference's schedule, and what the same
work would take under (a) layer-by-layer launches (every layer waits for its slowest source) and (b) chains (inside a
Cyclades batch every connected component runs its sources one after another, independently of the other components).
Run through gpurun; prints the two critical paths in Newton iterations."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.infer import default_infer_config, NUM_JOINT_VI_ITERS
from celeste_jl_amd.params import catalog_init_source, generic_init_source
from celeste_jl_amd.partition import partition_cyclades_dynamic

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
targets = list(range(S))
nmap = {t: list(fld.neighbors[t]) for t in targets}
batches = partition_cyclades_dynamic(targets, nmap, batch_size=400, rng=np.random.default_rng(42))
layers, tags = [], []          # tags[l] = (sweep, batch, j, [component index of every entry])
for sw in range(NUM_JOINT_VI_ITERS):
    for b, comps in enumerate(batches):
        for j in range(max(len(c) for c in comps)):
            idx = [k for k, c in enumerate(comps) if len(c) > j]
            layers.append([targets[comps[k][j]] for k in idx]); tags.append((sw, b, j, idx))
vp = np.stack([catalog_init_source(ce) for ce in fld.catalog])
for t in targets:
    vp[t] = generic_init_source(fld.catalog[t].pos)
centers = [vp[l, 0:2].copy() for l in layers]
new, its, evals, el, st = ctx.joint_infer(vp, layers, default_infer_config(), pos_centers=centers)
assert (st == 0).all()
off = np.cumsum([0] + [len(l) for l in layers])
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/joint_evals.npz", evals=evals, off=off, targets=np.concatenate([np.asarray(l) for l in layers]))
layered = 0
chain = {}
for l, (sw, b, j, idx) in enumerate(tags):
    e = evals[off[l]:off[l + 1]]
    layered += int(e.max())
    for k, n in zip(idx, e):
        chain[(sw, b, k)] = chain.get((sw, b, k), 0) + int(n)
chained = 0
for sw in range(NUM_JOINT_VI_ITERS):
    for b in range(len(batches)):
        chained += max(v for (s2, b2, k), v in chain.items() if s2 == sw and b2 == b)
print(json.dumps({"layers": len(layers), "entries": int(off[-1]), "mean_evals": float(evals.mean()), "max_evals": int(evals.max()),
                  "critical_path_evals_layered": layered, "critical_path_evals_chained_per_batch": chained,
                  "evals_by_sweep": [float(np.mean([evals[off[l]:off[l + 1]].mean() for l, t in enumerate(tags) if t[0] == sw])) for sw in range(3)],
                  "layer_max_by_sweep": [int(sum(evals[off[l]:off[l + 1]].max() for l, t in enumerate(tags) if t[0] == sw)) for sw in range(3)]}))

# how the trust-region sub-problems of this schedule were solved
import ctypes as C
from celeste_jl_amd import cabi
st5 = (C.c_uint64 * 5)()
lib = cabi.load_library()
lib.celeste_optim_stats(1, st5)
ctx.joint_infer(vp, layers, default_infer_config(), pos_centers=centers)
lib.celeste_optim_stats(0, st5)
print(json.dumps({"tr_interior": int(st5[0]), "tr_boundary": int(st5[1]), "tr_hard": int(st5[2]), "secular_iters_total": int(st5[3]),
                  "secular_iters_max": int(st5[4])}))
import time
for mode in ("0", "1"):
    os.environ["CELESTE_JOINT_DATAFLOW"] = mode
    for rep in range(2):
        t0 = time.perf_counter(); ctx.joint_infer(vp, layers, default_infer_config(), pos_centers=centers); dt = time.perf_counter() - t0
    print("joint_infer dataflow=%s: %.4f s" % (mode, dt))
