"""Who writes into freed host memory after a context dies?  (One medium-field fuzz scene in ~800 came out corrupted under pytest:
two words of a small numpy array allocated early in the NEXT test changed -- one decremented, one zeroed -- and an explicit
gc.collect() after every test made it go away.)  This loop makes the condition on purpose: a context is used, then left to the
CYCLIC collector (a self-reference), which destroys it at some allocation inside a burst of small numpy allocations of every
size class; the arrays take the blocks the destruction freed, and are checked a little later.
usage: python tools/gpu_stale_write_hunt.py <iterations> [mode] [tag]      mode: cycle (default) | refcount | nodestroy"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi

iters = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "cycle"; tag = sys.argv[3] if len(sys.argv) > 3 else mode
rng = np.random.default_rng(os.getpid())
scenes = []
for k in range(6):
    H, W, S = int(rng.integers(120, 240)), int(rng.integers(120, 240)), int(rng.integers(33, 91))
    f = synthetic.make_field(H, W, S, seed=4000 + k, nan_fraction=0.01, margin=8)
    scenes.append(f)
print(tag, "scenes built", flush=True)
kept = []
hits = 0
t0 = time.time()
for it in range(iters):
    f = scenes[it % len(scenes)]
    S = f.vp.shape[0]
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors, psf_K=2)
    tg = rng.permutation(S)[:int(rng.integers(33, S + 1))].tolist()
    flags = int(rng.choice([0, 4, 1, 5, 3, 7, 7, 7]))
    if rng.random() < 0.25:
        ctx.eval_batch(f.vp, tg, flags | cabi.FLAG_FP32, raise_on_error=False)
    ctx.eval_batch(f.vp, tg, flags, raise_on_error=False)
    if flags & 2:
        ctx.eval_batch(f.vp, tg, flags | cabi.FLAG_SPLIT, raise_on_error=False)
    if mode == "cycle":
        ctx._me = ctx          # only the cyclic collector can free it now
        del ctx
    elif mode == "refcount":
        del ctx
    else:
        kept.append(ctx); del ctx
    # the burst: small arrays of every size class; the collector runs somewhere in here
    arrays = []
    for rep in range(3):
        for size in range(32, 2048 + 1, 16):
            a = np.empty(size, dtype=np.uint8); a[:] = 0xA5
            arrays.append(a)
        junk = [object() for _ in range(400)]     # container allocations drive the generation-0 counter
        junk = [[i] for i in range(400)]
    time.sleep(0.002)
    for a in arrays:
        bad = np.flatnonzero(a != 0xA5)
        if bad.size:
            hits += 1
            w = a.view(np.uint32) if a.size % 4 == 0 else None
            print(tag, "iteration", it, "array of", a.size, "bytes changed at offsets", bad[:16].tolist(), "bytes", [hex(int(x)) for x in a[bad[:16]]],
                  "" if w is None else "words %s" % [(int(i), hex(int(w[i]))) for i in np.unique(bad // 4)[:8]], flush=True)
    del arrays
print(tag, "done", iters, "iterations,", hits, "hits, %.1f s" % (time.time() - t0), flush=True)
