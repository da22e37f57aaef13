"""Repeat one medium-field fuzz scenario (tests/test_gpu_parity.py::_randomised_field, seed0 = 3000) on fresh contexts and
report every call whose statuses are not all 0 -- the hunt for a rare, timing-dependent non-finite result.
usage (through gpurun, several copies side by side): python tools/gpu_flaky_hunt.py <seed> <iterations> [tag]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi

seed = int(sys.argv[1]); iters = int(sys.argv[2]); tag = sys.argv[3] if len(sys.argv) > 3 else "x"
seed0 = 3000; size_range = (120, 240); s_range = (33, 91)
rng = np.random.default_rng(seed0 + seed)
H, W = int(rng.integers(*size_range)), int(rng.integers(*size_range))
S = int(rng.integers(*s_range))
f = synthetic.make_field(H, W, S, seed=seed0 + 1000 + seed, nan_fraction=float(rng.choice([0.0, 0.01, 0.05])), margin=int(rng.integers(3, 27)))
for s_ in range(S):
    if rng.random() < 0.3:
        p = f.patches[s_][int(rng.integers(5))]
        if p.active_pixel_bitmap.size:
            p.active_pixel_bitmap &= rng.random(p.active_pixel_bitmap.shape) > 0.2
if S > 2 and rng.random() < 0.5:
    p = f.patches[int(rng.integers(S))][int(rng.integers(5))]
    (h0, h1), (w0, w1) = p.box
    p.box = ((h0, h0 - 1), (w0, w0 - 1))
    p.active_pixel_bitmap = np.zeros((0, 0), dtype=bool)
psf_K = 2
if rng.random() < 0.35:
    from celeste_jl_amd.model import render_psf
    psf_K = int(rng.choice([1, 3]))
    for row in f.patches:
        for p in row:
            w = rng.dirichlet(np.ones(psf_K) * 4)
            p.psf = np.array([[w[k], 0.2 * rng.normal(), 0.2 * rng.normal(), (1.1 + 0.8 * k) ** 2, 0.15 * rng.normal(),
                               (1.2 + 0.8 * k) ** 2] for k in range(psf_K)])
            p.stamp = render_psf(p.psf)
if rng.random() < 0.35:
    Jm = np.array([[1.0 + 0.1 * rng.normal(), 0.1 * rng.normal()], [0.1 * rng.normal(), 1.0 + 0.1 * rng.normal()]])
    Jinv = np.linalg.inv(Jm)
    for s_, row in enumerate(f.patches):
        pix = f.vp[s_, 0:2].copy()
        world = rng.normal(size=2) * 5
        f.vp[s_, 0:2] = world
        for p in row:
            p.wcs_jacobian = Jm.copy()
            p.world_center = world - Jinv @ (pix - p.pixel_center)
tg = rng.permutation(S)[:int(rng.integers(max(1, s_range[0] - 1), S + 1))].tolist()
flags = int(rng.choice([0, 4, 1, 5, 3, 7, 7, 7]))
print(tag, "seed", seed, "H W S psf_K", H, W, S, psf_K, "targets", len(tg), "flags", flags, flush=True)
ref = None
bad_calls = 0
t0 = time.time()
for it in range(iters):
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors, psf_K=psf_K)
    for rep in range(2):
        g = ctx.eval_batch(f.vp, tg, flags, raise_on_error=False)
        st = g[4]
        if ref is None and not st.any():
            ref = [None if a is None else np.array(a, copy=True) for a in g]
        if st.any():
            bad_calls += 1
            ks = [k for k in range(len(tg)) if st[k] != 0]
            print(tag, "iteration", it, "call", rep, "statuses != 0:", [(k, tg[k], int(st[k])) for k in ks], flush=True)
            for k in ks[:4]:
                print("   v", g[0][k], "ref v", None if ref is None else ref[0][k], "counters", g[3][k], None if ref is None else ref[3][k])
                if g[1] is not None:
                    print("   non-finite d entries", np.argwhere(~np.isfinite(g[1][k])).ravel().tolist())
                if g[2] is not None:
                    bh = np.argwhere(~np.isfinite(g[2][k]))
                    print("   non-finite h entries", len(bh), bh[:8].tolist())
        elif ref is not None:
            # all statuses 0: the results must be bit-identical from call to call
            for name, a, b in (("v", g[0], ref[0]), ("d", g[1], ref[1]), ("h", g[2], ref[2]), ("cnt", g[3], ref[3])):
                if a is not None and not np.array_equal(a, b):
                    w = np.argwhere(np.asarray(a) != np.asarray(b))
                    print(tag, "iteration", it, "call", rep, name, "differs from the first good call at", len(w), "places, first", w[:4].tolist(),
                          "max abs diff", float(np.nanmax(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)))), flush=True)
                    bad_calls += 1
    del ctx
print(tag, "done", iters, "iterations,", bad_calls, "bad calls, %.1f s" % (time.time() - t0), flush=True)
