import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic, cabi
from oracle import oracle
seed=int(sys.argv[1]); seed0=3000; size_range=(120,240); s_range=(33,91)
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
print("H W S psf_K", H, W, S, psf_K)
problem = cabi.Problem(f.images, f.patches, f.neighbors, psf_K=psf_K)
tg = rng.permutation(S)[:int(rng.integers(max(1, s_range[0] - 1), S + 1))].tolist()
flags = int(rng.choice([0, 4, 1, 5, 3, 7, 7, 7]))
print("targets", len(tg), "flags", flags)
r = oracle.elbo_batch(problem, f.vp, tg, flags)
v=r[0]
bad=[(k,t) for k,t in enumerate(tg) if not np.isfinite(v[k])]
print("oracle non-finite targets:", bad, "status", r[4] if len(r)>4 else None)
for k,t in bad[:3]:
    print(t, f.vp[t])
    print(" is finite vp:", np.isfinite(f.vp[t]).all(), "patch sizes", [pp.active_pixel_bitmap.shape for pp in f.patches[t]])
if os.environ.get("ON_GPU"):
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors, psf_K=psf_K)
    do32 = rng.random() < 0.25          # (the test's own sequence: a single-precision evaluation first, one time in four)
    print("single-precision call first:", do32)
    if do32:
        g32 = ctx.eval_batch(f.vp, tg, flags | cabi.FLAG_FP32, raise_on_error=False)
        print("  fp32 statuses != 0:", [(k, tg[k], int(g32[4][k])) for k in range(len(tg)) if g32[4][k] != 0])
    for rep in range(3):
        g = ctx.eval_batch(f.vp, tg, flags, raise_on_error=False)
        st = g[4]
        print("fp64 call %d: statuses != 0:" % rep, [(k, tg[k], int(st[k])) for k in range(len(tg)) if st[k] != 0])
        for k in range(len(tg)):
            if st[k] != 0:
                print("  device v", g[0][k], "oracle v", r[0][k])
                if g[1] is not None:
                    bad_d = np.argwhere(~np.isfinite(g[1][k])).ravel(); print("  non-finite d entries", bad_d[:12].tolist())
                if g[2] is not None:
                    bad_h = np.argwhere(~np.isfinite(g[2][k])); print("  non-finite h entries", len(bad_h), bad_h[:6].tolist())
                print("  counters device", g[3][k], "oracle", r[3][k])
    # a fresh context, fp64 only
    ctx2 = cel.FieldContext(f.images, f.patches, f.neighbors, psf_K=psf_K)
    g = ctx2.eval_batch(f.vp, tg, flags, raise_on_error=False)
    print("fresh context, fp64: statuses != 0:", [(k, tg[k], int(g[4][k])) for k in range(len(tg)) if g[4][k] != 0])
