"""Where an optimiser iteration goes (run through gpurun; CELESTE_MI355X_LIB = a -DOPTIM_TIMING build adds the step
kernel's per-section shader clocks).  Full batch (2000 targets) and Cyclades-sized batches (80 targets)."""
import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd import cabi

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
lib = cabi.load_library()
has_clk = hasattr(lib, "celeste_optim_clocks")
NAMES = ["chain rule: symmetrise (+ rest)", "accept/copy H", "sub-problem total", "  tridiagonalisation", "  Q'g", "  extreme eigenvalues",
         "  interior try + secular", "  model + Qy", "tail", "chain rule: loads, box / simplex transforms",
         "chain rule: simplex Jacobians, gradient", "chain rule: H J (rows, from HBM)", "chain rule: J' (H J) (columns)",
         "rejected steps (1 us = every step rejected)"]


def clocks(reset=True):
    if not has_clk:
        return None
    out = (C.c_uint64 * 16)()
    lib.celeste_optim_clocks(1 if reset else 0, out)
    return np.array(out[:], dtype=np.float64)


has_lift = hasattr(lib, "celeste_lift_clocks")
LIFT_NAMES = ["load + KL terms", "pass 1: records, brightness", "pass 2: Jacobians", "pass 3: gradient + Hessian", "KL value", "assemble + store"]


def lift_clocks():
    if not has_lift:
        return None
    out = (C.c_uint64 * 16)()
    lib.celeste_lift_clocks(1, out)
    return np.array(out[:], dtype=np.float64)


has_fused = hasattr(lib, "celeste_fused_clocks")
FUSED_NAMES = ["wait for a queue item", "chunk: descriptor, parameters, tables", "chunk: pixels + ordered adds", "chunk: fold, store, arrival",
               "step: lift", "step: chain rule + accept + sub-problem", "step: drain + queue"]


def fused_clocks():
    if not has_fused:
        return None
    out = (C.c_uint64 * 16)()
    lib.celeste_fused_clocks(1, out)
    return np.array(out[:], dtype=np.float64)


def run(tg, label, reps=3):
    cfg = cel.ElboConfig(max_iters=50)
    ctx.maximize_batch(fld.vp, tg, cfg)
    clocks(); lift_clocks(); fused_clocks()
    t0 = time.time()
    for _ in range(reps):
        vp, its, evals, elbo, st = ctx.maximize_batch(fld.vp, tg, cfg)
    dt = (time.time() - t0) / reps
    print("%s: %d targets, %.3f ms per call, max iters %d, mean %.1f -> %.1f us per Newton iteration of the longest target"
          % (label, len(tg), dt * 1e3, its.max(), its.mean(), dt * 1e6 / (its.max() + 1)))
    c = fused_clocks()
    if c is not None and c[15] > 0:
        for k, name in enumerate(FUSED_NAMES):
            n = c[14] if 1 <= k <= 3 else (c[15] if k >= 4 else c[14] + 0.0)
            print("    fused %-42s %8.0f cycles per %s (%.1f us at 2.4 GHz); total %.3g" %
                  (name, c[k] / max(n, 1), "chunk item" if k <= 3 else "step", c[k] / max(n, 1) / 2400, c[k]))
        print("    fused: %d chunk items, %d steps" % (c[14], c[15]))
    c = lift_clocks()
    if c is not None and c[15] > 0:
        for k, name in enumerate(LIFT_NAMES):
            print("    lift %-28s %8.0f cycles per workgroup (%.1f us at 2.4 GHz)" % (name, c[k] / c[15], c[k] / c[15] / 2400))
    c = clocks()
    if c is not None and c[15] > 0:
        n = c[15]
        for k, name in enumerate(NAMES):
            print("    %-28s %8.0f cycles per step-kernel workgroup (%.1f us at 2.4 GHz)" % (name, c[k] / n, c[k] / n / 2400))


if "small" not in sys.argv:
    run(np.arange(S, dtype=np.int32), "full batch")
rng = np.random.default_rng(5)
nb = [set(map(int, x)) for x in fld.neighbors]
for size in ((80,) if "small" in sys.argv else (80, 400)):
    # a conflict-free batch, as a Cyclades layer is
    order = rng.permutation(S); chosen = []; blocked = set()
    for t in order:
        if int(t) in blocked:
            continue
        chosen.append(int(t)); blocked |= nb[int(t)]; blocked.add(int(t))
        if len(chosen) == size:
            break
    run(np.array(chosen, dtype=np.int32), "conflict-free batch", reps=5)
