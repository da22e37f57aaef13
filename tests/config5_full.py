"""BASELINE.json configs[4] at full size on one GPU: 16 overlapping SDSS-size fields on a 4 x 4 grid (80 images of
2048 x 1489), ~30 k sources, all images resident, handed over through the sparse patch list of celeste_problem_t.
The fp32 component loop against the fp64 device path (all sources) and the CPU oracle (a sample) at the stated 1e-4,
fp64 against the oracle at 1e-8, and the rate of one rank's 1/8 shard.  Test infrastructure: run by
test_gpu_fullsize.py, or directly (`python tests/config5_full.py`, writes gpurun_out/config5_full.json)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(grid=(4, 4), n_src=30000, H=2048, W=1489, workers=None, log=print):
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.partition import shard_targets, estimate_time
    from oracle import oracle as orc
    workers = workers or min(16, len(os.sched_getaffinity(0)))
    out = {"grid": grid, "n_sources": n_src, "H": H, "W": W}
    t0 = time.time()
    f = synthetic.make_multifield(grid=grid, H=H, W=W, overlap=0.10, n_sources=n_src, seed=5, sparse=True, workers=workers)
    S, N = len(f.catalog), len(f.images)
    t1 = time.time()
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    t2 = time.time()
    seen = np.array([len(r.entries) for r in f.patches])
    out.update(n_images=N, generate_s=round(t1 - t0, 1), ctx_create_s=round(t2 - t1, 1), patch_entries=int(seen.sum()),
               images_per_source=[int(seen.min()), float(seen.mean()), int(seen.max())],
               neighbor_links=int(sum(len(r) for r in f.neighbors)))
    log(out, flush=True)
    tg = np.arange(S, dtype=np.int32)
    st = ctx.work_stats(tg)
    out["pixel_visits"] = st["active_pixel_visits"]
    ctx.enable_timing(True)
    ALL = 7
    res = {}
    for flags, name in ((ALL, "fp64"), (ALL | cabi.FLAG_FP32, "fp32"), (5 | cabi.FLAG_FP32, "fp32_grad")):
        ms = []
        for it in range(4):
            t = time.time()
            r = ctx.eval_batch(f.vp, tg, flags)
            wall = time.time() - t
            ms.append(ctx.last_kernel_ms())
        res[name] = r
        k = np.array(ms)[1:].mean(axis=0)
        out[name] = {"prep_ms": round(float(k[0]), 3), "pixel_ms": round(float(k[1]), 3), "lift_ms": round(float(k[2]), 3),
                     "sources_per_s_kernels": round(S / k.sum() * 1e3), "host_api_s": round(wall, 3),
                     "pixel_visits_per_s": round(st["active_pixel_visits"] / k[1] * 1e3)}
        log(name, out[name], flush=True)
    v64, d64, h64, c64, s64 = res["fp64"]
    v32, d32, h32, c32, s32 = res["fp32"]
    assert (s64 == 0).all() and (s32 == 0).all() and np.array_equal(c32, c64)
    ev = float(np.max(np.abs(v32 - v64) / np.abs(v64)))
    ed = float(max(np.abs(d32[t] - d64[t]).max() / np.abs(d64[t]).max() for t in tg))
    eh = float(max(np.abs(h32[t] - h64[t]).max() / np.abs(h64[t]).max() for t in tg))
    out["fp32_vs_fp64_device"] = {"v": ev, "d": ed, "h": eh}
    log("fp32 vs fp64 (all sources):", ev, ed, eh, flush=True)
    assert max(ev, ed, eh) <= 1e-4
    # one rank's share under the cost-balanced 8-way sharding (parallel.py): same numbers as in the full sweep
    costs = [estimate_time(row) for row in f.patches]
    shards = shard_targets(costs, 8)
    for flags, name in ((ALL | cabi.FLAG_FP32, "fp32"), (5 | cabi.FLAG_FP32, "fp32_grad")):
        ms = []
        for it in range(4):
            r = ctx.eval_batch(f.vp, shards[0], flags)
            ms.append(ctx.last_kernel_ms())
        k = np.array(ms)[1:].mean(axis=0)
        out["shard0_" + name] = {"targets": len(shards[0]), "kernels_ms": [round(float(x), 3) for x in k],
                                 "sources_per_s_kernels": round(len(shards[0]) / k.sum() * 1e3)}
        log("shard 0/8", name, out["shard0_" + name], flush=True)
        ref = res[name]
        assert np.array_equal(r[0], ref[0][shards[0]]) and np.array_equal(r[1], ref[1][shards[0]])
    # the CPU oracle on a sample (the most-imaged sources and a regular stride)
    sample = sorted(set(list(range(0, S, max(1, S // 40))) + [int(x) for x in np.argsort(-seen)[:8]]))
    t = time.time()
    ov, od, oh, ocnt, ost = orc.elbo_batch(ctx.problem, f.vp, sample, ALL)
    out["oracle_sample"] = {"n": len(sample), "seconds": round(time.time() - t, 1)}
    assert np.array_equal(c64[sample], ocnt)
    e = {"v32": float(np.max(np.abs(v32[sample] - ov) / np.abs(ov))),
         "d32": float(max(np.abs(d32[t] - od[k]).max() / np.abs(od[k]).max() for k, t in enumerate(sample))),
         "h32": float(max(np.abs(h32[t] - oh[k]).max() / np.abs(oh[k]).max() for k, t in enumerate(sample))),
         "v64": float(np.max(np.abs(v64[sample] - ov) / np.abs(ov))),
         "d64": float(max(np.abs(d64[t] - od[k]).max() / np.abs(od[k]).max() for k, t in enumerate(sample))),
         "h64": float(max(np.abs(h64[t] - oh[k]).max() / np.abs(oh[k]).max() for k, t in enumerate(sample)))}
    out["vs_oracle"] = e
    log("vs oracle:", e, flush=True)
    assert max(e["v32"], e["d32"], e["h32"]) <= 1e-4 and max(e["v64"], e["d64"], e["h64"]) <= 1e-8
    import ctypes as C
    free, total = C.c_size_t(), C.c_size_t()
    if C.CDLL("libamdhip64.so").hipMemGetInfo(C.byref(free), C.byref(total)) == 0:
        out["device_memory_GB"] = round((total.value - free.value) / 1e9, 2)
    return out


if __name__ == "__main__":
    grid = tuple(int(x) for x in os.environ.get("GRID", "4,4").split(","))
    out = run(grid, int(os.environ.get("NSRC", "30000")), int(os.environ.get("FIELD_H", "2048")),
              int(os.environ.get("FIELD_W", "1489")))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "config5_full.json"), "w"), indent=1)
    print(json.dumps(out))
