#!/usr/bin/env python3
"""bench.py -- sources/sec of the ELBO hot path (value + gradient + Hessian + KL) on MI355X.

One "step" = one sweep of elbo() over every source of one synthetic 2048x1489x5 SDSS-size field
(~2000 star+galaxy sources, BASELINE.json configs[2]), inputs resident in HBM.  With --gpus N each rank
owns one such field (sources shard with no data-path collective; weak scaling) and the per-source results
(value + 44-gradient) are all-gathered over RCCL after every sweep (the "catalog gather"), on a second stream
so that the gather of sweep k overlaps the kernels of sweep k + 1; every gather is complete before the clock stops.

Prints ONE JSON line on rank 0 (contract in the task description).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FLAGS_ALL = 1 | 2 | 4
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X FP64 vector (non-matrix) peak
FLOPS_PER_PIXEL_VISIT = 4046        # FP64 flops pixel_kernel<2, double> spends per visited pixel, counted in the ISA
                                    # (tools/count_flops.py: FMA = 2; psf_K = 2)


def profiled_traffic_bytes():
    """HBM bytes per pixel_kernel<2> launch from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE,
    separate passes, KiB -> bytes, uncorrected: see DESIGN.md 4.3).  None when no profile is present."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def build_field(H, W, n_sources, seed, cache=True):
    import pickle
    from celeste_jl_amd import synthetic
    path = "/tmp/celeste_field_%d_%d_%d_%d.pkl" % (H, W, n_sources, seed)
    if cache and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return pickle.load(f)
        except Exception:
            pass
    fld = synthetic.make_field(H, W, n_sources, seed=seed, name="synthetic_%dx%dx5_%dsrc" % (H, W, n_sources))
    if cache:
        try:
            with open(path + ".tmp%d" % os.getpid(), "wb") as f:
                pickle.dump(fld, f)
            os.replace(path + ".tmp%d" % os.getpid(), path)
        except Exception:
            pass
    return fld


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / p))))
        except Exception:
            pass
    return n


def cpu_baseline(problem, vp, targets, seconds_target=15.0):
    """The oracle (dense reference-faithful C restatement, "port") on this box's host cores, bounded sample."""
    from oracle import oracle
    cores = usable_cores()
    probe = targets[:: max(1, len(targets) // (2 * cores))][: 2 * cores]
    t0 = time.time()
    oracle.elbo_batch(problem, vp, probe, FLAGS_ALL, n_threads=cores)
    per = (time.time() - t0) / max(1, len(probe))
    n = int(min(len(targets), max(4 * cores, seconds_target / max(per, 1e-6))))
    sample = targets[:: max(1, len(targets) // n)][:n]
    t0 = time.time()
    _, _, _, _, st = oracle.elbo_batch(problem, vp, sample, FLAGS_ALL, n_threads=cores)
    dt = time.time() - t0
    assert (st == 0).all()
    # the same sample with the reduced-variable algorithm of the HIP engine on the CPU (oracle/celeste_reduced.c):
    # separates the algorithmic part of the GPU / CPU ratio from the hardware part
    oracle.reduced_elbo_batch(problem, vp, sample[: 2 * cores], FLAGS_ALL, n_threads=cores)
    t0 = time.time()
    _, _, _, _, st2 = oracle.reduced_elbo_batch(problem, vp, sample, FLAGS_ALL, n_threads=cores)
    dt2 = time.time() - t0
    assert (st2 == 0).all()
    return {"value": len(sample) / dt, "unit": "sources/sec", "cores": cores, "kind": "port",
            "reduced_algorithm_value": len(sample) / dt2,
            "sample": "%d of %d targets (every %d-th), value+grad+Hessian+KL, OpenMP dynamic over sources, %d threads "
                      "(affinity %d CPUs, cgroup quota respected), %.1f s; reduced_algorithm_value: same sample, same threads, "
                      "the engine's reduced-variable algorithm in plain C (%.1f s)"
                      % (len(sample), len(targets), max(1, len(targets) // n), cores,
                         len(os.sched_getaffinity(0)), dt, dt2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=2048)
    ap.add_argument("--width", type=int, default=1489)
    ap.add_argument("--sources", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = args.gpus > 1 or world > 1 or "RANK" in os.environ   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)

    # one field per rank (weak scaling): same shape, rank-dependent seed
    fld = build_field(args.height, args.width, args.sources, args.seed + rank)
    S = len(fld.catalog)
    ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors, device=dev.index)
    targets = np.arange(S, dtype=np.int32)
    stats = ctx.work_stats(targets)

    d_vp = torch.tensor(fld.vp, dtype=torch.float64, device=dev)
    d_tg = torch.tensor(targets, dtype=torch.int32, device=dev)
    # value / gradient outputs are double-buffered so that the catalog gather of sweep k (RCCL, on its own stream)
    # overlaps the kernels of sweep k + 1
    d_vs = [torch.zeros(S, dtype=torch.float64, device=dev) for _ in range(2)]
    d_ds = [torch.zeros(S, 44, dtype=torch.float64, device=dev) for _ in range(2)]
    d_v, d_d = d_vs[0], d_ds[0]
    d_h = torch.zeros(S, 44, 44, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(S, 2, dtype=torch.int64, device=dev)
    d_st = torch.zeros(S, dtype=torch.int32, device=dev)
    compute_stream = torch.cuda.current_stream(dev)
    stream = compute_stream.cuda_stream
    if use_dist:
        comm_stream = torch.cuda.Stream(dev)
        gather_in = [torch.zeros(S, 45, dtype=torch.float64, device=dev) for _ in range(2)]
        gather_out = torch.zeros(world * S, 45, dtype=torch.float64, device=dev)
        buf_free = [torch.cuda.Event(), torch.cuda.Event()]
        for e in buf_free:
            e.record(compute_stream)
    step_no = [0]

    def step():
        k = step_no[0] % 2
        step_no[0] += 1
        if use_dist:
            compute_stream.wait_event(buf_free[k])      # the gather that last read this buffer pair is through
        ctx.eval_batch_device(d_vp.data_ptr(), S, d_tg.data_ptr(), FLAGS_ALL, d_vs[k].data_ptr(), d_ds[k].data_ptr(),
                              d_h.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), stream)
        if use_dist:  # the catalog gather: value + 44-gradient of every source, RCCL over xGMI
            done = torch.cuda.Event()
            done.record(compute_stream)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(done)
                gather_in[k][:, 0] = d_vs[k]
                gather_in[k][:, 1:] = d_ds[k]
                dist.all_gather_into_tensor(gather_out, gather_in[k])
                buf_free[k].record(comm_stream)

    def sync():
        if use_dist:
            comm_stream.synchronize()
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert int((d_st != 0).sum().item()) == 0, "non-zero per-target status"
    pixel_visits = int(d_cnt[:, 0].sum().item())

    # secondary figure: value + gradient only (first-order mode; the headline includes the Hessian)
    FLAGS_GRAD = 1 | 4

    def step_grad():
        ctx.eval_batch_device(d_vp.data_ptr(), S, d_tg.data_ptr(), FLAGS_GRAD, d_v.data_ptr(), d_d.data_ptr(),
                              0, d_cnt.data_ptr(), d_st.data_ptr(), stream)
    for _ in range(3):
        step_grad()
    sync()
    tg0 = time.perf_counter()
    for _ in range(args.steps):
        step_grad()
    sync()
    dt_grad = time.perf_counter() - tg0

    # dominant-kernel duration: HIP events recorded by the library on the launch stream, averaged
    ctx.enable_timing(True)
    kms = []
    for _ in range(min(20, max(3, args.steps))):
        step()
        torch.cuda.synchronize(dev)
        kms.append(ctx.last_kernel_ms())
    ctx.enable_timing(False)
    kms = np.array(kms).mean(axis=0)

    # split variant: per-pixel records to HBM, then the streaming per-patch sum -- the one
    # HBM-bound kernel of the path; a measurement aid next to the fused throughput configuration
    split = None
    if world == 1:
        def step_split():
            ctx.eval_batch_device(d_vp.data_ptr(), S, d_tg.data_ptr(), FLAGS_ALL | cabi.FLAG_SPLIT, d_v.data_ptr(),
                                  d_d.data_ptr(), d_h.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), stream)
        ctx.enable_timing(True)
        sm = []
        for i in range(8):
            step_split()
            torch.cuda.synchronize(dev)
            if i >= 2:
                sm.append(ctx.last_kernel_ms() + [ctx.last_record_sum_ms()])
        ctx.enable_timing(False)
        sm = np.array(sm).mean(axis=0)
        rb = stats["record_bytes"]
        split = {"kernel": "record_sum_kernel", "bound": "hbm", "kernel_ms": float(sm[3]),
                 "algorithmic_bytes_per_launch": rb, "achieved": rb / (sm[3] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": rb / (sm[3] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "stored_record_bytes": stats["record_tiles"] * 68 * 64 * 8,
                 "traffic": (profiled_traffic_bytes() or {}).get("record_sum_bytes_per_launch"),
                 "record_write_kernel_ms": float(sm[1]), "lift_ms": float(sm[2]),
                 "note": "544 B (68 f64) per visited pixel + one 544 B result per patch; fused kernel stays the "
                         "throughput configuration"}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * S / (dt / args.steps)
        alg_bytes = stats["algorithmic_bytes"]
        traffic = profiled_traffic_bytes()
        achieved = alg_bytes / (kms[1] * 1e-3) / 1e9
        out = {
            "metric": "sources/sec (ELBO value+gradient+Hessian+KL per target source)",
            "value": value, "unit": "sources/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: synthetic %dx%dx5 SDSS-size field, %d star+galaxy "
                                   "sources per GPU, fp64, Sa=1 with value-only neighbours, psf_K=2"
                                   % (args.height, args.width, S),
                       "sources_per_gpu": S, "pixel_visits_per_sweep": pixel_visits,
                       "neighbor_links": stats["neighbor_links"], "parallelism": "sources sharded, 1 field per GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("pixel_kernel_bytes_per_launch"),
                         "traffic_source": (traffic or {}).get("source"),
                         "kernel": "pixel_kernel<2>", "kernel_ms": float(kms[1]),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "valu_utilization": (traffic or {}).get("pixel_kernel_valu_utilization"),
                         "fp64": {"achieved": FLOPS_PER_PIXEL_VISIT * stats["active_pixel_visits"] / (kms[1] * 1e-3) / 1e12,
                                  "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": FLOPS_PER_PIXEL_VISIT * stats["active_pixel_visits"] / (kms[1] * 1e-3) / 1e12
                                  / FP64_VECTOR_PEAK_TFLOPS,
                                  "flops_per_pixel_visit": FLOPS_PER_PIXEL_VISIT},
                         "note": "the fused kernel is FP64-VALU bound, not HBM- or MFMA-bound (SURVEY.md F8, DESIGN.md 4.3): "
                                 "valu_utilization = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles from the committed PMC pass; "
                                 "the HBM-bound kernel of the path is split_variant.kernel"},
            "kernels_ms": {"prep": float(kms[0]), "pixel": float(kms[1]), "lift": float(kms[2])},
            "pixel_visits_per_sec": pixel_visits / (kms[1] * 1e-3),
            "grad_only_sources_per_sec_rank0": S / (dt_grad / args.steps),
        }
        if split is not None:
            out["split_variant"] = split
        if world == 1:
            # secondary, end-to-end figure: ElboMaximize.maximize! (Newton trust region, <= 50 iterations, KL on)
            # for every source of the field, neighbours frozen; wall time includes H2D/D2H and allocations
            import celeste_jl_amd as _cel
            ctx.maximize_batch(fld.vp, targets[:64], _cel.ElboConfig(max_iters=3))  # warm-up
            t1 = time.perf_counter()
            _, its, evals, _, ost = ctx.maximize_batch(fld.vp, targets, _cel.ElboConfig())
            dt_opt = time.perf_counter() - t1
            out["optimizer"] = {"optimized_sources_per_sec": S / dt_opt, "seconds": dt_opt,
                                "mean_newton_iterations": float(its.mean()), "elbo_evaluations": int(evals.sum()),
                                "failed": int((ost != 0).sum())}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ctx.problem, fld.vp, targets)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
