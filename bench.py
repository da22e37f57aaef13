#!/usr/bin/env python3
"""bench.py -- sources/sec of the ELBO hot path (value + gradient + Hessian + KL) on MI355X.

One "step" = one sweep of elbo() over every source of the workload, inputs resident in HBM:

  --config 3 (default)  BASELINE.json configs[2] / configs[3]: ONE synthetic 2048x1489x5 SDSS-size field, ~2000
                        star+galaxy sources, fp64.
  --config 5            BASELINE.json configs[4]: 16 overlapping SDSS-size fields (80 images, ~30k sources, sparse
                        patch list), fp32 component loop (--dtype f32, the default for this config), checked inside the
                        run against the fp64 device path at 1e-4.

With --gpus N the SAME workload is sharded by source across the N ranks ("strong" scaling, the default: this is
BASELINE configs[3], the reference's one-field source list drained by N workers, ParallelRun.jl:546-607): every rank
holds the replicated images, evaluates its cost-balanced shard of the targets (estimate_time, ParallelRun.jl:45-56)
and the per-source results (value + 44-gradient) are all-gathered over RCCL after every sweep (the "catalog gather",
the only exchange of the path) on a second stream, overlapping the kernels of the next sweep; every gather is complete
before the clock stops.  --scaling weak gives every rank its own field instead (round-1 behaviour).
--backend gloo stages the gather through the host (lets two ranks share one GPU; used by the tests).
A plain `python bench.py --gpus N` (no launcher, no RANK in the environment) starts its N ranks itself under
torch.distributed.run; however it was started, the run refuses to go on unless WORLD_SIZE == --gpus, RCCL has a device per
rank, and an all_reduce over the gather's backend counts N ranks (`ranks_seen` in the line).

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
FP32_VECTOR_PEAK_TFLOPS = 157.3     # MI355X FP32 vector peak (packed)
# What a stream of independent FMAs SUSTAINS on this chip (tools/fp64_rate_probe.hip under rocprofv3 --pmc GRBM_GUI_ACTIVE,
# profiles/r04_fp64_issue_rates.txt): v_fma_f64 issues one wave-instruction per 4.94 cycles per SIMD at 2.08 GHz, v_pk_fma_f32
# one per 4.85 cycles at 2.32 GHz -- 64 lanes x 2 (x 2 packed) flop x 1024 SIMDs
FP64_SUSTAINED_FMA_TFLOPS = 64 * 2 / 4.94 * 1024 * 2.083e9 / 1e12      # 55.3
FP32_SUSTAINED_FMA_TFLOPS = 64 * 4 / 4.85 * 1024 * 2.317e9 / 1e12      # 125.2


def profile_facts():
    """Figures that only a rocprofv3 / ISA pass can give (HBM bytes per launch from the PMC passes, VALU issue
    utilisation, FP64 flops per visited pixel counted in the compiled ISA), committed under profiles/ by
    tools/profile_round.sh for the kernels as they are in this tree.  None when no profile is present."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def live_pmc(args, timeout_s=60, split=False):
    """HBM traffic and VALU issue utilisation of pixel_kernel<2, double> (split=True: HBM traffic of record_sum_kernel,
    the split variant's streaming sum) measured in THIS run: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ
    counters -- separate passes with --kernel-trace only, as the guide's HBM section prescribes) over a few sweeps of
    the same field in a child process.  None when rocprofv3 is absent, fails or times out; the line then falls back
    to the committed figures of profiles/hbm_traffic.json and says so."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = {}
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "split" if split else "fused", "--steps", "3", "--warmup", "1",
             "--height", str(args.height), "--width", str(args.width), "--sources", str(args.sources), "--seed", str(args.seed)]
    if getattr(args, "variable_psf", False):
        child.append("--variable-psf")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    kernel = "record_sum_kernel" if split else "pixel_kernel<2, double"
    passes = (("FETCH_SIZE",), ("WRITE_SIZE",)) if split else (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"))
    try:
        for counters in passes:
            with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
                cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--"] + child
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
                files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    return None
                rows = [row for f in files for row in csv.DictReader(open(f))
                        if row["Kernel_Name"].replace("void ", "").startswith(kernel)]
                if not rows:
                    return None
                gmax = max(int(row["Grid_Size"]) for row in rows)
                for c in counters:
                    vals = [float(row["Counter_Value"]) for row in rows if row["Counter_Name"] == c and int(row["Grid_Size"]) == gmax]
                    if not vals:
                        return None
                    out[c] = sum(vals) / len(vals)
    except Exception:
        return None
    if split:
        # 16 B/lane streaming loads: FETCH_SIZE counts half the bytes on gfx950 (guide, HBM section) -> doubled
        return {"record_sum_bytes_per_launch": (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0,
                "record_sum_source": "measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over 3 split "
                                     "sweeps in a child process; KiB x 1024, FETCH_SIZE x 2 (the gfx950 correction for 16 B/lane "
                                     "streaming reads)"}
    return {"pixel_kernel_bytes_per_launch": (out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0,
            "fetch_bytes": out["FETCH_SIZE"] * 1024.0, "write_bytes": out["WRITE_SIZE"] * 1024.0,
            "pixel_kernel_valu_utilization": out["SQ_ACTIVE_INST_VALU"] * 4.0 / (out["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE "
                      "(separate passes, --kernel-trace only) over 3 sweeps of the same field in a child process; "
                      "KiB x 1024, uncorrected (4-8 B/lane loads); per-launch means of the full-grid dispatches"}


def _cached(path, make, cache=True):
    import pickle
    if cache and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return pickle.load(f)
        except Exception:
            pass
    obj = make()
    if cache:
        try:
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as f:
                pickle.dump(obj, f, protocol=4)
            os.replace(tmp, path)
        except Exception:
            pass
    return obj


def build_field(H, W, n_sources, seed, cache=True, variable=False):
    from celeste_jl_amd import synthetic
    return _cached("/tmp/celeste_field_%d_%d_%d_%d%s.pkl" % (H, W, n_sources, seed, "_var" if variable else ""),
                   lambda: synthetic.make_field(H, W, n_sources, seed=seed, variable=variable,
                                                name="synthetic_%dx%dx5_%dsrc%s" % (H, W, n_sources, "_variable" if variable else "")), cache)


def build_multifield(grid, H, W, n_sources, seed, cache=True):
    from celeste_jl_amd import synthetic
    workers = max(1, min(16, usable_cores()))
    # (not pickled: 80 planes of 12 MB; generation takes ~15 s with 16 workers)
    return synthetic.make_multifield(grid=grid, H=H, W=W, overlap=0.10, n_sources=n_sources, seed=seed, sparse=True,
                                     workers=workers)


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


def read_sclk_mhz(device_index=0):
    """The shader clock (MHz) the driver reports for the HIP device right now: the starred level of pp_dpm_sclk of the card
    whose PCI address is the device's (sysfs: one file read, no tool start-up -- rocm-smi takes long enough for the chip to
    fall back to its idle state).  None when sysfs does not say."""
    import glob
    import re
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        bdf = None
    best = None
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            if bdf and bdf not in os.path.realpath(os.path.dirname(f)):
                continue
            for ln in open(f).read().splitlines():
                m = re.match(r"\s*\d+:\s*(\d+)\s*[Mm][Hh]z\s*\*", ln)
                if m:
                    best = int(m.group(1))
        except Exception:
            pass
    return best


def self_launch(n):
    """Re-run this command line as n ranks under torch.distributed.run on this node; returns the launcher's exit code.
    --standalone: the launcher itself binds a free port for the rendezvous (no bind-and-release race), on 127.0.0.1 (the
    container's hostname may not resolve).  stdout is inherited: rank 0's JSON line is this process's output."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def parity_pin_status():
    """Has the oracle been pinned by numbers the reference itself produced?  tools/reference_golden.jl writes them to
    tests/golden/ref/*.txt (one command, needs Julia 0.6 + the reference's packages: not in this image);
    tests/test_reference_outputs.py arms itself on them."""
    import glob
    return "present" if glob.glob(os.path.join(ROOT, "tests", "golden", "ref", "*.txt")) else "absent"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=(3, 5))
    ap.add_argument("--dtype", default=None, choices=("f64", "f32"))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"))
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--driver", default="ranks", choices=("ranks", "group"),
                    help="ranks: one process per GPU, torch.distributed (the default, what the launcher starts); group: ONE "
                         "process, the N devices behind the C ABI (celeste_group_*: worker threads, RCCL inside the library)")
    ap.add_argument("--group-devices", default=None,
                    help="--driver group: comma-separated HIP ordinals of the members (default 0..N-1; a repeated ordinal puts "
                         "several members on one device -- tests)")
    ap.add_argument("--height", type=int, default=2048)
    ap.add_argument("--width", type=int, default=1489)
    ap.add_argument("--sources", type=int, default=None)
    ap.add_argument("--grid", default="4,4", help="config 5: grid of overlapping fields")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed sweep and the roofline")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the rocprofv3 counter passes; report the committed profiles/hbm_traffic.json figures")
    ap.add_argument("--pmc-child", default=None, choices=("fused", "split", "optim"), help=argparse.SUPPRESS)   # the profiled child of live_pmc / tools/profile_round.sh
    ap.add_argument("--no-config5", action="store_true",
                    help="config 3, one GPU: do not append the config5 sub-record (a short full-size --config 5 --dtype f32 run)")
    ap.add_argument("--kernels-in-pass", action="store_true",
                    help="record the kernels' HIP-event durations inside the timed steps themselves (one synchronisation per "
                         "step) instead of in a pass of their own: kernels_ms then sums to at most ms_per_step by construction")
    ap.add_argument("--shard-projection", action="store_true",
                    help="one GPU: sweep rank 0's cost-balanced shard of this workload for N = 1, 2, 4, 8 ranks (fp32 and fp64) -- "
                         "a one-GPU bound on the strong-scaling efficiency (used for the config5 sub-record)")
    ap.add_argument("--variable-psf", action="store_true",
                    help="config 3 with the PSF the reference's production path uses: an SDSSPSFMap evaluated at every source "
                         "(SDSSIO.jl:239-299) -- one 51x51 stamp, hence one set of spline coefficients, per (source, band) -- plus a "
                         "varying sky plane and per-row calibration, instead of AccuracyBenchmark's ConstantPSFMap")
    ap.add_argument("--no-variable-psf", action="store_true", help="config 3, one GPU: do not append the variable_psf sub-record")
    ap.add_argument("--no-group-driver", action="store_true",
                    help="N > 1: do not let rank 0 repeat the sweep through celeste_group_* (the in-library driver) after the ranks' run")
    ap.add_argument("--check-dir", default=None,
                    help="every rank writes its gathered (v, d) of the last sweep to <dir>/rank<r>.npz (tests)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.driver == "group":
        if int(os.environ.get("RANK", "0")) != 0:
            return               # (started under a launcher: the group lives in ONE process -- rank 0's)
        return group_main(args)
    if args.gpus > 1 and "RANK" not in os.environ and not args.pmc_child:
        # a plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, the same launcher and
        # arguments the driver uses) -- the line must never describe fewer ranks than --gpus asked for
        sys.exit(self_launch(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        # never mislabel: the line's n_gpus is the number of ranks that swept, and it must be what --gpus asked for
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%s: launch %d ranks (or run the plain "
                         "`python bench.py --gpus %d`, which starts them itself)"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "1"), args.gpus, args.gpus))
    # Libraries write to file descriptor 1 behind Python's back (RCCL prints a five-line version banner when its first
    # communicator comes up): until the result line is due, fd 1 points at stderr, so that rank 0's stdout carries the
    # one JSON line and nothing else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.dtype is None:
        args.dtype = "f32" if args.config == 5 else "f64"
    if args.steps is None:
        args.steps = 200 if args.config == 3 else 10   # (config 3: a 0.7 ms step -- 50 steps gave 4 % box-to-box noise)
    if args.sources is None:
        args.sources = 2000 if args.config == 3 else 30000
    if args.seed is None:
        args.seed = 3 if args.config == 3 else 5

    import torch
    import torch.distributed as dist
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    from celeste_jl_amd.parallel import DeviceShardedSweep
    from celeste_jl_amd.partition import estimate_time

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    cabi.load_library()   # before any worker process is forked (the image generator of --config 5 uses a pool)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = args.gpus > 1 or world > 1 or "RANK" in os.environ   # launched by torch.distributed.run
    n_dev = torch.cuda.device_count()
    ranks_seen = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl" and (world > n_dev or local_rank >= n_dev):
            raise SystemExit("%d ranks on %d visible GPU(s): RCCL needs one device per rank; "
                             "use --backend gloo to share a device" % (world, n_dev))
        dev_index = local_rank % n_dev
        torch.cuda.set_device(dev_index)
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        # the ranks that really take part, counted over the backend the gather uses
        ones = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", dev_index) if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks (all_reduce saw %d), --gpus says %d"
                             % (dist.get_world_size(), ranks_seen, args.gpus))
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    strong = args.scaling == "strong"
    flags = FLAGS_ALL | (cabi.FLAG_FP32 if args.dtype == "f32" else 0)

    # ---- the workload: the same field(s) on every rank (strong), or one per rank (weak) ----
    seed = args.seed if strong else args.seed + rank

    def make():
        if args.config == 3:
            return build_field(args.height, args.width, args.sources, seed, variable=args.variable_psf)
        grid = tuple(int(x) for x in args.grid.split(","))
        return build_multifield(grid, args.height, args.width, args.sources, seed)
    if use_dist and strong and args.config == 3:
        if rank == 0:      # one rank renders the field and caches it; the others load the cache
            fld = make()
        dist.barrier()
        if rank != 0:
            fld = make()
    else:
        fld = make()
    S = len(fld.catalog)
    ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors, device=dev.index)
    # what celeste_ctx_create costs once the process has its runtime and code object (the first call of a process carries
    # ~0.1 s of one-time HIP initialisation): a second context of the same marshalled problem, created and released (rank 0)
    ctx_create_ms = None
    if rank == 0 and not args.pmc_child:
        c2 = cel.FieldContext(fld.images, None, fld.neighbors, device=dev.index, problem=ctx.problem)
        ctx_create_ms = c2.create_ms
        c2.close()
    targets = np.arange(S, dtype=np.int32)
    costs = [estimate_time(row) for row in fld.patches]
    sweep = DeviceShardedSweep(ctx, targets, costs, rank, world, flags, backend=args.backend if use_dist else None,
                               shards=None if strong else [list(range(S))] * world)
    mine = sweep.mine
    stats = ctx.work_stats(mine)
    d_vp = torch.tensor(fld.vp, dtype=torch.float64, device=dev)

    def sync():
        sweep.wait()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        sweep.step(d_vp.data_ptr())
    sync()
    kms_in_pass = []
    if args.kernels_in_pass:      # (the config-5 sub-record: steps of several ms, one synchronisation per step costs nothing)
        ctx.enable_timing(True)
    # per-step durations of the timed loop: one HIP event per step boundary on the stream the launch chain runs on (recorded,
    # never waited for, inside the loop: ~2 us of host time per step; read after the clock has stopped)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_ev[i].record(sweep.compute_stream)
        sweep.step(d_vp.data_ptr())
        if args.kernels_in_pass:
            sweep.wait()
            kms_in_pass.append(ctx.last_kernel_ms())
    step_ev[args.steps].record(sweep.compute_stream)
    sync()
    dt = time.perf_counter() - t0
    sclk_after = read_sclk_mhz(dev.index)      # (once, after the clock has stopped: what the chip reports right behind the loop)
    step_ms = np.array([step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)])
    if args.kernels_in_pass:
        ctx.enable_timing(False)
    if args.pmc_child == "split":   # the profiled child of live_pmc(split=True): a few sweeps of the split variant
        d_v0 = torch.zeros(S, dtype=torch.float64, device=dev)
        d_d0 = torch.zeros(S, 44, dtype=torch.float64, device=dev)
        for _ in range(3):
            ctx.eval_batch_device(d_vp.data_ptr(), S, sweep.d_tg.data_ptr(), FLAGS_ALL | cabi.FLAG_SPLIT, d_v0.data_ptr(),
                                  d_d0.data_ptr(), sweep.d_h.data_ptr(), sweep.d_cnt.data_ptr(), sweep.d_st.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        return
    if args.pmc_child == "optim":
        # the kernels that dominate a run with extras, alone under the profiler (tools/profile_round.sh): maximize! of every
        # source (lock-step driver: optim_step_kernel + the sweep's kernels), of a Cyclades-sized layer (optim_fused_kernel<false>)
        # and the joint-inference schedule as one dataflow launch (optim_fused_kernel<true>)
        from celeste_jl_amd.infer import one_node_joint_infer
        ctx.maximize_batch(fld.vp, targets, cel.ElboConfig())
        layer = conflict_free_layer(fld, S, 80)
        for _ in range(3):
            ctx.maximize_batch(fld.vp, layer, cel.ElboConfig())
        one_node_joint_infer(ctx, fld.catalog, [int(t) for t in targets], fld.neighbors)
        torch.cuda.synchronize(dev)
        return
    if args.pmc_child:     # the profiled child of live_pmc(): the sweeps are all rocprofv3 needs
        return
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    g_v, g_d, st_loc, cnt_loc = sweep.results()
    assert int((st_loc != 0).sum()) == 0, "non-zero per-target status"
    if strong:
        assert np.isfinite(g_v).all() and np.isfinite(g_d).all()
    if args.check_dir:
        os.makedirs(args.check_dir, exist_ok=True)
        np.savez(os.path.join(args.check_dir, "rank%d.npz" % rank), v=g_v, d=g_d, mine=mine, h=sweep.hessians())
    pixel_visits_local = int(cnt_loc[:, 0].sum())

    # ---- dominant-kernel duration: HIP events recorded by the library on the launch stream, averaged ----
    ctx.enable_timing(True)
    kms = []
    blk0 = sweep.blocks[0].data_ptr()
    for _ in range(min(20, max(3, args.steps))):
        if sweep.n > 0:
            ctx.eval_batch_device(d_vp.data_ptr(), sweep.n, sweep.d_tg.data_ptr(), flags, blk0, blk0 + 8 * sweep.width,
                                  sweep.d_h.data_ptr() if sweep.d_h is not None else 0, sweep.d_cnt.data_ptr(),
                                  sweep.d_st.data_ptr(), sweep.compute_stream.cuda_stream)
        sweep.wait()
        kms.append(ctx.last_kernel_ms())
    ctx.enable_timing(False)
    kms = np.array(kms_in_pass if kms_in_pass else kms).mean(axis=0)
    sync()
    # the same loop once more when the timed one was short: what a step costs once the chip has left its idle power state
    # (a secondary figure: `value` stays the K steps the caller asked for)
    steady = None
    if args.steps < 100 and args.config == 3 and not args.pmc_child:
        for _ in range(20):
            sweep.step(d_vp.data_ptr())
        sync()
        ts0 = time.perf_counter()
        for _ in range(200):
            sweep.step(d_vp.data_ptr())
        sync()
        steady = {"ms_per_step": (time.perf_counter() - ts0) / 200 * 1e3, "steps": 200, "warmup": 20,
                  "sclk_mhz_after_loop": read_sclk_mhz(dev.index),
                  "note": "the timed loop repeated with 200 steps behind everything else of this run (rank 0's clock)"}

    # N > 1: the OTHER driver of the same sweep in the same run -- rank 0 alone, ONE process, the N devices behind the C ABI
    # (celeste_group_*: worker threads, ncclCommInitAll + ncclAllGather inside the library) -- while the other ranks wait on the
    # host (the rendezvous store, not a collective: their GPUs stay idle for the group's members)
    group_rec = None
    if use_dist and world > 1 and not args.no_group_driver and not args.pmc_child:
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            try:
                if n_dev < world and args.backend == "nccl":
                    raise RuntimeError("rank 0 sees %d device(s), the group needs %d" % (n_dev, world))
                group_rec = group_driver_child(args, [r % n_dev for r in range(world)])
            except Exception as e:      # (the ranks' line does not depend on it)
                group_rec = {"error": repr(e)}
            store.set("celeste_group_driver_done", "1")
        else:
            store.wait(["celeste_group_driver_done"])
        sync()

    # N > 1: every rank's sweep of its own shard WITHOUT the catalog gather (same launches, wall clock between two
    # device synchronisations) -- what separates load imbalance / small-shard latency from the cost of the exchange
    sweep_only_ms = None
    if use_dist and world > 1:
        cs = sweep.compute_stream.cuda_stream

        def shard_only():
            if sweep.n > 0:
                ctx.eval_batch_device(d_vp.data_ptr(), sweep.n, sweep.d_tg.data_ptr(), flags, blk0, blk0 + 8 * sweep.width,
                                      sweep.d_h.data_ptr() if sweep.d_h is not None else 0, sweep.d_cnt.data_ptr(),
                                      sweep.d_st.data_ptr(), cs)
        for _ in range(3):
            shard_only()
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(args.steps):
            shard_only()
        torch.cuda.synchronize(dev)
        mine_ms = torch.tensor([(time.perf_counter() - ts) / args.steps * 1e3], dtype=torch.float64,
                               device=dev if args.backend == "nccl" else "cpu")
        allms = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(allms, mine_ms)
        sweep_only_ms = [float(t.item()) for t in allms]
        sync()

    extras = world == 1 and not args.no_extras
    d_tg = sweep.d_tg
    d_v = torch.zeros(S, dtype=torch.float64, device=dev)
    d_d = torch.zeros(S, 44, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(S, 2, dtype=torch.int64, device=dev)
    d_st = torch.zeros(S, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    out_extra = {}
    if extras:
        # secondary figure: value + gradient only (first-order mode; the headline includes the Hessian)
        fl_grad = (flags & ~2)

        def step_grad():
            ctx.eval_batch_device(d_vp.data_ptr(), S, d_tg.data_ptr(), fl_grad, d_v.data_ptr(), d_d.data_ptr(),
                                  0, d_cnt.data_ptr(), d_st.data_ptr(), stream)
        for _ in range(3):
            step_grad()
        torch.cuda.synchronize(dev)
        tg0 = time.perf_counter()
        for _ in range(args.steps):
            step_grad()
        torch.cuda.synchronize(dev)
        out_extra["grad_only_sources_per_sec_rank0"] = S / ((time.perf_counter() - tg0) / args.steps)

    if args.config == 5 and world == 1 and args.dtype == "f32":
        # the configuration's stated tolerance: fp32 component loop within 1e-4 of the fp64 device path, every source
        d_h64 = torch.zeros(S, 44, 44, dtype=torch.float64, device=dev)
        ctx.eval_batch_device(d_vp.data_ptr(), S, d_tg.data_ptr(), FLAGS_ALL, d_v.data_ptr(), d_d.data_ptr(),
                              d_h64.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        v64, d64 = d_v.cpu().numpy(), d_d.cpu().numpy()
        ev = float(np.max(np.abs(g_v - v64) / np.abs(v64)))
        ed = float(np.max(np.abs(g_d - d64).max(axis=1) / np.abs(d64).max(axis=1)))
        h32 = sweep.d_h
        eh = float(((h32 - d_h64).abs().amax(dim=(1, 2)) / d_h64.abs().amax(dim=(1, 2))).max().item())
        assert max(ev, ed, eh) <= 1e-4, (ev, ed, eh)
        out_extra["fp32_vs_fp64_device"] = {"v": ev, "d": ed, "h": eh, "tolerance": 1e-4, "sources_checked": S}
        # the same sweep with the fp64 component loop: what the precision mode buys
        ctx.enable_timing(True)
        k64 = []
        for _ in range(3):
            ctx.eval_batch_device(d_vp.data_ptr(), S, d_tg.data_ptr(), FLAGS_ALL, d_v.data_ptr(), d_d.data_ptr(),
                                  d_h64.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            k64.append(ctx.last_kernel_ms()[1])
        ctx.enable_timing(False)
        out_extra["fp64_pixel_kernel_ms"] = float(np.mean(k64[1:]))
        out_extra["fp32_speedup_over_fp64_pixel_kernel"] = float(np.mean(k64[1:]) / kms[1])
        del d_h64

    if args.shard_projection and world == 1:
        out_extra["shard_projection"] = shard_projection(ctx, fld, targets, costs, d_vp, dev,
                                                         (("f32", FLAGS_ALL | cabi.FLAG_FP32), ("f64", FLAGS_ALL)) if args.config == 5
                                                         else (("f64", FLAGS_ALL),))

    # split variant: per-pixel records to HBM, then the streaming per-patch sum -- the one
    # HBM-bound kernel of the path; a measurement aid next to the fused throughput configuration
    split = None
    facts = profile_facts() or {}
    # HBM bytes and VALU utilisation of the dominant kernel: measured now when rocprofv3 can run here, else the
    # committed figures (the flop count per pixel visit always comes from the ISA of the tree, tools/count_flops.py)
    pmc_live = pmc_split = None
    if world == 1 and rank == 0 and args.config == 3 and args.dtype == "f64" and (extras or args.variable_psf) and not args.no_live_pmc:
        pmc_live = live_pmc(args)
        if pmc_live:
            facts = dict(facts, **pmc_live)
        pmc_split = live_pmc(args, split=True) if extras else None
        if pmc_split:
            facts = dict(facts, **pmc_split)
    if extras and args.config == 3:
        d_h = sweep.d_h

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
                 "traffic": facts.get("record_sum_bytes_per_launch"), "traffic_measured_in_this_run": bool(pmc_split),
                 "traffic_source": facts.get("record_sum_source", facts.get("source")),
                 "record_write_kernel_ms": float(sm[1]), "lift_ms": float(sm[2]),
                 "note": "544 B (68 f64) per visited pixel + one 544 B result per patch; fused kernel stays the "
                         "throughput configuration"}

    # whole-job figures need every rank's share
    if use_dist:
        tot = torch.tensor([float(len(mine)), float(pixel_visits_local), float(stats["algorithmic_bytes"])],
                           dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        per_rank = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(per_rank, tot)
        per_rank = [[float(x) for x in t.cpu()] for t in per_rank]
    else:
        per_rank = [[float(len(mine)), float(pixel_visits_local), float(stats["algorithmic_bytes"])]]

    if rank == 0:
        n_total = int(sum(p[0] for p in per_rank))           # sources evaluated per step by the whole job
        ms_per_step = dt / args.steps * 1e3
        value = n_total / (dt / args.steps)
        # the star spline coefficients (53 x 53 f64 = 22 472 B per stamp) count once per band per sweep when
        # the PSF map is constant, per patch otherwise: the context's distinct stamps are exactly that
        spline_bytes = int(ctx.problem.c.n_stamps) * 53 * 53 * 8
        alg_bytes = stats["algorithmic_bytes"] + spline_bytes   # of rank 0's launch
        achieved = alg_bytes / (kms[1] * 1e-3) / 1e9
        fpp = facts.get("flops_per_pixel_visit_f32" if args.dtype == "f32" else "flops_per_pixel_visit")
        peak_fl = FP32_VECTOR_PEAK_TFLOPS if args.dtype == "f32" else FP64_VECTOR_PEAK_TFLOPS
        kname = "pixel_kernel<2, %s>" % ("float" if args.dtype == "f32" else "double")
        if args.config == 3:
            workload = ("BASELINE.json configs[%d]: synthetic %dx%dx5 SDSS-size field, %d star+galaxy sources%s, fp64, "
                        "Sa=1 with value-only neighbours, psf_K=2%s"
                        % (2 if world == 1 else 3, args.height, args.width, S,
                           "" if world == 1 else (" sharded by source across %d GPUs" % world if strong
                                                  else " per GPU (one field per rank)"),
                           ", SDSSPSFMap: one PSF stamp per (source, band), varying sky and calibration" if args.variable_psf else ""))
        else:
            workload = ("BASELINE.json configs[4]: %s grid of overlapping %dx%dx5 fields (%d images), %d sources in the "
                        "sparse patch list, %s component loop%s, tolerance 1e-4 vs fp64"
                        % (args.grid.replace(",", "x"), args.height, args.width, len(fld.images), S,
                           "fp32" if args.dtype == "f32" else "fp64",
                           "" if world == 1 else ", sharded by source across %d GPUs" % world))
        out = {
            "metric": "sources/sec (ELBO value+gradient+Hessian+KL per target source)",
            "value": value, "unit": "sources/sec", "n_gpus": world, "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "sources_per_step": n_total,
                       "shard_sizes": [int(p[0]) for p in per_rank],
                       "shard_pixel_visits": [int(p[1]) for p in per_rank],
                       "catalog_gather_bytes_per_step": sweep.gather_bytes, "gather_backend": sweep.backend,
                       "pixel_visits_per_sweep": int(sum(p[1] for p in per_rank)),
                       "neighbor_links_rank0": stats["neighbor_links"],
                       "parallelism": ("one field, targets sharded by estimated cost, images replicated, one all-gather of "
                                       "(v, d) per sweep" if strong else "one field per GPU, all-gather of (v, d) per sweep")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": facts.get("pixel_kernel_bytes_per_launch") if (world == 1 and args.config == 3 and
                                                                                     args.dtype == "f64") else None,
                         "traffic_source": facts.get("source"), "traffic_measured_in_this_run": bool(pmc_live),
                         "kernel": kname, "kernel_ms": float(kms[1]),
                         "algorithmic_bytes_per_launch": alg_bytes, "spline_coefficient_bytes": spline_bytes,
                         "psf_stamps": int(ctx.problem.c.n_stamps),
                         "valu_utilization": facts.get("pixel_kernel_valu_utilization") if args.dtype == "f64" else None},
            "timing_method": ("value / ms_per_step: wall clock around K sweeps (launch chain + catalog gather) between two device "
                              "synchronisations + barriers, max over ranks; kernels_ms / roofline.kernel_ms: HIP events recorded by "
                              "the library on the launch stream, " + ("inside the timed steps (one synchronisation per step)"
                                                                     if args.kernels_in_pass else "in a pass of their own after the timed region")),
            "parity_pin": parity_pin_status(),
            # celeste_ctx_create (the C call, second context of the process): image planes up, every PSF stamp conditioned +
            # spline-prefiltered on the device (the ImagePatch constructor, imaged_sources.jl:97-107), patch / neighbour / work tables
            "context_create_ms": ctx_create_ms,
            "kernels_ms": {"prep": float(kms[0]), "pixel": float(kms[1]), "lift": float(kms[2])},
            # where ms_per_step goes (rank 0): the steps of the timed loop one by one, the part of a step no kernel accounts
            # for, and the shader clock the chip reported right behind the loop -- a short run (--steps 20 after --warmup 5 is
            # 18 ms of work) measures a chip that is still leaving its idle power state; step_ms shows it
            "step_ms": {"min": float(step_ms.min()), "p50": float(np.median(step_ms)), "max": float(step_ms.max()),
                        "first": float(step_ms[0]), "last": float(step_ms[-1]),
                        "first_quarter_mean": float(step_ms[:max(1, len(step_ms) // 4)].mean()),
                        "last_quarter_mean": float(step_ms[-max(1, len(step_ms) // 4):].mean()),
                        "sum": float(step_ms.sum()), "wall_ms": dt * 1e3,
                        "source": "HIP events at the step boundaries of the timed loop, on the launch stream"},
            "gaps_ms": float(ms_per_step - (kms[0] + kms[1] + kms[2])),
            "sclk_mhz_after_loop": sclk_after,
            "pixel_visits_per_sec_rank0": pixel_visits_local / (kms[1] * 1e-3),
        }
        if fpp:
            # executed flops: the compiled ISA's count per visit x the (pixel, active source) pairs the kernel counted --
            # not the patches' areas, which include the last-column pixels that skip the component loop
            fl = fpp * pixel_visits_local / (kms[1] * 1e-3) / 1e12
            sus = FP32_SUSTAINED_FMA_TFLOPS if args.dtype == "f32" else FP64_SUSTAINED_FMA_TFLOPS
            hbm = out["roofline"]
            # the binding roofline on top: the fused kernel is vector-ALU bound, so `frac` is
            # the executed-flop fraction of the nominal vector peak; the HBM figures the metric's text asks for sit in `hbm`
            out["roofline"] = {"bound": "fp32_valu" if args.dtype == "f32" else "fp64_valu",
                               "achieved": fl, "peak": peak_fl, "unit": "TFLOP/s", "frac": fl / peak_fl,
                               "traffic": hbm["traffic"], "kernel": kname, "kernel_ms": float(kms[1]),
                               "sustained_fma_peak": sus, "frac_of_sustained": fl / sus,
                               "sustained_source": "profiles/r04_fp64_issue_rates.txt: independent FMAs issue every 4.9 cycles per "
                                                   "SIMD at the clock the chip holds under them, 0.70 (fp64) / 0.80 (packed fp32) of "
                                                   "the nominal peak (self-measured: `frac` against the nominal peak is the figure "
                                                   "of record)",
                               "flops_per_pixel_visit": fpp, "pixel_visits": pixel_visits_local, "dtype": args.dtype,
                               "flops_source": "tools/count_flops.py on the compiled ISA (profiles/hbm_traffic.json; "
                                               "tests/test_dpp_hazard.py holds the file to the tree); every basic block LLVM's loop "
                                               "annotations place in the pixel loop, guarded blocks (star spline, own-geometry region) "
                                               "included once per visit: a static count, an upper bound of the executed one by the "
                                               "rare trips that skip them",
                               "instruction_mix": facts.get("instruction_mix_f32" if args.dtype == "f32" else "instruction_mix"),
                               "valu_utilization": hbm["valu_utilization"],
                               "hbm": hbm,
                               "note": "valu_utilization = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles (hbm.traffic_source says where the "
                                       "counters come from); the HBM-bound kernel of the path is split_variant.kernel; figures are "
                                       "rank 0's launch"}
            out["roofline"]["valu"] = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "sustained_fma_peak",
                                                                       "frac_of_sustained", "flops_per_pixel_visit", "pixel_visits",
                                                                       "dtype", "instruction_mix")}   # (the rounds-1..4 location)
        if steady is not None:
            out["steady_state"] = steady
        if group_rec is not None:
            out["group_driver"] = group_rec
        if sweep_only_ms is not None:
            out["config"]["sweep_ms_without_gather_per_rank"] = sweep_only_ms
            out["config"]["gather_and_imbalance_ms"] = ms_per_step - max(sweep_only_ms)
        out.update(out_extra)
        if split is not None:
            out["split_variant"] = split
        if extras:
            out.update(secondary_figures(ctx, fld, targets, args, costs))
        if extras and args.config == 3 and not args.variable_psf and not args.no_variable_psf and args.height >= 2048:
            out["variable_psf"] = variable_psf_record(args, out)
        if extras and args.config == 3 and not args.no_config5 and args.height >= 2048:
            out["config5"] = config5_record(args)
        if not args.no_cpu_baseline and world == 1:
            tg_cpu = targets if args.config == 3 else targets[:: max(1, S // 2000)]
            out["cpu_baseline"] = cpu_baseline(ctx.problem, fld.vp, tg_cpu)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)    # (teardown messages, if any, stay off stdout as well)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def group_main(args):
    """--driver group: the same sweep, the same line, from ONE process through celeste_group_* -- the reference's own shape
    (N workers inside one process, ParallelRun.jl:546-607).  `ranks_seen` is ncclCommCount of the library's communicator."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.dtype is None:
        args.dtype = "f32" if args.config == 5 else "f64"
    if args.steps is None:
        args.steps = 200 if args.config == 3 else 10
    if args.sources is None:
        args.sources = 2000 if args.config == 3 else 30000
    if args.seed is None:
        args.seed = 3 if args.config == 3 else 5
    import torch
    from celeste_jl_amd import cabi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    cabi.load_library()
    devices = [int(x) for x in args.group_devices.split(",")] if args.group_devices else list(range(args.gpus))
    if len(devices) != args.gpus and not args.group_devices:
        raise SystemExit("--gpus %d but %d devices" % (args.gpus, len(devices)))
    if max(devices) >= torch.cuda.device_count():
        raise SystemExit("--driver group over devices %s: only %d visible" % (devices, torch.cuda.device_count()))
    flags = FLAGS_ALL | (cabi.FLAG_FP32 if args.dtype == "f32" else 0)
    if args.config == 3:
        fld = build_field(args.height, args.width, args.sources, args.seed, variable=args.variable_psf)
    else:
        fld = build_multifield(tuple(int(x) for x in args.grid.split(",")), args.height, args.width, args.sources, args.seed)
    out = group_measure(args, fld, devices, flags)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out))
    sys.stdout.flush()
    os.dup2(2, 1)


GROUP_CHILD_TIMEOUT_S = 300


def group_driver_child(args, devices):
    """The group driver's line for the ranks' run at N > 1, from a CHILD process of rank 0 (`bench.py --driver group` over the
    same devices, outside the launcher's environment) under a time limit: the in-library driver has never met more than one
    real device, and whatever it does on its first node -- a communicator that does not come up, a
    crash -- must cost the ranks' line a `group_driver.error`, not the line itself."""
    import subprocess
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
            "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, os.path.abspath(__file__), "--driver", "group", "--gpus", str(len(devices)),
           "--group-devices", ",".join(str(d) for d in devices), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--config", str(args.config), "--dtype", args.dtype, "--height", str(args.height), "--width", str(args.width),
           "--sources", str(args.sources), "--seed", str(args.seed), "--grid", args.grid, "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=GROUP_CHILD_TIMEOUT_S)
    except subprocess.TimeoutExpired:
        return {"error": "the group driver's process did not finish within %d s (killed)" % GROUP_CHILD_TIMEOUT_S}
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "the group driver's process ended with code %d" % r.returncode, "stderr_tail": r.stderr[-600:]}
    rec = json.loads(lines[-1])
    rec["process"] = "child of rank 0 (time limit %d s)" % GROUP_CHILD_TIMEOUT_S
    return rec


def group_measure(args, fld, devices, flags, check=True):
    """The sweep of `fld` through a celeste_group_t over `devices`: W warm-up sweeps, K timed ones between two
    celeste_group_sweep_wait, then a pass with HIP-event timing.  Returns the bench line of the group driver (also nested as
    `group_driver` in the ranks' line at N > 1)."""
    from celeste_jl_amd.group import FieldGroup
    S = len(fld.catalog)
    targets = np.arange(S, dtype=np.int32)
    g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=devices)
    info = g.info()
    g.plan(fld.vp, targets, flags)                       # everything resident: table, shards, gather blocks
    sizes, costs = g.shard_sizes()
    for _ in range(args.warmup):
        g.sweep()
    g.wait()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.sweep()
    g.wait()
    dt = time.perf_counter() - t0
    v, d, _, cnt, st = g.results(hessians=False)
    assert (st == 0).all() and np.isfinite(v).all() and np.isfinite(d).all()
    if args.check_dir and check:
        os.makedirs(args.check_dir, exist_ok=True)
        np.savez(os.path.join(args.check_dir, "group.npz"), v=v, d=d)
    g.enable_timing(True)
    kms, ev_ms, ga_ms = [], [], []
    for _ in range(min(20, max(3, args.steps))):
        g.sweep()
        g.wait()
        kms.append(g.last_kernel_ms(0))
        a, b = g.last_sweep_ms()
        ev_ms.append(a); ga_ms.append(b)
    g.enable_timing(False)
    kms = np.array(kms).mean(axis=0)
    ev_ms, ga_ms = np.array(ev_ms).mean(axis=0), np.array(ga_ms).mean(axis=0)
    pixel_visits = int(cnt[:, 0].sum())
    facts = profile_facts() or {}
    fpp = facts.get("flops_per_pixel_visit_f32" if args.dtype == "f32" else "flops_per_pixel_visit")
    peak_fl = FP32_VECTOR_PEAK_TFLOPS if args.dtype == "f32" else FP64_VECTOR_PEAK_TFLOPS
    # member 0's share of the visits (its launch is the one timed), by shard cost
    visits0 = pixel_visits * costs[0] / max(1, sum(costs))
    n = len(devices)
    enq, aborted = g.collectives()
    out = {"metric": "sources/sec (ELBO value+gradient+Hessian+KL per target source)",
           "value": S / (dt / args.steps), "unit": "sources/sec", "n_gpus": info["n_devices"], "ranks_seen": info["rccl_ranks"],
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "timing_method": "wall clock around K celeste_group_sweep calls between two celeste_group_sweep_wait (all members' "
                            "streams drained, every catalog gather complete); kernels_ms: HIP events in a pass of its own",
           "config": {"workload": ("BASELINE.json configs[%d]: synthetic %dx%dx5 field, %d sources, %s, one process, %d group "
                                   "member(s) on devices %s" % (2 if n == 1 else 3, args.height, args.width, S, args.dtype, n, devices))
                      if args.config == 3 else "BASELINE.json configs[4] through a device group of %d member(s) on %s" % (n, devices),
                      "driver": "group (celeste_group_*: one process, worker thread + stream per member, images replicated, "
                                "targets sharded by estimate_time, one all-gather of (v, d, counters, status) per sweep)",
                      "sources_per_step": S, "shard_sizes": sizes, "shard_costs": costs,
                      "gather_backend": info["exchange"], "members": n, "devices": devices,
                      "collectives_enqueued_per_member": enq, "aborted": aborted,
                      "catalog_gather_bytes_per_step": n * (max(sizes) * 47 + (max(sizes) + 1) // 2) * 8,
                      "pixel_visits_per_sweep": pixel_visits,
                      "member_eval_ms": [float(x) for x in ev_ms], "member_gather_ms": [float(x) for x in ga_ms]},
           "kernels_ms": {"prep": float(kms[0]), "pixel": float(kms[1]), "lift": float(kms[2])}}
    if fpp:
        fl = fpp * visits0 / (kms[1] * 1e-3) / 1e12
        out["roofline"] = {"bound": "fp32_valu" if args.dtype == "f32" else "fp64_valu", "achieved": fl, "peak": peak_fl,
                           "unit": "TFLOP/s", "frac": fl / peak_fl, "traffic": None, "kernel_ms": float(kms[1]),
                           "kernel": "pixel_kernel<2, %s>" % ("float" if args.dtype == "f32" else "double"),
                           "note": "member 0's launch; its share of the pixel visits by shard cost"}
    if check and not args.no_cpu_baseline and n == 1 and args.config == 3:
        out["cpu_baseline"] = cpu_baseline(g.problem, fld.vp, targets)
    g.close()
    return out


def conflict_free_layer(fld, S, n):
    """n sources no two of which are neighbours (a Cyclades-sized layer), drawn with a fixed seed"""
    rng = np.random.default_rng(5)
    nb = [set(map(int, x)) for x in fld.neighbors]
    chosen, blocked = [], set()
    for t in rng.permutation(S):
        if int(t) in blocked:
            continue
        chosen.append(int(t)); blocked |= nb[int(t)]; blocked.add(int(t))
        if len(chosen) == min(n, S):
            break
    return np.array(chosen, dtype=np.int32)


def secondary_figures(ctx, fld, targets, args, costs):
    """Figures next to the headline (rank 0, one GPU) that say how the engine behaves where the reference actually calls
    it: the host-pointer API the Julia shim binds, one elbo() per call (the literal drop-in of ElboMaximize.jl:166),
    maximize! for every source and for a Cyclades-sized layer, joint inference, and the sweep time of rank 0's shard for
    N = 2, 4, 8 ranks (what strong scaling can reach when the catalog gather hides under the next sweep)."""
    import torch
    import celeste_jl_amd as cel
    from celeste_jl_amd.partition import shard_targets
    out = {}
    S = len(targets)
    # host-pointer API (celeste_elbo_eval_batch): vp H2D, kernels, D2H of v / d / h / counters / status
    for packed in (False, True):
        fl = FLAGS_ALL | (cel.cabi.FLAG_PACKED_HESS if packed else 0)
        ctx.eval_batch(fld.vp, targets, fl)
        t1 = time.perf_counter()
        reps = 5 if S <= 4000 else 2
        for _ in range(reps):
            ctx.eval_batch(fld.vp, targets, fl)
        out["host_api_sources_per_sec" + ("_packed_hessian" if packed else "")] = S * reps / (time.perf_counter() - t1)
    if args.config != 3:
        return out
    # celeste_elbo_eval: ONE target per call (value + gradient + Hessian + KL), host pointers
    lat = []
    for k in range(210):
        t = int(targets[(k * 97) % S])
        t1 = time.perf_counter()
        ctx.eval_batch(fld.vp, [t], FLAGS_ALL, pinned=False)
        lat.append(time.perf_counter() - t1)
    lat = np.sort(np.array(lat[10:]))
    out["single_call_latency_us"] = {"median": float(np.median(lat) * 1e6), "p90": float(lat[int(0.9 * len(lat))] * 1e6),
                                     "calls": int(len(lat)), "what": "celeste_elbo_eval for one target (value + gradient + Hessian "
                                     "+ KL) on the whole-field context (2000 sources: the 704 KB table goes up with every call), "
                                     "results down included"}
    # the literal drop-in (ElboMaximize.jl:166 inside process_source, ParallelRun.jl:468-488): one context per source over
    # ElboArgs(images, patches[[s; neighbors], :], [1]) on a shared image handle -- the table is the source and its neighbours
    iset = cel.cabi.ImageSet(fld.images)
    lat, t_create, t_first, t_close = [], [], [], []
    for k in range(24):
        t = int(targets[(k * 97) % S])
        loc = [t] + [int(x) for x in fld.neighbors[t]]
        pctx = cel.FieldContext(fld.images, [fld.patches[s2] for s2 in loc], [list(range(1, len(loc)))] + [[] for _ in loc[1:]],
                                image_set=iset)
        vloc = np.ascontiguousarray(fld.vp[loc])
        for rep in range(12):
            t1 = time.perf_counter()
            pctx.eval_batch(vloc, [0], FLAGS_ALL, pinned=False)
            if rep >= 2:
                lat.append(time.perf_counter() - t1)
            elif rep == 0 and k >= 4:
                t_first.append(time.perf_counter() - t1)
        t1 = time.perf_counter()
        pctx.close()
        if k >= 4:      # (the first contexts of a process fill the library's stream / staging pools)
            t_close.append(time.perf_counter() - t1)
            t_create.append(pctx.create_ms * 1e-3)
    iset.close()
    lat = np.sort(np.array(lat))
    out["single_call_latency_us"]["per_source_context"] = {
        "median": float(np.median(lat) * 1e6), "p90": float(lat[int(0.9 * len(lat))] * 1e6), "calls": int(len(lat)),
        "what": "the same call on per-source contexts (the source and its neighbours) sharing one image handle",
        "lifecycle_us": {"celeste_ctx_create_on": float(np.median(t_create) * 1e6), "first_call": float(np.median(t_first) * 1e6),
                         "celeste_ctx_destroy": float(np.median(t_close) * 1e6),
                         "what": "medians over 20 contexts; streams and page-locked staging come from the library's pools"}}
    # ElboMaximize.maximize! (Newton trust region, <= 50 iterations, KL on) for every source of the field,
    # neighbours frozen; wall time includes H2D/D2H
    ctx.maximize_batch(fld.vp, targets, cel.ElboConfig(max_iters=1))  # warm-up: the call's buffers at full size
    t1 = time.perf_counter()
    _, its, evals, _, ost = ctx.maximize_batch(fld.vp, targets, cel.ElboConfig())
    dt_opt = time.perf_counter() - t1
    out["optimizer"] = {"optimized_sources_per_sec": S / dt_opt, "seconds": dt_opt,
                        "mean_newton_iterations": float(its.mean()), "elbo_evaluations": int(evals.sum()),
                        "evals_per_sec": float(evals.sum()) / dt_opt,
                        "failed": int((ost != 0).sum())}
    # a Cyclades-sized layer: 80 conflict-free targets in one call (the fused optimiser launch)
    layer = conflict_free_layer(fld, S, 80)
    ctx.maximize_batch(fld.vp, layer, cel.ElboConfig(max_iters=2))
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        _, lits, _, _, lst = ctx.maximize_batch(fld.vp, layer, cel.ElboConfig())
    dt_layer = (time.perf_counter() - t1) / reps
    out["optimizer"]["cyclades_layer"] = {"targets": int(layer.size), "seconds": dt_layer, "max_newton_iterations": int(lits.max()),
                                          "us_per_newton_iteration_of_the_slowest_target": dt_layer * 1e6 / (int(lits.max()) + 1),
                                          "driver": "fused" if os.environ.get("CELESTE_OPT_FUSED", "") != "0" else "chained",
                                          "failed": int((lst != 0).sum())}
    # ParallelRun.one_node_joint_infer: the reference's schedule (Cyclades batches of 400, 3 sweeps), one C call
    from celeste_jl_amd.infer import joint_layers, one_node_joint_infer
    tg = [int(t) for t in targets]
    layers = joint_layers(tg, fld.neighbors)
    one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors)
    t1 = time.perf_counter()
    failed = set()
    one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors, failed=failed)
    dt_joint = time.perf_counter() - t1
    # the C call alone (celeste_joint_infer on the prepared table and schedule), with its evaluation count
    from celeste_jl_amd.infer import default_infer_config
    from celeste_jl_amd.params import init_source_table
    vp_j = init_source_table(fld.catalog, tg)
    centers = [vp_j[layer, 0:2].copy() for layer in layers]
    t1 = time.perf_counter()
    _, _, jevals, _, _ = ctx.joint_infer(vp_j, layers, default_infer_config(), pos_centers=centers)
    dt_jcall = time.perf_counter() - t1
    # the same schedule layer by layer (one fused optimiser launch per layer): what the dataflow launch replaces
    os.environ["CELESTE_JOINT_DATAFLOW"] = "0"
    one_node_joint_infer(ctx, fld.catalog, tg[:50], fld.neighbors)
    t1 = time.perf_counter()
    one_node_joint_infer(ctx, fld.catalog, tg, fld.neighbors)
    dt_layered = time.perf_counter() - t1
    os.environ.pop("CELESTE_JOINT_DATAFLOW")
    out["joint_infer"] = {"seconds": dt_joint, "sources_per_sec": S / dt_joint, "schedule": "Cyclades batches of 400, 3 sweeps "
                          "(ParallelRun.jl:135-196), celeste_joint_infer: ONE dataflow launch -- an entry starts when the entries "
                          "it depends on have ended; the table stays in HBM; host time (initial rows, colouring) included",
                          "layers": len(layers), "entries": int(sum(map(len, layers))), "largest_layer": max(map(len, layers)),
                          "failed": len(failed), "layer_by_layer_seconds": dt_layered,
                          "c_call_seconds": dt_jcall, "elbo_evaluations": int(jevals.sum()),
                          "evals_per_sec": float(jevals.sum()) / dt_jcall}
    # rank 0's cost-balanced shard of THIS field for N ranks, swept on this GPU by the driver the ranks run
    dev = torch.device("cuda", ctx.device)
    d_vp = torch.tensor(fld.vp, dtype=torch.float64, device=dev)
    sp = shard_projection(ctx, fld, targets, costs, d_vp, dev, (("f64", FLAGS_ALL),), K=20)
    out["shard_projection"] = dict(sp["f64"], note=sp["note"])
    return out


def shard_projection(ctx, fld, targets, costs, d_vp, dev, modes, K=10):
    """rank 0's cost-balanced shard of the workload for N = 1, 2, 4, 8 ranks, swept on THIS GPU by the driver the ranks run
    (parallel.DeviceShardedSweep, no gather): the strong-scaling efficiency the kernels allow if the catalog gather hides
    under the next sweep.  Not a multi-GPU measurement."""
    import torch
    from celeste_jl_amd.parallel import DeviceShardedSweep
    from celeste_jl_amd.partition import shard_targets
    stream = torch.cuda.current_stream(dev)
    out = {}
    for name, fl in modes:
        proj = {}
        for world in (1, 2, 4, 8):
            sw = DeviceShardedSweep(ctx, targets, costs, 0, 1, fl, shards=[shard_targets(costs, world)[0]])
            for _ in range(3):
                sw.step(d_vp.data_ptr())
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(K):
                sw.step(d_vp.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize(dev)
            proj[str(world)] = {"targets_rank0": sw.n, "ms_per_sweep": e0.elapsed_time(e1) / K}
            del sw
        for world in (2, 4, 8):
            proj[str(world)]["compute_bound_efficiency"] = proj["1"]["ms_per_sweep"] / (world * proj[str(world)]["ms_per_sweep"])
        out[name] = proj
    out["note"] = ("rank 0's shard of this workload for N ranks, swept on ONE GPU: the strong-scaling efficiency the kernels allow "
                   "if the catalog gather hides under the next sweep; not a multi-GPU measurement")
    return out


def variable_psf_record(args, head):
    """configs[2]'s field with the PSF the reference's production path really uses -- an SDSSPSFMap evaluated at every source
    (SDSSIO.jl:239-299; one stamp per patch, imaged_sources.jl:97-107): ~8 800 stamps = ~200 MB of spline coefficients per sweep,
    gathered 4 x 4 per pixel (fsm_util.jl:225-248) instead of one 22 KB table per band that never leaves L2 -- swept exactly like
    the headline, as a child run of this script (`--variable-psf`), counters included."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--variable-psf", "--steps", "100", "--warmup", "5", "--no-cpu-baseline",
           "--no-extras", "--no-config5", "--height", str(args.height), "--width", str(args.width), "--sources", str(args.sources),
           "--seed", str(args.seed)] + (["--no-live-pmc"] if args.no_live_pmc else [])
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "child run failed (rc %d): %s" % (r.returncode, r.stderr[-300:])}
        d = json.loads(line[-1])
    except Exception as e:   # (the headline does not depend on it)
        return {"error": repr(e)}
    keep = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "kernels_ms", "gaps_ms", "step_ms",
                              "sclk_mhz_after_loop", "pixel_visits_per_sec_rank0", "context_create_ms") if k in d}
    keep["workload"] = d["config"]["workload"]
    keep["sources_per_step"] = d["config"]["sources_per_step"]
    keep["pixel_visits_per_sweep"] = d["config"]["pixel_visits_per_sweep"]
    rf = d["roofline"]
    hbm = rf.get("hbm", rf)
    keep["roofline_valu"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_of_sustained", "kernel", "kernel_ms",
                                                "valu_utilization") if k in rf}
    keep["roofline_hbm"] = {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_measured_in_this_run",
                                                "algorithmic_bytes_per_launch", "spline_coefficient_bytes", "psf_stamps") if k in hbm}
    if hbm.get("traffic") and hbm.get("algorithmic_bytes_per_launch"):
        keep["traffic_over_algorithmic"] = hbm["traffic"] / hbm["algorithmic_bytes_per_launch"]
    keep["slowdown_vs_constant_psf"] = {"step": d["ms_per_step"] / head["ms_per_step"],
                                        "pixel_kernel": d["kernels_ms"]["pixel"] / head["kernels_ms"]["pixel"],
                                        "prep_and_neighbour_light": d["kernels_ms"]["prep"] / head["kernels_ms"]["prep"]}
    keep["wall_s_including_field_generation"] = time.time() - t0
    return keep


def config5_record(args):
    """BASELINE configs[4] on this one GPU, as a child run of this script (`--config 5 --dtype f32`, full size: 16 fields,
    80 images, 30 000 sources): throughput, the fp32-vs-fp64 check on every source, the fp32 kernel's VALU roofline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "5", "--dtype", "f32", "--steps", "10", "--warmup", "2",
           "--kernels-in-pass", "--no-cpu-baseline", "--no-extras", "--shard-projection", "--height", str(args.height),
           "--width", str(args.width)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "child run failed (rc %d): %s" % (r.returncode, r.stderr[-300:])}
        d = json.loads(line[-1])
    except Exception as e:   # (the headline does not depend on it)
        return {"error": repr(e)}
    keep = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "kernels_ms", "fp32_vs_fp64_device",
                              "pixel_visits_per_sec_rank0", "fp64_pixel_kernel_ms", "fp32_speedup_over_fp64_pixel_kernel") if k in d}
    keep["workload"] = d["config"]["workload"]
    keep["sources_per_step"] = d["config"]["sources_per_step"]
    keep["pixel_visits_per_sweep"] = d["config"]["pixel_visits_per_sweep"]
    keep["roofline_valu"] = d["roofline"].get("valu")
    hbm = d["roofline"].get("hbm", d["roofline"])
    keep["roofline_hbm_frac"] = hbm["frac"]
    # counter traffic of the fp32 pixel kernel at this size: three rocprofv3 passes over 30 000 sources take minutes, so the figure
    # is the committed one (tools/pmc_config5.sh -> profiles/<tag>_config5_pmc.json), not measured in this run
    try:
        import glob
        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_config5_pmc.json")))[-1]
        pk = json.load(open(prof))["pixel_kernel<2, float, false>"]
        traffic = (pk["fetch_kib"] + pk["write_kib"]) * 1024.0
        keep["roofline_hbm"] = {"achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"],
                                "algorithmic_bytes_per_launch": hbm.get("algorithmic_bytes_per_launch"), "traffic": traffic,
                                "traffic_measured_in_this_run": False,
                                "traffic_over_algorithmic": traffic / hbm["algorithmic_bytes_per_launch"] if hbm.get("algorithmic_bytes_per_launch") else None,
                                "traffic_source": os.path.relpath(prof, ROOT) + " (FETCH_SIZE + WRITE_SIZE, KiB x 1024, uncorrected; the "
                                                  "writes include the kernel's 56 B per lane of scratch, stored once per work item)"}
    except Exception:
        pass
    if "shard_projection" in d:
        keep["shard_projection"] = d["shard_projection"]
    keep["wall_s_including_field_generation"] = time.time() - t0
    return keep


if __name__ == "__main__":
    main()
