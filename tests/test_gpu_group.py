"""-m gpu: one process, N devices behind the C ABI (celeste_group_*, csrc/group.h).

The group entry points must return what the one-device entry points return, BIT FOR BIT, whatever the number of members:
a target's evaluation / optimisation does not depend on what else is in its launch, and the exchange only moves bytes.
On a one-GPU box the members are
  * one member on device 0 -- the RCCL path (ncclCommInitAll over one device, ncclAllGather with one rank), and
  * two / three members that share device 0 -- worker threads, cost-sharding, the gather layout and the table updates, with
    plain device-to-device copies standing in for RCCL (which refuses duplicate devices).
The >= 2-device tests at the end arm themselves on a node that has the devices (RCCL between real ranks).

tests/test_gpu_group_rccl_branch.py runs THIS FILE a second time in a child process with CELESTE_GROUP_EXCHANGE=rccl and
tests/libfake_rccl.so preloaded (a strict host-rendezvous stand-in that accepts repeated devices): the two / three members then
exchange through the RCCL branch of csrc/group.h -- N threads enqueueing collectives on N communicators -- instead of peer copies."""
import os

import numpy as np
import pytest

# (worker threads + barriers: a hang must end the process, not the GPU box's lease -- the thread method kills from outside the
# blocked C call)
# CELESTE_FUZZ_SEEDS=N: every seeded fuzz test with N seeds instead of its default handful (a long run on a GPU box)
FUZZ_SEEDS = int(os.environ.get("CELESTE_FUZZ_SEEDS", "0"))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


FORCED_RCCL = os.environ.get("CELESTE_GROUP_EXCHANGE") == "rccl"      # (the child process of test_gpu_group_rccl_branch.py)


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="module")
def crowded():
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(220, 240, 40, seed=23, margin=30)
    return f, cel.FieldContext(f.images, f.patches, f.neighbors)


def _group(f, devices):
    from celeste_jl_amd.group import FieldGroup
    return FieldGroup(f.images, f.patches, f.neighbors, devices=devices)


def _same(a, b, what):
    for x, y, n in zip(a, b, ("v / table", "d / iterations", "h / f_evals", "counters / elbo", "status")):
        if x is None and y is None:
            continue
        assert np.array_equal(x, y, equal_nan=True), "%s: output %s differs (max |diff| %.3e)" % (
            what, n, np.nanmax(np.abs(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64))))


MEMBERS = [[0], [0, 0], [0, 0, 0]]


@pytest.mark.parametrize("devices", MEMBERS, ids=lambda d: "%d_member%s" % (len(d), "s" if len(d) > 1 else ""))
def test_group_eval_equals_the_one_device_entry_bit_for_bit(crowded, devices):
    from celeste_jl_amd import cabi
    from celeste_jl_amd.partition import shard_targets
    f, ctx = crowded
    S = len(f.catalog)
    g = _group(f, devices)
    info = g.info()
    assert info["n_members"] == len(devices) and info["devices"] == devices and info["n_devices"] == 1
    if len(devices) == 1 or FORCED_RCCL:
        assert info["exchange"] == "rccl" and info["rccl_ranks"] == len(devices)   # ncclCommCount of the group's communicator
    else:
        assert info["exchange"] == "peer_copy" and info["rccl_ranks"] == 0
    rng = np.random.default_rng(11)
    cases = [("every source", list(range(S)), cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL),
             ("shuffled, repeated", [int(t) for t in rng.integers(0, S, 57)], cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL),
             ("gradient only, no KL", list(range(0, S, 2)), cabi.FLAG_GRAD),
             ("value only", [5, 1, 9], 0),
             ("packed Hessians", list(range(S)), cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL | cabi.FLAG_PACKED_HESS),
             ("fewer targets than members", [7], cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL),
             ("single precision", list(range(S)), cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL | cabi.FLAG_FP32),
             # 1000 targets: the one-device batch is above the 768-target threshold where the neighbours' light switches from two
             # wavefronts per item to one, the members' shards (500 / 333) are below it -- same bits all the same
             ("single precision across the wide-value threshold", list(range(S)) * 25,
              cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL | cabi.FLAG_FP32)]
    for what, tg, flags in cases:
        ref = ctx.eval_batch(f.vp, tg, flags)
        got = g.eval_batch(f.vp, tg, flags)
        assert (ref[4] == 0).all()
        _same(ref, got, "%s, %d member(s)" % (what, len(devices)))
        # the shards: the reference's estimate_time, longest first onto the least loaded member (partition.shard_targets)
        sizes, costs = g.shard_sizes()
        assert sum(sizes) == len(tg)
        cost = [sum(int(p.active_pixel_bitmap.size) for p in f.patches[t]) for t in tg]
        expect = shard_targets(cost, len(devices))
        assert sizes == [len(s) for s in expect] and costs == [sum(cost[i] for i in s) for s in expect]
    # a non-finite source: its status, everyone else's result
    vp = f.vp.copy()
    vp[3, 7] = np.nan
    ref = ctx.eval_batch(vp, list(range(S)), raise_on_error=False)
    got = g.eval_batch(vp, list(range(S)), raise_on_error=False)
    assert (ref[4] != 0).any() and (ref[4] == 0).any()
    _same(ref, got, "a failing source")
    with pytest.raises(AssertionError):
        g.eval_batch(vp, list(range(S)))
    with pytest.raises(cabi.CelesteError):
        g.eval_batch(f.vp, [S])                     # out of range
    g.close()


def test_one_member_on_its_worker_thread_runs_rccl_off_the_calling_thread(crowded, monkeypatch):
    """CELESTE_GROUP_THREADS=1: a group of ONE member still gets its worker thread, so its launches AND its RCCL collectives
    (communicator from ncclCommInitAll on the calling thread, ncclAllGather with one rank enqueued by the worker) run the way
    every member of a multi-device group runs them -- the closest a one-GPU box comes to the real thing.  Sweep, maximize!
    and joint inference equal the one-device entry points bit for bit."""
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    from celeste_jl_amd.group import cyclades_schedule, schedule_layers
    f, ctx = crowded
    S = len(f.catalog)
    monkeypatch.setenv("CELESTE_GROUP_THREADS", "1")
    g = _group(f, [0])
    monkeypatch.delenv("CELESTE_GROUP_THREADS")
    info = g.info()
    assert info["exchange"] == "rccl" and info["rccl_ranks"] == 1 and info["n_members"] == 1
    flags = cabi.FLAG_GRAD | cabi.FLAG_HESS | cabi.FLAG_KL
    tg = list(range(S))
    _same(ctx.eval_batch(f.vp, tg, flags), g.eval_batch(f.vp, tg, flags), "sweep on the worker thread")
    g.plan(f.vp, tg, flags)
    for _ in range(5):
        g.sweep()                                        # (dispatched back to back: the worker overlaps gather k with sweep k + 1)
    g.wait()
    _same(ctx.eval_batch(f.vp, tg, flags), g.results(), "resident sweeps on the worker thread")
    cfg = cel.ElboConfig(max_iters=5)
    _same(ctx.maximize_batch(f.vp, tg, cfg), g.maximize_batch(f.vp, tg, cfg), "maximize! on the worker thread")
    b_off, c_off, flat = cyclades_schedule(tg, f.neighbors, batch_size=12, rng=np.random.default_rng(3))
    layers, entries = schedule_layers(b_off, c_off, flat, 1)
    ref = ctx.joint_infer(f.vp, layers, cfg)
    new, its, evals, el, st, nx = g.joint_infer(f.vp, b_off, c_off, flat, 1, cfg)
    assert nx == 1 and np.array_equal(new, ref[0])         # (one member: the whole schedule is one segment, one launch, one exchange)
    flat_entry = np.concatenate([np.asarray(e) for e in entries])
    assert np.array_equal(its.reshape(-1)[flat_entry], ref[1]) and np.array_equal(el.reshape(-1)[flat_entry], ref[3])
    g.close()


def test_group_resident_sweeps_track_the_table_and_overlap_their_gathers(crowded):
    """plan once, sweep many times (the gather of sweep k runs while sweep k + 1 computes; two blocks alternate): the last
    sweep's results are the one-device results; a new plan with another table and other targets replaces the first"""
    f, ctx = crowded
    S = len(f.catalog)
    for devices in ([0], [0, 0]):
        g = _group(f, devices)
        g.enable_timing(True)
        for tg, scale in ((list(range(S)), 1.0), (list(range(1, S, 3)), 1.0 + 1e-3)):
            vp = f.vp.copy()
            vp[:, 6:] *= scale
            g.plan(vp, tg)
            for _ in range(5):
                g.sweep()
            g.wait()
            ref = ctx.eval_batch(vp, tg)
            _same(ref, g.results(), "resident sweeps, %d member(s)" % len(devices))
            ev, ga = g.last_sweep_ms()
            assert len(ev) == len(devices) and all(x > 0 for x in ev) and all(x >= 0 for x in ga)
        g.close()


def test_group_maximize_equals_the_one_device_entry_bit_for_bit(crowded):
    import celeste_jl_amd as cel
    f, ctx = crowded
    S = len(f.catalog)
    rng = np.random.default_rng(5)
    cases = [("every source", list(range(S)), cel.ElboConfig(max_iters=9), {}),
             ("shuffled, pinned centres", [int(t) for t in rng.permutation(S)][:23], cel.ElboConfig(max_iters=6, loc_width=0.5),
              {"pos_centers": None}),
             ("frozen neighbours from another table", list(range(1, S, 3)), cel.ElboConfig(max_iters=5),
              {"vp_neighbors": f.vp * (1.0 + 1e-3 * rng.standard_normal(f.vp.shape) * (np.arange(44) >= 6))})]
    for devices in MEMBERS:
        g = _group(f, devices)
        for what, tg, cfg, kw in cases:
            kw = dict(kw)
            if "pos_centers" in kw:
                kw["pos_centers"] = f.vp[tg, 0:2] + 0.01
            ref = ctx.maximize_batch(f.vp, tg, cfg, **kw)
            got = g.maximize_batch(f.vp, tg, cfg, **kw)
            assert (ref[4] == 0).all() and ref[1].max() >= 5
            _same(ref, got, "%s, %d member(s)" % (what, len(devices)))
        # a failing target keeps its row, the others are optimised
        vp = f.vp.copy()
        bad = int(np.argmax([len(n) for n in f.neighbors]))     # NaN in the most connected source: it and its neighbours fail
        vp[bad, 10] = np.nan
        tg = list(range(S))
        ref = ctx.maximize_batch(vp, tg, cel.ElboConfig(max_iters=4), raise_on_error=False)
        got = g.maximize_batch(vp, tg, cel.ElboConfig(max_iters=4), raise_on_error=False)
        assert (ref[4] != 0).any() and (ref[4] == 0).any()
        _same(ref, got, "failing targets, %d member(s)" % len(devices))
        from celeste_jl_amd import cabi
        with pytest.raises(cabi.CelesteError):
            g.maximize_batch(f.vp, [1, 2, 1])           # duplicates
        g.close()


def test_group_joint_inference_shards_components_and_exchanges_once_per_batch(crowded):
    """celeste_group_joint_infer: the connected components of every Cyclades batch sharded over the members, one exchange
    per batch -- and the table, the per-entry iterations / evaluations / ELBO values / statuses of celeste_joint_infer on the
    flattened schedule, bit for bit"""
    import celeste_jl_amd as cel
    from celeste_jl_amd.group import cyclades_schedule, schedule_layers
    from celeste_jl_amd.params import catalog_init_source, generic_init_source
    f, ctx = crowded
    S = len(f.catalog)
    targets = [s for s in range(S) if s % 7 != 3]
    vp0 = np.stack([catalog_init_source(ce) for ce in f.catalog])
    for t in targets:
        vp0[t] = generic_init_source(f.catalog[t].pos)
    n_sweeps = 2
    b_off, c_off, flat = cyclades_schedule(targets, f.neighbors, batch_size=12, rng=np.random.default_rng(3))
    n_batches = len(b_off) - 1
    assert n_batches == 3 and sorted(flat.tolist()) == sorted(targets) and (np.diff(c_off) > 1).any()
    pos = vp0[flat, 0:2].copy()
    layers, entries = schedule_layers(b_off, c_off, flat, n_sweeps)
    cfg = cel.ElboConfig(max_iters=6)
    ref = ctx.joint_infer(vp0, layers, cfg, pos_centers=[pos[e] for e in entries])
    assert (ref[4] == 0).all() and ref[1].max() == 6
    flat_entry = np.concatenate([np.asarray(e) + (k // (len(layers) // n_sweeps)) * len(flat) for k, e in enumerate(entries)])
    for devices in MEMBERS:
        g = _group(f, devices)
        for dataflow in (None, 0):
            os.environ.pop("CELESTE_JOINT_DATAFLOW", None)
            if dataflow is not None:
                os.environ["CELESTE_JOINT_DATAFLOW"] = str(dataflow)
            try:
                new, its, evals, el, st, nx = g.joint_infer(vp0, b_off, c_off, flat, n_sweeps, cfg, pos_centers=pos)
            finally:
                os.environ.pop("CELESTE_JOINT_DATAFLOW", None)
            # exchanges: one per SEGMENT -- a group of one has nobody to wait for (one launch, one exchange); in this crowded
            # field every batch reads rows the batch before wrote on another member: one per batch, never one per layer
            assert (nx == 1) if len(devices) == 1 else (2 <= nx <= n_sweeps * n_batches), (devices, nx)
            enq, aborted = g.collectives()
            assert not aborted and len(set(enq)) == 1
            assert np.array_equal(new, ref[0]), (devices, np.abs(new - ref[0]).max())
            for got, want, what in ((its, ref[1], "iterations"), (evals, ref[2], "f_evals"), (el, ref[3], "elbo"), (st, ref[4], "status")):
                assert np.array_equal(got.reshape(-1)[flat_entry], want), (devices, what)
        # a batch whose components conflict is refused
        from celeste_jl_amd import cabi
        a = next(s for s in targets if any(n in targets for n in f.neighbors[s]))
        b = next(n for n in f.neighbors[a] if n in targets)
        with pytest.raises(cabi.CelesteError):
            g.joint_infer(vp0, [0, 2], [0, 1, 2], [a, b], 1, cfg)
        g.close()


def test_group_joint_inference_without_cross_member_reads_is_one_segment():
    """isolated sources (nobody has a neighbour): whatever the number of members, no member ever reads a row another member
    wrote -- the whole call is ONE segment: one launch chain per member, one exchange at the end, celeste_joint_infer's table"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.group import cyclades_schedule, schedule_layers
    f = synthetic.make_field(700, 700, 12, seed=5, margin=40)
    S = len(f.catalog)
    keep = [s for s in range(S) if not f.neighbors[s]]
    assert len(keep) >= 6
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    b_off, c_off, flat = cyclades_schedule(keep, f.neighbors, batch_size=3, rng=np.random.default_rng(2))
    assert len(b_off) - 1 >= 2
    layers, entries = schedule_layers(b_off, c_off, flat, 2)
    cfg = cel.ElboConfig(max_iters=5)
    ref = ctx.joint_infer(f.vp, layers, cfg)
    for devices in MEMBERS:
        g = _group(f, devices)
        new, its, evals, el, st, nx = g.joint_infer(f.vp, b_off, c_off, flat, 2, cfg)
        assert nx == 1 and np.array_equal(new, ref[0]) and (st == 0).all(), (devices, nx)
        g.close()
    ctx.close()


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS or 6))
def test_group_joint_inference_segments_fuzz(crowded, seed):
    """random Cyclades schedules (batch size, sweeps, target subset) over two or three members: wherever the host cuts the
    segments, table and per-entry outputs are celeste_joint_infer's on the flattened schedule, bit for bit, and every member has
    enqueued the same number of exchanges"""
    import celeste_jl_amd as cel
    from celeste_jl_amd.group import cyclades_schedule, schedule_layers
    f, ctx = crowded
    S = len(f.catalog)
    rng = np.random.default_rng(900 + seed)
    targets = sorted(rng.choice(S, int(rng.integers(8, S + 1)), replace=False).tolist())
    n_sweeps = int(rng.integers(1, 4))
    batch = int(rng.choice([3, 5, 9, 14, 40]))
    devices = [0] * int(rng.integers(2, 4))
    b_off, c_off, flat = cyclades_schedule(targets, f.neighbors, batch_size=batch, rng=np.random.default_rng(seed))
    layers, entries = schedule_layers(b_off, c_off, flat, n_sweeps)
    cfg = cel.ElboConfig(max_iters=4)
    pos = f.vp[flat, 0:2] + 0.01 * rng.standard_normal((len(flat), 2))
    ref = ctx.joint_infer(f.vp, layers, cfg, pos_centers=[pos[e] for e in entries])
    n_layers_per_sweep = len(layers) // n_sweeps
    flat_entry = np.concatenate([np.asarray(e) + (k // n_layers_per_sweep) * len(flat) for k, e in enumerate(entries)])
    g = _group(f, devices)
    new, its, evals, el, st, nx = g.joint_infer(f.vp, b_off, c_off, flat, n_sweeps, cfg, pos_centers=pos)
    assert 1 <= nx <= n_sweeps * (len(b_off) - 1)
    assert np.array_equal(new, ref[0]), (seed, devices, batch, n_sweeps, np.abs(new - ref[0]).max())
    for got, want in ((its, ref[1]), (evals, ref[2]), (el, ref[3]), (st, ref[4])):
        assert np.array_equal(got.reshape(-1)[flat_entry], want)
    enq, aborted = g.collectives()
    assert not aborted and len(set(enq)) == 1 and enq[0] == nx
    print("segments fuzz", seed, "members", len(devices), "batch", batch, "sweeps", n_sweeps, "batches", len(b_off) - 1, "exchanges", nx)
    g.close()


def test_group_on_the_bench_field_two_members_one_device():
    """config 3 at full size: 2000 targets, two members -- the sweep and one batch of joint inference"""
    import celeste_jl_amd as cel
    import bench
    from celeste_jl_amd.group import FieldGroup
    fld = bench.build_field(2048, 1489, 2000, 3)
    ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
    g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=[0, 0])
    tg = list(range(len(fld.catalog)))
    ref = ctx.eval_batch(fld.vp, tg)
    _same(ref, g.eval_batch(fld.vp, tg), "bench field")
    sizes, costs = g.shard_sizes()
    assert sum(sizes) == 2000 and abs(costs[0] - costs[1]) <= 0.002 * sum(costs)
    g.close()
    ctx.close()


def test_group_on_overlapping_fields_with_the_sparse_patch_list():
    """configs[4]'s shape (a source sees a few of many images: visit lists, sparse patch list) through a group of two: sweep,
    single-precision mode, maximize! and one Cyclades batch -- the one-device results"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.group import FieldGroup, cyclades_schedule, schedule_layers
    f = synthetic.make_multifield((2, 3), 160, 160, 0.10, 70, seed=11, sparse=True)
    S = len(f.catalog)
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    g = FieldGroup(f.images, f.patches, f.neighbors, devices=[0, 0])
    tg = list(range(S))
    for flags in (7, 7 | cabi.FLAG_FP32, 5):
        _same(ctx.eval_batch(f.vp, tg, flags), g.eval_batch(f.vp, tg, flags), "sparse patch list, flags %d" % flags)
    cfg = cel.ElboConfig(max_iters=5)
    _same(ctx.maximize_batch(f.vp, tg, cfg), g.maximize_batch(f.vp, tg, cfg), "sparse patch list, maximize")
    b_off, c_off, flat = cyclades_schedule(tg, f.neighbors, batch_size=30, rng=np.random.default_rng(1))
    layers, entries = schedule_layers(b_off, c_off, flat, 1)
    pos = f.vp[flat, 0:2].copy()
    ref = ctx.joint_infer(f.vp, layers, cfg, pos_centers=[pos[e] for e in entries])
    new, _, _, _, st, nx = g.joint_infer(f.vp, b_off, c_off, flat, 1, cfg, pos_centers=pos)
    assert 1 <= nx <= len(b_off) - 1 and np.array_equal(new, ref[0]) and (st == 0).all()
    g.close()
    ctx.close()


def test_infer_box_over_a_device_group_equals_the_one_device_run():
    """ParallelRun.infer_box with `devices=`: the node-level loops through celeste_group_* (two members on the one device)
    leave the optimised sources the one-device run leaves, joint and single"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(300, 340, 40, seed=77)
    box = cel.BoundingBox(0, 300, 0, 340)
    for method in ("joint_vi", "single_vi"):
        ref = cel.infer_box(f.images, box, f.catalog, method=method)
        got = cel.infer_box(f.images, box, f.catalog, method=method, devices=[0, 0])
        assert len(ref) == len(got) > 30
        for a, b in zip(ref, got):
            assert a.init_ra == b.init_ra and a.is_sky_bad == b.is_sky_bad and a.failed == b.failed
            assert np.array_equal(a.vs, b.vs), (method, np.abs(a.vs - b.vs).max())


@pytest.mark.skipif(FORCED_RCCL, reason="asserts the peer-copy backend of the default configuration")
def test_bench_group_driver_prints_the_line_and_the_single_rank_catalog(tmp_path):
    """`bench.py --driver group`: ONE process, the members behind the C ABI -- one member (RCCL, one rank: `ranks_seen` is
    ncclCommCount) and two members on the one device; both leave the catalog the torch driver's single rank leaves"""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--height", "300", "--width", "260", "--sources", "60", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}

    def run(extra, sub):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra + ["--check-dir", str(tmp_path / sub)],
                             capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1                      # RCCL's banner and everything else stay off stdout
        return json.loads(lines[0])
    ref = run(["--no-extras"], "ranks")
    d1 = run(["--driver", "group"], "g1")
    d2 = run(["--driver", "group", "--gpus", "2", "--group-devices", "0,0"], "g2")
    assert d1["n_gpus"] == 1 and d1["ranks_seen"] == 1 and d1["config"]["gather_backend"] == "rccl" and d1["config"]["shard_sizes"] == [60]
    assert d2["config"]["members"] == 2 and d2["config"]["gather_backend"] == "peer_copy" and sum(d2["config"]["shard_sizes"]) == 60
    for d in (d1, d2):
        for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline", "timing_method"):
            assert key in d, key
        assert d["value"] > 0 and d["roofline"]["bound"] == "fp64_valu" and d["unit"] == ref["unit"] and d["metric"] == ref["metric"]
    r = np.load(tmp_path / "ranks" / "rank0.npz")
    for sub in ("g1", "g2"):
        g = np.load(tmp_path / sub / "group.npz")
        assert np.array_equal(g["v"], r["v"]) and np.array_equal(g["d"], r["d"])


# ---- real devices: these arm themselves on a node that has them -------------------------------------------------------
@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 HIP devices (RCCL between real ranks)")
@pytest.mark.parametrize("n", [2, 4, 8])
def test_group_over_real_devices_rccl(crowded, n):
    if _n_devices() < n:
        pytest.skip("needs %d devices" % n)
    import celeste_jl_amd as cel
    from celeste_jl_amd.group import cyclades_schedule, schedule_layers
    f, ctx = crowded
    S = len(f.catalog)
    g = _group(f, list(range(n)))
    info = g.info()
    assert info["exchange"] == "rccl" and info["rccl_ranks"] == n and info["n_devices"] == n
    tg = list(range(S))
    _same(ctx.eval_batch(f.vp, tg), g.eval_batch(f.vp, tg), "eval over %d devices" % n)
    cfg = cel.ElboConfig(max_iters=6)
    _same(ctx.maximize_batch(f.vp, tg, cfg), g.maximize_batch(f.vp, tg, cfg), "maximize over %d devices" % n)
    b_off, c_off, flat = cyclades_schedule(tg, f.neighbors, batch_size=12, rng=np.random.default_rng(3))
    layers, entries = schedule_layers(b_off, c_off, flat, 2)
    pos = f.vp[flat, 0:2].copy()
    ref = ctx.joint_infer(f.vp, layers, cfg, pos_centers=[pos[e] for e in entries])
    new, _, _, _, st, nx = g.joint_infer(f.vp, b_off, c_off, flat, 2, cfg, pos_centers=pos)
    assert 1 <= nx <= 2 * (len(b_off) - 1) and np.array_equal(new, ref[0])
    g.close()


@pytest.mark.skipif(not FORCED_RCCL, reason="only in the child process of test_gpu_group_rccl_branch.py (fake RCCL preloaded)")
def test_zz_the_rccl_branch_really_ran_with_more_than_one_rank():
    """last in the file: the preloaded stand-in counted the collectives that completed on communicators of > 1 rank"""
    import ctypes as C
    fake = C.CDLL(os.environ["CELESTE_FAKE_RCCL"])
    out = (C.c_uint64 * 8)()
    fake.fake_rccl_stats(out)
    print("fake_rccl: %d all-gathers completed, %d of them between > 1 rank, %d aborts, %d count mismatches"
          % (out[0], out[1], out[2], out[4]))
    assert out[1] > 50 and out[2] == 0 and out[4] == 0
