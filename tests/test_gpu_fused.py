"""-m gpu: the fused optimiser launch (optim_fused_kernel: every target iterates at its own pace inside ONE persistent
launch, csrc/fused_kernels.h) against the chained driver (work list -> pixel -> lift -> step per Newton iteration).

Both drivers run the same device functions on the same 256-pixel chunk records, so everything they return must agree
BIT FOR BIT -- parameters, iteration counts, evaluation counts, ELBO values, status codes -- whatever the batch, and the
chained driver is in turn held to the CPU restatement by tests/test_gpu_optimizer.py.  Also here: the entry points built
on the fused launch (celeste_maximize_batch_device, celeste_joint_infer)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _both(ctx, vp, targets, cfg, **kw):
    """(chained, fused) results of the same call"""
    out = []
    for mode in (0, 1):
        with _env(CELESTE_OPT_FUSED=mode):
            out.append(ctx.maximize_batch(vp, targets, cfg, raise_on_error=False, **kw))
    return out


def _assert_identical(a, b, what):
    names = ("vp", "iterations", "f_evals", "elbo", "status")
    for x, y, n in zip(a, b, names):
        assert np.array_equal(x, y), "%s: %s differs (max |diff| %.3e)" % (
            what, n, np.abs(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)).max())


@pytest.fixture(scope="module")
def crowded():
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(220, 240, 40, seed=23, margin=30)
    return f, cel.FieldContext(f.images, f.patches, f.neighbors)


def test_fused_launch_equals_chained_driver_bit_for_bit(crowded):
    import celeste_jl_amd as cel
    f, ctx = crowded
    S = len(f.catalog)
    rng = np.random.default_rng(5)
    cases = [
        ("one target", [3], cel.ElboConfig(max_iters=7), {}),
        ("five targets, to convergence", [0, 9, 17, 25, 33], cel.ElboConfig(), {}),
        ("every source", list(range(S)), cel.ElboConfig(max_iters=12), {}),
        ("every source, shuffled, Optim's secular cap, no KL", [int(t) for t in rng.permutation(S)],
         cel.ElboConfig(max_iters=9, tr_secular_iters=5), {"include_kl": False}),
        ("wide position boxes, pinned centres", list(range(0, S, 2)), cel.ElboConfig(max_iters=10, loc_width=0.5),
         {"pos_centers": f.vp[0:S:2, 0:2] + 0.01}),
        ("frozen neighbours from another table", list(range(1, S, 3)), cel.ElboConfig(max_iters=8),
         {"vp_neighbors": f.vp * (1.0 + 1e-3 * rng.standard_normal(f.vp.shape) * (np.arange(44) >= 6))}),
    ]
    for what, tg, cfg, kw in cases:
        a, b = _both(ctx, f.vp, tg, cfg, **kw)
        assert (a[4] == 0).all(), what
        assert a[1].max() >= min(cfg.max_iters, 5), (what, a[1])       # the optimiser really iterated
        assert not np.array_equal(a[0][tg], f.vp[tg])
        _assert_identical(a, b, what)
        print("%-55s %3d targets, iterations %d..%d: identical" % (what, len(tg), a[1].min(), a[1].max()))


def test_fused_launch_with_visit_lists_and_eigen_solver(crowded):
    """the sparse-patch-list code path (visit items instead of the dense s * N + n tables) and the eigen-decomposition
    route of the sub-problem"""
    import celeste_jl_amd as cel
    f, _ = crowded
    with _env(CELESTE_FORCE_VISIT_LISTS=1):
        ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    tg = list(range(0, len(f.catalog), 2))
    a, b = _both(ctx, f.vp, tg, cel.ElboConfig(max_iters=8))
    _assert_identical(a, b, "visit lists")
    with _env(CELESTE_TR_SOLVER="eig"):
        c, d = _both(ctx, f.vp, tg, cel.ElboConfig(max_iters=8))
    _assert_identical(c, d, "eigen solver")
    assert np.abs(c[0] - a[0]).max() < 1e-6          # (two routes to the same step)


def test_fused_launch_survives_a_failing_target_and_refuses_duplicates(crowded):
    """ParallelRun.jl:582-597: a source whose ELBO is not finite (here: a neighbour with a non-finite parameter) keeps its
    row and gets its status, the others are optimised as if it were not there -- in both drivers, identically"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    f, ctx = crowded
    S = len(f.catalog)
    bad = int(np.argmax([len(n) for n in f.neighbors]))
    hit = set(f.neighbors[bad])
    targets = [t for t in range(S) if t != bad]
    nb = f.vp.copy()
    nb[bad, 7] = np.nan
    cfg = cel.ElboConfig(max_iters=6)
    a, b = _both(ctx, f.vp, targets, cfg, vp_neighbors=nb)
    _assert_identical(a, b, "failing targets")
    ok = [k for k, t in enumerate(targets) if t not in hit]
    ko = [k for k, t in enumerate(targets) if t in hit]
    assert len(ko) > 0 and len(ok) > 0
    assert np.isin(a[4][ko], (cabi.ERR_NONFINITE_INPUT, cabi.ERR_NONFINITE_RESULT)).all() and (a[4][ok] == 0).all()
    for k in ko:
        assert np.array_equal(a[0][targets[k]], f.vp[targets[k]]), "a failed target keeps its input row"
    good = [targets[k] for k in ok]
    clean, _ = _both(ctx, f.vp, good, cfg)
    assert np.array_equal(clean[0][good], a[0][good]) and np.array_equal(clean[3], a[3][ok])
    with _env(CELESTE_OPT_FUSED=1), pytest.raises(cabi.CelesteError):
        ctx.maximize_batch(f.vp, [3, 5, 3], cel.ElboConfig(max_iters=2))
    # the same through the joint-inference entry: the failed sources keep the rows they had before their layer
    from celeste_jl_amd.infer import joint_layers
    layers = joint_layers(targets, f.neighbors, batch_size=10, n_iters=1, rng=np.random.default_rng(1))
    vpb = f.vp.copy()
    vpb[bad, 7] = np.nan
    new, its, evals, el, st = ctx.joint_infer(vpb, layers, cfg)
    flat = [t for l in layers for t in l]
    for t, s1 in zip(flat, st):
        assert (s1 != 0) == (t in hit), (t, s1)
        if t in hit:
            assert np.array_equal(new[t], f.vp[t])
    assert np.isnan(new[bad, 7]) and np.isfinite(np.delete(new, bad, axis=0)).all()


def test_a_fused_launch_that_cannot_make_progress_gives_up_instead_of_hanging(crowded):
    """every wait inside the launch is bounded: with a time-out shorter than one Newton step the workgroups that wait for
    work abort the launch, the call fails loudly (CELESTE_ERR_HIP) and vp is untouched"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    f, ctx = crowded
    with _env(CELESTE_OPT_FUSED=1, CELESTE_FUSED_TIMEOUT_S=1e-6):
        with pytest.raises(cabi.CelesteError) as e:
            ctx.maximize_batch(f.vp, list(range(10)), cel.ElboConfig(max_iters=20))
    assert e.value.status == cabi.ERR_HIP
    a, b = _both(ctx, f.vp, list(range(10)), cel.ElboConfig(max_iters=5))     # the context is still usable
    _assert_identical(a, b, "after an aborted launch")


def test_maximize_batch_device_equals_the_host_pointer_call(crowded):
    """celeste_maximize_batch_device: the table is optimised in place in HBM; same rows, same per-target outputs"""
    import torch
    import celeste_jl_amd as cel
    f, ctx = crowded
    dev = torch.device("cuda", ctx.device)
    tg = np.arange(0, len(f.catalog), 3, dtype=np.int32)
    cfg = cel.ElboConfig(max_iters=9)
    pc = f.vp[tg, 0:2] - 0.02
    ref = ctx.maximize_batch(f.vp, tg, cfg, pos_centers=pc)
    for mode in (0, 1):
        d_vp = torch.tensor(f.vp, dtype=torch.float64, device=dev)
        d_tg = torch.tensor(tg, device=dev)
        d_pc = torch.tensor(pc, dtype=torch.float64, device=dev)
        d_it = torch.zeros(len(tg), dtype=torch.int32, device=dev)
        d_ev = torch.zeros_like(d_it)
        d_st = torch.zeros_like(d_it)
        d_el = torch.zeros(len(tg), dtype=torch.float64, device=dev)
        with _env(CELESTE_OPT_FUSED=mode):
            ctx.maximize_batch_device(d_vp.data_ptr(), len(tg), d_tg.data_ptr(), cfg, d_pos_centers=d_pc.data_ptr(),
                                      d_iterations=d_it.data_ptr(), d_f_evals=d_ev.data_ptr(), d_elbo=d_el.data_ptr(),
                                      d_status=d_st.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        got = (d_vp.cpu().numpy(), d_it.cpu().numpy(), d_ev.cpu().numpy(), d_el.cpu().numpy(), d_st.cpu().numpy())
        _assert_identical(ref, got, "device-pointer call, fused=%d" % mode)


def test_joint_infer_entry_equals_layer_by_layer_calls(crowded):
    """celeste_joint_infer keeps the table in HBM across all layers of the schedule; the same schedule driven from the
    host, one celeste_maximize_batch per layer with the table going up and down, must leave the same table"""
    import celeste_jl_amd as cel
    from celeste_jl_amd.infer import joint_infer_sweeps, joint_layers
    from celeste_jl_amd.params import catalog_init_source, generic_init_source
    f, ctx = crowded
    S = len(f.catalog)
    targets = [s for s in range(S) if s % 7 != 3]                    # a few sources stay frozen neighbours
    vp0 = np.stack([catalog_init_source(ce) for ce in f.catalog])
    for t in targets:
        vp0[t] = generic_init_source(f.catalog[t].pos)
    cfg = cel.ElboConfig(max_iters=6)
    for schedule, bs in (("cyclades", 12), ("coloring", 0)):
        layers = joint_layers(targets, f.neighbors, batch_size=bs, n_iters=2, rng=np.random.default_rng(3), schedule=schedule)
        assert len(layers) >= 4 and sorted(t for l in layers for t in l) == sorted(targets + targets)
        centers = [vp0[l, 0:2].copy() for l in layers]
        new, its, evals, el, st = ctx.joint_infer(vp0, layers, cfg, pos_centers=centers)
        assert (st == 0).all() and its.max() == 6

        def layer_host(vp, layer, pc):
            out, _, _, _, s = ctx.maximize_batch(vp, layer, cfg, pos_centers=pc)
            assert (s == 0).all()
            return out[layer]
        for mode in (0, 1):
            with _env(CELESTE_OPT_FUSED=mode):
                ref = joint_infer_sweeps(layer_host, vp0.copy(), targets, f.neighbors, batch_size=bs, n_iters=2,
                                         rng=np.random.default_rng(3), schedule=schedule)
            assert np.array_equal(ref, new), (schedule, mode, np.abs(ref - new).max())
        frozen = [s for s in range(S) if s not in targets]
        assert np.array_equal(new[frozen], vp0[frozen]) and not np.array_equal(new[targets], vp0[targets])
        print("joint inference, %s: %d layers of %d..%d sources: one call == layer-by-layer calls" %
              (schedule, len(layers), min(map(len, layers)), max(map(len, layers))))
    # a layer with two neighbouring sources is refused
    a = next(s for s in range(S) if f.neighbors[s])
    from celeste_jl_amd import cabi
    with pytest.raises(cabi.CelesteError):
        ctx.joint_infer(vp0, [[a, f.neighbors[a][0]]], cfg)


def test_joint_dataflow_launch_equals_the_layered_schedule(crowded):
    """celeste_joint_infer runs the whole schedule as one dataflow launch (entries start when the entries they depend on
    have ended); CELESTE_JOINT_DATAFLOW=0 runs it layer by layer.  Same table, same per-entry iterations, evaluations,
    ELBO values and statuses, bit for bit -- also when entries fail, and on the bench field's full schedule."""
    import time
    import celeste_jl_amd as cel
    from celeste_jl_amd.infer import joint_layers, default_infer_config
    from celeste_jl_amd.params import catalog_init_source, generic_init_source
    f, ctx = crowded
    S = len(f.catalog)
    targets = list(range(S))
    vp0 = np.stack([catalog_init_source(ce) for ce in f.catalog])
    for t in targets:
        vp0[t] = generic_init_source(f.catalog[t].pos)
    bad = int(np.argmax([len(n) for n in f.neighbors]))
    with _env(CELESTE_FORCE_VISIT_LISTS=1):
        ctx_lists = cel.FieldContext(f.images, f.patches, f.neighbors)      # the sparse-patch-list code path
    for case, cfg in (("clean", cel.ElboConfig(max_iters=8)), ("failing", cel.ElboConfig(max_iters=5)), ("eig", cel.ElboConfig(max_iters=4)),
                      ("visit lists", cel.ElboConfig(max_iters=5)), ("no iterations", cel.ElboConfig(max_iters=0))):
        vp = vp0.copy()
        ctx = ctx_lists if case == "visit lists" else crowded[1]
        tg = targets
        if case == "failing":
            vp[bad, 7] = np.nan
            tg = [t for t in targets if t != bad]
        layers = joint_layers(tg, f.neighbors, batch_size=12, n_iters=3, rng=np.random.default_rng(7))
        centers = [vp[l, 0:2].copy() for l in layers]
        with _env(CELESTE_TR_SOLVER="eig" if case == "eig" else None):
            with _env(CELESTE_JOINT_DATAFLOW=0):
                ref = ctx.joint_infer(vp, layers, cfg, pos_centers=centers)
            got = ctx.joint_infer(vp, layers, cfg, pos_centers=centers)
        for a, b, what in zip(ref, got, ("table", "iterations", "evaluations", "elbo", "status")):
            assert np.array_equal(a, b, equal_nan=True), (case, what)
        if case == "failing":
            assert (ref[4] != 0).any() and (ref[4] == 0).any()
        else:
            assert (ref[4] == 0).all()
    # the bench field: 2000 sources, Cyclades batches of 400, 3 sweeps
    import bench
    fld = bench.build_field(2048, 1489, 2000, 3)
    ctx2 = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
    tg = list(range(len(fld.catalog)))
    vp = np.stack([catalog_init_source(ce) for ce in fld.catalog])
    for t in tg:
        vp[t] = generic_init_source(fld.catalog[t].pos)
    layers = joint_layers(tg, fld.neighbors)
    centers = [vp[l, 0:2].copy() for l in layers]
    cfg = default_infer_config()
    out = {}
    for mode in (0, 1):
        with _env(CELESTE_JOINT_DATAFLOW=mode):
            ctx2.joint_infer(vp, layers[:2], cfg, pos_centers=centers[:2])
            t0 = time.perf_counter()
            out[mode] = ctx2.joint_infer(vp, layers, cfg, pos_centers=centers)
            dt = time.perf_counter() - t0
        print("bench field, %d layers, %d entries, %s: %.3f s" % (len(layers), sum(map(len, layers)),
                                                                 "one dataflow launch" if mode else "layer by layer", dt))
    for a, b, what in zip(out[0], out[1], ("table", "iterations", "evaluations", "elbo", "status")):
        assert np.array_equal(a, b), what
    assert (out[1][4] == 0).all()
    ctx2.close()



def test_the_persistent_launches_do_not_depend_on_how_much_of_the_grid_is_resident(crowded):
    """optim_fused_kernel never lets a workgroup wait for a particular other workgroup, so it must finish -- with the same
    bits -- on ONE workgroup, on a handful, and on more workgroups than the device can hold at once (the late ones only
    ever see the EXIT items).  Batch optimisation and the joint dataflow launch."""
    import celeste_jl_amd as cel
    from celeste_jl_amd.infer import joint_layers
    f, ctx = crowded
    S = len(f.catalog)
    tg = list(range(S))
    cfg = cel.ElboConfig(max_iters=6)
    layers = joint_layers(tg, f.neighbors, batch_size=12, n_iters=2, rng=np.random.default_rng(11))
    with _env(CELESTE_OPT_FUSED=1, CELESTE_JOINT_DATAFLOW=1):
        ref_b = ctx.maximize_batch(f.vp, tg[::2], cfg)
        ref_j = ctx.joint_infer(f.vp, layers, cfg)
        for grid in (1, 7, 1500):
            with _env(CELESTE_FUSED_GRID=grid):
                ctx2 = cel.FieldContext(f.images, f.patches, f.neighbors)     # (the grid is read once per context)
                got_b = ctx2.maximize_batch(f.vp, tg[::2], cfg)
                got_j = ctx2.joint_infer(f.vp, layers, cfg)
                ctx2.close()
            _assert_identical(ref_b, got_b, "batch, %d workgroups" % grid)
            _assert_identical(ref_j, got_j, "joint schedule, %d workgroups" % grid)



def test_one_node_joint_infer_uses_the_entry_and_reports_failures(crowded):
    import celeste_jl_amd as cel
    from celeste_jl_amd.infer import one_node_joint_infer
    f, ctx = crowded
    S = len(f.catalog)
    failed = set()
    vs = one_node_joint_infer(ctx, f.catalog, list(range(S)), f.neighbors, cel.ElboConfig(max_iters=4), batch_size=10,
                              n_iters=1, failed=failed)
    assert vs.shape == (S, 44) and np.isfinite(vs).all() and not failed


def test_fused_launch_on_the_bench_field():
    """config 3 (2048 x 1489 x 5, 2000 sources): a Cyclades-sized layer, a rank's N = 8 shard and the whole field"""
    import time
    import bench
    import celeste_jl_amd as cel
    fld = bench.build_field(2048, 1489, 2000, 3)
    ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
    rng = np.random.default_rng(11)
    for n, iters in ((80, 50), (250, 50), (2000, 50)):
        tg = np.sort(rng.choice(2000, n, replace=False)).astype(np.int32)
        res, dt = [], []
        for mode in (0, 1):
            with _env(CELESTE_OPT_FUSED=mode):
                ctx.maximize_batch(fld.vp, tg, cel.ElboConfig(max_iters=2))
                t0 = time.perf_counter()
                res.append(ctx.maximize_batch(fld.vp, tg, cel.ElboConfig(max_iters=iters)))
                dt.append(time.perf_counter() - t0)
        _assert_identical(res[0], res[1], "%d targets" % n)
        its = res[0][1]
        print("%4d targets, %d..%d Newton iterations (mean %.1f): chained %.2f ms (%.0f us per iteration of the slowest target), "
              "fused %.2f ms (%.0f us)" % (n, its.min(), its.max(), its.mean(), dt[0] * 1e3, dt[0] * 1e6 / (its.max() + 1),
                                           dt[1] * 1e3, dt[1] * 1e6 / (its.max() + 1)))


JOINT_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import celeste_jl_amd as cel
from celeste_jl_amd import synthetic
from celeste_jl_amd.infer import one_node_joint_infer
from celeste_jl_amd.partition import estimate_time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                       # both ranks share the one GPU: gloo, blocks staged through the host
dist.init_process_group("gloo", rank=rank, world_size=world)
f = synthetic.make_field(220, 240, 40, seed=23, margin=30)
ctx = cel.FieldContext(f.images, f.patches, f.neighbors, device=0)
costs = [float(estimate_time(row)) for row in f.patches]
failed = set()
vs = one_node_joint_infer(ctx, f.catalog, list(range(40)), f.neighbors, cel.ElboConfig(max_iters=5), batch_size=12, n_iters=2,
                          rank=rank, world=world, costs=costs, failed=failed)
assert not failed
np.save(os.path.join(%(out)r, "joint_rank%%d.npy" %% rank), vs)
dist.barrier(); dist.destroy_process_group()
'''


def test_joint_inference_two_ranks_on_one_gpu_equals_one_rank(crowded, tmp_path):
    """one_node_joint_infer with world = 2 (parallel.DeviceJointInfer: every layer sharded by cost, each rank optimises its
    shard in place in its device-resident table with celeste_maximize_batch_device, the optimised rows + status are
    all-gathered -- here over gloo, two processes sharing the one GPU; over RCCL in tests/test_gpu_multi_rccl.py) leaves
    every rank with the one-rank table, bit for bit"""
    import subprocess
    import sys
    import celeste_jl_amd as cel
    from celeste_jl_amd.infer import one_node_joint_infer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "joint_worker.py")
    with open(script, "w") as fh:
        fh.write(JOINT_WORKER % {"root": root, "out": str(tmp_path)})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 200), script]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    f, ctx = crowded
    ref = one_node_joint_infer(ctx, f.catalog, list(range(40)), f.neighbors, cel.ElboConfig(max_iters=5), batch_size=12, n_iters=2)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "joint_rank%d.npy" % r))
        assert np.array_equal(got, ref), "rank %d: max |diff| %.3e" % (r, np.abs(got - ref).max())


def test_eval_fused_kernel_equals_pixel_and_lift_kernels(crowded):
    """small evaluation batches (a rank's shard at N >= 4, one elbo() per call) run eval_fused_kernel -- one workgroup per
    chunk record, four wavefronts adding in iteration order, the lift by the workgroup that completes a target; everything
    it returns is pixel_kernel + lift_kernel's, bit for bit"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi, synthetic
    f, ctx = crowded
    S = len(f.catalog)
    rng = np.random.default_rng(2)
    with _env(CELESTE_FORCE_VISIT_LISTS=1):
        ctx_lists = cel.FieldContext(f.images, f.patches, f.neighbors)
    mf = synthetic.make_multifield(grid=(2, 2), H=150, W=160, n_sources=60, seed=8, sparse=True)   # sources in 1..4 of 20 images
    ctx_mf = cel.FieldContext(mf.images, mf.patches, mf.neighbors)
    cases = [("one target", ctx, f.vp, [7], 7), ("every source", ctx, f.vp, list(range(S)), 7),
             ("repeated targets, no KL", ctx, f.vp, [3, 3, 9, 3, 11], 3),
             ("packed Hessians", ctx, f.vp, [int(t) for t in rng.permutation(S)[:17]], 7 | cabi.FLAG_PACKED_HESS),
             ("visit lists", ctx_lists, f.vp, list(range(0, S, 2)), 7),
             ("overlapping fields, sparse patch list", ctx_mf, mf.vp, list(range(60)), 7)]
    for what, cx, vp, tg, flags in cases:
        res = []
        for mode in (0, 1):
            with _env(CELESTE_EVAL_FUSED=mode):
                res.append(cx.eval_batch(vp, tg, flags))
        assert (res[0][4] == 0).all(), what
        for a, b, name in zip(res[0], res[1], ("v", "d", "h", "counters", "status")):
            assert np.array_equal(a, b), "%s: %s differs (max |diff| %.3e)" % (what, name, np.abs(a - b).max())
    # a non-finite parameter: same status, same (non-finite) outputs
    bad = f.vp.copy()
    bad[4, 7] = np.inf
    out = []
    for mode in (0, 1):
        with _env(CELESTE_EVAL_FUSED=mode):
            out.append(ctx.eval_batch(bad, [4, 5], 7, raise_on_error=False))
    assert np.array_equal(out[0][4], out[1][4]) and out[0][4][0] == cabi.ERR_NONFINITE_INPUT


def test_eval_fused_kernel_with_repeated_targets_through_the_device_pointer_entry(crowded):
    """celeste_elbo_eval_batch_device does not know its targets on the host: a short list that repeats the chunk-richest
    source is LONGER than the bound for distinct targets, and eval_fused_kernel has no stride loop -- every slot must still
    be evaluated and lifted, and a second launch on the same context must not find stale arrival counters"""
    import torch
    f, ctx = crowded
    S = len(f.catalog)
    base = ctx.eval_batch(f.vp, list(range(S)), 7)
    cnt = base[3][:, 0] + base[3][:, 1]
    rich = int(np.argmax(cnt))              # (most pixel visits = most chunk records)
    dev = torch.device("cuda", ctx.device)
    d_vp = torch.tensor(f.vp, dtype=torch.float64, device=dev)
    for tg in ([rich] * 24, [rich, 1, rich, rich, 2, rich, rich, rich], [rich] * 32):
        n = len(tg)
        d_tg = torch.tensor(tg, dtype=torch.int32, device=dev)
        for rep in range(2):
            d_v = torch.full((n,), np.nan, dtype=torch.float64, device=dev)
            d_d = torch.full((n, 44), np.nan, dtype=torch.float64, device=dev)
            d_h = torch.full((n, 44, 44), np.nan, dtype=torch.float64, device=dev)
            d_c = torch.zeros(n, 2, dtype=torch.int64, device=dev)
            d_s = torch.full((n,), -1, dtype=torch.int32, device=dev)
            with _env(CELESTE_EVAL_FUSED=1):
                ctx.eval_batch_device(d_vp.data_ptr(), n, d_tg.data_ptr(), 7, d_v.data_ptr(), d_d.data_ptr(), d_h.data_ptr(),
                                      d_c.data_ptr(), d_s.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize(dev)
            assert (d_s.cpu().numpy() == 0).all(), (tg, rep, d_s.cpu().numpy())
            assert np.array_equal(d_v.cpu().numpy(), base[0][tg]) and np.array_equal(d_d.cpu().numpy(), base[1][tg])
            assert np.array_equal(d_h.cpu().numpy(), base[2][tg]) and np.array_equal(d_c.cpu().numpy(), base[3][tg])


def test_joint_infer_dataflow_equals_the_cpu_restatement(crowded, oracle):
    """celeste_joint_infer's dataflow launch (the production loop, ParallelRun.jl:135-196, 302-397) held DIRECTLY to the
    CPU restatement of maximize! driving the same schedule -- not through the layered device schedule: 40 crowded
    sources, Cyclades batches of 12, 2 sweeps, position boxes pinned at the initial positions.  Same bars as
    test_joint_inference_on_the_device_equals_the_cpu_restatement: every parameter within 1e-6 at three Newton
    iterations per entry (well-conditioned sub-problems); at six, the objective reached within 1e-9."""
    import celeste_jl_amd as cel
    from celeste_jl_amd.infer import joint_infer_sweeps, joint_layers
    from celeste_jl_amd.params import catalog_init_source, generic_init_source
    f, ctx = crowded
    S = len(f.catalog)
    targets = [t for t in range(S) if t % 8 != 5]              # five sources stay frozen neighbours
    frozen = [t for t in range(S) if t % 8 == 5]
    vp0 = np.stack([catalog_init_source(ce) for ce in f.catalog])
    for t in targets:
        vp0[t] = generic_init_source(f.catalog[t].pos)

    def run(max_iters):
        layers = joint_layers(targets, f.neighbors, batch_size=12, n_iters=2, rng=np.random.default_rng(3))
        centers = [vp0[layer, 0:2].copy() for layer in layers]
        with _env(CELESTE_JOINT_DATAFLOW=1, CELESTE_OPT_FUSED=1):
            dev, its, evals, el, st = ctx.joint_infer(vp0, layers, cel.ElboConfig(max_iters=max_iters), pos_centers=centers)
        assert (st == 0).all() and its.max() == max_iters

        def layer_cpu(vp, layer, pc):
            rows = []
            for t, c in zip(layer, pc):
                r = oracle.maximize(ctx.problem, vp, t, oracle.OptCfg(max_iters=max_iters), pos_center=c)
                assert r[4] == 0
                rows.append(r[0][t])
            return np.stack(rows)
        cpu = joint_infer_sweeps(layer_cpu, vp0.copy(), targets, f.neighbors, batch_size=12, n_iters=2,
                                 rng=np.random.default_rng(3))
        absdiff = np.abs(dev - cpu)
        med = np.median((absdiff / np.maximum(np.abs(cpu), 1e-3))[targets])
        moved = np.abs(cpu[targets] - vp0[targets]).max()
        worst = np.unravel_index(np.argmax(absdiff), absdiff.shape)
        print("dataflow launch vs CPU restatement, %d layers / %d entries, max_iters %d: max |diff| %.2e (source %d parameter "
              "%d), median relative %.1e; parameters moved by up to %.2g"
              % (len(layers), sum(map(len, layers)), max_iters, absdiff.max(), worst[0], worst[1], med, moved))
        assert moved > 0.1 and np.array_equal(dev[frozen], vp0[frozen])
        return dev, cpu, absdiff.max(), med
    _, _, absdiff, med = run(3)
    assert absdiff <= 1e-6 and med <= 1e-9, (absdiff, med)
    dev, cpu, absdiff, med = run(6)
    tg = np.array(targets, dtype=np.int32)
    e_dev = ctx.eval_batch(dev, tg, 4)[0]
    e_cpu = ctx.eval_batch(cpu, tg, 4)[0]
    rel = np.abs(e_dev - e_cpu) / np.abs(e_cpu)
    print("    ELBO reached, dataflow vs CPU tables: max relative difference %.1e" % rel.max())
    assert med <= 1e-9 and rel.max() <= 1e-9, (absdiff, med, rel.max())
