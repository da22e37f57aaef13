"""N > 1 path on CPU: two gloo ranks shard the targets, evaluate their shard (oracle injected as the
evaluator -- tests may do that) and all-gather the catalog; the result must equal the single-rank sweep."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


JOINT_TARGETS = [0, 3, 5, 8, 13, 21]


def _joint(f, pb, oracle, rank, world, costs):
    from celeste_jl_amd.infer import joint_infer_sweeps

    def maximize_layer(vp, layer, pc):
        rows = []
        for t, c in zip(layer, pc):   # first sweep: the position boxes are centred on the current positions
            assert np.array_equal(c, vp[t, 0:2])
            rows.append(oracle.maximize(pb, vp, t, oracle.OptCfg(max_iters=2))[0][t])
        return np.stack(rows)
    vp = f.vp.copy()
    return joint_infer_sweeps(maximize_layer, vp, JOINT_TARGETS, f.neighbors, batch_size=3, n_iters=1,
                              rng=np.random.default_rng(5), rank=rank, world=world, costs=costs)[JOINT_TARGETS]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from celeste_jl_amd import synthetic, cabi
    from celeste_jl_amd.parallel import sharded_sweep
    from celeste_jl_amd.partition import estimate_time
    from oracle import oracle
    f = synthetic.make_field(160, 200, 24, seed=11)
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    costs = [estimate_time(row) for row in f.patches]

    def evaluate(tg):
        v, d, _, _, st = oracle.elbo_batch(pb, f.vp, tg, 3 | 4, n_threads=1)
        assert (st == 0).all()
        return v, d

    v, d = sharded_sweep(evaluate, list(range(24)), costs, rank, world)

    # the optimiser shards the same way (single inference: neighbours frozen => shards are independent)
    from celeste_jl_amd.parallel import sharded_maximize
    opt_targets = [2, 7, 11, 19]

    def maximize(tg):
        return np.stack([oracle.maximize(pb, f.vp, t, oracle.OptCfg(max_iters=2))[0][t] for t in tg])

    vs = sharded_maximize(maximize, opt_targets, [costs[t] for t in opt_targets], rank, world)
    # joint inference: every layer of every Cyclades batch is sharded, updated rows all-gathered per layer
    from celeste_jl_amd.infer import joint_infer_sweeps
    vj = _joint(f, pb, oracle, rank, world, costs)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), v=v, d=d, vs=vs, vj=vj)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_equals_single_rank(tmp_path, oracle):
    import torch.multiprocessing as mp
    from celeste_jl_amd import synthetic, cabi
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f = synthetic.make_field(160, 200, 24, seed=11)
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    v, d, _, _, _ = oracle.elbo_batch(pb, f.vp, list(range(24)), 7, n_threads=2)
    vs = np.stack([oracle.maximize(pb, f.vp, t, oracle.OptCfg(max_iters=2))[0][t] for t in (2, 7, 11, 19)])
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["v"], v) and np.array_equal(z["d"], d)
        assert np.array_equal(z["vs"], vs)
    from celeste_jl_amd.partition import estimate_time
    vj = _joint(f, pb, oracle, 0, 1, [estimate_time(row) for row in f.patches])
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["vj"], vj)
    assert not np.array_equal(vj, f.vp[JOINT_TARGETS])
