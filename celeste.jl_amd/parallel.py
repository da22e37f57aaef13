"""Multi-GPU sweep: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Sources shard embarrassingly: every rank holds the replicated images, evaluates
its cost-balanced shard of targets, and the per-source results (value + 44-gradient, optionally the
Hessian) are all-gathered once per sweep -- the "catalog gather".  No other exchange exists on this
path, mirroring the reference's thread-level independence (ParallelRun.jl:546-607).
"""
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from .partition import shard_targets

P = 44


def sharded_sweep(evaluate: Callable[[Sequence[int]], Tuple[np.ndarray, np.ndarray]], targets: Sequence[int],
                  costs: Sequence[float], rank: int, world: int, all_gather: Optional[Callable] = None):
    """Evaluate `targets` sharded over `world` ranks and return (v[n], d[n,44]) for all targets, in order.

    evaluate(local_targets) -> (v, d) runs on this rank's device (FieldContext.eval_batch in production).
    all_gather(local_block: np.ndarray [max_shard, 45]) -> list of `world` blocks; defaults to
    torch.distributed.all_gather on the current default group.
    """
    targets = list(targets)
    shards = shard_targets(costs, world)
    mine = [targets[i] for i in shards[rank]]
    v, d = evaluate(mine) if mine else (np.zeros(0), np.zeros((0, P)))
    width = max(len(s) for s in shards)
    block = np.zeros((width, 1 + P))
    block[:len(mine), 0] = v
    block[:len(mine), 1:] = d
    if world == 1:
        blocks = [block]
    elif all_gather is not None:
        blocks = all_gather(block)
    else:
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.from_numpy(block).to(dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        blocks = [o.cpu().numpy() for o in outs]
    out_v = np.zeros(len(targets))
    out_d = np.zeros((len(targets), P))
    for r in range(world):
        idx = shards[r]
        out_v[idx] = blocks[r][:len(idx), 0]
        out_d[idx] = blocks[r][:len(idx), 1:]
    return out_v, out_d


def sharded_maximize(maximize: Callable[[Sequence[int]], np.ndarray], targets: Sequence[int], costs: Sequence[float],
                     rank: int, world: int, all_gather: Optional[Callable] = None) -> np.ndarray:
    """one_node_single_infer across ranks (ParallelRun.jl:546-607): every rank optimises its cost-balanced shard
    of targets against replicated images (neighbours frozen, so shards are independent) and the optimised
    44-vectors are all-gathered.  maximize(local_targets) -> [n_local, 44].  Returns [len(targets), 44]."""
    targets = list(targets)
    shards = shard_targets(costs, world)
    mine = [targets[i] for i in shards[rank]]
    res = maximize(mine) if mine else np.zeros((0, P))
    width = max(len(s) for s in shards)
    block = np.zeros((width, P))
    block[:len(mine)] = res
    if world == 1:
        blocks = [block]
    elif all_gather is not None:
        blocks = all_gather(block)
    else:
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.from_numpy(block).to(dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        blocks = [o.cpu().numpy() for o in outs]
    out = np.zeros((len(targets), P))
    for r in range(world):
        out[shards[r]] = blocks[r][:len(shards[r])]
    return out
