"""Multi-GPU sweep: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Sources shard embarrassingly: every rank holds the replicated images, evaluates
its cost-balanced shard of targets, and the per-source results (value + 44-gradient, optionally the
Hessian) are all-gathered once per sweep -- the "catalog gather".  No other exchange exists on this
path, mirroring the reference's thread-level independence (ParallelRun.jl:546-607).
"""
from typing import Callable, Optional, Sequence, Tuple

import os

import numpy as np

from .partition import shard_targets

P = 44


def _default_all_gather(block: np.ndarray, world: int, device: Optional[int]):
    """torch.distributed.all_gather of one equally-shaped block per rank.  With the nccl (= RCCL) backend the block is
    staged on `device` -- the HIP device of this rank's FieldContext, which must be given: the library only calls
    hipSetDevice internally, so torch's current device says nothing about where the rank computes."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        if device is None:
            raise ValueError("the nccl backend needs the device index of this rank's FieldContext (device=...)")
        t = torch.from_numpy(block).to("cuda:%d" % device)
    else:
        t = torch.from_numpy(block)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def sharded_sweep(evaluate: Callable[[Sequence[int]], Tuple[np.ndarray, np.ndarray]], targets: Sequence[int],
                  costs: Sequence[float], rank: int, world: int, all_gather: Optional[Callable] = None,
                  device: Optional[int] = None):
    """Evaluate `targets` sharded over `world` ranks and return (v[n], d[n,44]) for all targets, in order.

    evaluate(local_targets) -> (v, d) runs on this rank's device (FieldContext.eval_batch in production).
    all_gather(local_block: np.ndarray [max_shard, 45]) -> list of `world` blocks; defaults to
    torch.distributed.all_gather on the current default group (`device` = the HIP device of this rank's
    FieldContext, required with the nccl backend).
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
        blocks = _default_all_gather(block, world, device)
    out_v = np.zeros(len(targets))
    out_d = np.zeros((len(targets), P))
    for r in range(world):
        idx = shards[r]
        out_v[idx] = blocks[r][:len(idx), 0]
        out_d[idx] = blocks[r][:len(idx), 1:]
    return out_v, out_d


def sharded_maximize(maximize: Callable[[Sequence[int]], np.ndarray], targets: Sequence[int], costs: Sequence[float],
                     rank: int, world: int, all_gather: Optional[Callable] = None,
                     device: Optional[int] = None) -> np.ndarray:
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
        blocks = _default_all_gather(block, world, device)
    out = np.zeros((len(targets), P))
    for r in range(world):
        out[shards[r]] = blocks[r][:len(shards[r])]
    return out


def gather_status(local_status: Sequence[int], n_targets: int, costs: Sequence[float], rank: int, world: int,
                  all_gather: Optional[Callable] = None, device: Optional[int] = None) -> np.ndarray:
    """The per-target status codes of a sharded optimisation on EVERY rank (each rank only knows its shard's): the same
    partition as sharded_maximize, one more all-gather of one column.  Without it `failed` sets and warnings differ
    between ranks although the parameter tables agree."""
    shards = shard_targets(costs, world)
    width = max(len(s) for s in shards)
    block = np.zeros((width, 1))
    block[:len(shards[rank]), 0] = np.asarray(local_status, dtype=np.float64)
    if world == 1:
        blocks = [block]
    elif all_gather is not None:
        blocks = all_gather(block)
    else:
        blocks = _default_all_gather(block, world, device)
    out = np.zeros(n_targets, dtype=np.int32)
    for r in range(world):
        out[shards[r]] = np.rint(blocks[r][:len(shards[r]), 0]).astype(np.int32)
    return out


class DeviceShardedSweep:
    """BASELINE.json configs[3]: ONE field, its targets sharded by source across `world` ranks, on the device.

    The reference drains one field's source list with N workers (ParallelRun.jl:546-607); here every rank holds the
    replicated images (its own FieldContext over the whole field), evaluates its cost-balanced shard of the targets
    (`estimate_time`, ParallelRun.jl:45-56 -> partition.shard_targets) with one `celeste_elbo_eval_batch_device` launch,
    and the per-source results (value + 44-gradient, 360 B per target) are all-gathered -- the "catalog gather", the
    only exchange of the path.  Every shard is padded to the widest one so that one `all_gather_into_tensor` serves:
    the library writes the values and gradients of a rank straight into its gather block ([width] values followed by
    [width x 44] gradients), no packing kernel.  The gather of sweep k runs on its own stream and overlaps the kernels
    of sweep k + 1 (two blocks, alternating).

    backend "nccl" (= RCCL): device-side gather over xGMI.  backend "gloo": the block is staged through the host --
    this is how two ranks share ONE GPU in the tests (RCCL refuses two ranks on a device).
    Hessians stay on the rank that owns the target (`hessians()`), as the optimiser consumes them locally.
    """

    def __init__(self, ctx, targets: Sequence[int], costs: Sequence[float], rank: int, world: int, flags: int,
                 backend: Optional[str] = None, shards=None):
        """`shards` (index lists into `targets`, one per rank) overrides the cost-balanced partition -- bench.py's
        weak-scaling mode, where every rank sweeps all targets of its own field, uses it."""
        import torch
        self.torch = torch
        self.ctx, self.rank, self.world, self.flags = ctx, rank, world, flags
        self.targets = np.asarray(targets, dtype=np.int32)
        assert len(costs) == len(self.targets)
        self.shards = shards if shards is not None else shard_targets(costs, world)
        self.mine = self.targets[self.shards[rank]]
        self.n = int(self.mine.size)
        self.width = max(1, max(len(s) for s in self.shards))
        self.dev = torch.device("cuda", ctx.device)
        # CELESTE_GATHER_SINGLE=1 (tests): a process group of ONE rank still runs the catalog gather, so that the
        # RCCL path -- gather stream, events, all_gather_into_tensor -- is exercised on a one-GPU box
        self.gather = world > 1
        if world == 1 and os.environ.get("CELESTE_GATHER_SINGLE") == "1":
            import torch.distributed as dist
            self.gather = dist.is_available() and dist.is_initialized()
        if self.gather:
            import torch.distributed as dist
            self.dist = dist
            self.backend = backend or dist.get_backend()
        else:
            self.backend = backend or "none"
        W = self.width
        with torch.cuda.device(self.dev):
            self.d_tg = torch.tensor(self.mine, dtype=torch.int32, device=self.dev)
            self.blocks = [torch.zeros(W * (1 + P), dtype=torch.float64, device=self.dev) for _ in range(2)]
            want_h = bool(flags & 2)
            self.d_h = torch.zeros(max(self.n, 1), P, P, dtype=torch.float64, device=self.dev) if want_h else None
            self.d_cnt = torch.zeros(max(self.n, 1), 2, dtype=torch.int64, device=self.dev)
            self.d_st = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.dev)
            self.compute_stream = torch.cuda.current_stream(self.dev)
            if self.gather and self.backend == "nccl":
                self.comm_stream = torch.cuda.Stream(self.dev)
                self.gathered = torch.zeros(world * W * (1 + P), dtype=torch.float64, device=self.dev)
                self.buf_free = [torch.cuda.Event(), torch.cuda.Event()]
                for e in self.buf_free:
                    e.record(self.compute_stream)
            elif self.gather:
                self.h_block = torch.zeros(W * (1 + P), dtype=torch.float64).pin_memory()
                self.gathered = torch.zeros(world * W * (1 + P), dtype=torch.float64)
        self.k = 0
        self.last = 0
        self.gather_bytes = world * W * (1 + P) * 8 if self.gather else 0

    def step(self, d_vp_ptr: int):
        """One sweep: evaluate this rank's shard against the parameter table at device pointer `d_vp_ptr`
        (n_sources x 44 doubles), then start the catalog gather.  Asynchronous with the nccl backend."""
        torch = self.torch
        k = self.k & 1
        self.k += 1
        self.last = k
        blk = self.blocks[k]
        nccl = self.gather and self.backend == "nccl"
        if nccl:
            self.compute_stream.wait_event(self.buf_free[k])   # the gather that last read this block is through
        if self.n > 0:
            base = blk.data_ptr()
            self.ctx.eval_batch_device(d_vp_ptr, self.n, self.d_tg.data_ptr(), self.flags, base, base + 8 * self.width,
                                       self.d_h.data_ptr() if self.d_h is not None else 0, self.d_cnt.data_ptr(),
                                       self.d_st.data_ptr(), self.compute_stream.cuda_stream)
        if nccl:
            done = torch.cuda.Event()
            done.record(self.compute_stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(done)
                self.dist.all_gather_into_tensor(self.gathered, blk)
                self.buf_free[k].record(self.comm_stream)
        elif self.gather:   # gloo: stage through the host (two ranks on one GPU in the tests)
            self.h_block.copy_(blk, non_blocking=True)
            self.compute_stream.synchronize()
            outs = list(self.gathered.view(self.world, -1).unbind(0))
            self.dist.all_gather(outs, self.h_block)

    def wait(self):
        """Every launched sweep and every gather is complete when this returns."""
        if self.gather and self.backend == "nccl":
            self.comm_stream.synchronize()
        self.torch.cuda.synchronize(self.dev)

    def results(self):
        """(v[len(targets)], d[len(targets), 44]) of the last sweep for ALL targets, in the order of `targets`
        (identical on every rank), plus the status codes and pixel counters of this rank's shard."""
        self.wait()
        W = self.width
        g = (self.gathered if self.gather else self.blocks[self.last]).cpu().numpy().reshape(self.world, W * (1 + P))
        v = np.zeros(len(self.targets))
        d = np.zeros((len(self.targets), P))
        for r in range(self.world):
            idx = self.shards[r]
            v[idx] = g[r, :len(idx)]
            d[idx] = g[r, W:].reshape(W, P)[:len(idx)]
        return v, d, self.d_st[:self.n].cpu().numpy(), self.d_cnt[:self.n].cpu().numpy()

    def hessians(self):
        """[n_local, 44, 44] Hessians of this rank's shard (targets `self.mine`)."""
        self.wait()
        return None if self.d_h is None else self.d_h[:self.n].cpu().numpy()


class DeviceJointInfer:
    """Joint inference across ranks with the parameter table resident in HBM on every rank (ParallelRun.jl:135-196 on
    N GPUs): for every layer of the schedule each rank optimises its cost-balanced shard of the layer in place
    (`celeste_maximize_batch_device`: no table H2D / D2H), then the optimised rows -- 44 doubles + the status, padded to
    the widest shard -- are all-gathered with ONE `all_gather_into_tensor` on device blocks (RCCL over xGMI; 360 B per
    target and layer) and scattered into every rank's table.  With the gloo backend the blocks are
    staged through the host (two ranks sharing one GPU in the tests)."""

    def __init__(self, ctx, vp: np.ndarray, rank: int, world: int, cfg=None, backend: Optional[str] = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.rank, self.world, self.cfg = ctx, rank, world, cfg
        self.backend = backend or (dist.get_backend() if world > 1 else "none")
        self.dev = torch.device("cuda", ctx.device)
        with torch.cuda.device(self.dev):
            self.d_vp = torch.tensor(np.ascontiguousarray(vp, dtype=np.float64).reshape(ctx.S, P), device=self.dev)

    def layer(self, layer: Sequence[int], costs: Sequence[float], pos_centers: np.ndarray) -> np.ndarray:
        """Optimise the sources of `layer` (mutually non-neighbouring) against the shared table; returns their status
        codes, identical on every rank."""
        torch = self.torch
        layer = np.asarray(layer, dtype=np.int64)
        shards = shard_targets(costs, self.world)
        mine = shards[self.rank]
        W = max(1, max(len(s) for s in shards))
        with torch.cuda.device(self.dev):
            stream = torch.cuda.current_stream(self.dev)
            block = torch.zeros(W, P + 1, dtype=torch.float64, device=self.dev)
            if len(mine):
                d_tg = torch.tensor(layer[mine], dtype=torch.int32, device=self.dev)
                d_pc = torch.tensor(np.ascontiguousarray(pos_centers[mine], dtype=np.float64), device=self.dev)
                d_st = torch.zeros(len(mine), dtype=torch.int32, device=self.dev)
                self.ctx.maximize_batch_device(self.d_vp.data_ptr(), len(mine), d_tg.data_ptr(), self.cfg,
                                               d_pos_centers=d_pc.data_ptr(), d_status=d_st.data_ptr(),
                                               stream=stream.cuda_stream)
                block[:len(mine), :P] = self.d_vp[d_tg.long()]
                block[:len(mine), P] = d_st.double()
            if self.world == 1:
                gathered = block.unsqueeze(0)
            elif self.backend == "nccl":
                gathered = torch.empty(self.world, W, P + 1, dtype=torch.float64, device=self.dev)
                self.dist.all_gather_into_tensor(gathered, block)
            else:   # gloo: through the host
                h = block.cpu()
                outs = [torch.empty_like(h) for _ in range(self.world)]
                self.dist.all_gather(outs, h)
                gathered = torch.stack(outs).to(self.dev)
            status = np.zeros(len(layer), dtype=np.int32)
            for r in range(self.world):
                idx = shards[r]
                if not len(idx):
                    continue
                rows = torch.tensor(layer[idx], dtype=torch.int64, device=self.dev)
                if r != self.rank:
                    self.d_vp[rows] = gathered[r, :len(idx), :P]
                status[idx] = gathered[r, :len(idx), P].round().to(torch.int32).cpu().numpy()
        return status

    def table(self) -> np.ndarray:
        self.torch.cuda.synchronize(self.dev)
        return self.d_vp.cpu().numpy()
