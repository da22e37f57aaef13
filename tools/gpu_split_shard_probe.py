"""A rank's small shard (N = 4, 8: 500 / 250 targets of the bench field) swept as 1, 2, 4 parts on as many contexts (one
image handle) and streams: how much of the five dependent launches' latency hides when the parts' chains overlap.
Run through gpurun.

Measured (round 4, gpurun_out/r04c/split.txt): the split LOSES at every size -- 250 targets 0.176 / 0.193 / 0.217 ms as 1 / 2 / 4
parts, 500 targets 0.239 / 0.254 / 0.281, 1000 targets 0.388 / 0.397 / 0.431, and the full 2000-target sweep 0.733 -> 0.750 ms
with two parts: launches of different streams do not overlap usefully here (every kernel of a chain already spreads over the
chip; what a second stream adds is queue hand-over, not occupancy).  The product therefore keeps one chain per rank."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import celeste_jl_amd as cel
from celeste_jl_amd import cabi
from celeste_jl_amd.partition import shard_targets, estimate_time

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
dev = torch.device("cuda", 0)
costs = [estimate_time(fld.patches[s]) for s in range(S)]
P = 44
iset = cabi.ImageSet(fld.images)
ctxs = [cel.FieldContext(fld.images, fld.patches, fld.neighbors, image_set=iset) for _ in range(4)]
streams = [torch.cuda.Stream(dev) for _ in range(4)]
s0 = torch.cuda.current_stream(dev)
vp = torch.tensor(fld.vp, dtype=torch.float64, device=dev)
for world in (8, 4, 2):
    mine = np.asarray(shard_targets(costs, world)[0], dtype=np.int32)
    n = len(mine)
    blk = torch.zeros(n * (1 + P), dtype=torch.float64, device=dev)
    h = torch.zeros(n, P, P, dtype=torch.float64, device=dev)
    cnt = torch.zeros(n, 2, dtype=torch.int64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    ref = None
    for parts in (1, 2, 4):
        sub = shard_targets([costs[t] for t in mine], parts)      # cost-balanced parts of the shard
        order = np.concatenate([np.asarray(s, dtype=np.int64) for s in sub])
        tg = torch.tensor(mine[order], dtype=torch.int32, device=dev)
        offs = np.cumsum([0] + [len(s) for s in sub])

        def sweep():
            for k in range(parts):
                o, m = int(offs[k]), int(offs[k + 1] - offs[k])
                stream = streams[k] if parts > 1 else s0
                ctxs[k].eval_batch_device(vp.data_ptr(), m, tg.data_ptr() + 4 * o, 7, blk.data_ptr() + 8 * o,
                                          blk.data_ptr() + 8 * n + 8 * P * o, h.data_ptr() + 8 * P * P * o,
                                          cnt.data_ptr() + 16 * o, st.data_ptr() + 4 * o, stream.cuda_stream)
        for _ in range(3):
            sweep()
        torch.cuda.synchronize()
        K = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        for _ in range(K):
            if parts > 1:
                for k in range(parts):
                    streams[k].wait_stream(s0)
            sweep()
            if parts > 1:
                for k in range(parts):
                    s0.wait_stream(streams[k])
        e1.record(s0)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        got = np.empty(n); got[order] = blk[:n].cpu().numpy()
        if ref is None:
            ref = got
        print("N = %d: %4d targets as %d part(s): %.4f ms per sweep   (values identical to one part: %s)"
              % (world, n, parts, ms, np.array_equal(got, ref)))
