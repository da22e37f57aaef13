"""How much of a sweep's small kernels (work list, prep, value, lift) would hide under the pixel kernel if two halves of
the batch ran on two streams: two contexts over the same field, each sweeping one half of the targets on its own stream,
against one context sweeping everything.  Run through gpurun."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.partition import shard_targets, estimate_time

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
dev = torch.device("cuda", 0)
costs = [estimate_time(fld.patches[s]) for s in range(S)]
P = 44


def make(targets):
    ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
    n = len(targets)
    return dict(ctx=ctx, n=n, tg=torch.tensor(np.asarray(targets, dtype=np.int32), device=dev),
                vp=torch.tensor(fld.vp, dtype=torch.float64, device=dev), blk=torch.zeros(n * (1 + P), dtype=torch.float64, device=dev),
                h=torch.zeros(n, P, P, dtype=torch.float64, device=dev), cnt=torch.zeros(n, 2, dtype=torch.int64, device=dev),
                st=torch.zeros(n, dtype=torch.int32, device=dev))


def sweep(w, stream):
    w["ctx"].eval_batch_device(w["vp"].data_ptr(), w["n"], w["tg"].data_ptr(), 7, w["blk"].data_ptr(), w["blk"].data_ptr() + 8 * w["n"],
                               w["h"].data_ptr(), w["cnt"].data_ptr(), w["st"].data_ptr(), stream.cuda_stream)


whole = make(list(range(S)))
halves = [make(sh) for sh in shard_targets(costs, 2)]
s0 = torch.cuda.current_stream(dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for _ in range(5):
    sweep(whole, s0); sweep(halves[0], s1); sweep(halves[1], s2)
torch.cuda.synchronize()
K = 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s0)
for _ in range(K):
    sweep(whole, s0)
e1.record(s0); torch.cuda.synchronize()
t_whole = e0.elapsed_time(e1) / K
e0.record(s0)
s1.wait_stream(s0); s2.wait_stream(s0)
for _ in range(K):
    sweep(halves[0], s1); sweep(halves[1], s2)
s0.wait_stream(s1); s0.wait_stream(s2)
e1.record(s0); torch.cuda.synchronize()
t_two = e0.elapsed_time(e1) / K
print("one stream, 2000 targets: %.3f ms per sweep; two streams, 1000 targets each: %.3f ms per pair" % (t_whole, t_two))
