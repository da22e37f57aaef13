"""Strong-scaling estimate on ONE GPU (run through gpurun): rank 0's shard of the bench field for world = 1, 2, 4, 8 --
sweep time and its kernels; T(1) / (N T(N)) is the efficiency the compute allows (the gather overlaps the next sweep)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import celeste_jl_amd as cel
from celeste_jl_amd import cabi
from celeste_jl_amd.partition import shard_targets, estimate_time

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
costs = [estimate_time(fld.patches[s]) for s in range(S)]
dev = torch.device("cuda", 0)
d_vp = torch.tensor(fld.vp, dtype=torch.float64, device=dev)
P = 44
t1 = None
for world in [int(w) for w in os.environ.get('WORLDS', '1,2,4,8,16').split(',')]:
    shards = shard_targets(costs, world)
    mine = np.asarray(shards[0], dtype=np.int32)
    n = mine.size
    d_tg = torch.tensor(mine, dtype=torch.int32, device=dev)
    blk = torch.zeros(n * (1 + P), dtype=torch.float64, device=dev)
    d_h = torch.zeros(n, P, P, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(n, 2, dtype=torch.int64, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def sweep():
        ctx.eval_batch_device(d_vp.data_ptr(), n, d_tg.data_ptr(), 7, blk.data_ptr(), blk.data_ptr() + 8 * n, d_h.data_ptr(),
                              d_cnt.data_ptr(), d_st.data_ptr(), stream.cuda_stream)
    for _ in range(5):
        sweep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    e0.record(stream)
    for _ in range(K):
        sweep()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    ctx.enable_timing(True); sweep(); torch.cuda.synchronize(); km = ctx.last_kernel_ms(); ctx.enable_timing(False)
    if t1 is None:
        t1 = ms
    print("world %2d: rank 0 has %4d targets, %.3f ms per sweep -> efficiency %.2f | kernels %s"
          % (world, n, ms, t1 / (world * ms), [round(x, 4) for x in km]))
