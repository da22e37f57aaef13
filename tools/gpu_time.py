"""Time the three kernels on the bench field (run through gpurun).  usage: gpu_time.py [n_iter]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
tg = np.arange(S, dtype=np.int32)
ref = None
for cfg in os.environ.get("CHUNKS", "1024").split(","):
    chunk = int(cfg)
    os.environ["CELESTE_CHUNK_PX"] = str(chunk)
    ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
    ctx.enable_timing(True)
    ms = []
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        g = ctx.eval_batch(fld.vp, tg, int(os.environ.get('FLAGS', '7')))
        ms.append(ctx.last_kernel_ms())
    ms = np.array(ms)[2:].mean(axis=0)
    if ref is None:
        ref = g
    err = max(float(np.abs(g[i] - ref[i]).max() / np.abs(ref[i]).max()) for i in range(3) if g[i] is not None)
    print("chunk %5d: prep %.3f pixel %.3f lift %.3f ms  | visits %d inactive %d | vs first cfg %.1e"
          % (chunk, ms[0], ms[1], ms[2], g[3][:, 0].sum(), g[3][:, 1].sum(), err))
    ctx.close()
