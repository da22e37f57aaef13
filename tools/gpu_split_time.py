"""Time the split variant (record write + streaming record sum) on the bench field (run through gpurun)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd import cabi

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
tg = np.arange(S, dtype=np.int32)
for sum_tiles in os.environ.get("SUM_TILES", "16").split(","):
  os.environ["CELESTE_SUM_TILES"] = sum_tiles
  ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
  st = ctx.work_stats(tg)
  ctx.enable_timing(True)
  ref = ctx.eval_batch(fld.vp, tg, 7)
  ms = []
  for it in range(8):
      g = ctx.eval_batch(fld.vp, tg, 7 | cabi.FLAG_SPLIT)
      ms.append(ctx.last_kernel_ms() + [ctx.last_record_sum_ms()])
  ms = np.array(ms)[2:].mean(axis=0)
  err = max(float(np.abs(g[i] - ref[i]).max() / np.abs(ref[i]).max()) for i in range(3))
  stored = st["record_tiles"] * 68 * 64 * 8
  print("sum_tiles", sum_tiles, "split: prep %.3f  record-write %.3f  lift %.3f  record-sum %.3f ms | algorithmic %.3f GB -> %.0f GB/s (%.1f%% of 8 TB/s)"
        " | stored %.3f GB -> %.0f GB/s | vs fused %.1e"
        % (ms[0], ms[1], ms[2], ms[3], st["record_bytes"] / 1e9, st["record_bytes"] / ms[3] / 1e6,
           st["record_bytes"] / ms[3] / 1e6 / 80, stored / 1e9, stored / ms[3] / 1e6, err))
