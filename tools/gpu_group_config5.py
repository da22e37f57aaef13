"""BASELINE configs[4] at full size (16 overlapping fields, 80 images, 30 000 sources) through the device group: maximize! of every
source and one sweep of joint inference -- FieldContext against a group of one member (RCCL, one rank) and a group of two members
sharing the device (peer copies); results bit for bit, wall time of the C calls (gpurun; one GPU).
usage: python tools/gpu_group_config5.py [max_iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.group import FieldGroup, cyclades_schedule, schedule_layers

max_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t0 = time.time()
fld = bench.build_multifield((4, 4), 2048, 1489, 30000, 5)
S = len(fld.catalog)
tg = list(range(S))
print("generated %d sources on %d images in %.0f s" % (S, len(fld.images), time.time() - t0), flush=True)
cfg = cel.ElboConfig(max_iters=max_iters)
b_off, c_off, flat = cyclades_schedule(tg, fld.neighbors)
layers, entries = schedule_layers(b_off, c_off, flat, 1)
flat_entry = np.concatenate([np.asarray(e) for e in entries])
print("Cyclades: %d batches, %d components, %d layers" % (len(b_off) - 1, len(c_off) - 1, len(layers)), flush=True)


def timed(f):
    t = time.perf_counter(); r = f(); return r, time.perf_counter() - t


ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
ref_m, t_m = timed(lambda: ctx.maximize_batch(fld.vp, tg, cfg, raise_on_error=False))
ref_j, t_j = timed(lambda: ctx.joint_infer(fld.vp, layers, cfg))
print("%-40s maximize! %.3f s (%d iterations at most, %d failed)   joint sweep %.3f s" %
      ("FieldContext", t_m, int(ref_m[1].max()), int((ref_m[4] != 0).sum()), t_j), flush=True)
ctx.close()
for devices in ([0], [0, 0]):
    g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=devices)
    got_m, t_m = timed(lambda: g.maximize_batch(fld.vp, tg, cfg, raise_on_error=False))
    same_m = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref_m, got_m))
    (new, its, evals, el, st, nx), t_j = timed(lambda: g.joint_infer(fld.vp, b_off, c_off, flat, 1, cfg))
    same_j = (np.array_equal(new, ref_j[0], equal_nan=True) and np.array_equal(its.reshape(-1)[flat_entry], ref_j[1])
              and np.array_equal(el.reshape(-1)[flat_entry], ref_j[3], equal_nan=True))
    print("%-40s maximize! %.3f s identical %s   joint sweep %.3f s identical %s (%d exchanges)" %
          ("group of %d member(s), %s" % (len(devices), g.info()["exchange"]), t_m, same_m, t_j, same_j, nx), flush=True)
    g.close()
