"""BASELINE configs[4] at full size (16 overlapping fields, 80 images, 30 000 sources) through the device group: maximize! of every
source and one sweep of joint inference -- FieldContext against a group of one member (RCCL, one rank) and a group of two members
sharing the device (peer copies); results bit for bit, wall time of the C calls (gpurun; one GPU).
usage: python tools/gpu_group_config5.py [max_iters [batch sizes, comma separated]]"""
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
batch_sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [400]


def timed(f):
    t = time.perf_counter(); r = f(); return r, time.perf_counter() - t


def schedule(bs):
    b_off, c_off, flat = cyclades_schedule(tg, fld.neighbors, batch_size=bs)
    layers, entries = schedule_layers(b_off, c_off, flat, 1)
    print("Cyclades, batch size %d: %d batches, %d components, %d layers" % (bs, len(b_off) - 1, len(c_off) - 1, len(layers)), flush=True)
    return b_off, c_off, flat, layers, np.concatenate([np.asarray(e) for e in entries])


scheds = {bs: schedule(bs) for bs in batch_sizes}
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
ref_m, t_m = timed(lambda: ctx.maximize_batch(fld.vp, tg, cfg, raise_on_error=False))
print("%-40s maximize! %.3f s (%d iterations at most, %d failed)" % ("FieldContext", t_m, int(ref_m[1].max()), int((ref_m[4] != 0).sum())), flush=True)
ref_j = {}
for bs, (b_off, c_off, flat, layers, fe) in scheds.items():
    ref_j[bs], t_j = timed(lambda: ctx.joint_infer(fld.vp, layers, cfg))
    print("%-40s joint sweep, batch size %d: %.3f s" % ("FieldContext", bs, t_j), flush=True)
ctx.close()
for devices in ([0], [0, 0]):
    g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=devices)
    name = "group of %d member(s), %s" % (len(devices), g.info()["exchange"])
    got_m, t_m = timed(lambda: g.maximize_batch(fld.vp, tg, cfg, raise_on_error=False))
    same_m = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref_m, got_m))
    print("%-40s maximize! %.3f s identical %s" % (name, t_m, same_m), flush=True)
    for bs, (b_off, c_off, flat, layers, fe) in scheds.items():
        (new, its, evals, el, st, nx), t_j = timed(lambda: g.joint_infer(fld.vp, b_off, c_off, flat, 1, cfg))
        r = ref_j[bs]
        same_j = (np.array_equal(new, r[0], equal_nan=True) and np.array_equal(its.reshape(-1)[fe], r[1])
                  and np.array_equal(el.reshape(-1)[fe], r[3], equal_nan=True))
        print("%-40s joint sweep, batch size %d: %.3f s identical %s (%d exchanges)" % (name, bs, t_j, same_j, nx), flush=True)
    g.close()
