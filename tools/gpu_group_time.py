"""What the device group costs on top of the one-device entry points (gpurun; one GPU): the same calls through a FieldContext,
a group of one member (RCCL, one rank) and a group of two members sharing the device (peer copies), on the bench field."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel
from celeste_jl_amd.group import FieldGroup, cyclades_schedule, schedule_layers
from celeste_jl_amd.infer import default_infer_config
from celeste_jl_amd.params import init_source_table

fld = bench.build_field(2048, 1489, 2000, 3)
S = len(fld.catalog)
tg = list(range(S))
ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
cfg = default_infer_config()
vp_j = init_source_table(fld.catalog, tg)
b_off, c_off, flat = cyclades_schedule(tg, fld.neighbors)
layers, entries = schedule_layers(b_off, c_off, flat, 3)
pos = vp_j[flat, 0:2].copy()


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)


rows = [("FieldContext (one device, no group)", best(lambda: ctx.eval_batch(fld.vp, tg)), best(lambda: ctx.maximize_batch(fld.vp, tg, cel.ElboConfig())),
         best(lambda: ctx.joint_infer(vp_j, layers, cfg, pos_centers=[pos[e] for e in entries])))]
for devices in ([0], [0, 0]):
    g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=devices)
    rows.append(("group of %d member(s), %s" % (len(devices), g.info()["exchange"]), best(lambda: g.eval_batch(fld.vp, tg)),
                 best(lambda: g.maximize_batch(fld.vp, tg, cel.ElboConfig())),
                 best(lambda: g.joint_infer(vp_j, b_off, c_off, flat, 3, cfg, pos_centers=pos))))
    g.close()
print("%-42s %14s %14s %14s" % ("2000 sources, host pointers, best of 3", "elbo sweep ms", "maximize! ms", "joint (3 sw) ms"))
for name, a, b, c in rows:
    print("%-42s %14.2f %14.2f %14.2f" % (name, a * 1e3, b * 1e3, c * 1e3))
