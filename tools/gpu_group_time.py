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


def best(f, reps=6):
    f(); f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)


rows = [("FieldContext (one device, no group)", best(lambda: ctx.eval_batch(fld.vp, tg)), best(lambda: ctx.maximize_batch(fld.vp, tg, cel.ElboConfig())),
         best(lambda: ctx.joint_infer(vp_j, layers, cfg, pos_centers=[pos[e] for e in entries])))]
for devices in ([0], [0], [0, 0]):      # (the one-member group twice: a group's copy stream lands on whichever DMA engine the runtime hands out)
    g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=devices)
    rows.append(("group of %d member(s), %s" % (len(devices), g.info()["exchange"]), best(lambda: g.eval_batch(fld.vp, tg)),
                 best(lambda: g.maximize_batch(fld.vp, tg, cel.ElboConfig())),
                 best(lambda: g.joint_infer(vp_j, b_off, c_off, flat, 3, cfg, pos_centers=pos))))
    g.close()
# where a one-member group's host-pointer sweep spends its time: the resident calls one by one
g = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=[0])
def _phases():
    t = [time.perf_counter()]
    g.plan(fld.vp, tg); t.append(time.perf_counter())
    g.sweep(); g.wait(); t.append(time.perf_counter())
    g.results(hessians=False); t.append(time.perf_counter())
    g.results(); t.append(time.perf_counter())
    return np.diff(t) * 1e3
_phases()
ph = np.min([_phases() for _ in range(5)], axis=0)
print("one member, resident calls: plan %.3f ms, sweep+wait %.3f ms, results without Hessians %.3f ms, results with %.3f ms" % tuple(ph))
os.environ["CELESTE_GROUP_TRACE"] = "1"
g2 = FieldGroup(fld.images, fld.patches, fld.neighbors, devices=[0])
for _ in range(4):
    t0 = time.perf_counter(); g2.eval_batch(fld.vp, tg); print("  python call %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
del os.environ["CELESTE_GROUP_TRACE"]
ts = []
for _ in range(10):
    t0 = time.perf_counter(); g2.eval_batch(fld.vp, tg); ts.append((time.perf_counter() - t0) * 1e3)
print("one member, host-pointer eval_batch, 10 calls in a row (ms):", " ".join("%.2f" % t for t in ts))
ts = []
for _ in range(10):
    t0 = time.perf_counter(); ctx.eval_batch(fld.vp, tg); ts.append((time.perf_counter() - t0) * 1e3)
print("FieldContext, the same (ms):", " ".join("%.2f" % t for t in ts))
g2.close()
g.close()
print("%-42s %14s %14s %14s" % ("2000 sources, host pointers, best of 6", "elbo sweep ms", "maximize! ms", "joint (3 sw) ms"))
for name, a, b, c in rows:
    print("%-42s %14.2f %14.2f %14.2f" % (name, a * 1e3, b * 1e3, c * 1e3))
