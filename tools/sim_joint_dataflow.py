"""CPU model of the joint dataflow launch (fused_kernels.h, joint mode) on the bench field: P workgroups, one FIFO queue,
item durations from the per-phase shader clocks (tools/gpu_optim_sections.py).  Needs gpurun_out/joint_evals.npz
(tools/gpu_joint_sim.py: evaluations per entry of the real run).  Prints the modelled makespan for a few what-ifs."""
import sys, os, heapq, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

d = np.load("gpurun_out/joint_evals.npz")
ev, tg = d["evals"], d["targets"]
fld = bench.build_field(2048, 1489, 2000, 3)
nb = [list(map(int, x)) for x in fld.neighbors]
S, E = len(nb), len(tg)
chunks = []
for s in range(S):
    n = 0
    for p in fld.patches[s]:
        px = p.active_pixel_bitmap.shape[0] * p.active_pixel_bitmap.shape[1] if hasattr(p, "active_pixel_bitmap") else 0
        n += (px + 255) // 256
    chunks.append(n)
chunks = np.array(chunks)
print("chunks per source: mean %.1f" % chunks.mean())
last = [-1] * S; readers = [[] for _ in range(S)]; succ = [[] for _ in range(E)]; dep = np.zeros(E, int)
for e in range(E):
    t = int(tg[e]); ds = set()
    if last[t] >= 0: ds.add(last[t])
    for u in nb[t]:
        if last[u] >= 0: ds.add(last[u])
    ds.update(readers[t])
    for q in ds: succ[q].append(e)
    dep[e] = len(ds); last[t] = e; readers[t] = []
    for u in nb[t]: readers[u].append(e)


def simulate(P=512, t_chunk=19.5, t_step=130.0, t_start=5.0, t_render=13.0, n_render=5, t_end=9.0, busy_factor=1.0, label="", urgent=None, t_chunk_busy=None):
    """event simulation: a workgroup takes the queue's items in order; the last chunk item of an evaluation continues with
    the lift + step (t_step) and pushes the next evaluation's items; busy_factor stretches item times while every
    workgroup is busy (two waves per SIMD)."""
    depc = dep.copy()
    left = ev.copy()                       # evaluations left per entry
    q = collections.deque(); qh = collections.deque()
    def put(item):
        (qh if urgent is not None and urgent[item[1]] else q).append(item)
    for e in range(E):
        if depc[e] == 0: put(("S", e))
    free = [(0.0, w) for w in range(P)]    # (time a workgroup becomes free, id)
    heapq.heapify(free)
    pending = []                           # (time, seq, items to push)
    arrivals = collections.Counter(); rarr = collections.Counter()
    seq = 0; now = 0.0; done = 0; end = 0.0
    while done < E:
        # release pushes up to the time the next workgroup is free
        tfree, w = free[0]
        while pending and ((not q and not qh) or pending[0][0] <= tfree):
            tp, _, items = heapq.heappop(pending)
            for k, e in items: put((k, e, tp))
        if not q and not qh:
            if not pending: break
            continue
        item = qh.popleft() if qh else q.popleft()
        kind, e = item[0], item[1]
        tavail = item[2] if len(item) > 2 else 0.0
        tfree, w = heapq.heappop(free)
        t0 = max(tfree, tavail)
        stretch = busy_factor if len(q) + len(qh) > P else 1.0
        nch = max(1, int(chunks[tg[e]]))
        if kind == "S":
            t1 = t0 + t_start; heapq.heappush(pending, (t1, seq, [("R", e)] * n_render)); seq += 1
        elif kind == "R":
            t1 = t0 + t_render; rarr[e] += 1
            if rarr[e] == n_render:
                heapq.heappush(pending, (t1, seq, [("C", e)] * nch)); seq += 1
        else:
            t1 = t0 + (t_chunk_busy if (t_chunk_busy is not None and len(q) + len(qh) > P) else t_chunk * stretch); arrivals[e] += 1
            if arrivals[e] == nch:
                arrivals[e] = 0
                t1 += t_step * (stretch if busy_factor > 1 else 1.0)
                left[e] -= 1
                if left[e] > 0:
                    heapq.heappush(pending, (t1, seq, [("C", e)] * nch)); seq += 1
                else:
                    t1 += t_end; done += 1; end = max(end, t1)
                    ready = []
                    for s2 in succ[e]:
                        depc[s2] -= 1
                        if depc[s2] == 0: ready.append(("S", s2))
                    if ready: heapq.heappush(pending, (t1, seq, ready)); seq += 1
        heapq.heappush(free, (t1, w))
    print("%-58s makespan %.1f ms" % (label, end / 1e3))
    return end


simulate(label="measured phase times (latency-bound values)")
simulate(busy_factor=1.25, label="... items 25 % slower while the queue is longer than the grid")
simulate(t_step=117.0, label="lift - 13 us")
simulate(t_step=105.0, label="lift + step - 25 us")
simulate(t_chunk=16.0, label="chunk item - 3.5 us")
simulate(P=1024, label="1024 workgroups")
simulate(P=100000, label="unbounded workgroups (critical path)")

simulate(t_step=118.0, label="lift - 12 us (this round's lift)")
simulate(t_step=118.0, t_chunk_busy=14.3, label="... + four records per workgroup while the queue is long")
# priority: bottom level of every entry from A-PRIORI estimates of its evaluations (sweep means), urgent = the top share
per_sweep = E // 3
est = np.array([22.0 if e < per_sweep else (5.6 if e < 2 * per_sweep else 3.2) for e in range(E)])
for name, cost in (("a-priori estimates", est), ("true evaluation counts (oracle)", ev.astype(float))):
    bl = np.zeros(E)
    for e in range(E - 1, -1, -1):
        bl[e] = cost[e] + (max(bl[s2] for s2 in succ[e]) if succ[e] else 0.0)
    for share in (0.1, 0.25, 0.5):
        thr = np.quantile(bl, 1 - share)
        simulate(urgent=bl >= thr, label="two queues, urgent = top %d %% by bottom level (%s)" % (100 * share, name))
