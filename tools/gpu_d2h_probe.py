"""Device-to-host copy bandwidth per stream, in creation order (gpurun).  usage: gpu_d2h_probe.py [torch]
Every stream is created, then 32 MB are copied device -> page-locked host on it five times (HIP events); then pairs of streams
copy concurrently."""
import ctypes as C, sys, time
if "torch" in sys.argv:
    import torch
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
def chk(r):
    assert r == 0, r
N = 32 << 20
d = vp(); h = vp()
chk(hip.hipMalloc(C.byref(d), C.c_size_t(N))); chk(hip.hipHostMalloc(C.byref(h), C.c_size_t(2 * N), 0))
streams = []
def ev():
    e = vp(); chk(hip.hipEventCreate(C.byref(e))); return e
e0, e1 = ev(), ev()
def bw(s, off=0, reps=5):
    best = 1e9
    for _ in range(reps):
        chk(hip.hipEventRecord(e0, s))
        chk(hip.hipMemcpyAsync(vp(h.value + off), d, C.c_size_t(N), 2, s))
        chk(hip.hipEventRecord(e1, s)); chk(hip.hipStreamSynchronize(s))
        ms = C.c_float(); chk(hip.hipEventElapsedTime(C.byref(ms), e0, e1)); best = min(best, ms.value)
    return N / best / 1e6
for i in range(12):
    s = vp(); chk(hip.hipStreamCreateWithFlags(C.byref(s), 1)); streams.append(s)
    print("stream %2d: %.1f GB/s" % (i, bw(s)), flush=True)
print("again:", " ".join("%.1f" % bw(s) for s in streams))
for a, b in ((0, 1), (2, 3), (0, 4)):
    t0 = time.perf_counter()
    for _ in range(5):
        chk(hip.hipMemcpyAsync(h, d, C.c_size_t(N), 2, streams[a])); chk(hip.hipMemcpyAsync(vp(h.value + N), d, C.c_size_t(N), 2, streams[b]))
        chk(hip.hipStreamSynchronize(streams[a])); chk(hip.hipStreamSynchronize(streams[b]))
    print("streams %d + %d together: %.1f GB/s aggregate" % (a, b, 2 * N * 5 / (time.perf_counter() - t0) / 1e9))
