"""celeste_ctx_create for configs[4] (30 000 sources, 80 images, 178 636 visits; run through gpurun).  usage: gpu_ctx_create_config5.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from celeste_jl_amd import cabi
t0 = time.perf_counter()
fld = bench.build_multifield((4, 4), 2048, 1489, 30000, 5)
print("field generation %.1f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
problem = cabi.Problem(fld.images, fld.patches, fld.neighbors, psf_K=2, prior=None)
print("marshalling (Python) %.2f s" % (time.perf_counter() - t0), flush=True)
lib = cabi.load_library()
for rep in range(3):
    h = C.c_void_p()
    t2 = time.perf_counter()
    cabi.check(lib.celeste_ctx_create(C.byref(problem.c), 0, C.byref(h)), lib)
    t3 = time.perf_counter()
    lib.celeste_ctx_destroy(h)
    print("celeste_ctx_create %.1f ms, destroy %.1f ms" % ((t3 - t2) * 1e3, (time.perf_counter() - t3) * 1e3), flush=True)
