"""What a context costs to create (run through gpurun): celeste_images_create + celeste_ctx_create_on for the bench field, with the
constant PSF map (4 stamps) and with an SDSSPSFMap stamp per patch (8765 stamps), against one sweep and one joint inference.
usage: gpu_ctx_create_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import celeste_jl_amd as cel

for variable in (False, True):
    fld = bench.build_field(2048, 1489, 2000, 3, variable=variable)
    tg = np.arange(len(fld.catalog), dtype=np.int32)
    for rep in range(3):
        t0 = time.perf_counter()
        ctx = cel.FieldContext(fld.images, fld.patches, fld.neighbors)
        t1 = time.perf_counter()
        ctx.eval_batch(fld.vp, tg, 7)
        t2 = time.perf_counter()
        ctx.eval_batch(fld.vp, tg, 7)
        t3 = time.perf_counter()
        print("variable_psf=%s rep %d: FieldContext %.1f ms (stamps %d) | first host-pointer sweep %.2f ms, second %.2f ms"
              % (variable, rep, (t1 - t0) * 1e3, ctx.problem.c.n_stamps, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
        if rep == 2 and hasattr(ctx, "timings"):
            print("   ", ctx.timings)
        ctx.close()

# the C call alone (the Python marshalling of celeste_problem_t excluded)
import ctypes as C
from celeste_jl_amd import cabi
for variable in (False, True):
    fld = bench.build_field(2048, 1489, 2000, 3, variable=variable)
    t0 = time.perf_counter()
    problem = cabi.Problem(fld.images, fld.patches, fld.neighbors, psf_K=2, prior=None)
    t1 = time.perf_counter()
    lib = cabi.load_library()
    for rep in range(3):
        h = C.c_void_p()
        t2 = time.perf_counter()
        cabi.check(lib.celeste_ctx_create(C.byref(problem.c), 0, C.byref(h)), lib)
        t3 = time.perf_counter()
        lib.celeste_ctx_destroy(h)
        print("variable_psf=%s: marshalling (Python) %.1f ms | celeste_ctx_create %.1f ms" % (variable, (t1 - t0) * 1e3, (t3 - t2) * 1e3), flush=True)
