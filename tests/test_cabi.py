"""The C-ABI shared library loads and exports every symbol include/celeste_mi355x.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(lib):
    from celeste_jl_amd import cabi
    hdr = open(os.path.join(ROOT, "include", "celeste_mi355x.h")).read()
    declared = set(re.findall(r"\b(celeste_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(cabi.EXPORTED_SYMBOLS), declared ^ set(cabi.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.celeste_version() == cabi.ABI_VERSION
    assert lib.celeste_strerror(0) == b"ok" and b"CPU fallback" in lib.celeste_strerror(cabi.ERR_NO_DEVICE)


def test_struct_layout_matches_the_header():
    """ctypes mirrors of the header structs (sizes on the LP64 ABI)"""
    from celeste_jl_amd import cabi
    assert C.sizeof(cabi.ImageT) == 16 + 3 * 8
    assert C.sizeof(cabi.PatchT) == 16 + 8 + 8 * 8 + 8 + 8
    assert C.sizeof(cabi.PriorT) == 8 * (6 + 16 + 64 + 256 + 2)
    assert C.sizeof(cabi.ProblemT) == 16 + 6 * 8 + 3 * 8
    assert C.sizeof(cabi.WorkStatsT) == 56
    assert C.sizeof(cabi.OptimConfigT) == 2 * 8 + 2 * 4 + 5 * 8 + 2 * 4
    hdr = open(os.path.join(ROOT, "include", "celeste_mi355x.h")).read()
    for name, value in (("GRAD", cabi.FLAG_GRAD), ("HESS", cabi.FLAG_HESS), ("KL", cabi.FLAG_KL), ("FP32", cabi.FLAG_FP32),
                        ("SPLIT", cabi.FLAG_SPLIT), ("PACKED_HESS", cabi.FLAG_PACKED_HESS)):
        assert re.search(r"CELESTE_FLAG_%s = %du" % (name, value), hdr), name
    assert cabi.HP == 44 * 45 // 2 and "#define CELESTE_HP 990" in hdr


def test_unpack_hessian_layout():
    """CELESTE_FLAG_PACKED_HESS: element (i, j), i <= j, at j (j + 1) / 2 + i"""
    from celeste_jl_amd import cabi
    full = np.random.default_rng(0).normal(size=(3, 44, 44)); full = full + full.transpose(0, 2, 1)
    packed = np.zeros((3, cabi.HP))
    for j in range(44):
        for i in range(j + 1):
            packed[:, j * (j + 1) // 2 + i] = full[:, i, j]
    assert np.array_equal(cabi.unpack_hessian(packed), full)


def test_no_cpu_fallback(lib):
    """Without a HIP device the engine must refuse to compute (no silent fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic, cabi
    f = synthetic.make_sample_dataset("star")
    with pytest.raises(cabi.CelesteError) as e:
        cel.FieldContext(f.images, f.patches, f.neighbors)
    assert e.value.status == cabi.ERR_NO_DEVICE
    out = np.zeros(4); psf = np.ascontiguousarray(synthetic.band_psf(0)); g = np.zeros(2)
    dp = cabi.c_double_p
    st = lib.celeste_psf_raster(0, psf.ctypes.data_as(dp), 2, g.ctypes.data_as(dp), 2, g.ctypes.data_as(dp), 2,
                                out.ctypes.data_as(dp))
    assert st == cabi.ERR_NO_DEVICE


def test_invalid_arguments_are_rejected(lib):
    from celeste_jl_amd import cabi
    h = C.c_void_p()
    assert lib.celeste_ctx_create(None, 0, C.byref(h)) == cabi.ERR_INVALID_ARG
    assert lib.celeste_spline_prefilter(None, None) == cabi.ERR_INVALID_ARG
    assert lib.celeste_ctx_enable_timing(None, 1) == cabi.ERR_INVALID_ARG


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under celeste.jl_amd/ may import, link or call it"""
    pkg = os.path.join(ROOT, "celeste.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, fn)


def test_the_fake_rccl_of_the_tests_exports_every_rccl_symbol_the_library_imports(lib):
    """tests/fake_rccl.c (the strict stand-in preloaded under tests/test_gpu_group_rccl_branch.py) must cover exactly what
    libceleste_mi355x.so asks librccl for -- a collective the product starts calling and the stand-in lacks would silently bind to
    the real library inside the child process"""
    import subprocess
    import __graft_entry__ as g
    fake = os.path.join(g.ROOT, "tests", "libfake_rccl.so")
    if not os.path.exists(fake):
        g.build()
    undefined = subprocess.run(["nm", "-D", "--undefined-only", g.LIB], capture_output=True, text=True, check=True).stdout
    wanted = {ln.split()[-1] for ln in undefined.splitlines() if ln.split()[-1].startswith("nccl")}
    defined = subprocess.run(["nm", "-D", "--defined-only", fake], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in defined.splitlines()}
    assert wanted and wanted <= have, wanted - have
    assert {"ncclAllGather", "ncclCommInitAll", "ncclCommAbort", "ncclCommGetAsyncError"} <= wanted
    # and the product itself never mentions the stand-in
    for root, _, files in os.walk(os.path.join(g.ROOT, "celeste.jl_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                assert "fake_rccl" not in open(os.path.join(root, f)).read().replace("tests/fake_rccl.c", ""), f
