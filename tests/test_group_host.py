"""CPU: the host side of the device group (celeste_group_*, celeste.jl_amd/group.py) -- what can be held without a GPU:
the library exports only the C ABI, links RCCL, refuses to build a group without a device, and the Cyclades schedule in the
batch / component layout celeste_group_joint_infer takes flattens to exactly the layers celeste_joint_infer is given."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "celeste.jl_amd", "csrc", "libceleste_mi355x.so")


def test_the_library_exports_the_c_abi_and_nothing_else(lib):
    """csrc/exports.map: kernel handles, device stubs and C++ template instantiations stay local"""
    from celeste_jl_amd import cabi
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert names == sorted(cabi.EXPORTED_SYMBOLS), set(names) ^ set(cabi.EXPORTED_SYMBOLS)


def test_the_library_links_rccl_for_the_catalog_gather(lib):
    out = subprocess.run(["readelf", "-d", LIB], capture_output=True, text=True, check=True).stdout
    assert "librccl.so" in out and "libamdhip64.so" in out
    und = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True, check=True).stdout
    for sym in ("ncclCommInitAll", "ncclAllGather", "ncclCommCount", "ncclCommDestroy"):
        assert sym in und, sym


def test_group_create_refuses_without_a_device_and_checks_its_arguments(lib):
    import torch
    from celeste_jl_amd import cabi, synthetic
    h = C.c_void_p()
    assert lib.celeste_group_create(None, 1, None, C.byref(h)) == cabi.ERR_INVALID_ARG
    f = synthetic.make_sample_dataset("star")
    pb = cabi.Problem(f.images, f.patches, f.neighbors)
    assert lib.celeste_group_create(C.byref(pb.c), 0, None, C.byref(h)) == cabi.ERR_INVALID_ARG
    assert lib.celeste_group_create(C.byref(pb.c), 17, None, C.byref(h)) == cabi.ERR_INVALID_ARG
    for fn in (lib.celeste_group_sweep, lib.celeste_group_sweep_wait):
        assert fn(None) == cabi.ERR_INVALID_ARG
    assert lib.celeste_group_info(None, None) == cabi.ERR_INVALID_ARG
    lib.celeste_group_destroy(None)
    if not torch.cuda.is_available():
        assert lib.celeste_group_create(C.byref(pb.c), 1, None, C.byref(h)) == cabi.ERR_NO_DEVICE and not h.value
        from celeste_jl_amd.group import FieldGroup
        with pytest.raises(cabi.CelesteError) as e:
            FieldGroup(f.images, f.patches, f.neighbors, devices=[0, 0])
        assert e.value.status == cabi.ERR_NO_DEVICE


def test_cyclades_schedule_in_batch_component_layout_flattens_to_the_joint_layers():
    from celeste_jl_amd import synthetic
    from celeste_jl_amd.group import cyclades_schedule, schedule_layers
    from celeste_jl_amd.infer import joint_layers
    f = synthetic.make_field(220, 240, 40, seed=23, margin=30)
    S = len(f.catalog)
    targets = [s for s in range(S) if s % 7 != 3]
    for bs in (5, 12, 400):
        b_off, c_off, flat = cyclades_schedule(targets, f.neighbors, batch_size=bs, rng=np.random.default_rng(3))
        assert b_off[0] == 0 and c_off[0] == 0 and b_off[-1] == len(c_off) - 1 and c_off[-1] == len(flat) == len(targets)
        assert sorted(flat.tolist()) == sorted(targets)
        # components of a batch never conflict (partition.jl:173-236): no neighbour of a source sits in another component
        for b in range(len(b_off) - 1):
            comp_of = {}
            for k in range(b_off[b], b_off[b + 1]):
                for e in range(c_off[k], c_off[k + 1]):
                    comp_of[int(flat[e])] = k
            for s, k in comp_of.items():
                assert all(comp_of.get(n, k) == k for n in f.neighbors[s])
        layers, entries = schedule_layers(b_off, c_off, flat, 3)
        ref = joint_layers(targets, f.neighbors, batch_size=bs, n_iters=3, rng=np.random.default_rng(3))
        assert layers == ref
        assert all([int(flat[e]) for e in idx] == layer for idx, layer in zip(entries, layers))
