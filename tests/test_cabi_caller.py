"""A caller COMPILED against include/celeste_mi355x.h (tests/cabi_caller.c, gcc -std=c99) -- what the reference's `ccall`
would bind (ElboMaximize.jl:166; ParallelRun.jl:302-397): the compiler lays out the structs, not a ctypes mirror.

CPU: it compiles and links (the _Static_assert offsets hold), the ctypes mirrors agree with those offsets field by field,
and without a device it stops at celeste_ctx_create with CELESTE_ERR_NO_DEVICE.  -m gpu: it runs the committed raw
fixtures through celeste_elbo_eval / _batch / celeste_maximize_batch / celeste_joint_infer and its printed numbers are the
committed goldens' (1e-8) and, bit for bit, what the ctypes binding gets from the same calls."""
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util as gu
from parity_util import RTOL, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi_caller.c")
CSRC = os.path.join(ROOT, "celeste.jl_amd", "csrc")
RAW = os.path.join(ROOT, "tests", "golden", "raw")


@pytest.fixture(scope="module")
def exe(lib, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cabi") / "cabi_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", out, "-L", CSRC,
           "-lceleste_mi355x", "-Wl,-rpath," + CSRC, "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_caller_compiles_against_the_header_and_refuses_without_a_device(exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (the -m gpu tests run it)")
    r = subprocess.run([exe, os.path.join(RAW, "sample_star")], capture_output=True, text=True)
    assert r.returncode == 5 and "no HIP device" in r.stderr and r.stdout == ""


def test_ctypes_mirrors_have_the_offsets_the_compiler_checked():
    """every `_Static_assert(offsetof(T, f) == N)` / `sizeof(T) == N` of cabi_caller.c (which gcc verified against the header)
    holds for the ctypes Structure of the same name"""
    import ctypes as C
    from celeste_jl_amd import cabi
    mirror = {"celeste_image_t": cabi.ImageT, "celeste_patch_t": cabi.PatchT, "celeste_prior_t": cabi.PriorT,
              "celeste_problem_t": cabi.ProblemT, "celeste_work_stats_t": cabi.WorkStatsT,
              "celeste_optim_config_t": cabi.OptimConfigT, "celeste_group_info_t": cabi.GroupInfoT}
    txt = open(SRC).read()
    offs = re.findall(r"_Static_assert\(offsetof\((\w+), (\w+)\) == (\d+)", txt)
    sizes = re.findall(r"_Static_assert\(sizeof\((\w+)\) == (\d+)", txt)
    assert len(offs) >= 50 and {t for t, _ in sizes} == set(mirror)
    for t, n in sizes:
        assert C.sizeof(mirror[t]) == int(n), t
    for t, f, n in offs:
        assert getattr(mirror[t], f).offset == int(n), (t, f)
    # every field of every struct of the header is covered (reserved padding fields aside)
    hdr = open(os.path.join(ROOT, "include", "celeste_mi355x.h")).read()
    for t, cls in mirror.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (t, t), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = [re.sub(r"\[.*", "", d.strip().split()[-1].lstrip("*")) for d in body.split(";") if d.strip()]
        assert fields == [f for f, _ in cls._fields_], (t, fields)
        checked = {f for tt, f, _ in offs if tt == t}
        assert set(fields) - {"reserved"} == checked, (t, set(fields) ^ checked)


def _parse(out):
    rec = {"elbo": {}, "d": {}, "h": {}, "maximize": {}, "joint": {}, "joint_vp": {}}
    for ln in out.splitlines():
        tok = ln.split()
        if tok[0] == "batch_equal":
            rec["batch_equal"] = int(tok[1])
        elif tok[0] in ("group", "group_equal"):
            rec[tok[0]] = [int(x) for x in tok[1:]]
        elif tok[0] == "elbo":
            rec["elbo"][int(tok[1])] = (float(tok[2]), int(tok[3]), int(tok[4]))
        elif tok[0] in ("d", "h", "joint_vp"):
            rec[tok[0]][int(tok[1])] = np.array([float(x) for x in tok[2:]])
        elif tok[0] == "maximize":
            rec["maximize"][int(tok[1])] = (int(tok[2]), int(tok[3]), int(tok[4]), float(tok[5]), np.array([float(x) for x in tok[6:]]))
        elif tok[0] == "joint":
            rec["joint"][int(tok[1])] = (int(tok[2]), int(tok[3]), int(tok[4]), int(tok[5]), float(tok[6]))
    return rec


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sample_star", "sample_galaxy", "sample_two_body"])
def test_c_caller_reproduces_the_golden_and_the_ctypes_binding(exe, name):
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    r = subprocess.run([exe, os.path.join(RAW, name)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = _parse(r.stdout)
    z = np.load(gu.path(name))
    S = len(z["pos"])
    assert got["batch_equal"] == 1
    for s in range(S):
        v, na, ni = got["elbo"][s]
        assert abs(v - z["v7"][s]) <= RTOL * abs(z["v7"][s]) and [na, ni] == list(z["cnt"][s])
        assert rel_err(got["d"][s], z["d7"][s]) <= RTOL
        h = got["h"][s].reshape(44, 44)
        assert np.array_equal(h, h.T) and rel_err(h, z["h7"][s]) <= RTOL
    # the same calls through the ctypes binding, on patches the package's host logic builds: bit for bit (%.17g round-trips)
    f = gu.arrays_to_field(z)
    ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
    v, d, h, cnt, st = ctx.eval_batch(f.vp, list(range(S)), 7)
    for s in range(S):
        assert got["elbo"][s][0] == v[s] and np.array_equal(got["d"][s], d[s]) and np.array_equal(got["h"][s].reshape(44, 44), h[s])
    cfg = cel.ElboConfig(max_iters=4)
    vp, its, evals, el, ost = ctx.maximize_batch(f.vp, list(range(S)), cfg)
    for s in range(S):
        it, ev, stt, e, row = got["maximize"][s]
        assert (it, ev, stt) == (its[s], evals[s], ost[s]) and e == el[s] and np.array_equal(row, vp[s])
    layers = [[s] for s in range(S)] * 2
    jvp, jit, jev, jel, jst = ctx.joint_infer(f.vp, layers, cfg)
    for k in range(2 * S):
        src, it, ev, stt, e = got["joint"][k]
        assert (src, it, ev, stt) == (k % S, jit[k], jev[k], jst[k]) and e == jel[k]
    for s in range(S):
        assert np.array_equal(got["joint_vp"][s], jvp[s])
    # ... and through a device group of one member (RCCL: ncclCommInitAll over one device, one rank), the one-device numbers
    assert got["group"] == [1, cabi.EXCHANGE_RCCL, 1] and got["group_equal"] == [1, 1, 1, 1]   # (one member: one segment, one exchange)


@pytest.mark.gpu
def test_c_caller_drives_two_group_members_on_one_device(exe):
    """two members sharing the device (worker threads, shards, device-to-device exchange): still the one-device numbers"""
    from celeste_jl_amd import cabi
    r = subprocess.run([exe, os.path.join(RAW, "sample_two_body"), "0", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = _parse(r.stdout)
    assert got["group"] == [2, cabi.EXCHANGE_PEER_COPY, 0] and got["group_equal"] == [1, 1, 1, 1]   # (single-component batches all land on member 0: nobody reads a foreign row)


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 HIP devices (RCCL between real ranks, from a C caller)")
def test_c_caller_drives_a_group_over_real_devices(exe):
    """arms itself on a node with several devices: the compiled C caller, one member per device, RCCL between them"""
    from celeste_jl_amd import cabi
    n = min(_n_devices(), 8)
    r = subprocess.run([exe, os.path.join(RAW, "sample_two_body"), "0", str(n), "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = _parse(r.stdout)
    assert got["group"] == [n, cabi.EXCHANGE_RCCL, n] and got["group_equal"] == [1, 1, 1, 1]
