"""-m gpu: the RCCL branch of csrc/group.h with MORE THAN ONE RANK on a one-GPU box.

Real RCCL refuses a communicator that names a device twice, so until a node with several devices runs it, the multi-rank
ordering of the RCCL branch -- N worker threads enqueueing collectives on N communicators, the gather stream, the host barrier
in front of the row exchanges, the abort protocol -- would be product code no test has executed.  tests/fake_rccl.c is a
TEST-ONLY stand-in for librccl, preloaded into a child process: every ncclAllGather blocks on a host rendezvous until all ranks
of the communicator have made their k-th call with the same count, and only then copies the blocks.  That is stricter than RCCL:
a member that skips a collective, members that issue different numbers of collectives, or different counts, deadlock or fail in
the child (under a time limit) instead of on the node.

  1. tests/test_gpu_group.py, unchanged, runs in the child with CELESTE_GROUP_EXCHANGE=rccl: 2 / 3 members on device 0 through
     the RCCL branch, every result still the one-device result bit for bit; the fake's counters prove > 1 rank took part.
  2. tests/cabi_caller.c (compiled C, header only) drives two members the same way.
  3. The failure protocol, member by member: a member whose launch fails (takes part, error returned, group intact), a member
     that fails in front of the host barrier (everybody leaves, group intact), a member that skips the collective in sweep /
     elbo / maximize! / joint inference (CELESTE_ERR_ABORTED on the call within the time limit, communicators aborted, every
     later call CELESTE_ERR_ABORTED, destroy returns).
The product never sees the fake outside these child processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_SRC = os.path.join(ROOT, "tests", "fake_rccl.c")
FAKE_LIB = os.path.join(ROOT, "tests", "libfake_rccl.so")
IN_CHILD = os.environ.get("CELESTE_GROUP_EXCHANGE") == "rccl" and "CELESTE_FAKE_RCCL" in os.environ

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500, method="thread")]


def build_fake():
    if not os.path.exists(FAKE_LIB) or os.path.getmtime(FAKE_LIB) < os.path.getmtime(FAKE_SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", FAKE_SRC, "-o", FAKE_LIB,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return FAKE_LIB


def child_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    fake = build_fake()
    env.update({"LD_PRELOAD": fake, "CELESTE_FAKE_RCCL": fake, "CELESTE_GROUP_EXCHANGE": "rccl"})
    env.update(extra)
    return env


def _pytest_child(args, timeout):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-s"] + args, cwd=ROOT,
                       env=child_env(), capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-3000:] + "\n--- stderr ---\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    return r.stdout


@pytest.mark.skipif(IN_CHILD, reason="this is the child")
def test_the_group_tests_pass_through_the_rccl_branch_with_two_and_three_ranks():
    out = _pytest_child(["tests/test_gpu_group.py"], 1400)
    assert "fake_rccl:" in out and " passed" in out, out[-2000:]
    line = [ln for ln in out.splitlines() if "fake_rccl:" in ln][-1]
    print(line[line.index("fake_rccl:"):])
    print(out.strip().splitlines()[-1])


@pytest.mark.skipif(IN_CHILD, reason="this is the child")
def test_the_failure_protocol_through_the_rccl_branch():
    out = _pytest_child(["tests/test_gpu_group_rccl_branch.py", "-k", "child"], 900)
    assert " passed" in out and "fake_rccl:" in out, out[-2000:]
    line = [ln for ln in out.splitlines() if "fake_rccl:" in ln][-1]
    print(line[line.index("fake_rccl:"):])
    print(out.strip().splitlines()[-1])


@pytest.mark.skipif(IN_CHILD, reason="this is the child")
def test_c_caller_drives_two_members_through_the_rccl_branch(tmp_path):
    """tests/cabi_caller.c: two members on device 0, RCCL branch (fake preloaded): the one-device numbers, ncclCommCount = 2"""
    from celeste_jl_amd import cabi
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_cabi_caller as tc
    exe = str(tmp_path / "cabi_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), tc.SRC, "-o", exe, "-L", tc.CSRC,
           "-lceleste_mi355x", "-Wl,-rpath," + tc.CSRC, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    for members in (2, 3):
        r = subprocess.run([exe, os.path.join(tc.RAW, "sample_two_body"), "0", str(members)], capture_output=True, text=True, timeout=600,
                           env=child_env())
        assert r.returncode == 0, r.stderr[-2000:]
        got = tc._parse(r.stdout)
        assert got["group"] == [members, cabi.EXCHANGE_RCCL, members] and got["group_equal"] == [1, 1, 1, 1], got


# ---- the child's tests: member failures --------------------------------------------------------------------------------------
child = pytest.mark.skipif(not IN_CHILD, reason="runs in the child process (fake RCCL preloaded, RCCL branch forced)")


@pytest.fixture(scope="module")
def crowded():
    import celeste_jl_amd as cel
    from celeste_jl_amd import synthetic
    f = synthetic.make_field(220, 240, 40, seed=23, margin=30)
    return f, cel.FieldContext(f.images, f.patches, f.neighbors)


def _group(f, devices, fault=None):
    from celeste_jl_amd.group import FieldGroup
    os.environ.pop("CELESTE_GROUP_FAULT", None)
    if fault:
        os.environ["CELESTE_GROUP_FAULT"] = fault      # (read by celeste_group_create)
    try:
        return FieldGroup(f.images, f.patches, f.neighbors, devices=devices)
    finally:
        os.environ.pop("CELESTE_GROUP_FAULT", None)


def _assert_dead(g, f):
    """the group after an abort: flagged, every entry point refuses, destroy returns"""
    from celeste_jl_amd import cabi
    enq, aborted = g.collectives()
    assert aborted
    S = len(f.catalog)
    for call in (lambda: g.eval_batch(f.vp, list(range(S))), lambda: g.plan(f.vp, list(range(S))), lambda: g.maximize_batch(f.vp, [0, 1]),
                 lambda: g.joint_infer(f.vp, [0, 1], [0, 1], [0], 1)):
        with pytest.raises(cabi.CelesteError) as e:
            call()
        assert e.value.status == cabi.ERR_ABORTED
    g.close()


@child
@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["2_members", "3_members"])
def test_child_a_member_whose_launch_fails_still_takes_part(crowded, devices):
    """site `launch`: the member reports an error but enqueues every collective -- the call returns the error (not ABORTED), the
    counts stay equal, and the next call on the same group gives the one-device results"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    f, ctx = crowded
    S = len(f.catalog)
    tg = list(range(S))
    for what in ("eval", "sweep", "maximize", "joint"):
        g = _group(f, devices, "%d,launch,1" % (len(devices) - 1))
        with pytest.raises(cabi.CelesteError) as e:
            if what == "eval":
                g.eval_batch(f.vp, tg)
            elif what == "sweep":
                g.plan(f.vp, tg); g.sweep(); g.sweep(); g.wait()
            elif what == "maximize":
                g.maximize_batch(f.vp, tg, cel.ElboConfig(max_iters=3))
            else:
                from celeste_jl_amd.group import cyclades_schedule
                b_off, c_off, flat = cyclades_schedule(tg, f.neighbors, batch_size=12, rng=np.random.default_rng(3))
                g.joint_infer(f.vp, b_off, c_off, flat, 1, cel.ElboConfig(max_iters=3))
        assert e.value.status == cabi.ERR_HIP, (what, e.value.status)
        enq, aborted = g.collectives()
        assert not aborted and len(set(enq)) == 1, (what, enq)
        ref = ctx.eval_batch(f.vp, tg)
        got = g.eval_batch(f.vp, tg)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
        g.close()


@child
@pytest.mark.timeout(300, method="thread")
def test_child_a_member_that_fails_in_front_of_the_host_barrier_wakes_the_others(crowded):
    """site `barrier` (an allocation failing in front of a row exchange): nobody enters the collective, every member leaves with an
    error, the counts stay equal -- no abort, the group stays usable"""
    import celeste_jl_amd as cel
    from celeste_jl_amd import cabi
    f, ctx = crowded
    S = len(f.catalog)
    tg = list(range(S))
    g = _group(f, [0, 0, 0], "1,barrier,1")
    with pytest.raises(cabi.CelesteError) as e:
        g.maximize_batch(f.vp, tg, cel.ElboConfig(max_iters=3))
    assert e.value.status == cabi.ERR_HIP
    enq, aborted = g.collectives()
    assert not aborted and len(set(enq)) == 1
    cfg = cel.ElboConfig(max_iters=4)
    ref = ctx.maximize_batch(f.vp, tg, cfg)
    got = g.maximize_batch(f.vp, tg, cfg)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    g.close()


def _run_faulty(g, f, what):
    import celeste_jl_amd as cel
    S = len(f.catalog)
    tg = list(range(S))
    if what == "sweep":
        g.plan(f.vp, tg)
        for _ in range(5):
            g.sweep()              # (the third sweep's gather is the one the failing member never enqueues)
        g.wait()
    elif what == "eval":
        g.eval_batch(f.vp, tg)
    elif what == "maximize":
        g.maximize_batch(f.vp, tg, cel.ElboConfig(max_iters=3))
    else:
        from celeste_jl_amd.group import cyclades_schedule
        b_off, c_off, flat = cyclades_schedule(tg, f.neighbors, batch_size=12, rng=np.random.default_rng(3))
        g.joint_infer(f.vp, b_off, c_off, flat, 2, cel.ElboConfig(max_iters=3))


SITES = {"sweep": "sweep,3", "eval": "sweep,1", "maximize": "rows,1", "joint": "rows,2"}


@child
@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["2_members", "3_members"])
@pytest.mark.parametrize("what", ["sweep", "eval", "maximize", "joint"])
def test_child_a_member_that_skips_a_collective_aborts_the_group(crowded, devices, what):
    """sites `sweep` / `rows`: a HIP call fails between the point of no return and the collective, late enough (the injected
    fault waits 150 ms) that the others are inside theirs (with this stand-in: blocked on the host; with RCCL: their streams
    are).  The call must come back with CELESTE_ERR_ABORTED within the time limit, on every path, and the group must be dead but
    destroyable."""
    import time
    from celeste_jl_amd import cabi
    f, ctx = crowded
    g = _group(f, devices, "%d,%s,150" % (len(devices) - 1, SITES[what]))
    t0 = time.time()
    with pytest.raises(cabi.CelesteError) as e:
        _run_faulty(g, f, what)
    assert e.value.status == cabi.ERR_ABORTED, (what, e.value.status)
    assert time.time() - t0 < 60
    _assert_dead(g, f)


@child
@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize("what", ["sweep", "eval", "maximize", "joint"])
def test_child_a_member_that_fails_at_once_ends_either_way_but_consistently(crowded, what):
    """the same faults without the delay: the others may or may not have entered the collective when the failing member
    returns.  Either the group is aborted (CELESTE_ERR_ABORTED, dead), or the others saw the failure in time, enqueued nothing
    either (CELESTE_ERR_HIP, equal counts, group intact and still exact) -- never a hang, never a half-way state."""
    from celeste_jl_amd import cabi
    f, ctx = crowded
    S = len(f.catalog)
    for rep in range(3):
        g = _group(f, [0, 0, 0], "%d,%s" % (rep % 3, SITES[what]))
        with pytest.raises(cabi.CelesteError) as e:
            _run_faulty(g, f, what)
        enq, aborted = g.collectives()
        if e.value.status == cabi.ERR_ABORTED:
            assert aborted
            _assert_dead(g, f)
        else:
            assert e.value.status == cabi.ERR_HIP and not aborted and len(set(enq)) == 1, (e.value.status, enq)
            ref = ctx.eval_batch(f.vp, list(range(S)))
            got = g.eval_batch(f.vp, list(range(S)))
            for a, b in zip(ref, got):
                assert np.array_equal(a, b)
            g.close()


@child
def test_child_zz_the_fake_saw_the_aborts():
    import ctypes as C
    fake = C.CDLL(os.environ["CELESTE_FAKE_RCCL"])
    out = (C.c_uint64 * 8)()
    fake.fake_rccl_stats(out)
    print("fake_rccl: %d all-gathers completed, %d between > 1 rank, %d communicators aborted, %d rendezvous woken by an abort"
          % (out[0], out[1], out[2], out[3]))
    assert out[1] > 0 and out[2] >= 8 and out[4] == 0
