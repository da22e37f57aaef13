import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The HIP C-ABI library (host functions are usable without a GPU)."""
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "celeste.jl_amd", "csrc", "libceleste_mi355x.so")):
        g.build()
    from celeste_jl_amd import cabi
    return cabi.load_library()


@pytest.fixture(scope="session")
def oracle(lib):
    from oracle import oracle as o
    o.lib()
    return o


# ---- CELESTE_CANARY=1: who writes into freed host memory? -------------------------------------------------------------------
# After every test (garbage collected first) a set of small numpy arrays of every size class is filled with a pattern; the
# arrays take the heap blocks the test just freed.  They are checked after the NEXT test: a changed byte is a write through a
# stale pointer (or a late DMA) by something that test -- or the tail of the one before -- left behind.  Reports go to
# gpurun_out/flaky/canary_<pid>.txt and fail the test that was running.
_CANARY = {"sets": [], "log": []}


@pytest.fixture(autouse=True)
def _heap_canaries(request):
    if not os.environ.get("CELESTE_CANARY"):
        yield
        return
    import gc
    import numpy as np
    yield
    gc.collect()
    problems = []
    for made_after, arrays in _CANARY["sets"]:
        for a in arrays:
            bad = np.flatnonzero(a != 0xA5)
            if bad.size:
                words = a.view(np.uint32) if a.size % 4 == 0 else None
                problems.append("canary of %d bytes (made after %s) changed at byte offsets %s: bytes %s%s" % (
                    a.size, made_after, bad[:16].tolist(), [hex(int(x)) for x in a[bad[:16]]],
                    "" if words is None else "; words %s" % [(int(i), hex(int(words[i]))) for i in np.unique(bad // 4)[:8]]))
                a[:] = 0xA5
    _CANARY["sets"] = _CANARY["sets"][-2:]
    arrays = []
    for size in range(32, 4096 + 1, 16):
        for _ in range(6 if size <= 1536 else 2):
            a = np.empty(size, dtype=np.uint8)
            a[:] = 0xA5
            arrays.append(a)
    _CANARY["sets"].append((request.node.name, arrays))
    if problems:
        msg = "heap canaries changed during %s:\n  " % request.node.name + "\n  ".join(problems)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out", "flaky"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "flaky", "canary_%d.txt" % os.getpid()), "a") as fh:
                fh.write(msg + "\n")
        except OSError:
            pass
        pytest.fail(msg)


# ---- tools/heapwho preloaded with HEAPWHO_QUARANTINE=<library>: after every test, look at the blocks that library has freed ----
@pytest.fixture(autouse=True)
def _heapwho_quarantine(request):
    yield
    if not os.environ.get("HEAPWHO_QUARANTINE"):
        return
    import ctypes as C
    try:
        hw = C.CDLL(None)
        hw.heapwho_scan.argtypes = [C.c_char_p]
    except (AttributeError, OSError):
        return
    out = os.path.join(ROOT, "gpurun_out", "flaky")
    os.makedirs(out, exist_ok=True)
    n = hw.heapwho_scan(os.path.join(out, "quarantine_%d.txt" % os.getpid()).encode())
    if n:
        with open(os.path.join(out, "quarantine_%d.txt" % os.getpid()), "a") as fh:
            fh.write("    (after %s)\n" % request.node.name)
