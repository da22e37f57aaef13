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
