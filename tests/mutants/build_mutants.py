#!/usr/bin/env python3
"""Builds the mutated libraries of tests/test_mutants.py: the product source with -DCELESTE_MUTANT=k (see
csrc/elbo_kernels.h).  Test infrastructure; the outputs (tests/mutants/*.so) are git-ignored build artefacts that travel
to the GPU box with the snapshot.  usage: python tests/mutants/build_mutants.py [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "celeste.jl_amd", "csrc")
MUTANTS = {1: "iota_by_column", 2: "sky_transposed", 3: "one_stamp_for_all"}


def path(k):
    return os.path.join(HERE, "libceleste_mutant_%d_%s.so" % (k, MUTANTS[k]))


def build(force=False):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge     # the product's own compile and link flags (exports.map, RCCL)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc", ".map"))]
    newest = max(os.path.getmtime(s) for s in srcs)
    procs = []
    for k in MUTANTS:
        if force or not os.path.exists(path(k)) or os.path.getmtime(path(k)) < newest:
            procs.append(subprocess.Popen([ge.HIPCC] + ge.HIP_FLAGS + ["-DCELESTE_MUTANT=%d" % k, "-o", path(k),
                                           os.path.join(CSRC, "celeste_abi.hip")] + ge.LINK_FLAGS, cwd=CSRC))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("mutant build failed")
    return [path(k) for k in MUTANTS]


if __name__ == "__main__":
    print("\n".join(build("--force" in sys.argv)))
