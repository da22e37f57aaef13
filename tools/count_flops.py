"""FP64 operations per pixel visit of pixel_kernel<2, double>, counted in the compiled ISA (no GPU needed).

usage: python tools/count_flops.py   -> prints the static counts and the per-visit figure bench.py uses:
(outer pixel loop - one copy of the component loop) + 14 psf_K x (component loop), FMA = 2 flops, mul / add = 1."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "celeste.jl_amd", "csrc")


def stats(lines):
    f = fp = valu = 0
    for ln in lines:
        s = ln.strip().split()
        if not s or s[0].startswith((";", ".")) or s[0].endswith(":"):
            continue
        op = s[0]
        if op.startswith("v_fma"):
            f += 2; fp += 1
        elif op.startswith(("v_mul_f64", "v_add_f64")):
            f += 1; fp += 1
        valu += op.startswith("v_")
    return f, fp, valu


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, "celeste_abi.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    i = txt.index("_Z12pixel_kernelILi2EdLb0EEv")
    i = txt.index(":\n", i)
    lines = txt[i:txt.index("s_endpgm", i)].split("\n")
    # the component loop: the innermost basic-block run that holds the four ds_read_b128 of a component record
    reads = [k for k, ln in enumerate(lines) if "ds_read_b128" in ln]
    lo, hi = reads[0], reads[-1]
    while lo > 0 and not re.match(r"\s*(;\s*%bb|\.LBB)", lines[lo]):
        lo -= 1
    while hi < len(lines) and "s_cbranch" not in lines[hi]:
        hi += 1
    comp = stats(lines[lo:hi + 1])
    labels = {ln.strip()[:-1]: k for k, ln in enumerate(lines) if ln.strip().endswith(":") and ln.strip().startswith(".LBB")}
    outer = None
    for k, ln in enumerate(lines):
        s = ln.strip()
        if s.startswith(("s_cbranch", "s_branch")) and s.split()[-1] in labels and labels[s.split()[-1]] < k:
            if outer is None or k - labels[s.split()[-1]] > outer[1] - outer[0]:
                outer = (labels[s.split()[-1]], k)
    body = stats(lines[outer[0]:outer[1]])
    psf_k = 2
    per_visit = body[0] - comp[0] + 14 * psf_k * comp[0]
    print("component loop: %d flops, %d FP64 instructions, %d VALU" % comp)
    print("pixel loop body (one copy of the component loop inside): %d flops, %d FP64 instructions, %d VALU" % body)
    print("FP64 flops per pixel visit (psf_K = 2): %d" % per_visit)
    return per_visit


if __name__ == "__main__":
    main()
