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
    labels = {ln.strip().split(":")[0]: k for k, ln in enumerate(lines) if ln.strip().startswith(".LBB") and ":" in ln}
    loops = []   # (first line, last line) of every backward branch
    for k, ln in enumerate(lines):
        t = ln.strip().split()
        if t and t[0].startswith(("s_cbranch", "s_branch")) and t[-1] in labels and labels[t[-1]] < k:
            loops.append((labels[t[-1]], k))
    # the component loops: innermost loops that read a component record (ds_read_b128) -- one per profile type
    # (de Vaucouleurs: 8 psf_K components, exponential: 6 psf_K), same body
    comp = [(a, b) for a, b in loops if any("ds_read_b128" in l for l in lines[a:b])
            and not any(a < a2 and b2 < b for a2, b2 in loops)]
    # the pixel loop: the innermost loop around the component loops (the loop over a group's chunks encloses it)
    outer = min((l for l in loops if all(l[0] < a and b < l[1] for a, b in comp)), key=lambda l: l[1] - l[0])
    body = stats(lines[outer[0]:outer[1]])
    psf_k = 2
    n_comp = [8 * psf_k, 6 * psf_k] if len(comp) == 2 else [14 * psf_k]
    per_visit = body[0]
    for (a, b), n in zip(sorted(comp), n_comp):
        c = stats(lines[a:b + 1])
        unroll = max(1, sum("ds_read_b64" in l for l in lines[a:b + 1]))   # one exp-table read per component
        trips = n // unroll
        print("component loop (%d trips of %d component%s): %d flops, %d FP64 instructions, %d VALU per trip"
              % ((trips, unroll, "s" if unroll > 1 else "") + c))
        per_visit += (trips - 1) * c[0]
    print("pixel loop body (one copy of each component loop inside): %d flops, %d FP64 instructions, %d VALU" % body)
    print("FP64 flops per pixel visit (psf_K = 2): %d" % per_visit)
    return per_visit


if __name__ == "__main__":
    main()
