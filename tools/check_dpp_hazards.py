"""Static check of the compiled device code for the one hazard the compiler cannot see for us: the DPP instructions written
as inline asm (v_fmac_f64_dpp ... row_newbcast, csrc/optim_kernels.h) read their first source through the DPP path, which
needs two wait states behind a VALU write of that register -- the hardware does not interlock, and the compiler's hazard
recogniser does not look inside asm strings.  row_replicate() puts an s_nop behind the swaps that produce the operand; this
script makes sure no other VALU write of a DPP source (a copy the register allocator inserted, say) sits within two wait
states of its use.  It checks EVERY DPP instruction of the listing (the compiler's own included), basic block by basic block.

usage: python tools/check_dpp_hazards.py [listing.s]   (no argument: compiles csrc/celeste_abi.hip to a listing first)
exit code 1 if a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "celeste.jl_amd", "csrc")
DPP_WORDS = ("row_newbcast", "quad_perm", "row_mirror", "row_half_mirror", "row_bcast", "row_shr", "row_shl", "row_ror", "wave_shr", "wave_shl")


def vregs(tok):
    tok = tok.strip().rstrip(",").lstrip("-|").rstrip("|")
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(lines):
    """(number of DPP instructions, list of (line number, text, register, wait states seen))"""
    n, bad, hist = 0, [], []   # hist: (wait states the instruction provides, VGPRs it writes; None = block boundary)
    for no, line in enumerate(lines, 1):
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        if t.endswith(":"):
            hist = [(0, None)]
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        if any(w in t for w in DPP_WORDS):
            n += 1
            src = vregs(parts[2]) if len(parts) > 2 else set()
            ws = 0
            for w, wr in reversed(hist):
                if ws >= 2 or wr is None:
                    break
                if wr & src:
                    bad.append((no, t, sorted(wr & src), ws))
                    break
                ws += w
        if op.startswith("s_nop"):
            hist.append((int(parts[1]) + 1 if len(parts) > 1 else 1, set()))
        elif op.startswith("v_"):
            wr = vregs(parts[1]) if len(parts) > 1 else set()
            if "swap" in op and len(parts) > 2:
                wr |= vregs(parts[2])
            hist.append((1, wr))
        else:
            hist.append((1, set()))
        hist = hist[-8:]
    return n, bad


def listing():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, "celeste_abi.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def main():
    lines = open(sys.argv[1]).read().split("\n") if len(sys.argv) > 1 else listing()
    n, bad = check(lines)
    asm_dpp = sum("v_fmac_f64_dpp" in l for l in lines)
    for no, t, regs, ws in bad:
        print("line %d: %s  <- v%s written %d wait state(s) before" % (no, t, regs, ws))
    print("%d DPP instructions (%d v_fmac_f64_dpp from inline asm), %d hazards" % (n, asm_dpp, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
