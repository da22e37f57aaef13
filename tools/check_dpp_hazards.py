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


def _parse(lines):
    """instructions of the listing: (line number, text, opcode, VGPRs written, wait states it provides, DPP source or None,
    VALU write of EXEC), labels -> instruction index, branch sites per label"""
    ins, label_at, branches = [], {}, {}
    for no, line in enumerate(lines, 1):
        t = line.split(";")[0].strip()
        if t.endswith(":") and " " not in t:
            label_at[t[:-1]] = len(ins)
            continue
        if not t or t.startswith("."):
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        wr, ws, src, wexec = set(), 1, None, False
        if any(w in t for w in DPP_WORDS):
            src = vregs(parts[2]) if len(parts) > 2 else set()
        if op.startswith("s_nop"):
            ws = int(parts[1]) + 1 if len(parts) > 1 else 1
        elif op.startswith("v_"):
            wr = vregs(parts[1]) if len(parts) > 1 else set()
            if "swap" in op and len(parts) > 2:
                wr |= vregs(parts[2])
            # a VALU write of EXEC: v_cmpx*, or any VALU instruction whose destination is exec
            wexec = op.startswith("v_cmpx") or (len(parts) > 1 and parts[1].startswith("exec"))
        if op.startswith(("s_cbranch", "s_branch")) and len(parts) > 1:
            branches.setdefault(parts[-1], []).append(len(ins))
        ins.append((no, t, op, wr, ws, src, wexec))
    return ins, label_at, branches


def check(lines):
    """(number of DPP instructions, list of (line number, text, register, wait states seen)).  A DPP instruction needs two
    wait states behind a VALU write of its source and five behind a VALU write of EXEC.  Basic-block boundaries are NOT taken
    as safe: the walk back from a DPP instruction continues through the fall-through predecessor and through every branch
    that targets the block (the static_for bodies of the solver sit inside `if` regions, so a write at the end of a
    predecessor block can sit right in front of a DPP that opens the next one)."""
    ins, label_at, branches = _parse(lines)
    starts = {}
    for lab, idx in label_at.items():
        starts.setdefault(idx, []).append(lab)
    bad, n = [], 0

    def walk(i, ws, src, need, depth, seen):
        """instructions before index i, `ws` wait states already between them and the DPP; True if a hazard is found"""
        while i >= 0 and ws < need:
            if i + 1 in starts and depth < 4:           # instruction i + 1 opens a block: the branches that target it
                for lab in starts[i + 1]:
                    for b in branches.get(lab, ()):
                        if (b, ws) not in seen:
                            seen.add((b, ws))
                            hit = walk(b, ws, src, need, depth + 1, seen)
                            if hit:
                                return hit
                if ins[i][2].startswith(("s_branch", "s_endpgm", "s_setpc")):
                    return False                         # no fall-through into the block
            no, t, op, wr, w, _, wexec = ins[i]
            if ws < 2 and wr & src:
                return (no, t, sorted(wr & src), ws)
            if wexec and ws < 5:
                return (no, t, "exec", ws)
            ws += w
            i -= 1
        return False

    for i, (no, t, op, wr, w, src, wexec) in enumerate(ins):
        if src is None:
            continue
        n += 1
        hit = walk(i - 1, 0, src, 5, 0, set())
        if hit:
            bad.append((no, t, hit[2], hit[3]))
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
