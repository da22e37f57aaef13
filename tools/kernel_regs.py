"""Registers / scratch / LDS of every kernel in the compiled library, and VALU / LDS / FP64 instruction counts of the
pixel kernel's loops (static, from the ISA; no GPU needed).  usage: python tools/kernel_regs.py [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "celeste.jl_amd", "csrc")


def asm(extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, os.path.join(CSRC, "celeste_abi.hip")] + list(extra), stderr=subprocess.DEVNULL)
        return open(out).read()


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else "pixel_kernel"
    txt = asm()
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M):
        name = m.group(1)
        if flt not in name:
            continue
        blk = txt[m.start():m.start() + 6000]
        g = lambda k: (re.search(r"\.amdhsa_%s (\d+)" % k, blk) or [None, "?"])[1]
        body = txt[txt.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = [l.strip().split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        valu = sum(l.startswith("v_") for l in lines)
        print("%-60s vgpr %s (accum offset %s) sgpr %s scratch %s lds %s | static: %d instr, %d VALU, %d ds_, %d ds_add_f64, %d v_cndmask, %d dpp"
              % (name[:60], g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"), g("private_segment_fixed_size"),
                 g("group_segment_fixed_size"), len(lines), valu, sum(l.startswith("ds_") for l in lines),
                 sum(l.startswith("ds_add_f64") for l in lines), sum(l.startswith("v_cndmask") for l in lines),
                 sum("dpp" in l for l in body.split("\n"))))


if __name__ == "__main__":
    main()
