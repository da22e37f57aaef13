"""Opcode histogram of the pixel loop of pixel_kernel<2, float> outside its component loops, by category (no GPU needed).
usage: python tools/isa_hist.py [--blocks]"""
import collections
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import count_flops as cf

CATS = (("packed fp32 math (v_pk_fma / mul / add)", ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")),
        ("scalar fp32 math", ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_max_f32", "v_min_f32")),
        ("transcendental (exp, log, rcp, ldexp)", ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_ldexp_f32", "v_rsq_f32")),
        ("register moves", ("v_mov_b32", "v_mov_b64", "v_accvgpr")),
        ("row folds (permlane swaps)", ("v_permlane16_swap", "v_permlane32_swap")),
        ("selects and compares", ("v_cndmask", "v_cmp", "v_med3", "v_min_i32", "v_max_i32")),
        ("f64 <-> f32 / int conversions", ("v_cvt_",)),
        ("fp64 arithmetic", ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_floor_f64", "v_fmac_f64")),
        ("integer / address arithmetic", ("v_add_u32", "v_sub_u32", "v_lshl", "v_ashr", "v_mad_", "v_mul_lo", "v_mul_u32", "v_addc", "v_and_", "v_or_", "v_lshr", "v_add_co", "v_mul_hi", "v_sub_co", "v_subrev", "v_bfe", "v_xor", "v_not", "v_add3", "v_lshlrev")),
        ("lane reads", ("v_readlane", "v_readfirstlane", "v_writelane")))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                               "celeste_abi.hip"], cwd=cf.CSRC, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    sym = "_Z12pixel_kernelILi2EfLb0EEv"
    i = txt.index(sym); i = txt.index(":\n", i)
    lines = txt[i:txt.index("s_endpgm", i)].split("\n")
    blocks, parent = cf.blocks_of(lines)

    def own_lines(loop):
        return [ln for _, a, b, lp in blocks if lp == loop for ln in lines[a:b + 1]]
    inner = [h for h in parent if h not in parent.values()]
    comp = sorted((h for h in inner if sum("ds_read_b128" in l for l in own_lines(h)) >= 4), key=lambda h: next(a for lb, a, _, _ in blocks if lb == h))
    pixel = parent[parent[comp[0]]]
    c = collections.Counter()
    for ln in own_lines(pixel):
        t = ln.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        c[t[0]] += 1
    valu = {k: v for k, v in c.items() if k.startswith("v_")}
    total = sum(valu.values())
    print("pixel_kernel<2, float>: the pixel loop outside its component loops, per trip (two pixels per lane): %d VALU, %d s_nop, %d LDS, "
          "%d global / flat / scratch loads" % (total, c.get("s_nop", 0), sum(v for k, v in c.items() if k.startswith("ds_")),
                                                sum(v for k, v in c.items() if k.startswith(("global_load", "flat_load", "scratch_load")))))
    print("| category | VALU per trip | share |")
    print("|---|---|---|")
    left = dict(valu)
    for name, pre in CATS:
        n = 0
        for k in list(left):
            if k.startswith(pre):
                n += left.pop(k)
        print("| %s | %d | %.0f %% |" % (name, n, 100.0 * n / total))
    print("| other (%s) | %d | %.0f %% |" % (", ".join(sorted(left)) or "-", sum(left.values()), 100.0 * sum(left.values()) / total))
    for h in comp:
        cc = cf.stats(own_lines(h))
        print("component loop %s: %d VALU per trip of two components (x two pixels per lane)" % (h, cc["valu"]))


if __name__ == "__main__":
    main()
