"""Floating-point operations and VALU instructions per pixel visit of pixel_kernel<2, double> and pixel_kernel<2, float>,
counted in the compiled ISA (no GPU needed).

usage: python tools/count_flops.py [--write]
Prints the static counts and the per-visit figures bench.py uses: (pixel loop body - one copy of each component loop) +
trips x (component loop); FMA = 2 flops, multiply / add = 1, packed fp32 instructions count both halves.  --write stores
them in profiles/hbm_traffic.json (flops_per_pixel_visit, flops_per_pixel_visit_f32, instruction_mix, instruction_mix_f32)."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "celeste.jl_amd", "csrc")


def classify(op):
    """(fp64 flops, fp32 flops, is VALU, is FMA-class) of one instruction"""
    valu = op.startswith("v_")
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return 2, 0, valu, True
    if op.startswith(("v_mul_f64", "v_add_f64")):
        return 1, 0, valu, False
    if op.startswith("v_pk_fma_f32"):
        return 0, 4, valu, True
    if op.startswith(("v_pk_mul_f32", "v_pk_add_f32")):
        return 0, 2, valu, False
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mac_f32")):
        return 0, 2, valu, True
    if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_exp_f32")):
        return 0, 1, valu, False
    return 0, 0, valu, False


def stats(lines):
    s = {"f64": 0, "f32": 0, "valu": 0, "fma": 0, "fp_instr": 0}
    for ln in lines:
        t = ln.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        a, b, v, fma = classify(t[0])
        s["f64"] += a; s["f32"] += b; s["valu"] += v; s["fma"] += fma; s["fp_instr"] += (a + b) > 0
    return s


def loops_of(lines):
    labels = {ln.strip().split(":")[0]: k for k, ln in enumerate(lines) if ln.strip().startswith(".LBB") and ":" in ln}
    out = []   # (first line, last line) of every backward branch
    for k, ln in enumerate(lines):
        t = ln.strip().split()
        if t and t[0].startswith(("s_cbranch", "s_branch")) and t[-1] in labels and labels[t[-1]] < k:
            out.append((labels[t[-1]], k))
    return out


def analyse(txt, symbol, psf_k=2, pixels_per_lane=1):
    """pixels_per_lane: pixels a lane handles per trip of the pixel loop -- 2 in the single-precision kernel (pixel_iter_px2:
    two pixels in the halves of every float2), so its per-trip counts are halved to give per-pixel-visit figures"""
    i = txt.index(symbol)
    i = txt.index(":\n", i)
    lines = txt[i:txt.index("s_endpgm", i)].split("\n")
    loops = loops_of(lines)
    # the component loops: innermost loops that read component records (ds_read_b128): fp64 -- one per profile type
    # (8 psf_K de Vaucouleurs, 6 psf_K exponential components, two per trip); fp32 -- one loop, two components per trip
    comp = sorted((a, b) for a, b in loops if any("ds_read_b128" in l for l in lines[a:b])
                  and not any(a < a2 and b2 < b for a2, b2 in loops))
    outer = min((l for l in loops if all(l[0] < a and b < l[1] for a, b in comp)), key=lambda l: l[1] - l[0])
    body = stats(lines[outer[0]:outer[1]])
    n_comp = [8 * psf_k, 6 * psf_k] if len(comp) == 2 else [14 * psf_k]
    per_visit = dict(body)
    report = []
    for (a, b), n in zip(comp, n_comp):
        c = stats(lines[a:b + 1])
        # components per trip: the fp64 loop requests every record with six explicit ds_read_b128 (a run of 8 / 6
        # prototypes is unrolled: 8 / 6 per trip); the packed fp32 loop handles one pair per trip
        reads = sum("ds_read_b128" in l for l in lines[a:b + 1])
        per_trip = reads // 6 if reads >= 30 else 2
        trips = n // per_trip
        report.append("component loop (%d trips of %d components): %d fp64 + %d fp32 flops, %d VALU (%d FMA-class) per trip"
                      % (trips, per_trip, c["f64"], c["f32"], c["valu"], c["fma"]))
        for k in per_visit:
            per_visit[k] += (trips - 1) * c[k]
    report.append("pixel loop body (one copy of each component loop inside): %d fp64 + %d fp32 flops, %d VALU" %
                  (body["f64"], body["f32"], body["valu"]))
    if pixels_per_lane > 1:
        report.append("a trip of the pixel loop handles %d pixels per lane: per-visit figures = per-trip / %d" % (pixels_per_lane, pixels_per_lane))
        per_visit = {k: v / pixels_per_lane for k, v in per_visit.items()}
    rnd = (lambda x: int(round(x))) if pixels_per_lane == 1 else (lambda x: round(x, 1))
    mix = {"valu_per_pixel_visit": rnd(per_visit["valu"]), "fp_instructions": rnd(per_visit["fp_instr"]), "fma_class": rnd(per_visit["fma"]),
           "fma_share_of_valu": round(per_visit["fma"] / per_visit["valu"], 4),
           "fp64_flops": rnd(per_visit["f64"]), "fp32_flops": rnd(per_visit["f32"])}
    return rnd(per_visit["f64"] + per_visit["f32"]), mix, report


PIXELS_PER_LANE = {"": 1, "_f32": 2}


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, "celeste_abi.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    res = {}
    for key, sym, name in (("", "_Z12pixel_kernelILi2EdLb0EEv", "pixel_kernel<2, double>"),
                           ("_f32", "_Z12pixel_kernelILi2EfLb0EEv", "pixel_kernel<2, float>")):
        flops, mix, report = analyse(txt, sym, pixels_per_lane=PIXELS_PER_LANE[key])
        print(name)
        for r in report:
            print("  " + r)
        print("  per pixel visit (psf_K = 2): %g flops (%g fp64 + %g fp32), %g VALU instructions, %g of them FMA-class (%.0f %%)"
              % (flops, mix["fp64_flops"], mix["fp32_flops"], mix["valu_per_pixel_visit"], mix["fma_class"],
                 100 * mix["fma_share_of_valu"]))
        res["flops_per_pixel_visit" + key] = flops
        res["instruction_mix" + key] = mix
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        d = json.load(open(path)) if os.path.exists(path) else {}
        d.update(res)
        json.dump(d, open(path, "w"), indent=1)
        print("wrote", path)
    print("FP64 flops per pixel visit (psf_K = 2): %d" % res["flops_per_pixel_visit"])
    return res["flops_per_pixel_visit"]


if __name__ == "__main__":
    main()
