"""Floating-point operations and VALU instructions per pixel visit of pixel_kernel<2, double> and pixel_kernel<2, float>,
counted in the compiled ISA (no GPU needed).

usage: python tools/count_flops.py [--write]
Prints the static counts and the per-visit figures bench.py uses: every basic block LLVM's loop annotations place inside the
pixel loop, component-loop blocks times their trip counts; FMA = 2 flops, multiply / add = 1, packed fp32 instructions count
both halves.  The count is STATIC: a block behind an exec-mask guard (the star spline, the own-geometry region) is counted once
per visit although a trip in which no lane has a covered pixel skips it -- an upper bound of the executed count by that margin
(the hardware's SQ_INSTS_VALU per visit, prologues and partial waves included, sits ABOVE the static figure).  --write stores
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


def blocks_of(lines):
    """The function's basic blocks with LLVM's loop annotations: [(label or None, first line, last line, loop)], where loop is
    the header label of the innermost loop the block belongs to (None outside loops), and parent[header] = the header of the
    enclosing loop.  Membership comes from the assembler comments ("in Loop: Header=BBx_y", "Parent Loop BBx_y", "Loop
    Header"), not from address ranges: block placement moves rarely-skipped blocks (the star spline) behind a loop's first
    back edge, where a range-based count loses them."""
    import re
    starts = [k for k, ln in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", ln) or ln.startswith("; %bb.")]
    blocks, parent = [], {}
    for n, k in enumerate(starts):
        end = (starts[n + 1] if n + 1 < len(starts) else len(lines)) - 1
        m = re.match(r"\.(LBB\d+_\d+):", lines[k])
        label = m.group(1)[1:] if m else None
        note = [lines[k]]
        j = k + 1
        while j <= end and lines[j].strip().startswith(";") and not lines[j].startswith("; %bb."):
            note.append(lines[j]); j += 1
        note = "\n".join(note)
        if "Loop Header" in note:
            loop = label
            ps = re.findall(r"Parent Loop (BB\d+_\d+)", note)
            parent[label] = ps[-1] if ps else None
        else:
            m = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
            loop = m.group(1) if m else None
        blocks.append((label, k, end, loop))
    return blocks, parent


def analyse(txt, symbol, psf_k=2, pixels_per_lane=1):
    """pixels_per_lane: pixels a lane handles per trip of the pixel loop -- 2 in the single-precision kernel (pixel_iter_px2:
    two pixels in the halves of every float2), so its per-trip counts are halved to give per-pixel-visit figures"""
    i = txt.index(symbol)
    i = txt.index(":\n", i)
    lines = txt[i:txt.index("s_endpgm", i)].split("\n")
    blocks, parent = blocks_of(lines)

    def inside(loop, anc):     # is `loop` the loop `anc` or nested in it
        while loop is not None:
            if loop == anc:
                return True
            loop = parent.get(loop)
        return False

    def own_lines(loop):       # the blocks whose INNERMOST loop is `loop`
        return [ln for _, a, b, lp in blocks if lp == loop for ln in lines[a:b + 1]]

    # the component loops: innermost loops that read component records (ds_read_b128): fp64 -- one per profile type, a run of
    # 8 / 6 prototypes unrolled per trip; fp32 -- one per profile type inside a loop over the runs, two components per trip
    inner = [h for h in parent if h not in parent.values()]
    comp = sorted((h for h in inner if sum("ds_read_b128" in l for l in own_lines(h)) >= 4),
                  key=lambda h: next(a for lb, a, _, _ in blocks if lb == h))
    assert len(comp) == 2, comp
    nested = parent[comp[0]] != parent[comp[1]]          # fp32: each sits in its own loop over the runs
    pixel = parent[parent[comp[0]]] if nested else parent[comp[0]]
    assert pixel is not None and inside(comp[1], pixel)
    # multiplicity of every loop inside the pixel loop (other loops -- the neighbour gather -- count once)
    mult, report = {}, []
    for h, per_run in zip(comp, (8, 6)):
        reads = sum("ds_read_b128" in l for l in own_lines(h))
        per_trip = reads // 6 if reads >= 30 else 2
        c = stats(own_lines(h))
        if nested:
            mult[parent[h]] = psf_k
            mult[h] = psf_k * (per_run // per_trip)
        else:
            mult[h] = psf_k * per_run // per_trip
        report.append("component loop (%d trips of %d components): %d fp64 + %d fp32 flops, %d VALU (%d FMA-class) per trip"
                      % (mult[h], per_trip, c["f64"], c["f32"], c["valu"], c["fma"]))
    per_visit = {"f64": 0, "f32": 0, "valu": 0, "fma": 0, "fp_instr": 0}
    loops_in = sorted({lp for _, _, _, lp in blocks if inside(lp, pixel)}, key=str)
    for lp in loops_in:
        c = stats(own_lines(lp))
        for k in per_visit:
            per_visit[k] += mult.get(lp, 1) * c[k]
    body = stats(own_lines(pixel))
    report.append("pixel loop, the blocks outside the component loops: %d fp64 + %d fp32 flops, %d VALU" %
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
