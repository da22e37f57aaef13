"""The inline-asm DPP instructions of the trust-region solver (csrc/optim_kernels.h: v_fmac_f64_dpp row_newbcast) are outside
the compiler's hazard recogniser: tools/check_dpp_hazards.py looks at the compiled listing instead (CPU only, ~30 s)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_dpp_hazards as chk   # noqa: E402

_LISTING = []


def compiled_listing():
    """the device listing of the tree, compiled once per test session"""
    if not _LISTING:
        _LISTING.append(chk.listing())
    return _LISTING[0]


def test_the_checker_sees_a_hazard_when_there_is_one():
    lines = ["f:", "\tv_mov_b64_e32 v[4:5], v[8:9]", "\tv_fmac_f64_dpp v[0:1], v[4:5], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    n, bad = chk.check(lines)
    assert n == 1 and len(bad) == 1 and bad[0][2] == [4, 5]
    ok = [lines[0], lines[1], "\ts_nop 1", lines[2]]
    assert chk.check(ok) == (1, [])
    # a swap writes both of its operands
    sw = ["f:", "\tv_permlane32_swap_b32_e32 v7, v4", "\tv_add_f64 v[10:11], v[12:13], v[14:15]",
          "\tv_fmac_f64_dpp v[0:1], v[4:5], v[2:3] row_newbcast:0 row_mask:0xf bank_mask:0xf"]
    assert len(chk.check(sw)[1]) == 1
    # a block boundary is not safe: the write at the end of the predecessor block, and the one in front of a branch into it
    dpp = "\tv_fmac_f64_dpp v[0:1], v[4:5], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf"
    fall = ["f:", "\tv_mov_b64_e32 v[4:5], v[8:9]", ".LBB0_1:", dpp]
    assert len(chk.check(fall)[1]) == 1
    jump = ["f:", "\tv_mov_b64_e32 v[4:5], v[8:9]", "\ts_cbranch_vccz .LBB0_2", "\ts_nop 3", "\ts_branch .LBB0_3", ".LBB0_2:", dpp, ".LBB0_3:"]
    assert len(chk.check(jump)[1]) == 1              # (the branch itself is one wait state: one short)
    safe = ["f:", "\tv_mov_b64_e32 v[4:5], v[8:9]", "\ts_nop 0", "\ts_cbranch_vccz .LBB0_2", "\ts_branch .LBB0_3", ".LBB0_2:", dpp, ".LBB0_3:"]
    assert chk.check(safe) == (1, [])
    # a VALU write of EXEC needs five wait states before a DPP instruction
    ex = ["f:", "\tv_cmpx_lt_f64_e32 v[8:9], v[10:11]", "\ts_nop 2", dpp]
    assert len(chk.check(ex)[1]) == 1 and chk.check(ex)[1][0][2] == "exec"
    assert chk.check(["f:", "\tv_cmpx_lt_f64_e32 v[8:9], v[10:11]", "\ts_nop 4", dpp]) == (1, [])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_dpp_read_within_two_wait_states_of_a_write_in_the_compiled_library():
    lines = compiled_listing()
    n, bad = chk.check(lines)
    n_asm = sum("v_fmac_f64_dpp" in l for l in lines)
    assert n_asm > 2000, "the reduction's broadcast FMAs are in the listing (%d found)" % n_asm
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("src", ["tred_probe.hip", "fp64_rate_probe.hip"])
def test_the_probes_compile_against_the_tree(tmp_path, src):
    """tools/*.hip include the product's kernel headers (tred_probe) or stand alone (fp64_rate_probe): they must keep building"""
    import subprocess
    out = tmp_path / "probe"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", str(out), os.path.join(ROOT, "tools", src)],
                          stderr=subprocess.DEVNULL)
    assert out.exists()


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_register_and_scratch_budgets_of_the_hot_kernels():
    """what DESIGN.md states about the compiled kernels, read from the listing's kernel descriptors: the pixel kernels spill
    nothing and fit two (fp64) / three (fp32) wavefronts per SIMD; the persistent optimiser launch keeps its arguments out of
    scratch (592 B per lane before round 4's change, 128 after); the lock-step step kernel 96 B"""
    import re
    txt = "\n".join(compiled_listing())
    desc = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        g = lambda k: int(re.search(r"\.amdhsa_%s (\S+)" % k, m.group(2)).group(1))
        desc[m.group(1)] = (g("next_free_vgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
    def one(prefix):
        hits = [v for k, v in desc.items() if k.startswith(prefix)]
        assert len(hits) == 1, (prefix, len(hits))
        return hits[0]
    v, s, l = one("_Z12pixel_kernelILi2EdLb0EE")
    assert s == 0 and v <= 256 and l <= 16 * 1024
    # single precision, two pixels per lane: three waves per SIMD (168 VGPRs) with a few spills OUTSIDE the pixel loop
    # (measured faster than two waves and none: 5.37 against 5.72 ms on config 5)
    v, s, l = one("_Z12pixel_kernelILi2EfLb0EE")
    assert s <= 96 and v <= 168
    for k in ("_Z18optim_fused_kernelILb0EE", "_Z18optim_fused_kernelILb1EE"):
        v, s, l = one(k)
        assert s <= 128 and v <= 256 and 2 * l <= 160 * 1024, (k, v, s, l)
    v, s, l = one("_Z17optim_step_kernel")
    assert s <= 96 and v <= 256 and 8 * l <= 160 * 1024
    v, s, l = one("_Z17eval_fused_kernel")
    assert s == 0
    # the lift: 8 workgroups of 256 threads per CU (64 VGPRs); its spill is stored by every thread, i.e. it is HBM traffic
    # (88 B per thread = 45 MB per 2000-target sweep before round 5 moved the radius prior's logarithms out of the image loop)
    # Two instantiations since round 6: 8 waves per SIMD (one round for a 2000-target batch) and 7 (72 VGPRs, 8 B of spill) for the
    # batches that take many rounds anyway -- 36 B x 256 threads x 30 000 targets were 276 MB of config 5's lift writes
    v, s, l = one("_Z11lift_kernelILi8EE")
    assert s <= 48 and v <= 64 and 8 * l <= 160 * 1024
    v, s, l = one("_Z11lift_kernelILi7EE")
    assert s <= 8 and v <= 72 and 7 * l <= 160 * 1024


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_the_flop_counts_bench_reports_are_those_of_this_tree():
    """bench.py's roofline numerator (flops per pixel visit, instruction mix) is read from profiles/hbm_traffic.json; a kernel
    edit without `python tools/count_flops.py --write` would leave it stale -- the fresh count of the compiled listing must
    equal the committed one, fp64 and fp32"""
    import json
    import count_flops as cf
    txt = "\n".join(compiled_listing())
    committed = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for key, sym in (("", "_Z12pixel_kernelILi2EdLb0EEv"), ("_f32", "_Z12pixel_kernelILi2EfLb0EEv")):
        flops, mix, _ = cf.analyse(txt, sym, pixels_per_lane=cf.PIXELS_PER_LANE[key])
        assert flops == committed["flops_per_pixel_visit" + key], (key, flops, committed["flops_per_pixel_visit" + key])
        assert mix == committed["instruction_mix" + key], (key, mix, committed["instruction_mix" + key])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_the_fp32_pixel_kernels_scratch_stays_out_of_its_loops():
    """pixel_kernel<2, float> keeps 56 B per lane in scratch at three waves per SIMD (measured faster than two waves and none).
    Where the spill code sits, from LLVM's loop annotations in the listing: STORES only in the kernel's prologue and once per
    work item (depth <= 1: the work-list loop) -- never per chunk or per trip --; inside the pixel loop (depth 3) at most two
    8-byte reloads per 128-pixel trip; nothing at all inside the component loops (depth >= 4).  The fp64 kernel has no scratch."""
    import count_flops as cf
    txt = "\n".join(compiled_listing())

    def scratch_by_depth(sym):
        i = txt.index(sym); i = txt.index(":\n", i)
        lines = txt[i:txt.index("s_endpgm", i)].split("\n")
        blocks, parent = cf.blocks_of(lines)

        def depth(lp):
            return 0 if lp is None else 1 + depth(parent.get(lp))
        out = []
        for _, a, b, lp in blocks:
            for ln in lines[a:b + 1]:
                t = ln.strip().split()
                if t and t[0].startswith("scratch_"):
                    out.append((depth(lp), t[0]))
        return out
    f32 = scratch_by_depth("_Z12pixel_kernelILi2EfLb0EEv")
    assert f32, "the kernel no longer spills: tighten tests/test_dpp_hazard.py and DESIGN.md"
    assert all(d <= 1 for d, op in f32 if op.startswith("scratch_store")), f32
    assert sum(1 for d, op in f32 if d == 3) <= 2 and all(op.startswith("scratch_load") for d, op in f32 if d >= 2), f32
    assert not [x for x in f32 if x[0] >= 4], f32
    assert scratch_by_depth("_Z12pixel_kernelILi2EdLb0EEv") == []
