"""Summarise a tools/profile_round.sh run: python tools/summarize_profile.py <tag>
Reads gpurun_out/<tag>/, writes profiles/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.md and
refreshes profiles/hbm_traffic.json (the `traffic` figure bench.py reports)."""
import glob, json, os, shutil, sys
import pandas as pd

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name


def means(sub):
    out = {}
    for f in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
        df = pd.read_csv(f)
        df["k"] = df.Kernel_Name.map(short)
        # the bench's optimiser leg launches the same kernels on shrinking batches: keep full-batch dispatches only
        df = df[df.Grid_Size == df.groupby("k").Grid_Size.transform("max")]
        for (k, c), v in df.groupby(["k", "Counter_Name"]).Counter_Value.mean().items():
            out.setdefault(k, {})[c] = float(v)
    return out


fetch, write, sq = means("pmc_fetch"), means("pmc_write"), means("pmc_sq")
kt = pd.read_csv(glob.glob(os.path.join(src, "trace", "*kernel_trace.csv"))[0])
kt["k"] = kt.Kernel_Name.map(short)
kt["ns"] = kt.End_Timestamp - kt.Start_Timestamp
kt = kt[kt.Grid_Size_X == kt.groupby("k").Grid_Size_X.transform("max")]
ks = kt.groupby("k").ns.mean().reset_index().rename(columns={"ns": "AverageNs"})
lines = ["# %s PMC summary (rocprofv3 --pmc, separate passes; per-launch means; workload = bench.py config 3)" % tag, "",
         "Command: tools/profile_round.sh %s (FETCH_SIZE, WRITE_SIZE and SQ counters each in their own pass, "
         "--kernel-trace only).  FETCH_SIZE / WRITE_SIZE are KiB as reported; on gfx950 FETCH_SIZE counts half the "
         "bytes of a >= 16 B/lane coalesced streaming read (guide, HBM section), so the corrected column doubles it "
         "for record_sum_kernel (16 B/lane loads); the pixel kernel's loads are 4-8 B/lane and stay uncorrected." % tag, "",
         "| kernel | avg us (kernel-trace, full-batch launches) | FETCH_SIZE KiB | WRITE_SIZE KiB | fetch bytes (corrected) | write bytes |",
         "|---|---|---|---|---|---|"]
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "reduce_kernel" in k:
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0); w = write.get(k, {}).get("WRITE_SIZE", 0.0)
    corr = 2.0 if k.startswith("record_sum_kernel") else 1.0
    row = ks[ks.k == k]
    avg = float(row.AverageNs.iloc[0]) / 1e3 if len(row) else float("nan")
    lines.append("| %s | %.1f | %.0f | %.0f | %.0f | %.0f |" % (k, avg, f, w, f * 1024 * corr, w * 1024))
    traffic[k] = {"fetch_bytes": f * 1024 * corr, "write_bytes": w * 1024, "avg_us": avg}
lines += ["", "## SQ counters (per launch)", "", "```"]
for k in sorted(sq):
    if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k:
        continue
    c = sq[k]
    util = c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_BUSY_CYCLES", 1) * 4.0 / 1.0, 1)
    lines.append("%-28s %s" % (k, {n: round(v) for n, v in c.items()}))
lines += ["```", ""]
def find(table, prefix):
    for k, v in table.items():
        if k.startswith(prefix):
            return v
    return {}


pk = find(sq, "pixel_kernel<2, double")
valu_util = None
if pk:
    simd_cycles = pk["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    valu_util = pk["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
    lines += ["## Reading (pixel_kernel<2>, one launch = one sweep of the bench field)", "",
              "* GRBM_GUI_ACTIVE / 8 XCDs = %.3g cycles; x 1024 SIMDs = %.3g SIMD-cycles." % (pk["GRBM_GUI_ACTIVE"] / 8, simd_cycles),
              "* SQ_ACTIVE_INST_VALU = %.3g quad-cycles -> **%.0f %% of all SIMD issue cycles are VALU** (FP64-bound)."
              % (pk["SQ_ACTIVE_INST_VALU"], 100 * valu_util),
              "* SQ_WAVE_CYCLES / SIMD-cycles = %.2f resident waves per SIMD." % (pk["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles), ""]
# ---- the optimiser / joint-inference kernels (bench.py --pmc-child optim), when the round's run has those passes ----
def sums(sub):
    """per kernel: every counter SUMMED over all its dispatches of the child run, and the number of dispatches"""
    out = {}
    for f in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
        df = pd.read_csv(f)
        df["k"] = df.Kernel_Name.map(short)
        for (k, c), g in df.groupby(["k", "Counter_Name"]):
            out.setdefault(k, {})[c] = float(g.Counter_Value.sum())
            out[k]["dispatches"] = int(len(g))
    return out


opt_sq, opt_lds, opt_f, opt_w = sums("opt_sq"), sums("opt_lds"), sums("opt_fetch"), sums("opt_write")
optim = {}
if opt_sq:
    okt = glob.glob(os.path.join(src, "opt_trace", "*kernel_trace.csv"))
    dur = {}
    if okt:
        t = pd.read_csv(okt[0]); t["k"] = t.Kernel_Name.map(short); t["ns"] = t.End_Timestamp - t.Start_Timestamp
        dur = {k: (float(g.ns.sum()), int(len(g))) for k, g in t.groupby("k")}
        shutil.copy(glob.glob(os.path.join(src, "opt_trace", "*kernel_stats.csv"))[0], os.path.join(dst, tag + "_optim_kernel_stats.csv"))
    lines += ["## Optimiser and joint-inference kernels (`bench.py --pmc-child optim`: maximize! of all 2000 sources with the "
              "lock-step driver, three 80-target layers through the fused launch, the joint schedule as one dataflow launch; "
              "counters SUMMED over the kernel's dispatches, separate passes)", "",
              "| kernel | dispatches | total ms (kernel-trace) | VALU issue = ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024) | "
              "resident waves / SIMD | WAIT_INST_ANY / WAVE_CYCLES | WAIT_ANY / WAVE_CYCLES | FETCH MB | WRITE MB | "
              "LDS bank-conflict cycles / LDS active cycles |", "|---|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(opt_sq):
        if not (k.startswith("optim_") or k.startswith("pixel_kernel<2, double") or k.startswith("lift_kernel")):
            continue
        c, l = opt_sq[k], opt_lds.get(k, {})
        simd = c.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0
        wc = max(c.get("SQ_WAVE_CYCLES", 0), 1.0)
        wc2 = max(l.get("SQ_WAVE_CYCLES", wc), 1.0)
        row = {"dispatches": c.get("dispatches"), "total_ms": dur.get(k, (float("nan"), 0))[0] / 1e6,
               "valu_issue": c.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / max(simd, 1.0),
               "waves_per_simd": c.get("SQ_WAVE_CYCLES", 0) * 4.0 / max(simd, 1.0),
               "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / wc,
               "wait_any": (l.get("SQ_WAIT_ANY", float("nan")) / wc) if l else float("nan"),
               "fetch_mb": opt_f.get(k, {}).get("FETCH_SIZE", float("nan")) * 1024 / 1e6,
               "write_mb": opt_w.get(k, {}).get("WRITE_SIZE", float("nan")) * 1024 / 1e6,
               "lds_conflict": (l.get("SQ_LDS_BANK_CONFLICT", 0) / max(l.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)) if l else float("nan")}
        optim[k] = row
        lines.append("| %s | %s | %.2f | %.3f | %.2f | %.3f | %.3f | %.1f | %.1f | %.3f |"
                     % (k, row["dispatches"], row["total_ms"], row["valu_issue"], row["waves_per_simd"], row["wait_inst_any"],
                        row["wait_any"], row["fetch_mb"], row["write_mb"], row["lds_conflict"]))
    lines += ["", "(persistent launches: GRBM_GUI_ACTIVE covers the whole launch, so `VALU issue` is the fraction of the CHIP's "
              "issue slots the launch used -- a latency-bound dataflow fills few of them by design; scratch traffic shows up in "
              "FETCH / WRITE.)", ""]
    json.dump(optim, open(os.path.join(dst, tag + "_optim_pmc.json"), "w"), indent=1)
open(os.path.join(dst, tag + "_pmc_summary.md"), "w").write("\n".join(lines))
px = find(traffic, "pixel_kernel<2, double")
rs = find(traffic, "record_sum_kernel")
sys.path.insert(0, os.path.join(root, "tools"))
import count_flops
flops = count_flops.main()   # FP64 flops per visited pixel in the ISA of the kernels as they are in this tree
path = os.path.join(dst, "hbm_traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}      # (count_flops --write keeps its instruction-mix keys here too)
new = {"pixel_kernel_bytes_per_launch": px.get("fetch_bytes", 0) + px.get("write_bytes", 0),
       "flops_per_pixel_visit": flops,
       "fetch_bytes": px.get("fetch_bytes"), "write_bytes": px.get("write_bytes"),
       "pixel_kernel_valu_utilization": valu_util,
       "source": "profiles/%s_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KiB x 1024; "
                 "record_sum FETCH_SIZE doubled per the gfx950 16 B/lane correction, pixel kernel uncorrected)" % tag}
rsb = (rs.get("fetch_bytes", 0) + rs.get("write_bytes", 0)) or None
if rsb is not None or "record_sum_bytes_per_launch" not in d:     # (PMC passes without the split sweeps keep the last measured figure)
    new["record_sum_bytes_per_launch"] = rsb
d.update(new)
json.dump(d, open(path, "w"), indent=1)
print("\n".join(lines))
