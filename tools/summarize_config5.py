"""Summarise a tools/pmc_config5.sh run: python tools/summarize_config5.py <tag> [<out tag>]
Reads gpurun_out/<tag>/, writes profiles/<out tag>_config5_pmc_summary.md and profiles/<out tag>_config5_kernel_stats.csv."""
import glob, json, os, shutil, sys
import pandas as pd

tag = sys.argv[1]
out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
shutil.copy(glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0], os.path.join(dst, out_tag + "_config5_kernel_stats.csv"))
stats = pd.read_csv(os.path.join(dst, out_tag + "_config5_kernel_stats.csv"))


def short(name):
    return name.split("(")[0].replace("void ", "")


def means(sub):
    out = {}
    for f in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
        df = pd.read_csv(f)
        df["k"] = df.Kernel_Name.map(short)
        df = df[df.Grid_Size == df.groupby("k").Grid_Size.transform("max")]      # full-batch dispatches only
        for (k, c), v in df.groupby(["k", "Counter_Name"]).Counter_Value.mean().items():
            out.setdefault(k, {})[c] = float(v)
    return out


sq, lds, fetch, write = means("pmc_sq"), means("pmc_lds"), means("pmc_fetch"), means("pmc_write")
keep = [k for k in sorted(sq) if k.startswith(("pixel_kernel", "value_kernel", "lift_kernel", "prep_kernel", "setup", "work_"))]
lines = ["# %s: counters of the config-5 kernels (BASELINE configs[4] at full size: 16 fields, 80 images, 30 000 sources)" % out_tag, "",
         "Command: tools/pmc_config5.sh (bench.py --config 5 --dtype f32 --steps 3 under rocprofv3; SQ, LDS, FETCH_SIZE, WRITE_SIZE each in",
         "their own --pmc pass with --kernel-trace only).  Per-launch means over the full-batch dispatches.  The fp64 kernels appear",
         "because the run checks every source against the fp64 device path (fp32_vs_fp64_device).", "",
         "| kernel | VALU busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024) | waves per SIMD = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS | FETCH_SIZE KiB | WRITE_SIZE KiB | SQ_INSTS_VALU |",
         "|---|---|---|---|---|---|---|---|"]
summary = {}
for k in keep:
    c, l = sq[k], lds.get(k, {})
    simd_cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
    valu = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / simd_cycles if simd_cycles else float("nan")
    wps = c.get("SQ_WAVE_CYCLES", 0.0) / max(c.get("SQ_BUSY_CYCLES", 1.0), 1.0) / 4.0
    wait = c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    bank = l.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(l.get("SQ_ACTIVE_INST_LDS", 1.0), 1.0)
    f, w = fetch.get(k, {}).get("FETCH_SIZE", float("nan")), write.get(k, {}).get("WRITE_SIZE", float("nan"))
    lines.append("| %s | %.3f | %.2f | %.3f | %.3f | %.0f | %.0f | %.0f |" % (k, valu, wps, wait, bank, f, w, c.get("SQ_INSTS_VALU", 0.0)))
    summary[k] = {"valu_busy": valu, "waves_per_simd": wps, "wait_inst_any_share": wait, "lds_bank_conflict_share": bank,
                  "fetch_kib": f, "write_kib": w, "raw_sq": c, "raw_lds": l}
lines += ["", "## kernel-trace --stats of the same command", "", "```"]
for _, r in stats.iterrows():
    n = short(r["Name"])
    if n.startswith(("pixel_kernel", "value_kernel", "lift_kernel", "prep_kernel", "setup", "work_")):
        lines.append("%-40s calls %4d  avg %10.1f us  total %6.2f %%" % (n, r["Calls"], r["AverageNs"] / 1e3, r["Percentage"]))
lines.append("```")
open(os.path.join(dst, out_tag + "_config5_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(dst, out_tag + "_config5_pmc.json"), "w"), indent=1)
print("\n".join(lines))
