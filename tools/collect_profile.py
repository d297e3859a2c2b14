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
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERN = "noahmp_ranges_kernel"        # the step's column kernel: the class ranges of the sorted layout in one launch (round 6; before: noahmp_column_kernel<64, true, 1>)

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
        # share of the instructions that are neither float64 nor conversions nor transcendental and issue at half rate (compares, selects,
        # min / max, the division helpers, lane moves, packed float32): the static mix of the land kernel (tools/isa_stats.py on a
        # `hipcc -S` listing of noahmp_engine_d3_r1.hip: 4 470 of 11 150)
        "half_rate_share_of_rest": 0.40,
        "cycles_resident_per_wave": 4.0 * pmc.get("SQ_WAVE_CYCLES", 0) / waves,
    },
    "bench_line": bench,
}
json.dump(out, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
sys.path.insert(0, ROOT)
import bench as _bench  # noqa: E402
VR = _bench.valu_roofline(pmc, out["derived"], float(krow["AverageNs"]) / 1e6, "this profile")
L = ["# %s profile: `python3 bench.py` under rocprofv3 (MI355X, 1 GPU, %s: %d columns per launch of the dominant kernel)" % (tag, out["workload"], ncol), "",
     "## `rocprofv3 --kernel-trace --stats` (copied: %s_kernel_stats.csv)" % tag, "",
     "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for r in csv.DictReader(open(stats)):
    L.append("| %s | %s | %.0f | %s | %s | %s |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
L += ["", "Launch: grid %(Grid_Size)s, workgroup %(Workgroup_Size)s, LDS %(LDS_Block_Size)s B/block, scratch %(Scratch_Size)s B/lane, "
      "VGPR %(VGPR_Count)s, AGPR %(Accum_VGPR_Count)s, SGPR %(SGPR_Count)s (rocprofv3's fields; the compiler's "
      "`-Rpass-analysis=kernel-resource-usage` report for this kernel is 231 unified VGPRs, none spilled, no scratch (option-specialised land-only kernel), 2 waves/SIMD)." % meta, "",
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
      "- vector-ALU occupancy under the measured price list (profiles/r04_valu_issue.txt; `bench.py: valu_roofline`): **%.2f** of the kernel's duration "
      "(every instruction at the guide's 2 cycles: %.2f; every instruction at 4 cycles: %.2f); a wave issues one VALU instruction per %.2f cycles on average "
      "(4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU) and is resident %.0f cycles (4 x SQ_WAVE_CYCLES / waves)"
      % (VR["frac"], VR["frac_if_every_instruction_took_2_cycles"], VR["frac_if_every_instruction_took_4_cycles"],
         VR.get("wave_cadence_cycles_per_instruction", 0.0), out["derived"]["cycles_resident_per_wave"]),
      "- instruction cache: %.3g misses per %.3g requests (%.2f %%); LDS instructions per wave %.0f; float64 VALU instructions %.1f %% and conversions %.1f %% of all VALU instructions"
      % (pmc.get("SQC_ICACHE_MISSES", 0), pmc.get("SQC_ICACHE_REQ", 1), 100 * pmc.get("SQC_ICACHE_MISSES", 0) / max(pmc.get("SQC_ICACHE_REQ", 1), 1),
         pmc.get("SQ_INSTS_LDS", 0) / waves,
         100 * (pmc.get("SQ_INSTS_VALU_FMA_F64", 0) + pmc.get("SQ_INSTS_VALU_MUL_F64", 0) + pmc.get("SQ_INSTS_VALU_ADD_F64", 0)) / max(pmc.get("SQ_INSTS_VALU", 1), 1),
         100 * pmc.get("SQ_INSTS_VALU_CVT", 0) / max(pmc.get("SQ_INSTS_VALU", 1), 1)),
      "- the kernel is bound by the serial instruction streams of its two waves per SIMD (a wave alone issues at most one VALU instruction per ~4.2 cycles) and their stalls (every wave waits ~%.0f %% of its cycles, mostly on LDS look-ups of the libm tables and the layer arrays), not by HBM (SURVEY.md 8d) and not by vector-ALU throughput"
      % (100 * out["derived"]["wait_any_share_of_wave_cycles"]),
      "", "## bench.py line of the same build (un-profiled run)", "", "```", json.dumps(bench), "```", ""]
# ---- the per-step forcing permutation of the bench (five 2-D planes of the whole tile -- of the two-level T3D only level 1 travels:
# read + write + 6 B of plan per column)
ntile = bench["config"]["columns_per_gpu"]
for r in csv.DictReader(open(stats)):
    if "noahmp_scatter" in r["Name"]:
        b = ntile * (5 * 8 + 6)
        L += ["## Forcing permutation (`%s`, %s calls)" % (r["Name"].split("(")[0].split("::")[-1][:40], r["Calls"]), "",
              "five planes x %d columns x (4 B read + 4 B written) + 6 B of plan per column = %.0f MB in %.1f us = **%.2f TB/s = %.0f %% of 8 TB/s** (round 2: six planes in 203 us, 23 %%)"
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
# ---- config 5: traffic and SQ counters of its land kernel (separate passes of `bench.py --workload config5`)
p5 = {}
k5name = None
for d in ("fetch5", "write5", "sq5"):
    fs = glob.glob(os.path.join(src, d, "*_counter_collection.csv")) + glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if KERN in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            k5name = r["Kernel_Name"]
    for k, v in acc.items():
        p5[k] = sum(v) / len(v)
if p5:
    t5 = glob.glob(os.path.join(src, "trace5", "*_kernel_stats.csv")) + glob.glob(os.path.join(src, "trace5", "*", "*_kernel_stats.csv"))
    k5 = [r for r in csv.DictReader(open(t5[0])) if r["Name"] == k5name] if t5 else []
    w5 = p5.get("SQ_WAVES", 1)
    ncol5 = None
    try:
        b5 = [json.loads(l) for l in open(os.path.join(src, "bench_trace5.log")) if l.startswith("{")][-1]
        ncol5 = b5["roofline"]["columns_per_launch"]
    except Exception:                                     # noqa: BLE001
        b5 = None
    o5 = {"round": tag, "workload": "config5", "kernel": k5name, "columns_per_launch": ncol5,
          "avg_kernel_ns": float(k5[0]["AverageNs"]) if k5 else None, "calls": int(k5[0]["Calls"]) if k5 else None,
          "algorithmic_bytes_per_launch": 824 * ncol5 if ncol5 else None,
          "hbm_bytes_per_launch": p5.get("FETCH_SIZE", 0) * 1024 * 2 + p5.get("WRITE_SIZE", 0) * 1024,
          "pmc_mean_per_launch": p5,
          "derived": {"valu_insts_per_column_step": p5.get("SQ_INSTS_VALU", 0) / w5, "salu_insts_per_wave": p5.get("SQ_INSTS_SALU", 0) / w5,
                      "lane_utilisation": p5.get("SQ_THREAD_CYCLES_VALU", 0) / max(p5.get("SQ_ACTIVE_INST_VALU", 1) * 64, 1),
                      "wait_any_share_of_wave_cycles": p5.get("SQ_WAIT_ANY", 0) / max(p5.get("SQ_WAVE_CYCLES", 1), 1),
                      "cycles_resident_per_wave": 4.0 * p5.get("SQ_WAVE_CYCLES", 0) / w5},
          "bench_line": b5}
    json.dump(o5, open(os.path.join(dst, "%s_traffic5.json" % tag), "w"), indent=1)
    L += ["## config 5 land kernel: counters (copied: %s_traffic5.json)" % tag, "",
          "- %s columns per launch, %.3f ms per launch; HBM traffic %.1f MB = %.2fx algorithmic; %.0f VALU instructions per wave, lane utilisation **%.3f** "
          "(config 3: %.3f), waiting %.0f %% of wave cycles" % (ncol5, (o5["avg_kernel_ns"] or 0) / 1e6, o5["hbm_bytes_per_launch"] / 1e6,
                                                            o5["hbm_bytes_per_launch"] / max(o5["algorithmic_bytes_per_launch"] or 1, 1),
                                                            o5["derived"]["valu_insts_per_column_step"], o5["derived"]["lane_utilisation"],
                                                            out["derived"]["lane_utilisation"], 100 * o5["derived"]["wait_any_share_of_wave_cycles"]), ""]
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
