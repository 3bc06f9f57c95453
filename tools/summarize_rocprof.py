#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into profiles/<tag>_rocprof_summary.md and
profiles/<tag>_kernel_stats.csv -- the files the round's numbers are quoted from."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
lines = [f"# rocprofv3 summary, round tag {tag}", "",
         "Command: `python3 bench.py --steps 10 --warmup 3 --no-cpu` under `rocprofv3 --kernel-trace --stats` "
         "(PMC passes: `--steps 3 --warmup 1`, one counter set per run, `--pmc ... --kernel-trace`).", ""]


def short(name):
    name = name.split("(")[0]
    return name.replace("void hutk::", "").replace("hutk::", "")


j = None
bj = os.path.join(src, "bench_kt.json")
if os.path.exists(bj):
    txt = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if txt:
        j = json.loads(txt[-1])
        lines += ["## bench.py line of the traced run", "", "```json", json.dumps(j), "```", ""]

stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    lines += ["## kernel stats (`--kernel-trace --stats`)", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:12]:
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                     f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    lines.append("")

trace = glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    k = [r for r in rows if "k_tiles" in r["Kernel_Name"]]
    if k:
        r = k[-1]
        keys = [x for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                            "Workgroup_Size", "Grid_Size") if x in r]
        lines += ["## k_tiles dispatch", "", ", ".join(f"{x}={r[x]}" for x in keys), ""]
        durs = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in k]
        lines += [f"k_tiles durations (us), last {min(10, len(durs))}: " + ", ".join(f"{d:.1f}" for d in durs[-10:]), ""]
        gs = "Grid_Size" if "Grid_Size" in r else "Grid_Size_X" if "Grid_Size_X" in r else None
        if gs:
            full = max(int(x[gs]) for x in k)
            fd = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in k if int(x[gs]) == full]
            lines += [f"mean over the {len(fd)} full-size launches of the bench workload: {sum(fd) / len(fd):.1f} us "
                      f"(the --stats average above also counts the {len(k) - len(fd)} small launch(es) with which context "
                      "creation verifies its tables)", ""]

lines += ["## PMC counters (per full-size k_tiles dispatch, mean over the run's dispatches of the bench workload)", ""]
pmc = {}
pmc_ms = {}  # counter -> mean duration (ms) of the full-size k_tiles launches in the run that collected it
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        lines.append(f"- {os.path.basename(d)}: no counter file (set rejected?)")
        continue
    acc = defaultdict(list)
    rows = [r for r in csv.DictReader(open(files[0])) if "k_tiles" in r.get("Kernel_Name", "")]
    # only the full-size launches of the bench workload (context creation runs a small one to verify its tables)
    full = max((int(r["Grid_Size"]) for r in rows), default=0)
    for r in rows:
        if int(r["Grid_Size"]) == full:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    ms = None
    if kt:
        kr = [r for r in csv.DictReader(open(kt[0])) if "k_tiles" in r["Kernel_Name"]]
        gs = "Grid_Size" if kr and "Grid_Size" in kr[0] else "Grid_Size_X"
        if kr:
            fullk = max(int(r[gs]) for r in kr)
            du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in kr if int(r[gs]) == fullk]
            ms = sum(du) / len(du)
    for name, vals in acc.items():
        lines.append(f"- {name}: {sum(vals) / len(vals):.4g}  (n={len(vals)})")
        pmc[name] = sum(vals) / len(vals)
        pmc_ms[name] = ms
lines.append("")
# HBM-side traffic of one k_tiles launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE are kilobytes at the L2's memory side (Infinity-Cache hits included); on
# gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is exact.
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    tj = {"tag": tag, "kernel": "k_tiles", "fetch_size_kb": pmc["FETCH_SIZE"], "write_size_kb": pmc["WRITE_SIZE"],
          "traffic_bytes_per_launch": traffic,
          "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024; memory-side of L2, Infinity-Cache hits included",
          "workload_bytes": (j or {}).get("config", {}).get("bytes_per_gpu")}
    json.dump(tj, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    lines += [f"HBM-side traffic per k_tiles launch (corrected): {traffic / 1e9:.3f} GB", ""]
# Where a resident wavefront's cycles go.  SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY and SQ_ACTIVE_INST_* all count
# QUAD-cycles of one wavefront, summed over the wavefronts (MI355X_MICROARCH.md, PMC section: WAIT_ANY + WAIT_INST_ANY +
# ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint), so their quotients are shares of a wavefront's residence and cannot exceed 1.
# (Round 3 printed SQ_INSTS_VALU x an assumed 2..4 cycles over the SIMD cycles as an "issue roofline": a range whose upper
# end passed 1; dropped.  SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU comes out at one quad-cycle per instruction: the counter
# charges an instruction one quad-cycle whatever its opcode, so it does not price instructions either.)
if "SQ_INSTS_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0  # (summed over the 8 XCDs)
    wc = pmc.get("SQ_WAVE_CYCLES")
    share = lambda name: (pmc[name] / wc) if (wc and name in pmc) else None  # noqa: E731
    ij = {"tag": tag, "kernel": "k_tiles", "valu_insts_per_launch": pmc["SQ_INSTS_VALU"],
          "salu_insts_per_launch": pmc.get("SQ_INSTS_SALU"), "lds_insts_per_launch": pmc.get("SQ_INSTS_LDS"),
          "busy_cycles_per_launch": cycles, "kernel_ms_profiled": pmc_ms.get("GRBM_GUI_ACTIVE"),
          "unit_note": "SQ_* cycle counters are quad-cycles of one wavefront summed over wavefronts; shares are of SQ_WAVE_CYCLES",
          "wait_any_frac": share("SQ_WAIT_ANY"), "wait_inst_frac": share("SQ_WAIT_INST_ANY"),
          "valu_active_frac": share("SQ_ACTIVE_INST_VALU"),
          "lds_bank_conflict_frac": (pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"])
          if "SQ_LDS_BANK_CONFLICT" in pmc and pmc.get("SQ_LDS_IDX_ACTIVE") else None,
          "ta_busy_frac": (pmc["TA_BUSY_avr"] / cycles) if "TA_BUSY_avr" in pmc else None,
          "workload_bytes": (j or {}).get("config", {}).get("bytes_per_gpu")}
    json.dump(ij, open(os.path.join(dst, f"{tag}_issue.json"), "w"), indent=1)
    lines += [f"VALU wave-instructions per launch: {ij['valu_insts_per_launch']:.4g}; SALU {ij['salu_insts_per_launch'] or 0:.4g}; LDS {ij['lds_insts_per_launch'] or 0:.4g}", ""]
    if ij["valu_active_frac"] is not None:
        lines += [f"Share of a resident wavefront's cycles spent executing VALU instructions (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES): {ij['valu_active_frac']:.3f}", ""]
    if ij["wait_inst_frac"] is not None:
        lines += [f"... stalled at issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES): {ij['wait_inst_frac']:.3f}", ""]
    if ij["wait_any_frac"] is not None:
        lines += [f"Wavefronts parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES): {ij['wait_any_frac']:.3f}", ""]
    if ij["lds_bank_conflict_frac"] is not None:
        lines += [f"LDS bank conflicts: {ij['lds_bank_conflict_frac']:.3f} of the LDS-active cycles", ""]
    if ij["ta_busy_frac"] is not None:
        lines += [f"Texture-address units busy: {ij['ta_busy_frac']:.3f} of the launch", ""]
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
