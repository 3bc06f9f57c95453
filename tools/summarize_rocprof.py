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
    for name, vals in acc.items():
        lines.append(f"- {name}: {sum(vals) / len(vals):.4g}  (n={len(vals)})")
        pmc[name] = sum(vals) / len(vals)
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
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
