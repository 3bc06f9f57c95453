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

lines += ["## PMC counters (per k_tiles dispatch, mean over the run's dispatches)", ""]
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        lines.append(f"- {os.path.basename(d)}: no counter file (set rejected?)")
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "k_tiles" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, vals in acc.items():
        lines.append(f"- {name}: {sum(vals) / len(vals):.4g}  (n={len(vals)})")
lines.append("")
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
