#!/usr/bin/env python3
"""gpurun_out/var_<tag>/ (tools/profile_variants.sh) -> profiles/<tag>_<workload>_rocprof_summary.md: the run's JSON line,
the --stats table, and per-dispatch means of the PMC counters for the workload's dominant kernel (full-size launches)."""
import csv, glob, json, os, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = {"c5vl": "k_tiles", "merges": "k_tiles", "decode": "k_dec_tiles"}
CMD = {"c5vl": "bench.py --corpus C5 --vocab VL --no-cpu --no-extras", "merges": "bench.py --merges --no-cpu --no-extras",
       "decode": "tools/bench_decode.py"}
for name, kern in KERNEL.items():
    if not os.path.isdir(os.path.join(src, name)):
        continue
    lines = [f"# rocprofv3 summary, {tag}, workload `{name}`", "",
             f"Command: `python3 {CMD[name]}` under `rocprofv3 --kernel-trace --stats`; PMC passes: the same command with "
             "`--steps 3 --warmup 1`, one counter set per run (`--pmc ... --kernel-trace`).", ""]
    jf = os.path.join(src, name + ".json")
    if os.path.exists(jf):
        txt = [l for l in open(jf).read().splitlines() if l.startswith("{")]
        if txt:
            lines += ["## JSON line of the traced run", "", "```json", txt[-1], "```", ""]
    st = glob.glob(os.path.join(src, name, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        rows = list(csv.DictReader(open(st[0])))
        lines += ["## kernel stats", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
        for r in rows[:8]:
            nm = r["Name"].split("(")[0].replace("void hutk::", "").replace("hutk::", "")
            lines.append(f"| {nm} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
        lines.append("")
    tr = glob.glob(os.path.join(src, name, "kt", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        rows = [r for r in csv.DictReader(open(tr[0])) if kern in r["Kernel_Name"]]
        if rows:
            gs = "Grid_Size" if "Grid_Size" in rows[0] else "Grid_Size_X"
            full = max(int(r[gs]) for r in rows)
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if int(r[gs]) == full]
            r = rows[-1]
            keys = [x for x in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", gs) if x in r]
            lines += [f"## {kern} dispatch", "", r["Kernel_Name"].split("(")[0], "", ", ".join(f"{x}={r[x]}" for x in keys), "",
                      f"mean duration of the {len(d)} full-size launches: {sum(d) / len(d):.1f} us", ""]
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, name, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "")]
        full = max((int(r["Grid_Size"]) for r in rows), default=0)
        for r in rows:
            if int(r["Grid_Size"]) == full:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        lines += [f"## PMC counters per full-size {kern} dispatch (mean)", ""]
        m = {k: sum(v) / len(v) for k, v in acc.items()}
        for k in sorted(m):
            lines.append(f"- {k}: {m[k]:.4g}  (n={len(acc[k])})")
        lines.append("")
        if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
            lines.append(f"Wavefronts parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES): {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"LDS bank conflicts / LDS-active cycles: {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            lines.append(f"L2 hit rate: {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
        if m.get("TCP_TCC_READ_REQ_sum"):
            lines.append(f"mean L1->L2 read latency: {m['TCP_TCC_READ_REQ_LATENCY_sum'] / m['TCP_TCC_READ_REQ_sum']:.0f} cycles")
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:  # (the guide's correction, as in tools/summarize_rocprof.py)
            lines.append(f"HBM-side traffic per launch, (2 * FETCH_SIZE + WRITE_SIZE) * 1024: {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024 / 1e9:.3f} GB")
        lines.append("")
    open(os.path.join(root, "profiles", f"{tag}_{name}_rocprof_summary.md"), "w").write("\n".join(lines))
    print("wrote", f"profiles/{tag}_{name}_rocprof_summary.md")
