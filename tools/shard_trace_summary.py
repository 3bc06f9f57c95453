#!/usr/bin/env python3
"""Splits the steps of a kernel trace (rocprofv3 --kernel-trace of tools/shard_steps.py) into kernels and gaps.
usage: shard_trace_summary.py kernel_trace.csv [label]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "hutk" in r["Kernel_Name"]]
# a step begins with k_pre; drop the context's own small launches (grid of k_pre below the batch's) and the first 5 steps
pre = [i for i, r in enumerate(rows) if "k_pre" in r["Kernel_Name"]]
gs = "Grid_Size" if "Grid_Size" in rows[0] else "Grid_Size_X"
big = max(int(rows[i][gs]) for i in pre)
starts = [i for i in pre if int(rows[i][gs]) == big][5:]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    t0, t1 = int(ks[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = collections.OrderedDict()
    for r in ks:
        name = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        busy[name] = busy.get(name, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    gaps = (t1 - t0) - sum(busy.values())
    steps.append((t1 - t0, busy, gaps))
n = len(steps)
print(f"{sys.argv[2] if len(sys.argv) > 2 else ''}: {n} steps, mean step {sum(s[0] for s in steps) / n / 1e3:.1f} us (start of k_pre to start of the next step's k_pre)")
names = list(steps[0][1].keys())
for k in names:
    print(f"  {k:14s} {sum(s[1].get(k, 0) for s in steps) / n / 1e3:8.1f} us")
print(f"  {'gaps':14s} {sum(s[2] for s in steps) / n / 1e3:8.1f} us  ({len(names)} launches)")
