#!/usr/bin/env python3
"""tools/valu_issue_bench output (JSON lines) -> cost table: cycles per wave-instruction per SIMD at 1, 2, 4, 8 wavefronts
per SIMD, independent and dependent streams; with --isa <isa_histogram.json>: the static mix of a kernel priced with it.
usage: summarize_issue.py gpurun_out/r03_valu_issue.jsonl [--isa isa.json] [--json out.json]"""
import json, sys
from collections import defaultdict
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
hdr, rows = rows[0], rows[1:]
t = defaultdict(dict)
for r in rows:
    t[(r["op"], r["form"])][r["waves_per_simd"]] = r["cycles_per_inst_per_simd"]
clk = [r["clock_ghz_k1"] for r in rows if r["waves_per_simd"] == 1]
print("device", hdr["device"], " clock (s_memtime ticks / event time, k = 1): %.2f-%.2f GHz" % (min(clk), max(clk)))
print("%-16s %-6s %s" % ("op", "form", "cycles per wave-instruction per SIMD at 1, 2, 4, 8 wavefronts per SIMD"))
cost = {}
for (op, form), v in t.items():
    print("%-16s %-6s " % (op, form) + " ".join("%6.2f" % v[w] for w in (1, 2, 4, 8)))
    if form == "indep":
        cost[op] = v[8]
if "--json" in sys.argv:
    json.dump({"device": hdr["device"], "cycles_per_inst_per_simd_8_waves": cost,
               "table": {"%s/%s" % k: v for k, v in t.items()}}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
