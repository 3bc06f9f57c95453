#!/bin/bash
# Per-kernel device time of the exception-word workloads, LAST of five batches each (rocprofv3 --kernel-trace):
#   tools/exc_trace.sh TAG ["33 62" "70 120" ...]   -> gpurun_out/exc_trace_TAG.txt
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exc_trace_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
[ $# -eq 0 ] && set -- "33 62" "70 120" "130 250 50000" "300 900 20000"
: > "$OUT.txt"
for w in "$@"; do
  n=$(echo $w | tr " " _)
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/$n" -- python3 tools/kernel_times.py $( [[ "$w" == [0-9]* ]] && echo words ) $w > "$OUT/$n.log" 2>&1 || echo "failed: $w"
  python3 - "$OUT/$n" "$w" >> "$OUT.txt" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.OrderedDict()
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    d.setdefault(r["Kernel_Name"].split("(")[0][-40:], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("==", sys.argv[2], "(us, last batch)")
tot = 0
for k, v in d.items():
    if "k_" in k and "hutk" in k:
        print("  %-42s %8.1f" % (k, (v[-1][1] - v[-1][0]) / 1e3)); tot += (v[-1][1] - v[-1][0]) / 1e3
print("  sum %.1f" % tot)
PY
done
cat "$OUT.txt"
