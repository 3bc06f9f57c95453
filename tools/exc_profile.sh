#!/bin/bash
# Per-kernel device time (rocprofv3 --kernel-trace --stats) of the workloads that leave the tile kernel's fast path:
# random-letter words of 33-62, 70-120 and 300-900 letters (exception kernels) and CJK paragraphs (seams).
#   tools/exc_profile.sh TAG   -> gpurun_out/exc_TAG/*.txt
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 tools/kernel_times.py "$@" > "$OUT/$name.log" 2>&1 || echo "failed: $name"
  f=$(find "$OUT/$name" -name "*kernel_stats.csv" | head -1)
  { echo "== $name: $(head -1 $OUT/$name.log)"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("  %-60s calls %4s  total %10.3f ms  avg %9.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  } | tee -a "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
run w33_62 words 33 62
run w70_120 words 70 120
run w300_900 words 300 900
run cjk cjk 50000
