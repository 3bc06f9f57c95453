#!/bin/bash
# rocprofv3 evidence for the instantiations that tools/profile_rocprof.sh does not cover (run through gpurun from the repo root):
#   c5vl    k_tiles<uint16_t, false, true> on C5 x VL, 1 M documents (BASELINE config 5's single-GPU half)
#   merges  the id-keyed merge path: C3 x VG + merges file, 1 M documents
#   decode  the decode direction: tools/bench_decode.py, C3 1 M documents
# Per workload: one --kernel-trace --stats run and separate --pmc runs (kernel trace only beside them).  Raw output under
# gpurun_out/var_$TAG/, summaries by tools/summarize_variant.py (run it again on the merged output to write profiles/).
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/var_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum")
one() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/kt" -- python3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "failed: $name"; return 1; }
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name/pmc_$i" -- python3 "$@" --steps 3 --warmup 1 > "$OUT/${name}_pmc_$i.json" 2> "$OUT/${name}_pmc_$i.err" || echo "pmc set failed ($name): $set"
    i=$((i+1))
  done
  echo "$name done"
}
one c5vl bench.py --corpus C5 --vocab VL --no-cpu --no-extras
one merges bench.py --merges --no-cpu --no-extras
one decode tools/bench_decode.py
python3 tools/summarize_variant.py "$OUT" "$TAG"
