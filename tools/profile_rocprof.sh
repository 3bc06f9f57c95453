#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench.py workload
#   2. PMC passes (separate runs, --kernel-trace only): HBM traffic, LDS, wave cycles
# Raw output goes to gpurun_out/prof_$TAG/, the judged summary is written by
# tools/summarize_rocprof.py into profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="--steps 10 --warmup 3 --no-cpu --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err" || exit 1
echo "kernel trace done"
PARGS="--steps 3 --warmup 1 --no-cpu --no-verify --no-extras"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $PARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || echo "pmc set failed: $set"
  echo "pmc $set done"
done
python3 tools/summarize_rocprof.py "$OUT" "$TAG"
