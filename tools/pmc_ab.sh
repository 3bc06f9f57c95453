#!/bin/bash
# PMC counters of k_tiles for several library builds on ONE GPU box (gpurun, from the repo root):
#   tools/pmc_ab.sh TAG CORPUS N_DOCS VOCAB lib1.so lib2.so ...
# PMC_KERNEL=<substring> selects another kernel than k_tiles.  One counter set per run (rocprofv3 --pmc with --kernel-trace only); the per-dispatch means of the full-size
# k_tiles launches go to gpurun_out/pmc_<TAG>.txt.
set -o pipefail
TAG=$1; CORPUS=$2; NDOCS=$3; VOCAB=$4; shift 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG")
if [ -n "$PMC_SETS" ]; then IFS=';' read -r -a SETS <<< "$PMC_SETS"; fi  # e.g. PMC_SETS="SQ_INSTS_VALU SQ_INSTS_SALU;TA_BUSY_avr"
for lib in "$@"; do
  export HUTOKEN_AMD_LIB=$ROOT/$lib
  name=$(basename $lib .so)
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${name}_$i" -- python3 tools/profile_phases.py $CORPUS $NDOCS $VOCAB > "$OUT/${name}_$i.log" 2>&1 || echo "pmc set failed ($name): $set"
    i=$((i+1))
  done
done
python3 - "$OUT" "$@" <<'PY' | tee "$ROOT/gpurun_out/pmc_$TAG.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for lib in sys.argv[2:]:
    name = os.path.basename(lib)[:-3]
    print("==", name)
    for d in sorted(glob.glob(os.path.join(out, name + "_[0-9]*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if os.environ.get("PMC_KERNEL", "k_tiles") in r.get("Kernel_Name", "")]
            full = max((int(r["Grid_Size"]) for r in rows), default=0)
            acc = defaultdict(list)
            for r in rows:
                if int(r["Grid_Size"]) == full:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                print(f"  {k}: {sum(v)/len(v):.5g} (n={len(v)}, grid {full})")
PY
