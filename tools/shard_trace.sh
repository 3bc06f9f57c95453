#!/bin/bash
# kernel traces of the 1/8 shard, the 1/2 shard and the whole 1 M documents (gpurun, from the repo root) -> gpurun_out/shard_trace.txt
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
rm -f gpurun_out/shard_trace.txt
for n in 8 2 1; do
  rm -rf gpurun_out/shard_kt_$n
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shard_kt_$n -- python3 tools/shard_steps.py $n > gpurun_out/shard_kt_$n.log 2>&1 || exit 1
  f=$(find gpurun_out/shard_kt_$n -name "*kernel_trace.csv" | head -1)
  python3 tools/shard_trace_summary.py $f "C3 1/$n of 1 M documents" >> gpurun_out/shard_trace.txt
done
cat gpurun_out/shard_trace.txt
