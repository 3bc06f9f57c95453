#!/bin/bash
# Build a variant of the HIP library for A/B runs (tools/ab.py, tools/pmc_ab.sh):
#   tools/build_variant.sh NAME [-DHUTK_...=..] ...   ->  hutoken_amd/lib/ab/NAME.so
# Same flags as hutoken_amd/build.py:build_hip plus the given ones.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/hutoken_amd/lib/ab"
C=$ROOT/hutoken_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -I"$ROOT/include" -I"$C" "$@" \
  -o "$ROOT/hutoken_amd/lib/ab/$NAME.so" "$C/hutk_loader.cpp" "$C/hutk_api.cpp" "$C/hutk_kernels.hip" "$C/hutk_ptiles.hip" "$C/hutk_decode.hip" -lpthread
echo "built hutoken_amd/lib/ab/$NAME.so"
