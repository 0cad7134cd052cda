#!/bin/bash
# Builds a variant of the library for A/B timing: tools/build_variant.sh <name> <extra hipcc flags...>
# -> _ab/<name>/libocrf_hip.so (use with OCRF_HIP_SO=...).  Only raster_plan.hip is rebuilt with the flags.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/_ab/$NAME
mkdir -p "$OUT"
cd "$ROOT/ocrfdet_amd/csrc"
make -s libocrf_hip.so
/opt/rocm/bin/hipcc "$@" -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I"$ROOT/include" -c raster_plan.hip -o "$OUT/raster_plan.o"
OBJS=$(ls *.o | grep -v raster_plan.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libocrf_hip.so" $OBJS "$OUT/raster_plan.o"
echo "$OUT/libocrf_hip.so"
