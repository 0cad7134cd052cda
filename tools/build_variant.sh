#!/bin/bash
# Builds a variant of the whole library for A/B timing: tools/build_variant.sh <name> <extra hipcc flags...>
# -> _ab/<name>/libocrf_hip.so (use with OCRF_HIP_SO=...).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/_ab/$NAME
mkdir -p "$OUT"
cd "$ROOT/ocrfdet_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc "$@" -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I"$ROOT/include" -c $f -o "$OUT/${f%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libocrf_hip.so" "$OUT"/*.o
echo "$OUT/libocrf_hip.so"
