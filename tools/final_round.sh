#!/bin/bash
# One GPU-box session producing everything kept under profiles/ for a round:
#   tools/final_round.sh <tag>     -> gpurun_out/<tag>_*.json, gpurun_out/prof_<tag>/
set -eo pipefail
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out
mkdir -p $O
run() { name=$1; shift; echo "[final] $name: bench.py $*"; date
        timeout -k 10 400 python3 bench.py "$@" > $O/${TAG}_$name.log 2>&1
        grep '^{' $O/${TAG}_$name.log | tail -1 > $O/${TAG}_$name.json; cut -c1-400 $O/${TAG}_$name.json; }
run cfg2_bench
run cfg1_bench --config cfg1_6cam_256x704_bev128x128x8
run cfg4_bench --config cfg4_6cam_8frame_512x1408_bev200x200 --no-cpu-baseline
run neck_cfg2_bench --scope neck --no-cpu-baseline
run neck_cfg2_bench_perstep --scope neck --index-prep per_step --no-cpu-baseline
run neck_cfg2_bench_perstep_devgeom --scope neck --index-prep per_step --device-geometry --no-graph --no-cpu-baseline
run neck_cfg2_bench_perstep_devgeom_graph --scope neck --index-prep per_step --device-geometry --no-cpu-baseline
bash tools/collect_profiles.sh $TAG
