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
# counters first: bench.py reads profiles/<tag>_pmc_mean.csv (valid only for the sources it was collected from)
STEPS=20 bash tools/collect_profiles.sh $TAG --no-per-step --blocks 1
cp $O/prof_$TAG/pmc_mean.csv profiles/${TAG}_pmc_mean.csv
cp $O/prof_$TAG/pmc_raw.csv profiles/${TAG}_pmc_raw.csv
cp $O/prof_$TAG/pmc_meta.json profiles/${TAG}_pmc_meta.json
run cfg2_bench
run cfg2_bench_per_call --render-mode per_call --no-cpu-baseline --no-per-step
run cfg2_bench_guard_device --render-guard device --no-cpu-baseline --no-per-step
run cfg1_bench --config cfg1_6cam_256x704_bev128x128x8
run cfg4_bench --config cfg4_6cam_8frame_512x1408_bev200x200 --no-cpu-baseline
run neck_cfg2_bench --scope neck --no-cpu-baseline
run neck_cfg2_bench_perstep --scope neck --index-prep per_step --no-cpu-baseline
run neck_cfg2_bench_perstep_devgeom --scope neck --index-prep per_step --device-geometry --no-graph --no-cpu-baseline
run neck_cfg2_bench_perstep_devgeom_graph --scope neck --index-prep per_step --device-geometry --no-cpu-baseline
python3 tools/time_render_plan.py > $O/${TAG}_time_render_plan.txt 2>&1 || true
python3 tools/time_pool_mfma.py --group 8 > $O/${TAG}_time_pool_mfma_g8.txt 2>&1 || true
python3 tools/time_pool_mfma.py --group 2 > $O/${TAG}_time_pool_mfma_g2.txt 2>&1 || true
python3 tools/timeline_hotpath.py > $O/${TAG}_timeline_hotpath.txt 2>&1 || true
python3 tools/diag_plan_blend.py --grid 896 > $O/${TAG}_blend_stats_grid896.txt 2>&1 || true
python3 tools/time_hoa.py > $O/${TAG}_time_hoa.txt 2>&1 || true
python3 tools/time_plan_build.py 50 > $O/${TAG}_time_plan_build.txt 2>&1 || true
python3 tools/chains_r4.py > $O/${TAG}_chains.txt 2>&1 || true
python3 tools/time_pool_panel.py --group 8 --unit-cost 8 > $O/${TAG}_time_pool_panel.txt 2>&1 || true
python3 tools/host_cost_sections.py > $O/${TAG}_host_cost.txt 2>&1 || true
python3 tools/time_step_blocks.py > $O/${TAG}_step_blocks.txt 2>&1 || true
OCRF_BENCH_SINGLE_DEVICE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_2rank_gloo.log 2>&1 || true
grep '^{' $O/${TAG}_bench_2rank_gloo.log | tail -1 > $O/${TAG}_bench_2rank_gloo.json || true
python3 -m pytest tests/test_rasterize_gpu.py tests/test_full_size_gpu.py -q -s 2>&1 | grep "rasteriser parity" > $O/${TAG}_raster_parity_counts.txt || true
