#!/bin/bash
# One GPU-box session producing everything kept under profiles/ for round 6:
#   tools/final_round_r6.sh [tag]     -> gpurun_out/<tag>_*, gpurun_out/prof_<tag>/
set -eo pipefail
TAG=${1:-r6}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=gpurun_out
mkdir -p $O
run() { name=$1; shift; echo "[final] $name: bench.py $*"; date
        timeout -k 10 400 python3 bench.py "$@" > $O/${TAG}_$name.log 2>&1 || true
        grep '^{' $O/${TAG}_$name.log | tail -1 > $O/${TAG}_$name.json; cut -c1-300 $O/${TAG}_$name.json; }
PART=${PART:-0}
if [ $PART != 2 ]; then
# counters first: bench.py reads profiles/<tag>_pmc_mean.csv (valid only for the sources it was collected from)
STEPS=20 bash tools/collect_profiles.sh $TAG --no-per-step --blocks 1
cp $O/prof_$TAG/pmc_mean.csv profiles/${TAG}_pmc_mean.csv
cp $O/prof_$TAG/pmc_raw.csv profiles/${TAG}_pmc_raw.csv
cp $O/prof_$TAG/pmc_meta.json profiles/${TAG}_pmc_meta.json
run cfg2_bench
run cfg2_bench_rep1 --no-cpu-baseline --no-per-step
run cfg2_bench_per_call --render-mode per_call --no-cpu-baseline --no-per-step
run cfg2_bench_guard_device --render-guard device --no-cpu-baseline --no-per-step
run cfg2_bench_objects --gaussians objects --no-cpu-baseline --no-per-step --no-gaussian-sets
run cfg1_bench --config cfg1_6cam_256x704_bev128x128x8
run cfg4_bench --config cfg4_6cam_8frame_512x1408_bev200x200 --no-cpu-baseline
run neck_cfg2_bench --scope neck --no-cpu-baseline
run neck_cfg2_bench_perstep --scope neck --index-prep per_step --no-cpu-baseline
run neck_cfg2_bench_perstep_eager --scope neck --index-prep per_step --no-graph --no-cpu-baseline
run neck_cfg2_bench_perstep_hostgeom --scope neck --index-prep per_step --host-geometry --no-cpu-baseline
fi
if [ $PART = 1 ]; then echo '[final] part 1 done'; exit 0; fi
t() { name=$1; shift; echo "[final] $name"; date; timeout -k 10 300 python3 "$@" 2>&1 | grep -v "^/opt/amdgpu" > $O/${TAG}_$name.txt || true; }
t time_blend tools/time_blend_r5.py
t chains tools/chains_r5.py
t step_blocks tools/step_blocks_r5.py
t gauss_sets tools/gauss_sets.py --steps 20
t modes_objects tools/time_modes_sets.py objects
t modes_stress tools/time_modes_sets.py stress
t blend_stats_grid704 tools/diag_plan_blend.py --fuse 1 --grid 704
t time_hoa tools/time_hoa.py
t time_plan_build tools/time_plan_build.py 50
t time_pool_panel tools/time_pool_panel.py --group 8 --unit-cost 8
t time_render_plan tools/time_render_plan.py
t time_render_per_call tools/time_render.py
t time_index_prep tools/time_index_prep.py
t modes tools/modes_r5.py neck
t time_heads_train tools/time_heads_train.py
t neck_train_cfg2_profile tools/time_neck_train.py --profile
t neck_train_cfg2_layers tools/time_neck_train.py --layers
t neck_train_cfg2_hostgeom tools/time_neck_train.py --host-geometry
t neck_train_cfg2_sources tools/prof_neck_train_sources.py
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/${TAG}_trace_step -o x -- python3 $ROOT/tools/_steps_only.py > $ROOT/$O/${TAG}_trace_step.log 2>&1 ) || true
python3 tools/step_timeline.py $O/${TAG}_trace_step/x_kernel_trace.csv > $O/${TAG}_timeline_hotpath.txt 2>&1 || true
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/${TAG}_trace_neck -o x -- python3 $ROOT/tools/_neck_only.py graph 30 > $ROOT/$O/${TAG}_trace_neck.log 2>&1 ) || true
python3 tools/step_timeline.py $O/${TAG}_trace_neck/x_kernel_trace.csv neck_prefilter 3 > $O/${TAG}_timeline_neck.txt 2>&1 || true
cp $O/${TAG}_trace_neck/x_kernel_stats.csv $O/${TAG}_neck_cfg2_kernel_stats.csv 2>/dev/null || true
OCRF_BENCH_SINGLE_DEVICE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_2rank_gloo.log 2>&1 || true
grep '^{' $O/${TAG}_bench_2rank_gloo.log | tail -1 > $O/${TAG}_bench_2rank_gloo.json || true
python3 -m pytest tests/test_rasterize_gpu.py tests/test_full_size_gpu.py tests/test_raster_plan_gpu.py -q -s -m gpu 2>&1 | grep "rasteriser parity" > $O/${TAG}_raster_parity_counts.txt || true
echo "[final] done"; date
