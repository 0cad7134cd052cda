set -e
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_raster_plan_gpu.py tests/test_full_size_gpu.py -q -m gpu 2>&1 | tail -2
timeout -k 10 100 python tools/diag_plan_blend.py --grid 896 2>&1 | grep -v amdgpu | tail -9
timeout -k 10 100 python tools/chains_r4.py 2>&1 | grep -v amdgpu | head -7
for rep in 1 2; do timeout -k 10 150 python tools/ab_step_knobs.py 2>&1 | grep median | sed 's/defaults.*issue=None//'; done
