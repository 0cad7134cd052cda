# A/B of the cfg2 step: pooling back ends (tile LSS + MFMA HT = round 4's default; panel = bev_pool_panel.hip)
set -e
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_bev_pool_panel_gpu.py -q 2>&1 | tail -2
timeout -k 10 120 python tools/time_pool_panel.py --group 8 --unit-cost 8 | grep -v amdgpu.ids
for rep in 1 2; do
  timeout -k 10 150 python tools/ab_step_knobs.py --ht mfma --lss tile 2>&1 | grep median
  timeout -k 10 150 python tools/ab_step_knobs.py --ht panel --lss panel 2>&1 | grep median
done
timeout -k 10 150 python tools/ab_step_knobs.py --ht panel --lss tile 2>&1 | grep median
timeout -k 10 150 python tools/ab_step_knobs.py --ht panel --lss panel --bw 832 2>&1 | grep median
timeout -k 10 150 python tools/ab_step_knobs.py --ht panel --lss panel --bw 960 2>&1 | grep median
