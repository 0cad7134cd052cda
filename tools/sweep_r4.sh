#!/bin/bash
# round 4: the blend's share of the chip with the 95-VGPR blend; four records per trip
cd "$(dirname "$0")/.."
for bw in 704 768 832 896 960; do
  python3 tools/ab_step_knobs.py --bw $bw --steps 100 2>&1 | tail -1 | cut -c1-45,100-112,190-260
done
python3 tools/ab_step_knobs.py --bw 768 --hoa-first 1 --steps 100 2>&1 | tail -1 | cut -c1-45,100-112,190-260
export OCRF_HIP_SO=$PWD/_ab/trip4/libocrf_hip.so
echo "== trip4"
python3 tools/chains_r4.py 2>&1 | grep "render chain alone bw=0"
for bw in 768 896; do
  python3 tools/ab_step_knobs.py --bw $bw --steps 100 2>&1 | tail -1 | cut -c1-45,100-112,190-260
done
