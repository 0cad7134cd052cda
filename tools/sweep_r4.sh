#!/bin/bash
# round 4: wave priority of the main-chain kernels (A/B in one call, two passes)
cd "$(dirname "$0")/.."
for pass in 1 2; do
for v in base prio_hoa prio_hoa1 prio_all; do
  if [ $v = base ]; then unset OCRF_HIP_SO; else export OCRF_HIP_SO=$PWD/_ab/$v/libocrf_hip.so; fi
  for bw in 896 1024; do
    echo -n "$v bw=$bw: "; python3 tools/ab_step_knobs.py --bw $bw --steps 200 2>&1 | tail -1 | cut -c1-45,190-260
  done
done
done
