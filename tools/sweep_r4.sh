#!/bin/bash
# round 4: HOA-2's output conv folded into the HOA-3 gate, A/B in one call (two passes)
cd "$(dirname "$0")/.."
run() { timeout -k 5 120 python3 tools/ab_step_knobs.py "$@" --steps 200 2>&1 | tail -1 | cut -c1-45,100-125,200-270; }
for pass in 1 2 3; do
  for fo in 1 0; do
    for bw in 896 1024; do run --fuse-out $fo --bw $bw; done
  done
done
python3 tools/time_hoa.py 2>&1 | grep -v amdgpu
