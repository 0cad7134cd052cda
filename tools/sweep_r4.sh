#!/bin/bash
# round 4: which stream carries the render chain, where HOA-1/2 go, the blend's share of the chip
cd "$(dirname "$0")/.."
for caller in 1 0; do
  for hf in 1 0; do
    for bw in auto 640; do
      python3 tools/ab_step_knobs.py --bw $bw --hoa-first $hf --caller $caller --steps 100 2>&1 | tail -1
    done
  done
done
