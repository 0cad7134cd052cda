#!/bin/bash
# round 4: the blend's share of the chip and where HOA-1/2 go, with the shorter main chain
cd "$(dirname "$0")/.."
for bw in auto 640 768 1024 0; do
  for hs in "0 0" "1 0" "0 1"; do
    set -- $hs
    python3 tools/ab_step_knobs.py --bw $bw --hoa-first $1 --hoa-stream $2 --steps 100 2>&1 | tail -1
  done
done
