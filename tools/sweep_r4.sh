#!/bin/bash
# round 4: two free-running chains vs two phases (pools alone, then blend beside HOA)
cd "$(dirname "$0")/.."
python3 tools/ab_step_knobs.py --schedule overlap --steps 100 2>&1 | tail -1
for bw in auto 640 896 1024 0; do
  for hs in 0 1; do
    python3 tools/ab_step_knobs.py --schedule phased --bw $bw --hoa-stream $hs --steps 100 2>&1 | tail -1
  done
done
