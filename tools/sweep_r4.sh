#!/bin/bash
# round 4: the update kernel without its conservative early-out
cd "$(dirname "$0")/.."
for pass in 1 2; do
for v in base noearly; do
  if [ $v = base ]; then unset OCRF_HIP_SO; else export OCRF_HIP_SO=$PWD/_ab/$v/libocrf_hip.so; fi
  echo -n "$v: "; python3 tools/time_render_plan.py 2>&1 | grep "planned chain"
done
done
