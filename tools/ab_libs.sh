#!/bin/bash
# A/B of library builds on the cfg2 step in ONE box session: tools/ab_libs.sh <label>:<so or -> ... (env ARGS: time_step.py flags)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for spec in "$@"; do
    IFS=: read -r label so <<< "$spec"
    if [ "$so" = "-" ]; then unset OCRF_HIP_SO; else export OCRF_HIP_SO=$PWD/$so; fi
    echo -n "$label rep$rep: "; timeout -k 10 200 python tools/time_step.py $ARGS 2>&1 | grep "step median"
  done
done
