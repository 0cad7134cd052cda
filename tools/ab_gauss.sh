#!/bin/bash
# A/B of library variants / bin shapes on the three Gaussian sets in ONE box session:
#   tools/ab_gauss.sh "<label>:<so or ->:<bins>" ...      e.g.  "tree:-:4x2" "scan1:_ab/scan1/libocrf_hip.so:4x2" "nobins:-:none"
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    IFS=: read -r label so bins <<< "$spec"
    if [ "$so" = "-" ]; then unset OCRF_HIP_SO; else export OCRF_HIP_SO=$PWD/$so; fi
    timeout -k 10 300 python tools/gauss_sets.py --steps 20 --bins "$bins" --sets "${SETS:-init,objects}" 2>&1 | grep -E "^(init|stress|objects) " | \
      python -c "
import sys, json
for l in sys.stdin:
    k, j = l.split(' ', 1); r = json.loads(j)
    print('$label rep$rep', k, 'step %.4f ms  blend in-step %.1f alone %.1f us  tile pair %.1f us  shares %s  scanned %.0f staged %.0f' % (r['ms_per_step'], r['blend_us'], r['blend_alone_us'], r['tile_pair_us'], r['phase_shares_scan_stage_blend'], r['scanned_per_tile_pair'], r['staged_per_tile_pair']), flush=True)"
  done
done
