#!/bin/bash
# bench.py's timed region vs tools/ab_step_knobs.py in one call
cd "$(dirname "$0")/.."
for i in 1 2; do
  timeout -k 5 200 python3 bench.py --no-cpu-baseline --no-per-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 200x5', d['ms_per_step_blocks'])"
  timeout -k 5 200 python3 bench.py --no-cpu-baseline --no-per-step --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20x5 ', d['ms_per_step_blocks'])"
  timeout -k 5 120 python3 tools/ab_step_knobs.py --steps 200 2>&1 | tail -1 | cut -c1-45,190-260
done
