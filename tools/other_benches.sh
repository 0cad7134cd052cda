#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $O/r4pre_$name.log 2>&1; grep '^{' $O/r4pre_$name.log | tail -1 > $O/r4pre_$name.json; python3 -c "
import json,sys
d=json.loads(open('$O/r4pre_$name.json').read()); print('$name', d.get('ms_per_step'), d.get('per_step_ms'), d.get('per_sample_ms'))" || tail -5 $O/r4pre_$name.log; }
run cfg1 --config cfg1_6cam_256x704_bev128x128x8 --no-cpu-baseline --steps 20 --warmup 5
run cfg4 --config cfg4_6cam_8frame_512x1408_bev200x200 --no-cpu-baseline --steps 20 --warmup 5
run neck --scope neck --no-cpu-baseline --steps 20 --warmup 5
run neck_perstep --scope neck --index-prep per_step --no-cpu-baseline --steps 20 --warmup 5
run cfg2_guard --render-guard device --no-cpu-baseline --no-per-step --steps 20 --warmup 5
run cfg2_percall --render-mode per_call --no-cpu-baseline --no-per-step --steps 20 --warmup 5
