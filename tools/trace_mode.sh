#!/bin/bash
# Kernel timeline of one step of a HotPath mode from rocprofv3 --kernel-trace (csv):
#   tools/trace_mode.sh <mode of tools/_steps_only.py> <marker kernel substring>   -> gpurun_out/trace_<mode>/step.txt
set -eo pipefail
MODE=${1:?mode}; MARK=${2:?marker}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/trace_$MODE
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o x -- python3 "$ROOT/tools/_steps_only.py" 80 "$MODE" > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 - "$F" "$MARK" > "$OUT/step.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2]
nm = lambda r: r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
idx = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
idx = idx[len(idx) // 2:]
steps = []
for a, b in zip(idx, idx[1:]):
    seg = rows[a:b]
    t0 = int(seg[0]['Start_Timestamp'])
    steps.append((int(rows[b]['Start_Timestamp']) - t0, seg))
steps.sort(key=lambda s: s[0])
print(len(steps), 'steps; period us: min %.1f median %.1f max %.1f' % (steps[0][0] / 1e3, steps[len(steps) // 2][0] / 1e3, steps[-1][0] / 1e3))
span, seg = steps[len(steps) // 2]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    running = sum(1 for q in seg if int(q['Start_Timestamp']) <= s < int(q['End_Timestamp'])) - 1
    print('%8.1f -> %8.1f  (%6.1f)  q%-3s ||%d  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], running, nm(r)))
PY
cat "$OUT/step.txt"
