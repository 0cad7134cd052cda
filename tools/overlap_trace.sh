#!/bin/bash
# Kernel timeline of one OVERLAPPED step (two HIP streams) of the bench, from rocprofv3 --kernel-trace:
#   tools/overlap_trace.sh <tag> [bench.py flags ...]   -> gpurun_out/overlap_<tag>/step.txt
set -eo pipefail
TAG=${1:?tag}
shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/overlap_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o x -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --blocks 2 --no-cpu-baseline --no-per-step "$@" > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 - "$F" "$OUT" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
idx = [i for i, r in enumerate(rows) if 'bev_pool_tile_kernel' in r['Kernel_Name']]
steps = []
for a, b in zip(idx, idx[1:]):
    seg = rows[a:b]
    if len({r['Queue_Id'] for r in seg}) < 2 or len(seg) > 20:
        continue
    t0 = int(seg[0]['Start_Timestamp'])
    steps.append((max(int(r['End_Timestamp']) for r in seg) - t0, seg))
steps.sort(key=lambda s: s[0])
print(len(steps), 'two-queue steps; spans us: min %.1f median %.1f max %.1f' % (steps[0][0] / 1e3, steps[len(steps) // 2][0] / 1e3, steps[-1][0] / 1e3))
span, seg = steps[len(steps) // 2]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    print('%8.1f -> %8.1f  (%6.1f)  q%s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], nm(r)))
PY
cp "$F" "$OUT/kernel_trace.csv"
rm -rf "$OUT/trace"
