#!/bin/bash
# Quick look on the GPU box: serial rocprofv3 kernel stats of the bench step + the default bench line.
#   tools/quick_trace.sh <tag> [bench.py flags ...]
set -eo pipefail
TAG=${1:?tag}
shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/quick_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -o x -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-overlap "$@" > "$OUT/trace_serial.log" 2>&1
F=$(find "$OUT/trace_serial" -name '*kernel_stats.csv' | head -1)
cp "$F" "$OUT/kernel_stats_serial.csv"
python3 - "$OUT/kernel_stats_serial.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print('%-90s calls %5s  avg %8.1f us  total %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
"
