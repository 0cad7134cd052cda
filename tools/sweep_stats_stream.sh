set -e
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  timeout -k 10 150 python tools/ab_step_knobs.py 2>&1 | grep median | sed 's/defaults.*stats_stream/stats_stream/'
  timeout -k 10 150 python tools/ab_step_knobs.py --stats-stream render 2>&1 | grep median | sed 's/defaults.*stats_stream/stats_stream/'
  timeout -k 10 150 python tools/ab_step_knobs.py --stats-stream own 2>&1 | grep median | sed 's/defaults.*stats_stream/stats_stream/'
done
