set -e
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for o in render_first lss_first pools_first; do
    timeout -k 10 150 python tools/ab_step_knobs.py --issue $o 2>&1 | grep median | sed 's/defaults.*stats_stream=None//'
  done
done
for bw in 960 1024; do timeout -k 10 150 python tools/ab_step_knobs.py --issue lss_first --bw $bw 2>&1 | grep median | sed 's/defaults.*bw=/bw=/'; done
