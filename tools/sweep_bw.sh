set -e
cd "$(dirname "$0")/.."
for rep in 1 2; do
for bw in 704 768 800 832 864; do
  timeout -k 10 150 python tools/ab_step_knobs.py --bw $bw 2>&1 | grep median | sed 's/defaults.*bw=/bw=/; s/fuse_out.*: median/median/'
done
done
