set -e
cd "$(dirname "$0")/.."
for g in 0 384 512 640 768 1024; do
  timeout -k 10 150 python tools/ab_step_knobs.py 2=$g 2>&1 | grep median | sed 's/ht=mfma.*issue=None//'
done
