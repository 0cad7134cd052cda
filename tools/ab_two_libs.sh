# A/B of two builds of the library in ONE box session: tools/ab_two_libs.sh <other libocrf_hip.so> [ab_step_knobs flags]
set -e
cd "$(dirname "$0")/.."
OTHER=$1; shift || true
for rep in 1 2 3; do
  echo -n "in-tree : "; timeout -k 10 150 python tools/ab_step_knobs.py "$@" 2>&1 | grep median | sed 's/defaults.*: median/median/'
  echo -n "other   : "; OCRF_HIP_SO=$OTHER timeout -k 10 150 python tools/ab_step_knobs.py "$@" 2>&1 | grep median | sed 's/defaults.*: median/median/'
done
