#!/bin/bash
# Collects, on the GPU box, everything bench.py's roofline block is recomputed from, as CSV:
#   tools/collect_profiles.sh <tag> [bench.py flags ...]
#     1. rocprofv3 --kernel-trace --stats of `python3 bench.py --no-overlap ...`  (every kernel alone)
#        and of the default (renders on the side stream) run;
#     2. one rocprofv3 --pmc pass per counter group (separate runs; --pmc is never combined with a trace
#        domain other than the kernel trace) of the --no-overlap run: HBM bytes (FETCH_SIZE, WRITE_SIZE),
#        the SQ instruction / cycle counters of the blend roofline, LDS counters, GRBM_GUI_ACTIVE (clock);
#     3. tools/pmc_reduce.py -> gpurun_out/prof_<tag>/{kernel_stats*.csv, pmc_raw.csv, pmc_mean.csv}
# Copy the three reduced files into profiles/ (named per round) to have them judged.
set -eo pipefail
TAG=${1:?tag}
shift || true
EXTRA="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-20}
B="$ROOT/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-pool-backends $EXTRA"

echo "[collect] kernel trace, serial" && date
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_serial" -o x -- python3 $B --no-overlap > "$OUT/trace_serial.log" 2>&1
echo "[collect] kernel trace, overlapped" && date
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_overlap" -o x -- python3 $B > "$OUT/trace_overlap.log" 2>&1

i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  echo "[collect] pmc pass $i: $GROUP" && date
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc_$i" -o x -- python3 $B --no-overlap > "$OUT/pmc_$i.log" 2>&1 \
    || echo "[collect] pmc pass $i FAILED (see pmc_$i.log)"
done
python3 "$ROOT/tools/pmc_reduce.py" "$OUT"
# what these passes ran: bench.py uses the counters only for the same config / render mode / library sources
python3 - "$OUT" $EXTRA <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), '..'))
root = os.path.abspath(os.path.join(sys.argv[1], '..', '..'))
sys.path.insert(0, root)
import bench
flags = sys.argv[2:]
def opt(name, default):
    return flags[flags.index(name) + 1] if name in flags else default
meta = dict(config=opt('--config', bench.DEFAULT_CONFIG), render_mode=opt('--render-mode', 'planned'),
            source_hash=bench.source_hash(), flags=' '.join(flags), note='rocprofv3 --pmc passes of bench.py --no-overlap')
json.dump(meta, open(os.path.join(sys.argv[1], 'pmc_meta.json'), 'w'), indent=1)
print('wrote pmc_meta.json', meta)
PY
echo "[collect] done" && date
