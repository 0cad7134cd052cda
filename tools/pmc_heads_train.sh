#!/bin/bash
# PMC passes over ocrf_gauss_heads_backward alone (tools/time_heads_train.py): what its waves wait for.
#   tools/pmc_heads_train.sh <tag>
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_heads_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_WAVES" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" \
             "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc_$i" -o x -- python3 $ROOT/tools/time_heads_train.py > "$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($GROUP) failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'heads_backward' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    v = v[2:] or v
    print('%-32s %16.0f  (%d launches)' % (k, sum(v) / len(v), len(v)))
PY
