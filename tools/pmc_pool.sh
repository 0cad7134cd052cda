#!/bin/bash
# PMC passes over the pooling kernel alone (tools/_pools_only.py): where the memory pipe stalls.
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_pool_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_LFIFO_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc_$i" -o x -- python3 $ROOT/tools/_pools_only.py ${2:-4} > "$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($GROUP) failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bev_pool_tile' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    # launches alternate lss, ht
    lss, ht = v[0::2], v[1::2]
    print('%-42s lss %14.0f   ht %14.0f' % (k, sum(lss) / max(len(lss), 1), sum(ht) / max(len(ht), 1)))
PY
