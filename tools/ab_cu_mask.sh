#!/bin/bash
# every variant of tools/ab_cu_mask.py in a process of its own -> gpurun_out/ab_cu_mask.log
mkdir -p gpurun_out
out=gpurun_out/ab_cu_mask.log
: > $out
python tools/ab_cu_mask.py --variant where >> $out 2>/dev/null
for v in plain prio_render prio_main split:224 split:208 split:192 split:176 split:160 split:144 split:128 split:96 \
         split:0:64 split:0:96 split:0:128 split:192:0 split:224:0 split:224:64 split:208:96 split:192:128 plain; do
  python tools/ab_cu_mask.py --variant $v >> $out 2>/dev/null || echo "$v failed" >> $out
done
cat $out
