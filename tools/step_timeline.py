"""Diagnostic: timeline of ONE overlapped step from a rocprofv3 --kernel-trace csv (start offset, duration, stream of
every kernel between two successive launches of the step's first kernel).
    rocprofv3 --kernel-trace --output-format csv -d OUT -o x -- python3 tools/_steps_only.py
    python tools/step_timeline.py OUT/x_kernel_trace.csv [first_kernel_substring] [which step from the end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else 'raster_plan_head_kernel'
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
i0, i1 = marks[-back - 1], marks[-back]
t0 = int(rows[i0]['Start_Timestamp'])
print('step of %d kernels, %.1f us from its first start to the next step\'s first start' % (
    i1 - i0, (int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    print('%8.1f -> %8.1f  (%6.1f us)  q%-3s  %s' % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
