"""Prints the kernels of ONE step (start offset, duration, queue) from a rocprofv3 --kernel-trace CSV:
    python tools/trace_step.py <x_kernel_trace.csv> [anchor kernel substring = lss_keys] [which occurrence = 10]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'lss_keys'
which = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def nm(r):
    return r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]


idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    print('%8.1f %7.1f  q%s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3,
                                   (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), nm(r)))
