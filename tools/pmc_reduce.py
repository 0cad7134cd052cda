"""Reduces what tools/collect_profiles.sh left under gpurun_out/prof_<tag>/ to three small CSV files kept
under profiles/ (the raw numbers bench.py's `roofline` block is recomputed from):

  kernel_stats_serial.csv / kernel_stats_overlap.csv   rocprofv3 --stats per kernel (calls, total, mean, min, max; ns)
  pmc_raw.csv     rows of rocprofv3's counter_collection.csv (dispatch, counter, value) for THIS library's kernels
                  (torch / MIOpen / rocprim kernels dropped, argument lists cut off), the first RAW_KEEP dispatches
                  of every (kernel, counter) — the means below are over ALL dispatches
  pmc_mean.csv    mean / min / max per (kernel, counter) + launches

    python tools/pmc_reduce.py gpurun_out/prof_<tag>
"""
import collections
import csv
import glob
import os
import sys

RAW_KEEP = 8
OURS = ('raster_', 'bev_pool', 'hoa', 'neck_', 'lss_', 'ht_', 'radix_', 'scan_', 'lower_bound', 'zero_words', 'geom_',
        'camera_')


def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0].strip()


def ours(name):
    s = short(name)
    return any(s.startswith(p) or ('::' + p) in s for p in OURS)


def stats(root, sub, out):
    files = glob.glob(os.path.join(root, sub, '**', '*kernel_stats.csv'), recursive=True)
    if not files:
        return
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append(r)
    with open(out, 'w', newline='') as fo:
        w = csv.writer(fo)
        w.writerow(['kernel', 'calls', 'total_ns', 'mean_ns', 'min_ns', 'max_ns', 'percent'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['MinNs'], r['MaxNs'],
                        r['Percentage']])
    print('wrote', out, len(rows), 'kernels')


def main():
    root = sys.argv[1]
    stats(root, 'trace_serial', os.path.join(root, 'kernel_stats_serial.csv'))
    stats(root, 'trace_overlap', os.path.join(root, 'kernel_stats_overlap.csv'))
    acc = collections.defaultdict(list)
    n_raw = 0
    with open(os.path.join(root, 'pmc_raw.csv'), 'w', newline='') as fo:
        w = csv.writer(fo)
        w.writerow(['pass', 'dispatch_id', 'kernel', 'grid', 'workgroup', 'lds_bytes', 'vgpr', 'sgpr', 'counter', 'value'])
        for d in sorted(glob.glob(os.path.join(root, 'pmc_*'))):
            if not os.path.isdir(d):
                continue
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if not ours(r['Kernel_Name']):
                        continue
                    k = short(r['Kernel_Name'])
                    v = float(r['Counter_Value'])
                    acc[(k, r['Counter_Name'])].append(v)
                    if len(acc[(k, r['Counter_Name'])]) > RAW_KEEP:
                        continue
                    w.writerow([os.path.basename(d), r.get('Dispatch_Id', ''), k, r.get('Grid_Size', ''),
                                r.get('Workgroup_Size', ''), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''),
                                r.get('SGPR_Count', ''), r['Counter_Name'], r['Counter_Value']])
                    n_raw += 1
    with open(os.path.join(root, 'pmc_mean.csv'), 'w', newline='') as fo:
        w = csv.writer(fo)
        w.writerow(['kernel', 'counter', 'launches', 'mean', 'min', 'max'])
        for (k, c), v in sorted(acc.items()):
            w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
    print('pmc rows kept:', n_raw, 'kernel x counter pairs:', len(acc))


if __name__ == '__main__':
    main()
