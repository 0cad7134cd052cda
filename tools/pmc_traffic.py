"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output) into the
per-kernel HBM traffic record kept under profiles/ (mean per dispatch; gfx950 correction of
MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced read stream).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<note>"
"""
import collections
import csv
import glob
import json
import sys


def read(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
            acc[k].append(float(row['Counter_Value']))
    return acc


def main():
    fetch, write = read(sys.argv[1], 'FETCH_SIZE'), read(sys.argv[2], 'WRITE_SIZE')
    out = {'_note': sys.argv[4]}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
        w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        out[k] = {'launches': max(len(fetch.get(k, [])), len(write.get(k, []))), 'fetch_kib_raw': round(f, 1),
                  'write_kib': round(w, 1), 'hbm_bytes_corrected': int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_corrected'] if isinstance(kv[1], dict) else 0)[:25]:
        if isinstance(v, dict):
            print('%-50s %8.2f MB per launch (%d launches)' % (k[:50], v['hbm_bytes_corrected'] / 1e6, v['launches']))


if __name__ == '__main__':
    main()
