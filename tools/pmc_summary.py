"""Summarise rocprofv3 --pmc CSV passes (one directory per counter group) per kernel: mean counter
value per launch.  usage: python tools/pmc_summary.py <dir> [kernel-name-substring ...]"""
import csv, glob, collections, sys
root = sys.argv[1]
filt = sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/*/x_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        short = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if filt and not any(s in short for s in filt):
            continue
        res[short][row['Counter_Name']].append(float(row['Counter_Value']))
names = sorted({c for v in res.values() for c in v})
print('%-46s' % 'kernel' + ''.join('%14s' % n[:13] for n in names))
for k, v in sorted(res.items()):
    print('%-46s' % k[:45] + ''.join('%14.4g' % (sum(v[n]) / len(v[n]) if n in v else float('nan')) for n in names))
