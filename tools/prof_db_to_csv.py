"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database; this turns its `top_kernels` view into the
CSV summary kept under profiles/.   python tools/prof_db_to_csv.py results.db out.csv [--print N]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    n = int(sys.argv[sys.argv.index('--print') + 1]) if '--print' in sys.argv else 0
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute('select * from top_kernels'))
    cols = [d[0] for d in c.description]
    with open(out, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(cols + ['unit'])
        for r in rows:
            w.writerow(list(r) + ['us'])
    for r in rows[:n]:
        print('%-90s calls %5d  total %10.1f us  avg %9.2f us  %5.1f %%' % (r[0].replace('(anonymous namespace)::', '')[:90], r[1], r[2], r[3], r[4]))


if __name__ == '__main__':
    main()
