"""Timeline of ONE step from a rocprofv3 rocpd database: the kernel dispatches between the last two
launches of a marker kernel, with start offset, duration, queue and how many other kernels were
running when each started — shows what actually overlaps inside a hipGraph replay.

    python tools/prof_db_timeline.py results.db --marker neck_prefilter_kernel [--index K] [--schema]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    marker = sys.argv[sys.argv.index('--marker') + 1] if '--marker' in sys.argv else 'neck_prefilter_kernel'
    c = sqlite3.connect(db).cursor()
    if '--schema' in sys.argv:
        for (n, t) in c.execute("select name, type from sqlite_master where type in ('table','view')"):
            print(t, n)
    c.execute('select * from kernels limit 1')
    cols = [d[0] for d in c.description]
    if '--schema' in sys.argv:
        print('kernels:', cols)
    pick = lambda *names: next(n for n in names if n in cols)
    c_name, c_start, c_end = pick('name', 'kernel_name'), pick('start', 'start_timestamp'), pick('end', 'end_timestamp')
    c_q = next((n for n in ('queue_id', 'queue', 'stream_id', 'stream') if n in cols), None)
    q = f'select {c_name}, {c_start}, {c_end}' + (f', {c_q}' if c_q else ', 0') + f' from kernels order by {c_start}'
    rows = list(c.execute(q))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3:
        sys.exit('marker kernel launched fewer than 3 times')
    k = int(sys.argv[sys.argv.index('--index') + 1]) if '--index' in sys.argv else len(marks) // 2
    lo, hi = marks[k], marks[k + 1]
    step = rows[lo:hi]
    t0 = step[0][1]
    busy = 0.0
    for i, (name, s, e, qid) in enumerate(step):
        running = sum(1 for (_, s2, e2, _) in step if s2 <= s < e2) - 1
        nm = name.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        print(f'{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  q{qid}  ||{running}  {nm}')
        busy += (e - s) / 1e3
    print(f'{len(step)} kernels, {busy:.1f} us of kernel time, span {(max(r[2] for r in step) - t0) / 1e3:.1f} us, '
          f'next step starts at {(rows[hi][1] - t0) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
