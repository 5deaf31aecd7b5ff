"""Per-kernel table from a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db or *_kernel_stats.csv).
usage: python tools/prof_summary.py <results.db> <steps-in-trace> [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name if len(name) <= 88 else name[:85] + '...'


def main():
    db, steps = sys.argv[1], float(sys.argv[2])
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    print('kernel time total %.1f ms = %.2f ms/step over %g steps' % (tot / 1e6, tot / 1e6 / steps, steps))
    print('| kernel | calls/step | ms/step | avg us | % |')
    print('|---|---|---|---|---|')
    for name, n, dur, avg in rows[:40]:
        print('| `%s` | %.1f | %.2f | %.1f | %.1f |' % (short(name), n / steps, dur / 1e6 / steps, avg / 1e3, 100 * dur / tot))
    if len(sys.argv) > 3:
        with open(sys.argv[3], 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
            for name, n, dur, avg in rows:
                w.writerow([name, n, dur, '%.1f' % avg, '%.3f' % (100 * dur / tot)])


if __name__ == '__main__':
    main()
