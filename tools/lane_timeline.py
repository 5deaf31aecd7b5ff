"""Which hardware queue / stream did every kernel of the last steps of a rocprofv3 --kernel-trace run on, and how busy was
each?  usage: python tools/lane_timeline.py <results.db> [kernels-per-step]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 1005
    c = sqlite3.connect(db)
    cur = c.execute('select * from kernels limit 1')
    cols = [d[0] for d in cur.description]
    print('columns:', cols)
    qcol = next((k for k in ('queue_id', 'queue', 'Queue_Id') if k in cols), None)
    scol = next((k for k in ('stream_id', 'stream', 'Stream_Id') if k in cols), None)
    sel = 'start, end, name' + (', %s' % qcol if qcol else ', 0') + (', %s' % scol if scol else ', 0')
    rows = c.execute('select %s from kernels order by start' % sel).fetchall()
    rows = rows[-4 * per_step:]            # the last four steps
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print('%d kernels over %.2f ms' % (len(rows), (t1 - t0) / 1e6))
    by = {}
    for s, e, name, q, st in rows:
        by.setdefault((q, st), []).append((s, e))
    print('| queue | stream | kernels | busy ms | busy / span |')
    for key in sorted(by, key=lambda k: -len(by[k])):
        iv = sorted(by[key])
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print('| %s | %s | %d | %.2f | %.2f |' % (key[0], key[1], len(iv), busy / 1e6, busy / (t1 - t0)))
    # every stream's kernels by name
    import re
    for main in sorted(by, key=lambda k: -len(by[k])):
        names = {}
        for s, e, name, q, st in rows:
            if (q, st) == main:
                n = re.sub(r'\(.*$', '', name)[:70]
                d = names.setdefault(n, [0, 0])
                d[0] += 1
                d[1] += e - s
        print('kernels of stream %s (per step, 4 steps traced):' % (main,))
        for n, (cnt, dur) in sorted(names.items(), key=lambda kv: -kv[1][1])[:18]:
            print('  %-72s %6.1f calls %7.3f ms %6.1f us' % (n, cnt / 4, dur / 4e6, dur / cnt / 1e3))
    # union over everything: how much of the span has at least one kernel running
    iv = sorted((s, e) for s, e, *_ in rows)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('any kernel running: %.2f of the span; sum of kernel durations / span = %.2f' % (
        busy / (t1 - t0), sum(e - s for s, e, *_ in rows) / (t1 - t0)))


if __name__ == '__main__':
    main()
