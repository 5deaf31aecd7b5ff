"""Where does the wall time of one training step go?  Reads a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv), takes
the LAST step (delimited by the optimiser launches), and prints per queue: busy time, time alone, overlapped time, idle
gaps, plus the largest gaps on the queue that carries the forward pass.
usage: python tools/timeline.py <kernel_trace.csv> [optimiser-kernel-substring] [print the last N launches] [kernel-name substring: print the launches around two of its occurrences]"""
import csv
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    path = sys.argv[1]
    opt = sys.argv[2] if len(sys.argv) > 2 else 'sgd'
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if opt in r['Kernel_Name'].lower()]
    assert len(marks) >= 2, 'need two optimiser launches (%r) to delimit a step' % opt
    step = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = int(rows[marks[-2]]['End_Timestamp'])
    t1 = int(step[-1]['End_Timestamp'])
    print('step: %d launches, wall %.3f ms' % (len(step), (t1 - t0) / 1e6))
    q = defaultdict(list)
    for r in step:
        q[(r.get('Queue_Id'), r.get('Stream_Id'))].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    un = {k: union([(a, b) for a, b, _ in v]) for k, v in q.items()}
    allbusy = union([(a, b) for v in q.values() for a, b, _ in v])
    print('any queue busy %.3f ms, all idle %.3f ms' % (length(allbusy) / 1e6, (t1 - t0 - length(allbusy)) / 1e6))
    keys = sorted(q, key=lambda k: -length(un[k]))
    for k in keys:
        others = union([iv for k2 in keys if k2 != k for iv in un[k2]])
        ov = length(intersect(un[k], others))
        ks = sum(b - a for a, b, _ in q[k])
        print('queue %s: %4d launches, kernel time %.3f ms, busy %.3f ms (alone %.3f, beside another queue %.3f)'
              % (k, len(q[k]), ks / 1e6, length(un[k]) / 1e6, (length(un[k]) - ov) / 1e6, ov / 1e6))
    # the matrix pipe: when is at least one conv kernel (gather / wgrad) running, when are two, when none -- forward (up to the
    # loss kernel) and backward apart
    conv = lambda n: ('gather_' in n or 'wgrad_' in n) and 'slab' not in n
    loss = [int(r['Start_Timestamp']) for r in step if 'ce_up' in r['Kernel_Name'] or 'cross_entropy' in r['Kernel_Name']]
    tl = loss[0] if loss else t0
    for name, lo, hi in (('forward', t0, tl), ('backward', tl, t1)):
        per_q = {k: union([(max(a, lo), min(b, hi)) for a, b, n in v if conv(n) and b > lo and a < hi]) for k, v in q.items()}
        qs = [k for k in per_q if per_q[k]]
        any_c = union([iv for k in qs for iv in per_q[k]])
        two = []
        for i, k in enumerate(qs):
            for k2 in qs[i + 1:]:
                two += intersect(per_q[k], per_q[k2])
        two = union(two)
        ksum = sum(min(b, hi) - max(a, lo) for v in q.values() for a, b, n in v if conv(n) and b > lo and a < hi)
        print('%s %.3f ms: a conv kernel running %.3f ms (two queues at once %.3f), none %.3f; conv kernel time %.3f ms'
              % (name, (hi - lo) / 1e6, length(any_c) / 1e6, length(two) / 1e6, (hi - lo - length(any_c)) / 1e6, ksum / 1e6))
    main_q = keys[0]
    v = sorted(q[main_q])
    gaps = []
    for (a0, b0, n0), (a1, b1, n1) in zip(v, v[1:]):
        if a1 > b0:
            gaps.append((a1 - b0, n0[:60], n1[:60], (b0 - t0) / 1e6))
    print('gaps on %s: %d, total %.3f ms; median %.1f us' % (main_q, len(gaps), sum(g[0] for g in gaps) / 1e6,
                                                            sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3))
    for g in sorted(gaps, reverse=True)[:12]:
        print('  %.1f us at %.2f ms: %s -> %s' % (g[0] / 1e3, g[3], g[1], g[2]))
    tail = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if tail:
        print('last %d launches (queue, start ms, duration us, kernel):' % tail)
        for r in step[-tail:]:
            print('  q%s %8.3f %8.1f  %s' % (r.get('Queue_Id'), (int(r['Start_Timestamp']) - t0) / 1e6,
                                           (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:100]))
    focus = sys.argv[4] if len(sys.argv) > 4 else ''
    if focus:
        idx = [i for i, r in enumerate(step) if focus in r['Kernel_Name']]
        for j in idx[len(idx) // 2:len(idx) // 2 + 2]:      # two occurrences from the middle of the step
            print('around %r (launch %d of the step):' % (focus, j))
            for r in step[max(0, j - 6):j + 7]:
                print('  q%s %8.3f -> %8.3f (%7.1f us)  %s' % (r.get('Queue_Id'), (int(r['Start_Timestamp']) - t0) / 1e6,
                      (int(r['End_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                      r['Kernel_Name'][:90]))
    # time profile in 1 ms bins: how much of each bin each queue is busy
    nb = int((t1 - t0) / 1e6) + 1
    print('per-ms occupancy (queues in the order above):')
    for bi in range(nb):
        lo, hi = t0 + bi * 1e6, t0 + (bi + 1) * 1e6
        print('  %2d ms: %s' % (bi, '  '.join('%3d%%' % (100 * length(intersect(un[k], [[lo, hi]])) / 1e6) for k in keys)))


if __name__ == '__main__':
    main()
