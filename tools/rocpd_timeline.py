#!/usr/bin/env python
"""Timeline view of a rocprofv3 --kernel-trace database (rocpd SQLite): for the steady-state part of a bench run (the last
`frac` of the dispatches) how much of the wall time the GPU ran 0 / 1 / >= 2 kernels at once, per HIP stream (queue) busy
time, and the longest stretches with nothing running, each with the kernel that ended before and the one that started after.

    python tools/rocpd_timeline.py <db> [frac=0.5] [gaps=25]
"""
import sqlite3
import sys


def main(path, frac=0.5, ngaps=25):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = cur.execute("select start, end, %s, %s from kernels order by start" % (name_col, qcol or '0')).fetchall()
    rows = rows[int(len(rows) * (1.0 - frac)):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, {}
    for t, d in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        last = t
        depth += d
    wall = float(t1 - t0)
    print('window: %d dispatches, %.3f ms wall, sum of kernel durations %.3f ms' % (len(rows), wall / 1e6, sum(r[1] - r[0] for r in rows) / 1e6))
    for k in sorted(hist):
        print('  %s kernels running: %8.3f ms  %5.1f %%' % ('>=3' if k == 3 else ' %d ' % k, hist[k] / 1e6, 100.0 * hist[k] / wall))
    if qcol:
        per = {}
        for s, e, _, q in rows:
            n, t = per.get(q, (0, 0))
            per[q] = (n + 1, t + e - s)
        print('per %s: ' % qcol + ', '.join('%s: %d launches %.2f ms' % (q, n, t / 1e6) for q, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])))
    # idle gaps
    gaps = []
    cur_end, cur_name = rows[0][1], rows[0][2]
    for s, e, n, _ in rows[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    tot = sum(g[0] for g in gaps)
    print('idle: %d gaps, %.3f ms in total; by the kernel that FOLLOWS the gap:' % (len(gaps), tot / 1e6))
    by = {}
    for g, before, after in gaps:
        k = after.split('(')[0][:60]
        n, t = by.get(k, (0, 0))
        by[k] = (n + 1, t + g)
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:ngaps]:
        print('  %-62s %5d gaps %8.3f ms (avg %.1f us)' % (k, n, t / 1e6, t / n / 1e3))


def one_step(path, marker='preprocess_kernel', min_gap_us=15.0):
    """The idle gaps of the LAST complete step (from one `marker` kernel to the next), in launch order."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select start, end, name from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
    i0, i1 = marks[-2], marks[-1]
    t0 = rows[i0][0]
    print('last complete step: %d dispatches, %.3f ms' % (i1 - i0, (rows[i1][0] - t0) / 1e6))
    cur_end, cur_name = rows[i0][1], rows[i0][2]
    for s, e, n in rows[i0 + 1:i1 + 1]:
        if s - cur_end > min_gap_us * 1e3:
            print('  t=%8.3f ms  gap %7.1f us   after %-50s before %s' % ((cur_end - t0) / 1e6, (s - cur_end) / 1e3, cur_name.split('(')[0][:50], n.split('(')[0][:60]))
        if e > cur_end:
            cur_end, cur_name = e, n


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == 'step':
        one_step(sys.argv[1])
        sys.exit(0)
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, int(sys.argv[3]) if len(sys.argv) > 3 else 25)
