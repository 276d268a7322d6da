#!/usr/bin/env python
"""Where do the launches that compute nothing come from?  For the LAST complete step of a rocprofv3 --kernel-trace database (rocpd SQLite):
every dispatch whose kernel name matches one of the patterns (default: fill / copy / ATen element-wise kernels), with its duration, its
grid size and the two kernels launched before and the two after it -- enough to name the call site.  Then the counts per (previous kernel,
this kernel, next kernel) triple.

    python tools/rocpd_context.py <db> [pattern,pattern,...]
"""
import collections
import sqlite3
import sys


def main(path, pats):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    rows = cur.execute("select start, end, name%s from kernels order by start" % (', ' + gx if gx else ', 0')).fetchall()
    marks = [i for i, r in enumerate(rows) if r[2].startswith('preprocess_kernel')]
    i0, i1 = marks[-2], marks[-1]
    step = rows[i0:i1]
    short = lambda n: n.replace('void ', '').split('(')[0][:48]
    print('last complete step: %d dispatches' % len(step))
    trip = collections.Counter()
    tot = collections.Counter()
    for i, (s, e, n, g) in enumerate(step):
        if not any(p in n for p in pats):
            continue
        prev = [short(step[j][2]) for j in range(max(0, i - 2), i)]
        nxt = [short(step[j][2]) for j in range(i + 1, min(len(step), i + 3))]
        print('%4d %7.1f us grid %-8s %-40s | after: %s | before: %s' % (i, (e - s) / 1e3, g, short(n), ' <- '.join(reversed(prev)), ' -> '.join(nxt)))
        trip[(prev[-1] if prev else '', short(n), nxt[0] if nxt else '')] += 1
        tot[short(n)] += (e - s) / 1e3
    print()
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print('%-48s %8.1f us per step' % (k, v))
    print()
    for (a, b, c), n in sorted(trip.items(), key=lambda kv: -kv[1]):
        print('%3d x  %-44s -> [%s] -> %s' % (n, a, b, c))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2].split(',') if len(sys.argv) > 2 else ['rocclr_fill', 'rocclr_copy', 'at::native'])
