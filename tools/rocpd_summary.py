#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`)
into a per-kernel table: calls, total / average / min / max duration, share of GPU kernel time.

    python tools/rocpd_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats_1080p.md
"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from kernels").fetchone()
    print('# rocprofv3 kernel-trace summary: %s' % path)
    print('')
    print('total kernel time %.3f ms over %d dispatches; first-to-last span %.3f ms' % (
        total / 1e6, sum(r[1] for r in rows), (span[1] - span[0]) / 1e6))
    print('')
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for name, n, tot, avg, mn, mx in rows[:top]:
        short = name if len(name) < 90 else name[:87] + '...'
        print('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |' % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    rest = rows[top:]
    if rest:
        print('| (%d more kernels) | %d | %.3f | | | | %.1f |' % (len(rest), sum(r[1] for r in rest), sum(r[2] for r in rest) / 1e6,
                                                               100.0 * sum(r[2] for r in rest) / total))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
