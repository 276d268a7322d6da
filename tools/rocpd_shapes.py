#!/usr/bin/env python
"""Per-launch-shape breakdown of a rocprofv3 rocpd database: groups dispatches by (kernel, grid) so that the
individual layer shapes behind one igemm template instantiation can be told apart.

    python tools/rocpd_shapes.py gpurun_out/prof/x_results.db igemm 60
"""
import re
import sqlite3
import sys


def main(path, pat='', top=60):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(end-start), avg(end-start), min(end-start) "
                       "from kernels group by name, grid_x, grid_y, grid_z order by 7 desc").fetchall()
    total = sum(r[6] for r in rows)
    print('| kernel | grid (workgroups) | calls | total ms | avg us | min us | % |')
    print('|---|---|---:|---:|---:|---:|---:|')
    n = 0
    for name, gx, gy, gz, wx, cnt, tot, avg, mn in rows:
        if pat and not re.search(pat, name):
            continue
        short = re.sub(r'\(.*', '', name).replace('void ', '')
        print('| `%s` | %dx%dx%d | %d | %.3f | %.1f | %.1f | %.1f |' % (short, gx // max(wx, 1), gy, gz, cnt, tot / 1e6, avg / 1e3,
                                                                     mn / 1e3, 100.0 * tot / total))
        n += 1
        if n >= top:
            break


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '', int(sys.argv[3]) if len(sys.argv) > 3 else 60)
