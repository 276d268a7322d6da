#!/usr/bin/env python
"""Per-kernel sums of whatever counters a rocprofv3 --pmc pass collected (rocpd SQLite): one row per kernel, one column per
counter (value per launch), sorted by total duration.

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d out -o c -- python bench.py ...
    python tools/rocpd_counters.py out/.../c_results.db [top]"""
import sqlite3
import sys


def main(path, top=25):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(end-start) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    data, counters = {}, []
    for k, c, n, v, ns in rows:
        data.setdefault(k, {})[c] = (n, v, ns)
        if c not in counters:
            counters.append(c)
    names = sorted(data, key=lambda k: -max(x[2] for x in data[k].values()))
    print('| kernel | launches | avg us | ' + ' | '.join('%s / launch' % c for c in counters) + ' |')
    print('|---|---:|---:|' + '---:|' * len(counters))
    for k in names[:top]:
        n, _, ns = next(iter(data[k].values()))
        short = k if len(k) < 64 else k[:61] + '...'
        print('| `%s` | %d | %.1f | ' % (short, n, ns / n / 1e3) + ' | '.join('%.4g' % (data[k].get(c, (1, 0.0, 0))[1] / n) for c in counters) + ' |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
