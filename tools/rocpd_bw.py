#!/usr/bin/env python
"""Per-kernel achieved HBM bandwidth from the two PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE: KiB per dispatch,
fetch doubled as in rocpd_pmc.py) and the dispatch durations recorded in the same databases: (fetch + write) / time, aggregated over
all launches of a kernel.  For the streaming kernels (BatchNorm / GroupNorm passes) this is the number to hold against ~5.5 TB/s.

    python tools/rocpd_bw.py <fetch.db> <write.db> [top]
"""
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value), sum(end-start) from counters_collection "
                       "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def main(fdb, wdb, top=25):
    f, w = per_kernel(fdb, 'FETCH_SIZE'), per_kernel(wdb, 'WRITE_SIZE')
    rows = []
    for k, (n, kib, ns) in f.items():
        wn, wkib, wns = w.get(k, (n, 0.0, ns))
        gib = (2.0 * kib + wkib) / 1048576.0
        ms = 0.5 * (ns + wns) / 1e6
        rows.append((ms, k, n, gib))
    rows.sort(reverse=True)
    print('| kernel | launches | GiB (fetch x2 + write) | ms | TB/s |')
    print('|---|---:|---:|---:|---:|')
    for ms, k, n, gib in rows[:top]:
        short = k if len(k) < 70 else k[:67] + '...'
        print('| `%s` | %d | %.2f | %.2f | %.2f |' % (short, n, gib, ms, gib * 1.073741824 / ms if ms else 0.0))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25)
