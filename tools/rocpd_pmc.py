#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md §HBM prescribes).  Units: the counters are KiB per dispatch; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced streaming read, so the fetch side is doubled (guide's correction);
WRITE_SIZE is used as reported (uncalibrated).

    python tools/rocpd_pmc.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db
"""
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value), sum(end-start) from counters_collection "
                       "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def main(fdb, wdb, top=25):
    f = per_kernel(fdb, 'FETCH_SIZE')
    w = per_kernel(wdb, 'WRITE_SIZE')
    names = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0, 0))[1] + w.get(k, (0, 0, 0))[1]))
    print('| kernel | launches | fetch MiB/launch (x2 corrected) | write MiB/launch | total GiB (all launches) |')
    print('|---|---:|---:|---:|---:|')
    tot = 0.0
    for k in names[:top]:
        nf, vf, _ = f.get(k, (0, 0.0, 0))
        nw, vw, _ = w.get(k, (0, 0.0, 0))
        n = max(nf, nw, 1)
        fetch = 2.0 * vf / 1024.0
        write = vw / 1024.0
        tot += fetch + write
        short = k if len(k) < 70 else k[:67] + '...'
        print('| `%s` | %d | %.2f | %.2f | %.3f |' % (short, n, fetch / n, write / n, (fetch + write) / 1024.0))
    rest = sum(2.0 * f.get(k, (0, 0.0, 0))[1] / 1024.0 + w.get(k, (0, 0.0, 0))[1] / 1024.0 for k in names[top:])
    print('| (others) | | | | %.3f |' % (rest / 1024.0))
    print('')
    print('total HBM traffic over the traced process: %.2f GiB' % ((tot + rest) / 1024.0))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25)
