#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md §HBM prescribes).  Units: the counters are KiB per dispatch; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced streaming read, so the fetch side is doubled (guide's correction);
WRITE_SIZE is used as reported (uncalibrated).

    python tools/rocpd_pmc.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db [top [bench_line.json]]

With a bench.py JSON line as the 4th argument the table gains the algorithmic bytes per launch (every operand of the conv / GEMM
launch once, from the descriptors: roofline.algorithmic_mib_per_launch) and the counter / algorithmic ratio.
"""
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value), sum(end-start) from counters_collection "
                       "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def _norm(kernel_name):
    """'void igemm_nt_kernel<128, 128, 64, 32, 2>(unsigned short const*, ...' -> 'igemm_nt<128,128,64,32,2>' (the variant strings
    of bench.py's roofline.algorithmic_mib_per_launch)."""
    k = kernel_name.replace('void ', '')
    depth, out = 0, []
    for ch in k:
        if ch == '(' and depth == 0:
            break
        depth += ch == '<'
        depth -= ch == '>'
        out.append(ch)
    return ''.join(out).replace(' ', '').replace('_kernel', '')


def main(fdb, wdb, top=25, algo=None):
    """algo: {variant: algorithmic MiB per launch} from a bench.py JSON line (roofline.algorithmic_mib_per_launch: every operand
    of a conv / GEMM launch once): adds the algorithmic column and the counter / algorithmic ratio for the kernels it names."""
    f = per_kernel(fdb, 'FETCH_SIZE')
    w = per_kernel(wdb, 'WRITE_SIZE')
    names = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0, 0))[1] + w.get(k, (0, 0, 0))[1]))
    algo = algo or {}
    fam_counter = {}                       # family (name without template arguments) -> (launches, counter MiB): for family-level rows
    for k in names:
        nf, vf, _ = f.get(k, (0, 0.0, 0))
        nw, vw, _ = w.get(k, (0, 0.0, 0))
        fam = _norm(k).split('<')[0]
        a, b = fam_counter.get(fam, (0, 0.0))
        fam_counter[fam] = (a + max(nf, nw, 1), b + 2.0 * vf / 1024.0 + vw / 1024.0)
    extra = ' algorithmic MiB/launch | counter / algorithmic |' if algo else ''
    print('| kernel | launches | fetch MiB/launch (x2 corrected) | write MiB/launch | total GiB (all launches) |' + extra)
    print('|---|---:|---:|---:|---:|' + ('---:|---:|' if algo else ''))
    tot = 0.0
    for k in names[:top]:
        nf, vf, _ = f.get(k, (0, 0.0, 0))
        nw, vw, _ = w.get(k, (0, 0.0, 0))
        n = max(nf, nw, 1)
        fetch = 2.0 * vf / 1024.0
        write = vw / 1024.0
        tot += fetch + write
        short = k if len(k) < 70 else k[:67] + '...'
        cols = ''
        if algo:
            key, fam = _norm(k), _norm(k).split('<')[0]
            if key in algo:
                cols = ' %.2f | %.2f |' % (algo[key], (fetch + write) / n / max(algo[key], 1e-9))
            elif fam in algo:          # the library names the family only (gemm_nt256): launch-weighted over its instantiations
                fn, fc = fam_counter[fam]
                cols = ' %.2f (family) | %.2f (family) |' % (algo[fam], fc / fn / max(algo[fam], 1e-9))
            else:
                cols = ' | |'
        print('| `%s` | %d | %.2f | %.2f | %.3f |%s' % (short, n, fetch / n, write / n, (fetch + write) / 1024.0, cols))
    rest = sum(2.0 * f.get(k, (0, 0.0, 0))[1] / 1024.0 + w.get(k, (0, 0.0, 0))[1] / 1024.0 for k in names[top:])
    print('| (others) | | | | %.3f |' % (rest / 1024.0))
    print('')
    print('total HBM traffic over the traced process: %.2f GiB' % ((tot + rest) / 1024.0))


if __name__ == '__main__':
    import json
    algo = None
    if len(sys.argv) > 4:                  # a file holding bench.py's JSON line
        for line in open(sys.argv[4]):
            if line.startswith('{'):
                algo = json.loads(line).get('roofline', {}).get('algorithmic_mib_per_launch')
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25, algo)
