#!/usr/bin/env python
"""Per-kernel MFMA utilisation from a rocprofv3 PMC pass that collected SQ_VALU_MFMA_BUSY_CYCLES (+ GRBM_GUI_ACTIVE,
SQ_WAVE_CYCLES, SQ_BUSY_CYCLES where available):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d out -o sq -- python bench.py ...
    python tools/rocpd_mfma.py out/sq_results.db > profiles/rNN_mfma_busy.md

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over every SIMD of the device (32 per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md); the device has 256 CUs x 4 SIMDs.  busy% = busy cycles / (1024 SIMDs x kernel cycles), with the
kernel's cycles taken as duration x 2.4 GHz (peak clock: a lower bound of the true utilisation under PMC, where the
clock sags) and, when GRBM_GUI_ACTIVE was collected, from that counter."""
import sqlite3
import sys


def main(path, top=30):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(end-start) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    data = {}
    for k, c, n, v, ns in rows:
        data.setdefault(k, {})[c] = (n, v, ns)
    names = sorted(data, key=lambda k: -data[k].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0, 0))[1])
    print('| kernel | launches | avg us (under PMC) | MFMA busy Mcycles/launch | busy % of 1024 SIMDs @2.4 GHz | busy % vs GRBM_GUI_ACTIVE |')
    print('|---|---:|---:|---:|---:|---:|')
    for k in names[:top]:
        n, busy, ns = data[k].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0, 0))
        if not n or busy <= 0:
            continue
        gui = data[k].get('GRBM_GUI_ACTIVE', (0, 0.0, 0))[1]
        us = ns / n / 1e3
        pct = 100.0 * busy / (1024.0 * ns * 2.4)
        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: cycles of one XCD = gui / 8
        pct_gui = 100.0 * busy / (1024.0 * gui / 8.0) if gui else float('nan')
        short = k if len(k) < 72 else k[:69] + '...'
        print('| `%s` | %d | %.1f | %.2f | %.1f | %.1f |' % (short, n, us, busy / n / 1e6, pct, pct_gui))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
