#!/usr/bin/env python
"""Every conv-engine launch of one 1080p training step with its shape, kernel variant, HIP-event time, TFLOP/s and algorithmic GB/s
(bench.py's instrumented step, not aggregated): where the small-K / strided layers sit against the HBM and MFMA rooflines."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
import tcvom_amd._lib as L                                           # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402

dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 1088, 1920, 0, config=os.environ.get('CONV_LAUNCHES_CONFIG', 'gca'))


def step():
    loss = train_step_loss(model(a, fg, bg))
    model.zero_grad(set_to_none=True)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
rec = []
L.PROFILE = rec
step()
torch.cuda.synchronize()
L.PROFILE = None
rows = []
for name, d, e0, e1 in rec:
    if 'bytes' in d:
        continue
    ms = e0.elapsed_time(e1)
    taps = sum(1 for t in range(d['ntaps']) if d['tap_w'][t] >= 0)
    gf = d['gflop'] if 'gflop' in d else 2.0 * d['P'] * d['K'] * taps * d['C'] * max(d['batch'], 1) / 1e9
    rows.append((ms, name.replace('tcvom_', ''), d['variant'], d['P'], d['K'], d['C'], taps, d['batch'], d.get('phases', 1), gf, d.get('algo_bytes', 0)))
sel = sys.argv[1] if len(sys.argv) > 1 else ''
print('%-26s %-30s %8s %5s %5s %4s %3s %3s %8s %8s %8s %8s' % ('entry', 'variant', 'P', 'K', 'C', 'taps', 'b', 'ph', 'us', 'TFLOP/s', 'MiB', 'GB/s'))
for ms, name, var, P, K, Cc, taps, b, ph, gf, nb in rows:
    if sel and sel not in var:
        continue
    print('%-26s %-30s %8d %5d %5d %4d %3d %3d %8.1f %8.1f %8.1f %8.1f' % (name, var, P, K, Cc, taps, b, ph, ms * 1e3, gf / ms, nb / 2 ** 20, nb / ms / 1e6))
