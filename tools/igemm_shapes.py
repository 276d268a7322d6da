#!/usr/bin/env python
"""One event-instrumented training step of the headline workload (bench.py's), every conv / GEMM launch listed by the kernel
instantiation the library selected and its shape: launches, total ms, TFLOP/s.  Shows which layer shapes the time of a kernel
family sits in (the per-family totals are bench.py's `all_igemm`)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
import tcvom_amd._lib as L                                # noqa: E402
from tcvom_amd.facade import train_step_loss              # noqa: E402
from tcvom_amd.optim import FusedAdam                     # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else 'gca'
    dev = torch.device('cuda', 0)
    model, a, fg, bg = bench.build(dev, bench.FULL_H, bench.FULL_W, 0, config)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)

    def step():
        loss = train_step_loss(model(a, fg, bg))
        model.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rec = []
    L.PROFILE = rec
    step()
    torch.cuda.synchronize()
    L.PROFILE = None
    agg = {}
    for name, d, e0, e1 in rec:
        if 'bytes' in d:
            continue
        taps = sum(1 for t in range(d['ntaps']) if d['tap_w'][t] >= 0)
        key = (d['variant'], name[6:], d['P'], d['K'], d['C'], taps, d['ntaps'], max(d['batch'], 1), d.get('phases', 1))
        n, ms, gf = agg.get(key, (0, 0.0, 0.0))
        agg[key] = (n + 1, ms + e0.elapsed_time(e1), gf + 2.0 * d['P'] * d['K'] * taps * d['C'] * max(d['batch'], 1) / 1e9)
    tot = sum(v[1] for v in agg.values())
    print('%-30s %-18s %9s %5s %5s %4s/%-4s %3s %2s | %3s %8s %7s' % ('variant', 'entry', 'P', 'K', 'C', 'tap', 'pad', 'b', 'ph', 'n', 'ms', 'TF/s'))
    for key, (n, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-30s %-18s %9d %5d %5d %4d/%-4d %3d %2d | %3d %8.3f %7.0f' % (key + (n, ms, gf / ms)))
    print('total %.3f ms' % tot)


if __name__ == '__main__':
    main()
