#!/usr/bin/env python
"""BASELINE config 2 (forward-only 3 x 512 x 512 window, train-mode statistics) as a HIP graph: does the forward capture, and what does a
replay cost against the eager forward (whose ~300 launches of 3-8 us kernels are enqueued by Python at ~10 us each)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, H, W, 0)


def fwd():
    with torch.no_grad():
        out = model(a, fg, bg)
        return train_step_loss(out), out[7]


def timed(fn, n=50):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for _ in range(5):
    fwd()
print('eager forward    %.3f ms' % timed(fwd))
t0 = time.time()
for _ in range(20):
    fwd()
print('host enqueue     %.3f ms' % ((time.time() - t0) / 20 * 1e3))
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss, alpha = fwd()
    torch.cuda.synchronize()
    print('captured')
    g.replay()
    torch.cuda.synchronize()
    l1, a1 = float(loss), alpha.clone()
    print('graph replay     %.3f ms' % timed(g.replay))
    le, ae = fwd()
    print('loss replay %.6f eager %.6f; alpha max diff %.3e' % (l1, float(le), float((a1 - ae).abs().max())))
except Exception as ex:                                             # noqa: BLE001
    print('capture failed: %r' % (ex,))
