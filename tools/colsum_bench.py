#!/usr/bin/env python
"""tcvom_colsum (bias gradients: column sums of a [P][K] 16-bit matrix) timed with HIP events; TCVOM_LIB selects the build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402

for P, K in ((32640, 128), (97920, 64), (97920, 128), (130560, 32), (2088960, 32)):
    dy = torch.randn(P, K, device='cuda').to(L.ACT_DTYPE)
    out = torch.empty(K, device='cuda')
    st = L.stream_ptr()
    for _ in range(3):
        L.call('tcvom_colsum', L.ptr(dy), L.ptr(out), P, K, K, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.call('tcvom_colsum', L.ptr(dy), L.ptr(out), P, K, K, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    err = float((out - dy.float().sum(0)).abs().max() / dy.float().sum(0).abs().max())
    print('P %8d K %4d  %7.1f us  %6.1f GB/s  rel err %.1e' % (P, K, us, P * K * 2 / us / 1e3, err))
