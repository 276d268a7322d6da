#!/usr/bin/env python
"""The Temporal Attention Module launches alone (tcvom_tam_fwd / tcvom_tam_bwd through the C ABI, memsets and copies of the launchers
included, no autograd / ATen work around them): HIP-event time per call at 136 x 240 x 128, window 7, for an all-unknown window, a
band-shaped trimap and random masks.  TCVOM_TAM_KEY_VALU=1: backward pass B on the one-wave-per-key kernel (A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402

B, H, W, C, win = 1, 136, 240, 128, 7
w2 = win * win
torch.manual_seed(0)
mk = lambda: torch.randn(B, H, W, C, device='cuda').to(L.ACT_DTYPE)   # noqa: E731
q, kb, kf, v, dout = mk(), mk(), mk(), mk(), mk()
out, dq, dkb, dkf = (torch.empty_like(q) for _ in range(4))
attb = torch.empty(B, w2, H * W, device='cuda')
attf = torch.empty_like(attb)
datt = torch.randn(B, w2, H * W, device='cuda')
pbuf = torch.empty(B, 2, w2, H * W, device='cuda')
dsbuf = torch.empty_like(pbuf)
work = torch.empty(B * H * W + 1, dtype=torch.int32, device='cuda')
st = L.stream_ptr()
yy, xx = torch.meshgrid(torch.arange(H, device='cuda'), torch.arange(W, device='cuda'), indexing='ij')
rr = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2).float().sqrt()
masks = {'all unknown': torch.ones(B, H, W, dtype=torch.uint8, device='cuda'),
         'band (3.3 %)': ((rr - 34).abs() < 2.5).to(torch.uint8)[None].contiguous(),
         'random 25 %': (torch.rand(B, H, W, device='cuda') < 0.25).to(torch.uint8),
         'none': torch.zeros(B, H, W, dtype=torch.uint8, device='cuda')}
fb = 5 * H * W * C * 2 + H * W + 2 * w2 * H * W * 4
bb = 7 * H * W * C * 2 + H * W + 2 * w2 * H * W * 4 * 3


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


only = os.environ.get('TAM_ONLY')
for name, m in masks.items():
    if only and not name.startswith(only):
        continue
    fwd = lambda: L.call('tcvom_tam_fwd', L.ptr(q), L.ptr(kb), L.ptr(kf), L.ptr(v), L.ptr(m), L.ptr(out), L.ptr(attb), L.ptr(attf),   # noqa: E731
                         L.ptr(work), B, H, W, C, win, st)
    bwd = lambda: L.call('tcvom_tam_bwd', L.ptr(q), L.ptr(kb), L.ptr(kf), L.ptr(m), L.ptr(dout), L.ptr(datt), L.ptr(datt), L.ptr(dq),   # noqa: E731
                         L.ptr(dkb), L.ptr(dkf), L.ptr(pbuf), L.ptr(dsbuf), L.ptr(work), B, H, W, C, win, st)
    tf = timeit(fwd)
    tb = timeit(bwd)
    print('%-14s fwd %6.1f us (%.3f of 8 TB/s)   bwd %6.1f us   fwd+bwd %6.1f us (%.3f of 8 TB/s)'
          % (name, tf, fb / tf / 8e6, tb, tf + tb, (fb + bb) / (tf + tb) / 8e6))
