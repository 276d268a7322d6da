#!/usr/bin/env python
"""TAM core (tcvom_tam_fwd / tcvom_tam_bwd) at the 1080p os8 size (136 x 240, C = 128, window 7) as a function of the
fraction of unknown pixels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import ops                                            # noqa: E402

H, W, C = 136, 240, 128
torch.manual_seed(0)
mk = lambda: torch.randn(1, H, W, C, device='cuda').to(ops.H16)
q, kb, kf, v = (mk().requires_grad_(True) for _ in range(4))
for frac in (0.0, 0.03, 0.25, 1.0, -1.0):
    if frac < 0:                                      # a band of unknown pixels along a circle (what a trimap looks like)
        yy, xx = torch.meshgrid(torch.arange(H, device='cuda'), torch.arange(W, device='cuda'), indexing='ij')
        rr = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2).float().sqrt()
        mask = ((rr - 34).abs() < 2.5).to(torch.uint8)[None]
    else:
        mask = (torch.rand(1, H, W, device='cuda') < frac).to(torch.uint8)
    n = int(mask.sum())

    def fwd():
        return ops.tam_attention(q, kb, kf, v, mask, 7)

    def both():
        out, ab, af = fwd()
        (out.float().sum() + ab.sum() + af.sum()).backward()
    for fn, name in ((fwd, 'fwd'), (both, 'fwd+bwd')):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print('unknown %6d px (%5.1f %%)  %-8s %8.1f us' % (n, 100.0 * n / (H * W), name, e0.elapsed_time(e1) * 100))
