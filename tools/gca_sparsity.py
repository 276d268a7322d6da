#!/usr/bin/env python
"""How sparse is the guided-contextual-attention matrix P (8160 x 8160 per frame at 1080p) on the bench window: exact zeros (after
the 2^-25 flush of the bf16 build / fp16's rounding), and all-zero tiles at the granularities a GEMM could skip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd import ops                                           # noqa: E402

dev = torch.device('cuda', 0)
H, W = int(os.environ.get('GS_H', 1088)), int(os.environ.get('GS_W', 1920))
model, a, fg, bg = bench.build(dev, H, W, 0)
ops.GCA_KEEP = []
with torch.no_grad():
    model(a, fg, bg)
torch.cuda.synchronize()
for idx, P in enumerate(ops.GCA_KEEP):
    B, N, ld = P.shape
    nz = (P != 0)
    print('GCA call %d: P %s, nonzero entries %.2f %%, row max mean %.3f, entries > 1e-3: %.3f %%'
          % (idx, tuple(P.shape), 100.0 * nz.float().mean().item(), P.float().amax(2).mean().item(), 100.0 * (P.float() > 1e-3).float().mean().item()))
    for tr, tc in ((256, 64), (256, 256), (64, 64), (32, 64)):
        n_r, n_c = N // tr, ld // tc
        t = nz[:, :n_r * tr, :n_c * tc].reshape(B, n_r, tr, n_c, tc).any(4).any(2)
        print('   tiles %3d x %3d (rows x cols): %.1f %% contain a nonzero' % (tr, tc, 100.0 * t.float().mean().item()))
