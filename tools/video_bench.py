#!/usr/bin/env python
"""Inference throughput on a clip (pred_test.py path, EvalModel('vmn_gca'), 1088x1920): one EvalModel call per 3-frame
sample (the reference loop) vs EvalModel.forward_video (encoder + decoder-front once per frame, features shared by the
three windows that contain the frame)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from models.model import EvalModel, FullModel_VMD                        # noqa: E402
from tcvom_amd.synthetic import formula_tensor, synthetic_window         # noqa: E402

dev = torch.device('cuda', 0)
T, H, W = 12, 1088, 1920
fm = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
fm.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in fm.NET.state_dict().items()})
fm = fm.to(dev).train()
with torch.no_grad():
    for _ in range(2):                                                   # calibrate the BatchNorm running statistics
        fm(*(t.to(dev) for t in synthetic_window(1, 3, 256, 320, seed=0)))
em = EvalModel('vmn_gca', agg_window=7, dilate_kernel=2)
em.NET.load_state_dict(fm.NET.state_dict())
em = em.to(dev).eval()
a, fg, bg = synthetic_window(1, T, H, W, seed=5)
al = a / 255.0
imgs = torch.round(fg * al + bg * (1 - al))[0].to(dev)
tris = torch.where(a <= 0, torch.zeros_like(a), torch.where(a >= 255, torch.full_like(a, 255.0), torch.full_like(a, 128.0)))[0].to(dev)


def per_sample():
    out = []
    for c in range(T):
        p = c + 1 if c == 0 else c - 1
        n = c - 1 if c == T - 1 else c + 1
        idx = [p, c, n]
        out.append(em(imgs[idx].unsqueeze(0), tris[idx].unsqueeze(0))[0, 1])
    return torch.stack(out)


for name, fn in (('per-sample windows', per_sample), ('forward_video (cached features)', lambda: em.forward_video(imgs, tris))):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print('%-34s %.1f ms / frame (%.1f frames/s), mean alpha %.4f' % (name, dt / T * 1e3, T / dt, float(out.mean())))
