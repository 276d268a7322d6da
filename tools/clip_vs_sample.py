#!/usr/bin/env python
"""EvalModel.forward_video (features of a frame computed once, `chunk` frames per launch) against per-sample EvalModel calls
(3-frame windows): the two run the same layers with different frame-batch sizes, hence different tile configurations and fp32
summation orders; this prints how far the uint8 alphas move, by size."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from models.model import EvalModel, FullModel_VMD
from tcvom_amd.synthetic import formula_tensor, synthetic_window

dev = 'cuda'
fm = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
fm.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in fm.NET.state_dict().items()})
fm = fm.to(dev).train()
with torch.no_grad():
    for _ in range(2):
        fm(*(t.to(dev) for t in synthetic_window(1, 3, 256, 320, seed=0)))
em = EvalModel('vmn_gca', agg_window=7, dilate_kernel=2)
em.NET.load_state_dict(fm.NET.state_dict())
em = em.to(dev).eval()
for H, W, T in ((96, 128, 4), (256, 320, 5), (544, 960, 5)):
    a, fg, bg = synthetic_window(1, T, H, W, seed=5)
    al = a / 255.0
    imgs = torch.round(fg * al + bg * (1 - al))[0].to(dev)
    tris = torch.where(a <= 0, torch.zeros_like(a), torch.where(a >= 255, torch.full_like(a, 255.0), torch.full_like(a, 128.0)))[0].to(dev)
    with torch.no_grad():
        clip = em.forward_video(imgs, tris)
        clip2 = em.forward_video(imgs, tris, chunk=3)
        per = []
        for c in range(T):
            p = c + 1 if c == 0 else c - 1
            n = c - 1 if c == T - 1 else c + 1
            x = torch.stack([imgs[p], imgs[c], imgs[n]]).unsqueeze(0)
            t = torch.stack([tris[p], tris[c], tris[n]]).unsqueeze(0)
            per.append(em(x, t).squeeze(0)[1])
        per = torch.stack(per)
    q = lambda t: torch.floor(t.clamp(0, 1) * 255)
    d = (q(clip) - q(per)).abs()
    d2 = (q(clip) - q(clip2)).abs()
    unk = (tris == 128)
    print('%dx%d T=%d: clip(chunk 4) vs per-sample: mean %.4f max %.0f grey levels, %d of %d unknown pixels differ by > 1; chunk 4 vs chunk 3: mean %.4f max %.0f'
          % (H, W, T, float(d.mean()), float(d.max()), int((d[unk] > 1).sum()), int(unk.sum()), float(d2.mean()), float(d2.max())))
