"""Evaluation metrics of calc_metric.py:22-46 on the GPU: SAD, MSE, SSDA, dtSSD and the flow-warped MESSDdt of one frame
in ONE kernel launch (the reference computes them with numpy / CPU grid_sample after PNG round trips).

`frame_metrics` takes device tensors: a, g [H,W] float (0..1), tri [H,W] uint8 (unknown = neither 0 nor 255), optionally
the adjacent frame (ha, hg) and the optical flow [H,W,2] (x, y displacement; NaN = invalid) as the reference's
`_flow_read` produces it.  Returns python floats with the reference's definitions:
    SAD  = mean |a - g| over the unknown region          MSE  = mean (a - g)^2
    SSDA = sqrt(sum (a - g)^2)                           dtSSD = sqrt(sum ((a - ha) - (g - hg))^2)
    MESSDdt = (sum |(a-g) - (pa-pg)|, sum |(a-g)^2 - (pa-pg)^2|, valid pixels), pa / pg = ha / hg warped by the flow
"""
import math

import torch

from . import _lib as L


def frame_metrics(a, g, tri, ha=None, hg=None, flow=None):
    if not a.is_cuda:
        raise RuntimeError('tcvom_amd.metrics runs on the GPU through libtcvom_hip.so only (no CPU fallback)')
    H, W = a.shape[-2:]
    f32 = lambda t: None if t is None else t.reshape(H, W).float().contiguous()
    a, g, ha, hg = f32(a), f32(g), f32(ha), f32(hg)
    tri = tri.reshape(H, W).to(torch.uint8).contiguous()
    fl = None
    if flow is not None:
        fl = flow.reshape(H, W, 2).permute(2, 0, 1).float().contiguous()
    acc = torch.zeros(8, dtype=torch.float64, device=a.device)
    L.call('tcvom_matting_metrics', L.ptr(a), L.ptr(g), L.ptr(tri), L.ptr(ha), L.ptr(hg), L.ptr(fl), L.ptr(acc), H, W, L.stream_ptr())
    n, sad, sq, dt, fix, org, valid = acc[:7].tolist()
    out = {'pixels': int(n), 'SAD': sad / n if n else float('nan'), 'MSE': sq / n if n else float('nan'), 'SSDA': math.sqrt(sq)}
    if ha is not None:
        out['dtSSD'] = math.sqrt(dt)
    if flow is not None:
        out['MESSDdt'] = (fix, org, int(valid))
    return out
