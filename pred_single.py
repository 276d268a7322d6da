#!/usr/bin/env python
"""Single-image inference entry point with the reference's call surface (pred_single.py:28-140): --model / --load /
--trimap, `FullModel(model, dilate_kernel in {5,12,20}, agg_window)`, eval mode, the 3-loss dict of `forward_pretrain`
(:56-67) and the per-sample SAD / MSE over the trimap's unknown region (:45-54) — on the MI355X HIP path with
synthetic frames (the LMDB datasets are out of scope).  BASELINE.json config 1: `--model dim`, one
512 x 512 frame + trimap.

    python pred_single.py --model dim --trimap medium [--load checkpoint.pth] [--size 512 512] [--frames 2]
"""
import argparse
import os

import numpy as np
import torch

from models.model import FullModel
from tcvom_amd.synthetic import synthetic_window

DILATE = {'narrow': 5, 'medium': 12, 'wide': 20}          # pred_single.py:79-84
SUB_LOSSES = ('L_alpha', 'L_comp', 'L_grad')


def SAD(a, g, m):
    return float(np.mean(np.abs(np.float32(a)[m] / 255.0 - np.float32(g)[m] / 255.0))) if m.any() else 0.0


def MSE(a, g, m):
    return float(np.mean((np.float32(a)[m] / 255.0 - np.float32(g)[m] / 255.0) ** 2)) if m.any() else 0.0


def forward_pretrain(model, a, fg, bg):
    with torch.no_grad():
        out = model(a, fg, bg)
    loss = {k: out[i].sum().item() for i, k in enumerate(SUB_LOSSES)}
    loss['L_total'] = sum(loss.values())
    return [*out[3:], loss]


def main(args):
    device = torch.device('cuda', 0)
    model = FullModel(model=args.model, dilate_kernel=DILATE[args.trimap], agg_window=int(args.agg_window))
    if args.load:
        dct = torch.load(args.load, map_location='cpu')
        dct = dct['state_dict'] if 'state_dict' in dct else dct
        missing, unexpected = model.NET.load_state_dict(dct, strict=False)
        print('Missing keys: ' + str(sorted(missing)))
        print('Unexpected keys: ' + str(sorted(unexpected)))
    model = model.to(device).eval()
    H, W = args.size
    totals = {k: 0.0 for k in SUB_LOSSES + ('L_total', 'mSAD', 'MSE')}
    for i in range(args.frames):
        a, fg, bg = (t.to(device) for t in synthetic_window(1, 1, H, W, seed=i))
        _imgs, tris, alphas, _comps, gts, _fs, _bs, loss = forward_pretrain(model, a, fg, bg)
        pred = np.uint8(alphas[0, 0, 0].float().cpu().numpy() * 255)
        gt = np.uint8(gts[0, 0, 0].float().cpu().numpy() * 255)
        unknown = np.uint8(tris[0, 0, 0].float().cpu().numpy() * 255) == 128
        for k in SUB_LOSSES + ('L_total',):
            totals[k] += loss[k]
        totals['mSAD'] += SAD(pred, gt, unknown)
        totals['MSE'] += MSE(pred, gt, unknown)
        if args.save:
            os.makedirs(args.save, exist_ok=True)
            np.save(os.path.join(args.save, 'frame%03d_pred.npy' % i), pred)
    out = {k: v / args.frames for k, v in totals.items()}
    for k in sorted(out):
        print('%s: %.6f' % (k, out[k]))
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', required=True, choices=['dim', 'gca', 'fba'], help='single-image base on the HIP path')
    ap.add_argument('--load', default=None, help='checkpoint (NET.state_dict layout of the reference)')
    ap.add_argument('--trimap', required=True, choices=list(DILATE))
    ap.add_argument('--agg_window', default=7)
    ap.add_argument('--size', type=int, nargs=2, default=(512, 512))
    ap.add_argument('--frames', type=int, default=2)
    ap.add_argument('--save', default=None)
    return ap.parse_args(argv)


if __name__ == '__main__':
    main(parse())
