#!/usr/bin/env python
"""Single-image inference entry point with the reference's call surface (pred_single.py:28-140): --model / --load /
--trimap, `FullModel(model, dilate_kernel in {5,12,20}, agg_window)`, eval mode, the 3-loss dict of `forward_pretrain`
(:56-67) and the per-sample SAD / MSE over the trimap's unknown region (:45-54) — on the MI355X HIP path.  With `--data`
the samples are the 3-frame validation clips of `dataset.VMD.VideoMattingDataset` (the `vmd` branch, pred_single.py:104-114,
167-192: centre frame scored on rows :1080, `_tri.png` / `_pred.png` written under `--save`); without it, synthetic frames
(the Adobe-DIM evaluation set of the `dim` branch is not read).  BASELINE.json config 1: `--model dim`, one 512 x 512
frame + trimap.

    python pred_single.py --model dim --trimap medium [--load checkpoint.pth] [--size 512 512] [--frames 2]
    python pred_single.py --model gca --trimap medium --data /data/VideoMatting108_val --save out [--subset]
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
    if getattr(args, 'data', None):
        return predict_directory(model, args, device, totals)
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


def predict_directory(model, args, device, totals):
    from PIL import Image
    from torch.utils.data import DataLoader
    from dataset.VMD import VideoMattingDataset
    ds = VideoMattingDataset(data_root=args.data, image_shape=(1088, 1920), mode='val', use_subset=args.subset,
                             plus1=args.model.startswith('vmn_res'), no_flow=True, sample_length=3, precomputed_val=args.data,
                             device=device)
    c = ds.sample_length // 2
    raws = DataLoader(ds.raw_view(), batch_size=None, shuffle=False, num_workers=args.n_threads)
    for raw in raws:
        fg, bg, gt, idx = ds.transform(raw)
        _imgs, tris, alphas, _comps, _gts, _fs, _bs, loss = forward_pretrain(model, gt[None], fg[None], bg[None])
        a = np.uint8(alphas[0, c, 0, :1080].float().cpu().numpy() * 255)
        t = np.uint8(tris[0, c, 0, :1080].float().cpu().numpy() * 255)
        g = np.uint8(gt[c, 0, :1080].cpu().numpy())
        m = (t > 0) * (t < 255)
        for k in SUB_LOSSES + ('L_total',):
            totals[k] += loss[k]
        totals['mSAD'] += SAD(a, g, m)
        totals['MSE'] += MSE(a, g, m)
        if args.save:
            fn = os.path.splitext(ds.samples[int(idx)][c])[0]
            os.makedirs(os.path.join(args.save, os.path.dirname(fn)), exist_ok=True)
            Image.fromarray(t, 'L').save(os.path.join(args.save, fn + '_tri.png'))
            Image.fromarray(a, 'L').save(os.path.join(args.save, fn + '_pred.png'))
    out = {k: v / float(len(ds)) for k, v in totals.items()}
    for k in sorted(out):
        print('%s: %.6f' % (k, out[k]))
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', required=True, choices=['dim', 'gca', 'fba', 'index'], help='single-image base on the HIP path')
    ap.add_argument('--load', default=None, help='checkpoint (NET.state_dict layout of the reference)')
    ap.add_argument('--trimap', required=True, choices=list(DILATE))
    ap.add_argument('--agg_window', default=7)
    ap.add_argument('--size', type=int, nargs=2, default=(512, 512))
    ap.add_argument('--frames', type=int, default=2)
    ap.add_argument('--save', default=None)
    ap.add_argument('--data', default=None, help='precomputed validation tree of VideoMatting108 (FG_done/, BG_done/, frame_corr.json, val_videos.txt)')
    ap.add_argument('--subset', action='store_true')
    ap.add_argument('--n_threads', type=int, default=4, help='PNG-decoding worker processes')
    return ap.parse_args(argv)


if __name__ == '__main__':
    main(parse())
