#!/usr/bin/env python
"""Inference entry point with the reference's call surface (pred_vmn.py:28-140): --model/--load/--trimap/--agg_window,
`FullModel_VMD('vmn_' + model, dilate_kernel in {5,12,20}, agg_window)`, eval mode, 3-frame windows at 1088x1920
(1080 is not a multiple of 32), the 5-loss dict, and the centre-frame `uint8(alpha*255)` crop [:1080, :1920] —
on the MI355X HIP path.  With `--data` the clips come from a precomputed validation tree through
`dataset.VMD.VideoMattingDataset(mode='val', precomputed_val=data, sample_length=3)` (pred_vmn.py:88-102) and
`<frame>_pred.png` / `<frame>_tri.png` are written under `--save` (pred_vmn.py:127-134); without it, synthetic clips.

    python pred_vmn.py --model gca --trimap medium [--load checkpoint.pth.tar] [--clips 2] [--save out_dir]
    python pred_vmn.py --model gca --trimap medium --data /data/VideoMatting108_val --save out_dir [--subset] [--n_threads 4]
"""
import argparse
import os

import numpy as np
import torch

from models.model import FullModel_VMD
from tcvom_amd.synthetic import synthetic_window

DILATE = {'narrow': 5, 'medium': 12, 'wide': 20}          # pred_vmn.py:68-75


def forward_pretrain(model, dp):
    fg, bg, a, idx = dp
    with torch.no_grad():
        out = model(a, fg, bg)
    loss = {k: out[i].sum().item() for i, k in enumerate(('L_alpha', 'L_comp', 'L_grad', 'L_dt', 'L_att'))}
    loss['L_total'] = sum(loss.values())
    return [*out[5:], loss, idx]


def main(args):
    device = torch.device('cuda', 0)
    model = FullModel_VMD(model='vmn_' + args.model, dilate_kernel=DILATE[args.trimap], agg_window=int(args.agg_window))
    if args.load:
        missing, unexpected = model.NET.load_state_dict(torch.load(args.load, map_location='cpu'), strict=False)
        print('Missing keys: ' + str(sorted(missing)))
        print('Unexpected keys: ' + str(sorted(unexpected)))
    model = model.to(device).eval()
    H, W, h, w = 1088, 1920, 1080, 1920
    totals = {}
    if getattr(args, 'data', None):
        return predict_directory(model, args, device, (H, W), (h, w))
    for clip in range(args.clips):
        a, fg, bg = (t.to(device) for t in synthetic_window(1, 3, H, W, seed=clip))
        _, tris, alphas, _, _, _, _, loss, _ = forward_pretrain(model, (fg, bg, a, torch.tensor([clip])))
        for k, v in loss.items():
            totals[k] = totals.get(k, 0.0) + v
        c = 1
        alpha = np.uint8(alphas[0, c, 0, :h, :w].float().cpu().numpy() * 255)
        tri = np.uint8(tris[0, c, 0, :h, :w].float().cpu().numpy() * 255)
        if args.save:
            os.makedirs(args.save, exist_ok=True)
            np.save(os.path.join(args.save, 'clip%03d_pred.npy' % clip), alpha)
            np.save(os.path.join(args.save, 'clip%03d_tri.npy' % clip), tri)
    for k in sorted(totals):
        print('%s: %.6f' % (k, totals[k] / args.clips))


def predict_directory(model, args, device, padded, frame):
    from PIL import Image
    from torch.utils.data import DataLoader
    from dataset.VMD import VideoMattingDataset
    ds = VideoMattingDataset(data_root=args.data, image_shape=padded, mode='val', use_subset=args.subset,
                             plus1=args.model.startswith('vmn_res'), precomputed_val=args.data, sample_length=3, no_flow=True,
                             device=device)
    c, (h, w) = ds.sample_length // 2, frame
    totals = {}
    raws = DataLoader(ds.raw_view(), batch_size=None, shuffle=False, num_workers=args.n_threads)     # PNG decode in workers
    for raw in raws:
        fg, bg, a, idx = ds.transform(raw)                                                            # the rest on the GPU
        _, tris, alphas, _, _, _, _, loss, _ = forward_pretrain(model, (fg[None], bg[None], a[None], idx[None]))
        for k, v in loss.items():
            totals[k] = totals.get(k, 0.0) + v
        if args.save:
            fn = os.path.splitext(ds.samples[int(idx)][c])[0]
            os.makedirs(os.path.join(args.save, os.path.dirname(fn)), exist_ok=True)
            for arr, suffix in ((tris, '_tri.png'), (alphas, '_pred.png')):
                Image.fromarray(np.uint8(arr[0, c, 0, :h, :w].float().cpu().numpy() * 255), 'L').save(os.path.join(args.save, fn + suffix))
    for k in sorted(totals):
        print('%s: %.6f' % (k, totals[k] / float(len(ds))))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', required=True, choices=['gca', 'fba', 'dim', 'index'], help='base matting network (HIP path: gca, fba, dim, index)')
    ap.add_argument('--load', default=None, help='checkpoint (NET.state_dict layout of the reference)')
    ap.add_argument('--trimap', required=True, choices=list(DILATE))
    ap.add_argument('--agg_window', default=7)
    ap.add_argument('--clips', type=int, default=1)
    ap.add_argument('--data', default=None, help='precomputed validation tree (FG_done/, BG_done/, frame_corr.json, val_videos.txt)')
    ap.add_argument('--subset', action='store_true')
    ap.add_argument('--n_threads', type=int, default=4, help='PNG-decoding worker processes')
    ap.add_argument('--save', default=None)
    main(ap.parse_args())
