#!/usr/bin/env python
"""Counterpart of the reference's pred_test.py (real-video inference from `<video>/*_rgb.png` + `*_trimap.png`
folders) on the HIP path: same sample construction (previous / current / next frame with the clip ends mirrored,
pred_test.py:27-41), reflect padding to multiples of 32 (:47-66), EvalModel forward, crop, uint8(alpha * 255) PNGs.

    python pred_test.py --data <root> --load <NET state_dict .pth> --save <out dir> [--videos v1 v2] [--dilation 5]

PNG I/O uses Pillow (OpenCV is not in this image); frames are converted to the BGR channel order cv2.imread yields.
"""
import argparse
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F

from models.model import EvalModel


class TestFolder(object):
    SAMPLE_LENGTH = 3

    def __init__(self, data_root, videos):
        self.data_root = data_root
        if not videos:
            videos = [os.path.basename(f) for f in sorted(glob.glob(os.path.join(data_root, '*'))) if os.path.isdir(f)]
        self.samples = []
        for v in sorted(videos):
            src = sorted(glob.glob(os.path.join(data_root, v, '*_rgb.png')))
            tri = sorted(glob.glob(os.path.join(data_root, v, '*_trimap.png')))
            assert len(src) == len(tri) and len(src) >= 2, 'video %s: %d frames, %d trimaps' % (v, len(src), len(tri))
            frames = list(zip(src, tri))
            for c in range(len(frames)):
                p = c + 1 if c == 0 else c - 1
                n = c - 1 if c == len(frames) - 1 else c + 1
                self.samples.append((frames[p], frames[c], frames[n]))

    def __len__(self):
        return len(self.samples)

    @staticmethod
    def possible_pad(t):
        H, W = t.shape[-2:]
        NH, NW = (H + 31) // 32 * 32, (W + 31) // 32 * 32
        t = t.float()
        if H == NH and W == NW:
            return t
        return F.pad(t.unsqueeze(0), (0, NW - W, 0, NH - H), mode='reflect').squeeze(0)

    def _frame(self, rgb, tri):
        from PIL import Image
        im = np.asarray(Image.open(rgb).convert('RGB'))[..., ::-1].copy()            # BGR, as cv2.imread
        tr = np.asarray(Image.open(tri).convert('L'))[..., None]
        return (self.possible_pad(torch.from_numpy(im).permute(2, 0, 1)), self.possible_pad(torch.from_numpy(tr.copy()).permute(2, 0, 1)),
                im.shape[:2])

    def centre(self, idx):
        """Only the centre frame of sample idx (the whole-clip path needs every frame once, not three times)."""
        return self._frame(*self.samples[idx][self.SAMPLE_LENGTH // 2])

    def __getitem__(self, idx):
        frames = [self._frame(rgb, tri) for rgb, tri in self.samples[idx]]
        return torch.stack([f[0] for f in frames]).float(), torch.stack([f[1] for f in frames]).float(), frames[-1][2]


def pred(dataset, indices, device, args):
    from PIL import Image
    torch.cuda.set_device(device)
    c = dataset.SAMPLE_LENGTH // 2
    model = EvalModel(model=args.model, agg_window=args.agg_window, dilate_kernel=args.dilation)
    model.NET.load_state_dict(torch.load(args.load, map_location='cpu'), strict=True)
    model.to(device).eval()
    out = []

    def run_sample(imgs, tris):
        res = model(imgs.to(device).unsqueeze(0), tris.to(device).unsqueeze(0))
        return (res[0] if isinstance(res, tuple) else res).squeeze()         # FBA returns (alphas, Fs, Bs)

    def save(i, alpha):
        info = os.path.normpath(dataset.samples[i][c][0]).split(os.sep)
        outfn = os.path.join(args.save, info[-2], info[-1][:-8] + '_alpha.png')
        os.makedirs(os.path.dirname(outfn), exist_ok=True)
        Image.fromarray(np.uint8(alpha * 255)).save(outfn)
        out.append(outfn)

    first, last = indices
    if args.model == 'vmn_gca' and not args.per_sample:
        # whole clips: every frame goes through the encoder once and its features serve the three windows containing
        # it (EvalModel.forward_video) -- same outputs as the per-sample loop below at a third of the encoder work
        i = first
        while i < last:
            vid = os.path.dirname(dataset.samples[i][c][0])
            j = i
            while j < last and os.path.dirname(dataset.samples[j][c][0]) == vid:
                j += 1
            whole = (i == 0 or os.path.dirname(dataset.samples[i - 1][c][0]) != vid) and \
                    (j == len(dataset) or os.path.dirname(dataset.samples[j][c][0]) != vid)
            if whole and j - i >= 2:
                frames = [dataset.centre(k) for k in range(i, j)]     # one decode per frame
                H, W = frames[0][2]
                alphas = model.forward_video(torch.stack([f[0] for f in frames]), torch.stack([f[1] for f in frames]))
                for k in range(i, j):
                    save(k, alphas[k - i, 0, :H, :W].cpu().numpy())
            else:                                       # a clip split across workers: per-sample windows
                for k in range(i, j):
                    imgs, tris, (H, W) = dataset[k]
                    save(k, run_sample(imgs, tris)[c][:H, :W].cpu().numpy())
            i = j
        return out
    for i in range(first, last):
        imgs, tris, (H, W) = dataset[i]
        save(i, run_sample(imgs, tris)[c][:H, :W].cpu().numpy())
    return out


def main(args):
    if args.save is None:
        args.save = 'test_results/{}'.format(os.path.splitext(args.load)[0])
    os.makedirs(args.save, exist_ok=True)
    dataset = TestFolder(args.data, args.videos)
    return pred(dataset, (0, len(dataset)), torch.device('cuda', args.gpu), args)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data', required=True)
    ap.add_argument('--videos', nargs='*', default=[])
    ap.add_argument('--load', required=True)
    ap.add_argument('--save', default=None)
    ap.add_argument('--model', default='vmn_gca')
    ap.add_argument('--agg_window', type=int, default=7)
    ap.add_argument('--dilation', type=int, default=None)
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--per_sample', action='store_true', help='one EvalModel call per 3-frame sample (the reference loop) instead of whole clips with cached features')
    return ap.parse_args(argv)


if __name__ == '__main__':
    main(parse())
