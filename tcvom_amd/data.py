"""Data front-end with the reference's loader contract (dataset/VMD.py: `VideoMattingDataset`), MI355X-first: the host only
decodes PNG files into uint8 frames (`load_raw`, safe in DataLoader workers); crop + bilinear resize + rounding, the
unknown-pixel test of the crop search and the validation padding run on the device over the S frames of a clip per launch
(`transform`, csrc/frontend.hip), so a 1080p clip crosses PCIe once as 7 bytes / pixel instead of 28.

    ds = VideoMattingDataset(root, [320, 320], plus1=False, mode='train', no_flow=True)
    fg, bg, a, idx = ds[0]          # float32 0..255, BGR, [S, 3|1, H, W] on the device  (VMD.py:293-301)

    # with worker processes: decode in the workers, finish on the device in the training process
    loader = DataLoader(ds.raw_view(), batch_size=None, num_workers=4)
    for raw in loader: fg, bg, a, idx = ds.transform(raw)

Directory layout, file lists, frame neighbourhoods, the random draws of the crop search (python `random`, same order as the
reference) and every rounding step follow dataset/VMD.py; line references below.  Not provided: the optical-flow branch
(`no_flow=False`; every caller in the reference passes `no_flow=True`) and the imgaug colour / JPEG augmentation
(VMD.py:50-55,253-262; imgaug is not part of this image) — training crops are geometric only.
"""
import ctypes as C
import json
import os
import random

import numpy as np
import torch
import torch.utils.data

from . import _lib as L

IMG_PADDING_VALUE = [103.53, 116.28, 123.675]        # BGR, VMD.py:265


def _read_png(path, mode):
    from PIL import Image                             # decode only; PIL returns RGB(A), the kernels reorder to BGR
    with Image.open(path) as im:
        return np.asarray(im.convert(mode))


class _RawView(torch.utils.data.Dataset):
    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, idx):
        return self.ds.load_raw(idx)


class VideoMattingDataset(torch.utils.data.Dataset):
    VIDEO_SHAPE = (1080, 1920)
    FG_FOLDER = 'FG_done'
    BG_FOLDER = 'BG_done'
    SCALES = [1.0, 1.25, 1.5, 1.75, 2.0]              # VMD.py:131

    def __init__(self, data_root, image_shape, plus1, mode, use_subset=False, no_flow=False, precomputed_val=None,
                 sample_length=5, device=None, worker_arithmetic=True):
        if not no_flow:
            raise NotImplementedError('the optical-flow branch of dataset/VMD.py is not part of this front-end: pass no_flow=True '
                                      '(as train_ddp.py, pred_vmn.py and pred_single.py do)')
        assert mode in ('train', 'val')
        if precomputed_val is not None:
            assert mode == 'val'
        self.no_flow, self.mode, self.precomputed_val, self.sample_length = no_flow, mode, precomputed_val, sample_length
        self.data_root = data_root
        self.image_shape = [image_shape[0] + 1, image_shape[1] + 1] if plus1 else list(image_shape)
        setname = ('{}_videos_subset.txt' if use_subset else '{}_videos.txt').format(mode)
        with open(os.path.join(data_root, 'frame_corr.json'), 'r') as f:
            self.frame_corr = json.load(f)
        with open(os.path.join(data_root, setname), 'r') as f:
            self.samples = self.parse(f)
        self.dataset_length = len(self.samples)
        self.device = torch.device(device) if device is not None else None
        # F.interpolate on the host rounds 3-channel images differently in a single-threaded process (ATen picks its
        # channels-last kernel there, csrc/frontend.hip): the reference loads through DataLoader workers, which are
        # single-threaded, so that is the default; False = the arithmetic of a num_workers=0 run on a multi-core host.
        self.image_form = 1 if worker_arithmetic else 0

    def __len__(self):
        return self.dataset_length

    # ------------------------------------------------------------------ file lists (host)
    def parse(self, f, length=None):
        """One sample per frame: the frame in the middle, length // 2 neighbours on each side, mirrored at the clip ends
        (VMD.py:167-181)."""
        length = self.sample_length if length is None else length
        by_dir = {}
        for k in sorted(self.frame_corr.keys()):
            by_dir.setdefault(os.path.dirname(k), []).append(k)
        samples = []
        half = length // 2
        for line in f:
            fns = by_dir.get(line.strip(), [])
            n = len(fns)
            for i in range(n):
                sample = [None] * length
                sample[half] = fns[i]
                for j in range(half):
                    lo, hi = i - j - 1, i + j + 1
                    sample[half - j - 1] = fns[lo] if lo >= 0 else fns[-lo]
                    sample[half + j + 1] = fns[hi] if hi < n else fns[n - hi - 2]
                samples.append(sample)
        return samples

    def load_raw(self, idx):
        """PNG decode of one sample: uint8 RGBA foregrounds [S, H, W, 4] and RGB backgrounds [S, H, W, 3] (VMD.py:218-236)."""
        sample = self.samples[idx]
        if self.mode == 'train' and random.random() > 0.5:
            sample = sample[::-1]
        root = self.data_root if self.precomputed_val is None else self.precomputed_val
        fg, bg = [], []
        for fn in sample:
            fg.append(_read_png(os.path.join(root, self.FG_FOLDER, fn), 'RGBA'))
            bgp = os.path.join(root, self.BG_FOLDER, self.frame_corr[fn])
            if not os.path.exists(bgp):
                bgp = os.path.splitext(bgp)[0] + '.png'
            bg.append(_read_png(bgp, 'RGB'))
            assert bg[-1].shape[:2] == fg[-1].shape[:2]
        return {'fg': torch.from_numpy(np.stack(fg)), 'bg': torch.from_numpy(np.stack(bg)), 'idx': idx}

    def raw_view(self):
        return _RawView(self)

    # ------------------------------------------------------------------ device side
    def _dev(self):
        if self.device is not None:
            return self.device
        if not torch.cuda.is_available():
            raise RuntimeError('VideoMattingDataset.transform runs on the MI355X (csrc/frontend.hip); no GPU is visible')
        return torch.device('cuda', torch.cuda.current_device())

    @staticmethod
    def _crop_resize(src, chan, ph, pw, nh, nw, Ho, Wo, form=0):
        S, Hs, Ws, Cs = src.shape
        out = torch.empty(S, len(chan), Ho, Wo, dtype=torch.float32, device=src.device)
        ch = (C.c_int32 * len(chan))(*chan)
        L.call('tcvom_crop_resize_u8', L.ptr(src), L.ptr(out), S, Hs, Ws, Cs, C.cast(ch, C.c_void_p), len(chan), ph, pw, nh, nw, Ho, Wo,
               form, L.stream_ptr())
        return out

    @staticmethod
    def _pad(t, Ho, Wo, value):
        S, Cc, H, W = t.shape
        if H == Ho and W == Wo:
            return t
        assert H <= Ho and W <= Wo                    # VMD.py:191
        out = torch.empty(S, Cc, Ho, Wo, dtype=torch.float32, device=t.device)
        v = (C.c_float * Cc)(*value) if value is not None else None
        L.call('tcvom_pad_bottom_right', L.ptr(t), L.ptr(out), S, Cc, H, W, Ho, Wo, C.cast(v, C.c_void_p) if v is not None else None,
               L.stream_ptr())
        return out

    def shape_aug(self, fg_u8):
        """Crop search of VMD.py:131-152: draw (scale, top, left) until every frame's resized alpha has an unknown pixel.
        Returns (ph, pw, nsize, alpha [S, 1, h, w])."""
        H, W = self.VIDEO_SHAPE
        assert self.image_shape[0] == self.image_shape[1]
        S = fg_u8.shape[0]
        counts = torch.empty(S, dtype=torch.int32, device=fg_u8.device)
        while True:
            scale = random.choice(self.SCALES)
            nsize = (int(self.image_shape[0] * scale), int(self.image_shape[1] * scale))
            ph = random.randint(0, H - nsize[0] - 1)
            pw = random.randint(0, W - nsize[1] - 1)
            pa = self._crop_resize(fg_u8, [3], ph, pw, nsize[0], nsize[1], self.image_shape[0], self.image_shape[1])
            L.call('tcvom_count_unknown', L.ptr(pa), S, pa[0].numel(), L.ptr(counts), L.stream_ptr())
            if min(counts.tolist()) >= 1:
                return ph, pw, nsize, pa

    def transform(self, raw):
        """uint8 frames -> (fg, bg, a, idx) of the reference's loader (VMD.py:250-301, no_flow)."""
        dev = self._dev()
        fg_u8 = raw['fg'].to(dev, non_blocking=True)
        bg_u8 = raw['bg'].to(dev, non_blocking=True)
        S, Hs, Ws, _ = fg_u8.shape
        Ho, Wo = self.image_shape
        bgr = [2, 1, 0]
        if self.mode == 'train':
            assert (Hs, Ws) == self.VIDEO_SHAPE, 'training clips are %dx%d (VMD.py:22)' % self.VIDEO_SHAPE
            ph, pw, nsize, a = self.shape_aug(fg_u8)
            fg = self._crop_resize(fg_u8, bgr, ph, pw, nsize[0], nsize[1], Ho, Wo, self.image_form)
            bg = self._crop_resize(bg_u8, bgr, ph, pw, nsize[0], nsize[1], Ho, Wo, self.image_form)
        elif self.precomputed_val is not None:
            fg = self._pad(self._crop_resize(fg_u8, bgr, 0, 0, Hs, Ws, Hs, Ws), Ho, Wo, IMG_PADDING_VALUE)
            bg = self._pad(self._crop_resize(bg_u8, bgr, 0, 0, Hs, Ws, Hs, Ws), Ho, Wo, IMG_PADDING_VALUE)
            a = self._pad(self._crop_resize(fg_u8, [3], 0, 0, Hs, Ws, Hs, Ws), Ho, Wo, None)
        else:
            fg = self._crop_resize(fg_u8, bgr, 0, 0, Hs, Ws, Ho, Wo, self.image_form)
            bg = self._crop_resize(bg_u8, bgr, 0, 0, Hs, Ws, Ho, Wo, self.image_form)
            a = self._crop_resize(fg_u8, [3], 0, 0, Hs, Ws, Ho, Wo)
        return fg, bg, a, torch.tensor(raw['idx'])

    def __getitem__(self, idx):
        return self.transform(self.load_raw(idx))
