#!/usr/bin/env python
"""Generate tests/golden/data_loader.npz: outputs of the REFERENCE loader (dataset/VMD.py) on a tiny synthetic clip tree.

Runs ONLY in the build container.  The reference module imports cv2 and imgaug, which this image lacks; both are stubbed at
import time (as gen_golden.py does for cv2):
  * cv2.imread -> PIL decode, channels swapped to cv2's BGR(A) order (PNG is lossless: same uint8 values);
  * imgaug's augmenters -> identity (`to_deterministic().augment_image(x) == x`): the fixtures pin the loader's own arithmetic
    -- file-list neighbourhoods, crop + bilinear resize + round, padding, the crop search and its use of python `random` --
    not imgaug's colour / JPEG augmentation (oracle/data.py restates the loader without it too).
The reference functions are called as they are: VideoMattingDataset.parse / img_crop_and_resize / possible_pad / shape_aug /
__getitem__ (dataset/VMD.py:62-66, 128-152, 167-181, 187-200, 202-301).  Stored: the synthetic frames (inputs) and the
reference's outputs.  tests/test_oracle_golden.py::test_data_loader_golden replays them through oracle/data.py.

    python tests/golden/gen_data_golden.py
"""
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def _import_reference_loader():
    from PIL import Image
    cv2 = types.ModuleType('cv2')
    cv2.IMREAD_UNCHANGED, cv2.IMREAD_COLOR, cv2.IMREAD_GRAYSCALE = -1, 1, 0

    def imread(path, flag=1):
        if os.path.basename(path).startswith('flow_'):          # 16-bit 3-channel flow file: PIL cannot decode it
            from tcvom_amd.data import png16_decode              # (a plain PNG decoder; returns FILE order R, G, B -> cv2's B, G, R)
            return np.ascontiguousarray(png16_decode(path)[..., ::-1])
        with Image.open(path) as im:
            if flag == cv2.IMREAD_GRAYSCALE:
                return np.asarray(im.convert('L')).copy()
            if flag == cv2.IMREAD_COLOR or im.mode == 'RGB':
                return np.asarray(im.convert('RGB'))[..., ::-1].copy()
            assert im.mode == 'RGBA', im.mode
            return np.asarray(im)[..., [2, 1, 0, 3]].copy()
    cv2.imread = imread
    cv2.setNumThreads = lambda n: None

    class _Identity(object):
        def __init__(self, *a, **k):
            pass

        def to_deterministic(self):
            return self

        def augment_image(self, x):
            return x
    imgaug = types.ModuleType('imgaug')
    iaa = types.ModuleType('imgaug.augmenters')
    iap = types.ModuleType('imgaug.parameters')
    for name in ('Sequential', 'MultiplyHueAndSaturation', 'GammaContrast', 'AddToHue', 'Sometimes', 'JpegCompression'):
        setattr(iaa, name, _Identity)
    iap.TruncatedNormal = lambda *a, **k: None
    imgaug.augmenters, imgaug.parameters = iaa, iap
    tv = types.ModuleType('torchvision')
    tv.utils = types.ModuleType('torchvision.utils')
    sys.modules.update({'cv2': cv2, 'imgaug': imgaug, 'imgaug.augmenters': iaa, 'imgaug.parameters': iap,
                        'torchvision': tv, 'torchvision.utils': tv.utils})
    torch.cuda.current_device = lambda: torch.device('cpu')
    sys.path = [p for p in sys.path if os.path.abspath(p or '.') != REPO]
    sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k == 'dataset' or k.startswith('dataset.') or k == 'utils' or k.startswith('utils.')]:
        del sys.modules[m]
    import dataset.VMD as ref_vmd
    sys.path.remove(REF)
    sys.path.insert(0, REPO)
    return ref_vmd


def make_frames(videos, nframes, H, W, seed):
    """uint8 RGBA foregrounds (soft disc alpha: foreground, unknown ring, background) and RGB backgrounds."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    fg = np.zeros((len(videos), nframes, H, W, 4), np.uint8)
    bg = np.zeros((len(videos), nframes, H, W, 3), np.uint8)
    for v in range(len(videos)):
        for k in range(nframes):
            fg[v, k, ..., :3] = rng.randint(0, 256, (H, W, 3))
            d = np.sqrt((xx - W * (0.4 + 0.03 * k + 0.1 * v)) ** 2 + (yy - H * 0.5) ** 2)
            fg[v, k, ..., 3] = np.clip((min(H, W) * 0.3 - d) * 12 + 128, 0, 255)
            bg[v, k] = rng.randint(0, 256, (H, W, 3))
    return fg, bg


def write_tree(root, videos, fg, bg):
    """The VideoMatting108 layout the loader reads (tests/test_oracle_golden.py writes the same tree from the fixture)."""
    from PIL import Image
    corr = {}
    for v, name in enumerate(videos):
        os.makedirs(os.path.join(root, 'FG_done', name), exist_ok=True)
        os.makedirs(os.path.join(root, 'BG_done', 'bg_' + name), exist_ok=True)
        for k in range(fg.shape[1]):
            Image.fromarray(fg[v, k], 'RGBA').save(os.path.join(root, 'FG_done', name, '%04d.png' % k))
            Image.fromarray(bg[v, k], 'RGB').save(os.path.join(root, 'BG_done', 'bg_' + name, '%04d.png' % k))
            corr['%s/%04d.png' % (name, k)] = 'bg_%s/%04d.jpg' % (name, k)      # names a .jpg: the .png fallback of VMD.py:229-231
    with open(os.path.join(root, 'frame_corr.json'), 'w') as f:
        json.dump(corr, f)
    for mode in ('train', 'val'):
        with open(os.path.join(root, '%s_videos.txt' % mode), 'w') as f:
            f.write('\n'.join(videos) + '\n')
    return corr


VIDEOS = ['va', 'vb']
NFRAMES, H, W = 4, 48, 64
CROP = [16, 16]                     # train crop (square: shape_aug asserts it)
PAD_SHAPE = [56, 72]                # precomputed-validation canvas
VAL_SHAPE = [24, 40]
SEEDS = (1234, 7, 99)


def main(ref):
    torch.set_num_threads(1)        # the loader runs in DataLoader workers, which torch pins to one thread (ATen's 3-channel
    # bilinear kernel rounds differently with several threads; oracle/data.py: `threads`)
    DS = ref.VideoMattingDataset
    fg, bg = make_frames(VIDEOS, NFRAMES, H, W, seed=0)
    root = tempfile.mkdtemp()
    try:
        write_tree(root, VIDEOS, fg, bg)
        out = {'fg': fg, 'bg': bg}
        for length in (3, 5):
            ds = DS(root, VAL_SHAPE, False, 'val', no_flow=True, sample_length=length)
            out['parse_%d' % length] = np.frombuffer(json.dumps(ds.samples).encode(), dtype=np.uint8)
        dsp1 = DS(root, CROP, True, 'val', no_flow=True, sample_length=3)
        out['plus1_shape'] = np.array(dsp1.image_shape)
        # img_crop_and_resize
        ds = DS(root, CROP, False, 'train', no_flow=True, sample_length=3)
        img = np.float32(fg[0, 1][..., [2, 1, 0]])
        alpha = np.float32(fg[0, 1][..., 3:])
        cases = [(0, 0, None), (5, 9, (32, 32)), (3, 7, (20, 20)), (10, 20, (24, 24))]
        out['resize_cases'] = np.array([[c[0], c[1]] + list(c[2] or (-1, -1)) for c in cases])
        for i, (ph, pw, n) in enumerate(cases):
            out['resize_img_%d' % i] = ds.img_crop_and_resize(img, ph, pw, n).numpy()
            out['resize_a_%d' % i] = ds.img_crop_and_resize(alpha, ph, pw, n).numpy()
        # possible_pad
        dpad = DS(root, PAD_SHAPE, False, 'val', no_flow=True, precomputed_val=root, sample_length=3)
        t3 = torch.from_numpy(img).permute(2, 0, 1)
        t1 = torch.from_numpy(alpha).permute(2, 0, 1)
        out['pad_img'] = dpad.possible_pad(t3.clone(), [103.53, 116.28, 123.675]).numpy()
        out['pad_a'] = dpad.possible_pad(t1.clone()).numpy()
        # shape_aug: the crop search and its consumption of python `random`
        DS.VIDEO_SHAPE = (H, W)
        f3 = [np.float32(fg[1, k][..., [2, 1, 0]]) for k in range(3)]
        b3 = [np.float32(bg[1, k][..., ::-1]) for k in range(3)]
        a3 = [np.float32(fg[1, k][..., 3:]) for k in range(3)]
        for s in SEEDS:
            random.seed(s)
            pfg, pbg, pa, _, _ = ds.shape_aug(f3, b3, a3)
            out['aug_fg_%d' % s] = torch.stack(pfg).numpy()
            out['aug_bg_%d' % s] = torch.stack(pbg).numpy()
            out['aug_a_%d' % s] = torch.stack(pa).numpy()
            out['aug_next_random_%d' % s] = np.array([random.random()])       # where the python RNG stands afterwards
        # __getitem__: validation by resize, validation by padding (precomputed), training (flip + crop search)
        dval = DS(root, VAL_SHAPE, False, 'val', no_flow=True, sample_length=3)
        for idx in (0, 5):
            g = dval[idx]
            out['val_%d_fg' % idx], out['val_%d_bg' % idx], out['val_%d_a' % idx] = [t.numpy() for t in g[:3]]
            assert int(g[3]) == idx
            g = dpad[idx]
            out['pad_%d_fg' % idx], out['pad_%d_bg' % idx], out['pad_%d_a' % idx] = [t.numpy() for t in g[:3]]
        for s in SEEDS:
            random.seed(s)
            g = ds[3]
            out['train_%d_fg' % s], out['train_%d_bg' % s], out['train_%d_a' % s] = [t.numpy() for t in g[:3]]
        np.savez_compressed(os.path.join(HERE, 'data_loader.npz'), **out)
        print('wrote data_loader.npz with %d arrays' % len(out))
    finally:
        shutil.rmtree(root)


def make_flows(nframes, H, W, seed):
    """int16 flow x 100 (x, y) + validity for every ordered pair of adjacent frames: a smooth field, a motion boundary (the
    'gradient check' of VMD.py:76-92 must reject its neighbourhood), a block of invalid pixels, a patch pointing out of the frame."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    flows = {}
    for a in range(nframes):
        for b in (a - 1, a + 1):
            if not 0 <= b < nframes:
                continue
            sgn = 1.0 if b > a else -1.0
            fx = sgn * (2.0 + 0.05 * xx + 0.5 * np.sin(yy / 7.0 + a)) + rng.uniform(-0.2, 0.2, (H, W))
            fy = sgn * (-1.0 + 0.03 * yy + 0.5 * np.cos(xx / 9.0 + b)) + rng.uniform(-0.2, 0.2, (H, W))
            fx[:, W // 2:] += sgn * 9.0 * (yy[:, W // 2:] > H // 3)          # a motion boundary
            fy[H // 2:, : W // 4] = -sgn * 40.0                              # points out of the frame after the crop
            valid = np.ones((H, W), bool)
            valid[5 + a:11 + a, 20:33] = False
            q = np.stack([np.round(fx * 100), np.round(fy * 100)], -1).astype(np.int16)
            flows[(a, b)] = (q, valid)
    return flows


def write_flows(root, videos, flows):
    """flow_png/<video>/flow_<a>_<b>.png in OpenCV's layout: the FILE holds R = validity, G = y, B = x (cv2 reads B, G, R)."""
    from tcvom_amd.data import write_png16
    for name in videos:
        os.makedirs(os.path.join(root, 'flow_png', name), exist_ok=True)
        for (a, b), (q, valid) in flows.items():
            img = np.stack([np.where(valid, 65535, 0).astype(np.uint16), q[..., 1].view(np.uint16), q[..., 0].view(np.uint16)], -1)
            write_png16(os.path.join(root, 'flow_png', name, 'flow_%04d_%04d.png' % (a, b)), img)


def gen_flow(ref):
    """tests/golden/data_loader_flow.npz: the reference loader's optical-flow branch (VMD.py:68-126 flow_crop_and_resize, the flow
    files of a sample :203-245, training crop :153-165, validation :274-291) on the synthetic tree + synthetic flow files."""
    torch.set_num_threads(1)
    DS = ref.VideoMattingDataset
    fg, bg = make_frames(VIDEOS, NFRAMES, H, W, seed=0)
    flows = make_flows(NFRAMES, H, W, seed=5)
    root = tempfile.mkdtemp()
    try:
        write_tree(root, VIDEOS, fg, bg)
        write_flows(root, VIDEOS, flows)
        out = {'flow_pairs': np.array(sorted(flows)), 'flow_q': np.stack([flows[k][0] for k in sorted(flows)]),
               'flow_valid': np.stack([flows[k][1] for k in sorted(flows)])}
        DS.VIDEO_SHAPE = (H, W)
        ds = DS(root, CROP, False, 'train', no_flow=False, sample_length=3)
        q, valid = flows[(1, 2)]
        fl = np.float32(q)
        fl[~valid] = np.nan
        fl = torch.from_numpy(fl) / 100
        cases = [(0, 0, None), (5, 9, (32, 32)), (3, 7, (20, 20)), (10, 20, (24, 24)), (0, 0, (16, 16))]
        out['fcr_cases'] = np.array([[c[0], c[1]] + list(c[2] or (-1, -1)) for c in cases])
        for i, (ph, pw, n) in enumerate(cases):
            out['fcr_%d' % i] = ds.flow_crop_and_resize(fl.clone(), ph, pw, n).numpy()
        dval = DS(root, VAL_SHAPE, False, 'val', no_flow=False, sample_length=3)
        out['fcr_val'] = dval.flow_crop_and_resize(fl.clone(), 0, 0).numpy()
        # __getitem__ with flows: validation by resize (S = 3 and 5), validation by padding, training
        for length in (3, 5):
            dv = DS(root, VAL_SHAPE, False, 'val', no_flow=False, sample_length=length)
            for idx in (0, 2, 5):
                g = dv[idx]
                assert len(g) == 6 and int(g[5]) == idx
                out['val%d_%d_wb' % (length, idx)], out['val%d_%d_wf' % (length, idx)] = g[3].numpy(), g[4].numpy()
                out['val%d_%d_a' % (length, idx)] = g[2].numpy()
        dpad = DS(root, PAD_SHAPE, False, 'val', no_flow=False, precomputed_val=root, sample_length=3)
        g = dpad[1]
        out['pad_1_wb'], out['pad_1_wf'] = g[3].numpy(), g[4].numpy()
        for length in (3, 5):
            dt = DS(root, CROP, False, 'train', no_flow=False, sample_length=length)
            for s in SEEDS:
                random.seed(s)
                g = dt[3]
                out['train%d_%d_wb' % (length, s)], out['train%d_%d_wf' % (length, s)] = g[3].numpy(), g[4].numpy()
                out['train%d_%d_a' % (length, s)] = g[2].numpy()
                out['train%d_next_random_%d' % (length, s)] = np.array([random.random()])
        np.savez_compressed(os.path.join(HERE, 'data_loader_flow.npz'), **out)
        print('wrote data_loader_flow.npz with %d arrays' % len(out))
    finally:
        shutil.rmtree(root)


if __name__ == '__main__':
    sys.path.insert(0, REPO)
    import tcvom_amd.data  # noqa: F401  (the PNG16 helpers of the cv2 stub, imported before the reference takes over `dataset` / `utils`)
    _ref = _import_reference_loader()
    main(_ref)
    gen_flow(_ref)
