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


def main():
    torch.set_num_threads(1)        # the loader runs in DataLoader workers, which torch pins to one thread (ATen's 3-channel
    ref = _import_reference_loader()    # bilinear kernel rounds differently with several threads; oracle/data.py: `threads`)
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


if __name__ == '__main__':
    main()
