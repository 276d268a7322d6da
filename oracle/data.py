"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's loader math, dataset/VMD.py, for
the no_flow path — file-list neighbourhoods (`parse`, VMD.py:167-181), `img_crop_and_resize` (VMD.py:62-66),
`possible_pad` (VMD.py:187-200), the crop search of `shape_aug` (VMD.py:131-152) and `__getitem__` (VMD.py:202-301)
without the imgaug colour / JPEG augmentation.

Parity status: PINNED.  tests/golden/gen_data_golden.py imports the reference module in the build container behind import-time
stubs of cv2 (imread through PIL, channels swapped to BGR(A)) and imgaug (identity augmenters) and stores what its parse /
img_crop_and_resize / possible_pad / shape_aug / __getitem__ return on a tiny synthetic clip tree (tests/golden/data_loader.npz);
tests/test_oracle_golden.py::test_data_loader_golden replays the same inputs through this file: bit exact, including the
position of python's `random` after the crop search.  Every numeric step is the SAME torch call the reference makes
(`F.interpolate(..., mode='bilinear', align_corners=True)`, `torch.floor(x + 0.5)`, `F.pad`) on the same uint8 pixel values
(PIL decodes RGB(A); cv2 would give BGR(A), the channel order is swapped back below).
"""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

IMG_PADDING_VALUE = [103.53, 116.28, 123.675]


def _mirrored(i, n):
    """Frame index i of an n-frame clip with the clip reflected about its first / last frame (no repeated end frame)."""
    if i < 0:
        return -i
    if i >= n:
        return 2 * (n - 1) - i
    return i


def parse(frame_corr, video_names, length):
    """VMD.py:167-181: one sample per frame, the frame in the middle and length // 2 neighbours per side; neighbours beyond the
    ends of the clip are the mirrored frames."""
    frames = {}
    for name in sorted(frame_corr):
        frames.setdefault(os.path.dirname(name), []).append(name)
    half = length // 2
    out = []
    for v in video_names:
        clip = frames.get(v.strip(), [])
        for centre in range(len(clip)):
            out.append([clip[_mirrored(centre + off, len(clip))] for off in range(-half, length - half)])
    return out


def img_crop_and_resize(img, image_shape, ph, pw, nsize=None, threads=1):
    """img: float32 numpy [H, W, C] -> [1, C, h, w].  `threads`: ATen rounds 3-channel images differently when the process has
    one thread (channels-last kernel) than when it has several (generic kernel); the reference's loader runs inside DataLoader
    workers, which torch pins to one thread, hence the default."""
    img2 = img[ph:ph + nsize[0], pw:pw + nsize[1]] if nsize is not None else img
    img2 = torch.from_numpy(img2).permute(2, 0, 1).unsqueeze(0)
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        return torch.floor(F.interpolate(img2, list(image_shape), mode='bilinear', align_corners=True) + 0.5)
    finally:
        torch.set_num_threads(before)


def possible_pad(t, image_shape, padvalue=0):
    """VMD.py:187-200: grow [C, H, W] to image_shape at the bottom / right; the new pixels get padvalue (scalar or per channel)."""
    C, H, W = t.shape[-3:]
    Ho, Wo = image_shape
    if (H, W) == (Ho, Wo):
        return t
    assert H <= Ho and W <= Wo
    fill = torch.as_tensor(padvalue, dtype=t.dtype).reshape(-1, 1, 1).expand(C, Ho, Wo)
    out = fill.clone()
    out[:, :H, :W] = t
    return out


def shape_aug(fg, bg, a, image_shape, video_shape, scales=(1.0, 1.25, 1.5, 1.75, 2.0)):
    """VMD.py:131-152: draw (scale, top, left) with python `random` until EVERY frame's resized alpha keeps an unknown pixel
    (0 < a < 255); the reference stops resizing a candidate at its first frame without one, which draws nothing further."""
    H, W = video_shape
    while True:
        s = random.choice(list(scales))
        n = (int(image_shape[0] * s), int(image_shape[1] * s))
        top = random.randint(0, H - n[0] - 1)
        left = random.randint(0, W - n[1] - 1)
        pa = [img_crop_and_resize(x, image_shape, top, left, n).squeeze(0) for x in a]
        if all(int(((p > 0) & (p < 255)).sum()) >= 1 for p in pa):
            break
    pfg = [img_crop_and_resize(x, image_shape, top, left, n).squeeze(0) for x in fg]
    pbg = [img_crop_and_resize(x, image_shape, top, left, n).squeeze(0) for x in bg]
    return pfg, pbg, pa


def get_item(root, frame_corr, sample, mode, image_shape, video_shape, precomputed=False):
    """The (fg, bg, a) of one sample; `sample` = list of frame names; consumes python `random` like the reference."""
    from PIL import Image
    if mode == 'train' and random.random() > 0.5:
        sample = sample[::-1]
    fg, bg, a = [], [], []
    for fn in sample:
        with Image.open(os.path.join(root, 'FG_done', fn)) as im:
            f = np.asarray(im.convert('RGBA'))
        bgp = os.path.join(root, 'BG_done', frame_corr[fn])
        if not os.path.exists(bgp):
            bgp = os.path.splitext(bgp)[0] + '.png'
        with Image.open(bgp) as im:
            b = np.asarray(im.convert('RGB'))
        fg.append(np.float32(f[..., [2, 1, 0]]))          # cv2 channel order
        bg.append(np.float32(b[..., [2, 1, 0]]))
        a.append(np.float32(f[..., 3:]))
    if mode == 'train':
        fg, bg, a = shape_aug(fg, bg, a, image_shape, video_shape)
    elif precomputed:
        fg = [possible_pad(torch.from_numpy(x).permute(2, 0, 1), image_shape, IMG_PADDING_VALUE) for x in fg]
        bg = [possible_pad(torch.from_numpy(x).permute(2, 0, 1), image_shape, IMG_PADDING_VALUE) for x in bg]
        a = [possible_pad(torch.from_numpy(x).permute(2, 0, 1), image_shape) for x in a]
    else:
        fg = [img_crop_and_resize(x, image_shape, 0, 0).squeeze(0) for x in fg]
        bg = [img_crop_and_resize(x, image_shape, 0, 0).squeeze(0) for x in bg]
        a = [img_crop_and_resize(x, image_shape, 0, 0).squeeze(0) for x in a]
    return torch.stack(fg).float(), torch.stack(bg).float(), torch.stack(a).float()
