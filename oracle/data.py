"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's loader math, dataset/VMD.py, file-list neighbourhoods (`parse`, VMD.py:167-181), `img_crop_and_resize` (VMD.py:62-66),
`possible_pad` (VMD.py:187-200), the crop search of `shape_aug` (VMD.py:131-152) and `__getitem__` (VMD.py:202-301)
without the imgaug colour / JPEG augmentation, and (round 3) its optical-flow branch: `flow_crop_and_resize` (VMD.py:68-126), the flow files
of a sample (VMD.py:203-213, 236-245) and their crop / resize / padding in training and validation (VMD.py:153-165, 267-291).

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


def shape_aug(fg, bg, a, image_shape, video_shape, scales=(1.0, 1.25, 1.5, 1.75, 2.0), return_crop=False):
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
    if return_crop:
        return pfg, pbg, pa, (top, left, n)          # the flows of the sample are cropped with the same window (VMD.py:153-160)
    return pfg, pbg, pa


def get_item(root, frame_corr, sample, mode, image_shape, video_shape, precomputed=False, read_flow=None):
    """The (fg, bg, a) of one sample; `sample` = list of frame names; consumes python `random` like the reference.
    read_flow(name_a, name_b) -> what cv2.imread returns for flow_png/<clip>/flow_<a>_<b>.png (uint16 [H, W, 3], B, G, R): the
    optical-flow branch, (fg, bg, a, wb, wf) is returned (VMD.py:236-245, 293-300)."""
    from PIL import Image
    if mode == 'train' and random.random() > 0.5:
        sample = sample[::-1]
    crop = None
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
        fg, bg, a, crop = shape_aug(fg, bg, a, image_shape, video_shape, return_crop=True)
    elif precomputed:
        fg = [possible_pad(torch.from_numpy(x).permute(2, 0, 1), image_shape, IMG_PADDING_VALUE) for x in fg]
        bg = [possible_pad(torch.from_numpy(x).permute(2, 0, 1), image_shape, IMG_PADDING_VALUE) for x in bg]
        a = [possible_pad(torch.from_numpy(x).permute(2, 0, 1), image_shape) for x in a]
    else:
        fg = [img_crop_and_resize(x, image_shape, 0, 0).squeeze(0) for x in fg]
        bg = [img_crop_and_resize(x, image_shape, 0, 0).squeeze(0) for x in bg]
        a = [img_crop_and_resize(x, image_shape, 0, 0).squeeze(0) for x in a]
    out = (torch.stack(fg).float(), torch.stack(bg).float(), torch.stack(a).float())
    if read_flow is None:
        return out
    base = [os.path.splitext(os.path.basename(fn))[0] for fn in sample]
    slots_b, slots_f = flow_slots(len(sample))
    fb = {i: flow_from_png16(read_flow(base[p], base[q])) for i, (p, q) in slots_b.items()}
    ff = {i: flow_from_png16(read_flow(base[p], base[q])) for i, (p, q) in slots_f.items()}
    wb, wf = flows_of_sample(fb, ff, mode, image_shape, crop, precomputed)
    return out + (wb, wf)


# ----------------------------------------------------------------------------- optical-flow branch (VMD.py:68-126, 203-291)
FLOW_QUANTIZATION_SCALE = 100
NAN = float('nan')


def _pixel_grid(h, w):
    """[1, 2, h, w]: channel 0 = column index, channel 1 = row index (utils/utils.py:69-72)."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    return torch.stack([xs, ys], 0).unsqueeze(0)


def _sample_at(img, xy):
    """Bilinear lookup of img [1, C, H, W] at pixel coordinates xy [1, 2, h, w] (utils/utils.py:74-88: grid_sample,
    align_corners=True, on coordinates scaled to [-1, 1])."""
    H, W = img.shape[-2:]
    gx = 2 * xy[:, :1] / (W - 1) - 1
    gy = 2 * xy[:, 1:] / (H - 1) - 1
    return F.grid_sample(img, torch.cat([gx, gy], 1).permute(0, 2, 3, 1), mode='bilinear', align_corners=True)


def _neighbours_agree(fa, fb):
    """The reference's 'gradient check' between two adjacent flow vectors (VMD.py:76-92): directions within 45 degrees (or one
    of them shorter than a pixel on average / zero) AND lengths within 50 pixels.  Returns int [h, w]."""
    fa, fb = fa.squeeze(), fb.squeeze()
    dot = (fa * fb).sum(0)
    na, nb = torch.sqrt((fa ** 2).sum(0)), torch.sqrt((fb ** 2).sum(0))
    nab = na * nb
    angle = torch.acos((dot / nab).abs().clamp(0, 1.0 - 1e-6))
    ok = angle <= np.pi / 4
    ok[nab == 0] = True
    ok[(na + nb) < 2] = True
    return (ok * (torch.abs(na - nb) < 50)).int()


def flow_crop_and_resize(flow, image_shape, ph, pw, nsize=None):
    """flow [H, W, 2] (pixels, NaN = invalid) -> [1, 2, h, w] for the crop (ph, pw, nsize) resized to image_shape: bilinear
    resampling (corner-aligned), vectors rescaled to the new pixel size, NaN where the source neighbourhood is not smooth
    (a motion boundary) or the vector points outside the new frame."""
    if nsize is not None:
        flow = flow[ph:ph + nsize[0], pw:pw + nsize[1]]
    else:
        nsize = (flow.shape[0], flow.shape[1])
    Ho, Wo = image_shape
    flow = flow.permute(2, 0, 1).unsqueeze(0)
    down = F.pad(_neighbours_agree(flow[:, :, :-1, :], flow[:, :, 1:, :]), (0, 0, 0, 1), value=1)
    right = F.pad(_neighbours_agree(flow[:, :, :, :-1], flow[:, :, :, 1:]), (0, 1, 0, 0), value=1)
    smooth = right * down                                                  # [H, W]
    grid = _pixel_grid(Ho, Wo).float()
    src = torch.cat([grid[:, :1] * ((nsize[1] - 1) / float(Wo - 1)), grid[:, 1:] * ((nsize[0] - 1) / float(Ho - 1))], 1)
    out = _sample_at(flow, src)
    cw, ch = torch.floor(src).split(1, dim=1)
    keep = smooth[(ch.squeeze().long(), cw.squeeze().long())][None, None, ...]
    out = torch.where(keep.bool(), out, torch.tensor(NAN))
    out[:, 0] /= nsize[1] / float(Wo)
    out[:, 1] /= nsize[0] / float(Ho)
    landed = (grid + out).squeeze(0)
    outside = ((landed[0] < 0) + (landed[1] < 0) + (landed[0] > Wo - 1) + (landed[1] > Ho - 1)).bool()
    out[outside[None, None, ...].repeat(1, 2, 1, 1)] = torch.tensor(NAN)
    return out


def flow_from_png16(x_bgr):
    """What the reference makes of cv2.imread(flow png, IMREAD_UNCHANGED) (VMD.py:204-213): channels 0, 1 (cv2 order: B, G) are
    the int16 displacement x 100, the last channel is the validity mask; invalid pixels become NaN."""
    flow = np.float32(np.int16(x_bgr[..., :-1]))
    flow[x_bgr[..., -1] == 0] = np.nan
    return torch.from_numpy(flow) / FLOW_QUANTIZATION_SCALE


def flow_slots(length):
    """Which (frame, direction) pairs of a sample carry a flow (VMD.py:241-245): forward wf[i] = i -> i + 1 and backward
    wb[i] = i -> i - 1 for the inner frames 2 .. length - 3, plus wf[1] and wb[length - 2]."""
    wf = {i: (i, i + 1) for i in range(2, length - 2)}
    wb = {i: (i, i - 1) for i in range(2, length - 2)}
    wf[1] = (1, 2)
    wb[length - 2] = (length - 2, length - 3)
    return wb, wf


def flows_of_sample(flows_b, flows_f, mode, image_shape, crop=None, precomputed=False):
    """flows_b / flows_f: {frame index: [H, W, 2] tensor} of the slots of `flow_slots`; returns (wb, wf) [S, 2, h, w] with NaN
    planes in the empty slots (VMD.py:153-165 training, 274-291 validation).  crop = (ph, pw, nsize) of the training crop."""
    S = max(list(flows_b) + list(flows_f)) + 2
    out = []
    for slots in (flows_b, flows_f):
        planes = {}
        for i, fl in slots.items():
            if mode == 'train':
                planes[i] = flow_crop_and_resize(fl, image_shape, crop[0], crop[1], crop[2]).squeeze(0)
            elif precomputed:
                planes[i] = possible_pad(fl.permute(2, 0, 1), image_shape, NAN)
            else:
                planes[i] = flow_crop_and_resize(fl, image_shape, 0, 0).squeeze(0)
        ref = next(iter(planes.values()))
        out.append(torch.stack([planes[i] if i in planes else torch.full_like(ref, NAN) for i in range(S)]).float())
    return out[0], out[1]
