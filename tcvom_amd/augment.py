"""Colour / JPEG augmentation of training samples (dataset/VMD.py:50-55, 253-262): per clip, foreground and background each
draw ONE parameter set (`to_deterministic()`), applied to every frame of the clip after the geometric crop:

    MultiplyHueAndSaturation(mul ~ TN(1.0, 0.2, [0.5, 1.5]))      hue and saturation of every pixel times mul
    GammaContrast(gamma ~ TN(1.0, 0.2, [0.5, 1.5]))               255 * (v / 255) ** gamma through a 256-entry table
    AddToHue(value ~ TN(0, 10, [-51, 51]))                        hue shifted by value / 255 * 180 (of 180) with wrap-around
    foreground only, with probability 0.6: JpegCompression(compression ~ U(70, 99))  -> JPEG quality 100 - compression

The reference runs imgaug (OpenCV colour conversions, PIL's JPEG codec) on the uint8 crops in its loader workers, and hands imgaug
the frames in OpenCV's B, G, R order although imgaug assumes R, G, B -- so "hue" there is the hue of the channel-swapped image;
the same is done here (the ops act on the loader's BGR tensors as they are).  The colour ops are tensor expressions on whatever
device holds the crops (the loader is plumbing, not the hot path): 8-bit HSV with H in [0, 180) as OpenCV defines it, values
rounded to integers after every stage like the uint8 round trips of the reference.  The JPEG round trip goes through Pillow on
the host (imgaug does the same): its codec IS the reference's codec.

PARITY UNPINNED for the colour ops: neither imgaug nor OpenCV exists in the build image, so no vector of the reference could be
generated; OpenCV's fixed-point RGB <-> HSV tables may differ from the float arithmetic here by one level.  The random draws use
python's `random` (truncated normals by rejection), not imgaug's numpy generator: the augmentation DISTRIBUTION is the
reference's, the individual draws are not reproducible against it.
"""
import io
import random

import numpy as np
import torch


def truncated_normal(mean, std, low, high):
    """One draw of imgaug.parameters.TruncatedNormal(mean, std, low, high)."""
    while True:
        v = random.gauss(mean, std)
        if low <= v <= high:
            return v


class ClipAugmentation(object):
    """The parameter set of one `pixel_aug.to_deterministic()` (+ the JPEG decision of `jpeg_aug.to_deterministic()`)."""

    def __init__(self, jpeg):
        self.mul = truncated_normal(1.0, 0.2, 0.5, 1.5)
        self.gamma = truncated_normal(1.0, 0.2, 0.5, 1.5)
        self.hue_add = truncated_normal(0.0, 0.1 * 100, -0.2 * 255, 0.2 * 255)
        self.jpeg_quality = None
        if jpeg and random.random() < 0.6:
            compression = random.uniform(70, 99)
            self.jpeg_quality = int(np.clip(np.round(100 - compression), 1, 100))


def rgb_to_hsv8(img):
    """float tensor [..., 3, H, W] holding integers 0..255 -> (h in [0, 180), s, v in [0, 255]) as OpenCV's 8-bit COLOR_RGB2HSV
    defines them: v = max, s = 255 (max - min) / max, h = 30 * sector position (degrees / 2), each rounded to an integer."""
    r, g, b = img[..., 0, :, :], img[..., 1, :, :], img[..., 2, :, :]
    v = torch.maximum(torch.maximum(r, g), b)
    mn = torch.minimum(torch.minimum(r, g), b)
    d = v - mn
    s = torch.where(v > 0, torch.round(255.0 * d / v.clamp_min(1.0)), torch.zeros_like(v))
    dd = d.clamp_min(1.0)
    h = torch.where(v == r, (g - b) / dd, torch.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd))
    h = torch.where(d > 0, h * 30.0, torch.zeros_like(h))
    h = torch.where(h < 0, h + 180.0, h)
    h = torch.round(h)
    h = torch.where(h >= 180.0, h - 180.0, h)
    return h, s, v


def hsv8_to_rgb(h, s, v):
    """Inverse of rgb_to_hsv8 (OpenCV's 8-bit COLOR_HSV2RGB): integers in, integers 0..255 out, stacked on dim -3."""
    hh = h / 30.0                                       # sector 0..6
    i = torch.floor(hh).clamp(0, 5)
    f = hh - i
    sf = s / 255.0
    p = v * (1.0 - sf)
    q = v * (1.0 - sf * f)
    t = v * (1.0 - sf * (1.0 - f))
    sel = lambda *c: sum(torch.where(i == k, c[k], torch.zeros_like(v)) for k in range(6))       # noqa: E731
    r = sel(v, q, p, p, t, v)
    g = sel(t, v, v, q, p, p)
    b = sel(p, p, t, v, v, q)
    return torch.round(torch.stack([r, g, b], -3)).clamp(0, 255)


def color_augment(frames, aug):
    """frames: float tensor [S, 3, H, W] of integers 0..255 (any device) -> the same after the three colour ops of `aug`."""
    x = frames.float()
    # MultiplyHueAndSaturation(mul): hue is stretched to 0..255 for the children, multiplied, wrapped; saturation multiplied, clipped
    h, s, v = rgb_to_hsv8(x)
    h255 = torch.floor(h / 180.0 * 255.0)
    h255 = torch.remainder(torch.round(h255 * aug.mul), 255.0)
    h = torch.floor(h255 / 255.0 * 180.0)
    s = torch.round(s * aug.mul).clamp(0, 255)
    x = hsv8_to_rgb(h, s, v)
    # GammaContrast: a 256-entry table, values truncated to integers
    table = torch.floor((((torch.arange(256, dtype=torch.float32, device=x.device) / 255.0) ** float(aug.gamma)) * 255.0).clamp(0, 255))
    x = table[x.long()]
    # AddToHue: the drawn value (of 255) becomes a shift of value / 255 * 180 (of 180), wrapped
    shift = float(int(aug.hue_add / 255.0 * 180.0))
    h, s, v = rgb_to_hsv8(x)
    h = torch.remainder(h + shift, 180.0)
    return hsv8_to_rgb(h, s, v)


def jpeg_round_trip(frames, quality):
    """[S, 3, H, W] integers 0..255 -> the same after PIL's JPEG encoder / decoder at `quality` (imgaug.JpegCompression does
    exactly this: Image.save(format='jpeg', quality=q), reload); host side."""
    from PIL import Image
    dev = frames.device
    arr = frames.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().numpy()
    out = np.empty_like(arr)
    for k in range(arr.shape[0]):
        buf = io.BytesIO()
        Image.fromarray(arr[k], 'RGB').save(buf, format='jpeg', quality=int(quality))
        buf.seek(0)
        out[k] = np.asarray(Image.open(buf).convert('RGB'))
    return torch.from_numpy(out).permute(0, 3, 1, 2).float().to(dev)


def augment_clip(fg, bg):
    """VMD.py:253-262: fg, bg float [S, 3, H, W] (BGR, integers 0..255) -> augmented (fg colour ops + optional JPEG, bg colour
    ops), each with its own parameter set, the same for all frames of the clip.  Consumes python `random`."""
    fa = ClipAugmentation(jpeg=True)
    ba = ClipAugmentation(jpeg=False)
    fg = color_augment(fg, fa)
    if fa.jpeg_quality is not None:
        fg = jpeg_round_trip(fg, fa.jpeg_quality)
    return fg, color_augment(bg, ba)
