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
reference) and every rounding step follow dataset/VMD.py; line references below.  The optical-flow branch (`no_flow=False`:
`fg, bg, a, wb, wf, idx`, VMD.py:203-213, 236-245, 153-165, 274-300) decodes the 16-bit flow files on the host and runs
`flow_crop_and_resize` (VMD.py:68-126) on the device.  NOT built: the colour / JPEG augmentation of training samples
(VMD.py:50-55, 253-262 -- imgaug's AddToHueAndSaturation / JpegCompression): it is outside SURVEY.md section 8, and imgaug /
OpenCV are not in this image, so its arithmetic cannot be pinned against the reference; a restatement without vectors was
removed in round 4.  Training samples get the geometric augmentation only.
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


def read_png16(path):
    """Multi-channel 16-bit PNG -> uint16 [H, W, C] in FILE channel order (R, G, B[, A]).  Pillow cannot decode these and
    OpenCV / imageio are optional: a dependency-free decoder (zlib + the five PNG row filters) is the last resort."""
    try:
        import imageio.v3 as iio
        return np.asarray(iio.imread(path))
    except ImportError:
        pass
    try:
        import cv2
        x = cv2.imread(path, cv2.IMREAD_UNCHANGED)
        if x is None:
            raise IOError('cv2 could not read %s' % path)
        return x[..., ::-1] if x.shape[-1] == 3 else x[..., [2, 1, 0, 3]]
    except ImportError:
        pass
    return png16_decode(path)


def png16_decode(path):
    """The dependency-free decoder behind read_png16."""
    import struct
    import zlib
    with open(path, 'rb') as f:
        raw = f.read()
    if raw[:8] != b'\x89PNG\r\n\x1a\n':
        raise IOError('%s is not a PNG file' % path)
    pos, idat, hdr = 8, [], None
    while pos < len(raw):
        n, typ = struct.unpack('>I4s', raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'IDAT':
            idat.append(body)
        pos += 12 + n
    W, H, depth, ctype, _, _, interlace = hdr
    nch = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 16 or nch is None or interlace:
        raise IOError('%s: unsupported PNG flavour (depth %d, colour type %d, interlace %d)' % (path, depth, ctype, interlace))
    bpp, stride = 2 * nch, 2 * nch * W
    data = zlib.decompress(b''.join(idat))
    out = np.zeros((H, stride), np.uint8)
    prev = np.zeros(stride, np.int64)
    for y in range(H):
        ft = data[y * (stride + 1)]
        line = np.frombuffer(data, np.uint8, stride, y * (stride + 1) + 1).astype(np.int64)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:                    # Sub: a running sum per byte lane
            cur = (np.cumsum(line.reshape(W, bpp), axis=0) & 255).reshape(-1)
        else:                            # Average / Paeth depend on the reconstructed left neighbour: pixel by pixel
            cur = np.zeros(stride, np.int64)
            lp, pp = line.tolist(), prev.tolist()
            c = [0] * stride
            for i in range(stride):
                left = c[i - bpp] if i >= bpp else 0
                up = pp[i]
                if ft == 3:
                    pred = (left + up) >> 1
                else:
                    ul = pp[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(up - ul), abs(left - ul), abs(left + up - 2 * ul)
                    pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
                c[i] = (lp[i] + pred) & 255
            cur = np.asarray(c, np.int64)
        out[y] = cur
        prev = cur
    return out.reshape(H, W, nch, 2).astype(np.uint16)[..., 0] << 8 | out.reshape(H, W, nch, 2)[..., 1]


def read_flow_png(path):
    """Optical-flow PNG of the VideoMatting108 tree -> float [H, W, 2] (x, y displacement in pixels, NaN where invalid), or
    None when the pair has no flow file.  The reference decodes it with cv2.imread(IMREAD_UNCHANGED) and takes x[..., :-1] as
    the flow and x[..., -1] as the validity mask (calc_metric.py:65-71); cv2 hands channels over in B, G, R order, so in FILE
    order (what every other reader returns) the layout is R = mask, G = y flow, B = x flow: convert first, slice second.
    A file that exists but cannot be decoded is an error (a silent None would zero MESSDdt)."""
    if not os.path.exists(path):
        return None
    x = read_png16(path)
    if x.ndim != 3 or x.shape[-1] not in (3, 4):
        raise IOError('%s: expected a 3- or 4-channel 16-bit flow image, got shape %s' % (path, x.shape))
    x = x[..., ::-1] if x.shape[-1] == 3 else x[..., [2, 1, 0, 3]]          # file order -> cv2 order
    flow = np.float32(np.ascontiguousarray(x[..., :-1][..., :2]).astype(np.uint16).view(np.int16))
    flow[x[..., -1] == 0] = np.nan
    return flow / 100.0


def write_png16(path, img):
    """uint16 [H, W, 3 | 4] in FILE channel order -> 16-bit PNG (no row filtering).  The flow files of a VideoMatting108 tree are
    such files (R = validity, G = y displacement, B = x displacement, see read_flow_png); Pillow cannot write them."""
    import struct
    import zlib
    img = np.ascontiguousarray(img, dtype=np.uint16)
    H, W, Cn = img.shape
    rows = img.astype('>u2').tobytes()
    stride = W * Cn * 2
    raw = b''.join(b'\x00' + rows[y * stride:(y + 1) * stride] for y in range(H))

    def chunk(t, b):
        return struct.pack('>I', len(b)) + t + b + struct.pack('>I', zlib.crc32(t + b) & 0xffffffff)
    with open(path, 'wb') as fh:
        fh.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 16, {3: 2, 4: 6}[Cn], 0, 0, 0)) +
                 chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


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
    FLOW_FOLDER = 'flow_png'                          # VMD.py:25: 16-bit PNGs, (x, y) displacement x 100 as int16 + validity
    SCALES = [1.0, 1.25, 1.5, 1.75, 2.0]              # VMD.py:131

    def __init__(self, data_root, image_shape, plus1, mode, use_subset=False, no_flow=False, precomputed_val=None,
                 sample_length=5, device=None, worker_arithmetic=True, color_aug=False):
        """color_aug: the reference's imgaug colour / JPEG augmentation of training samples (VMD.py:50-55, 253-262) is not built
        (see the module docstring); asking for it raises."""
        if color_aug:
            raise NotImplementedError('dataset.VMD: the imgaug colour / JPEG augmentation (VMD.py:253-262) is not part of this build '
                                      '(outside SURVEY.md section 8; no imgaug / OpenCV here to pin it against)')
        assert mode in ('train', 'val')
        if mode == 'train':
            import warnings
            warnings.warn('dataset.VMD (tcvom_amd.data): train-mode samples get the GEOMETRIC augmentation only (crop search, scale, '
                          'flip, pad). The reference ALWAYS applies imgaug pixel_aug + jpeg_aug to train samples '
                          '(dataset/VMD.py:50-55, 253-262); imgaug / OpenCV are absent from this image, parity could not be pinned, so '
                          'the colour / JPEG step is NOT built: a training run from disk differs from the reference recipe here. '
                          'Open gap, listed in README.md and DESIGN.md section 8.', stacklevel=2)
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
        self.color_aug = bool(color_aug)

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
        raw = {'fg': torch.from_numpy(np.stack(fg)), 'bg': torch.from_numpy(np.stack(bg)), 'idx': idx}
        if not self.no_flow:
            # VMD.py:236-245: forward flows wf[i] = i -> i + 1 and backward flows wb[i] = i -> i - 1 of the inner frames
            # 2 .. S - 3, plus wf[1] and wb[S - 2]; the other slots stay NaN
            S = len(sample)
            base = [os.path.splitext(os.path.basename(fn))[0] for fn in sample]
            dn = os.path.dirname(sample[0])
            slots = [('f', i, i + 1) for i in range(2, S - 2)] + [('b', i, i - 1) for i in range(2, S - 2)]
            slots += [('f', 1, 2), ('b', S - 2, S - 3)]
            flows = []
            for _, i, j in slots:
                path = os.path.join(root, self.FLOW_FOLDER, dn, 'flow_%s_%s.png' % (base[i], base[j]))
                fl = read_flow_png(path)
                if fl is None:
                    raise IOError('missing flow file %s' % path)
                flows.append(fl)
            raw['flow'] = torch.from_numpy(np.stack(flows))                     # [nflow, H, W, 2] fp32, NaN = invalid
            raw['flow_slots'] = [(d, i) for d, i, _ in slots]
        return raw

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

    def _flows(self, raw, dev, S, crop):
        """(wb, wf) fp32 [S, 2, Ho, Wo] from the decoded flow files of the sample: training crop / validation resize through
        tcvom_flow_crop_resize (VMD.py:68-126), precomputed validation padded with NaN (VMD.py:274-280); empty slots NaN."""
        fl = raw['flow'].to(dev, non_blocking=True).contiguous()
        n, Hs, Ws, _ = fl.shape
        Ho, Wo = self.image_shape
        if self.mode == 'val' and self.precomputed_val is not None:
            planes = fl.permute(0, 3, 1, 2).contiguous()
            if (Hs, Ws) != (Ho, Wo):
                planes = self._pad(planes, Ho, Wo, [float('nan'), float('nan')])
        else:
            ph, pw, nh, nw = crop if crop is not None else (0, 0, Hs, Ws)
            planes = torch.empty((n, 2, Ho, Wo), dtype=torch.float32, device=dev)
            L.call('tcvom_flow_crop_resize', L.ptr(fl), L.ptr(planes), n, Hs, Ws, ph, pw, nh, nw, Ho, Wo, L.stream_ptr())
        wb = torch.full((S, 2, Ho, Wo), float('nan'), dtype=torch.float32, device=dev)
        wf = torch.full((S, 2, Ho, Wo), float('nan'), dtype=torch.float32, device=dev)
        for k, (d, i) in enumerate(raw['flow_slots']):
            (wf if d == 'f' else wb)[i] = planes[k]
        return wb, wf

    def transform(self, raw):
        """uint8 frames -> (fg, bg, a, idx) of the reference's loader (VMD.py:250-301, no_flow)."""
        dev = self._dev()
        fg_u8 = raw['fg'].to(dev, non_blocking=True)
        bg_u8 = raw['bg'].to(dev, non_blocking=True)
        S, Hs, Ws, _ = fg_u8.shape
        Ho, Wo = self.image_shape
        bgr = [2, 1, 0]
        crop = None
        if self.mode == 'train':
            assert (Hs, Ws) == self.VIDEO_SHAPE, 'training clips are %dx%d (VMD.py:22)' % self.VIDEO_SHAPE
            ph, pw, nsize, a = self.shape_aug(fg_u8)
            crop = (ph, pw, nsize[0], nsize[1])
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
        if not self.no_flow:
            wb, wf = self._flows(raw, dev, S, crop)
            return fg, bg, a, wb, wf, torch.tensor(raw['idx'])          # VMD.py:293-300
        return fg, bg, a, torch.tensor(raw['idx'])

    def __getitem__(self, idx):
        return self.transform(self.load_raw(idx))
