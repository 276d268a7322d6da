#!/usr/bin/env python
"""Counterpart of the reference's calc_metric.py on the HIP path: SAD / MSE / SSDA / dtSSD / MESSDdt of a folder of
predictions, one fused kernel launch per frame (tcvom_amd.metrics.frame_metrics) instead of numpy + CPU grid_sample
in a process pool.  Same inputs and output as calc_metric.py:47-232:

    python calc_metric.py --pred <dir with <video>/<frame>_pred.png, _tri.png> --data <dataset root> [--output metric.json]

`--data` holds frame_corr.json (frame list), FG_done/<video>/<frame>.png (RGBA, ground-truth alpha in the last channel)
and optionally flow_png/<video>/flow_<a>_<b>.png (16-bit, written by OpenCV: B = int16 x displacement * 100, G = y, R =
validity, calc_metric.py:65-71) — without a flow file the frame pair contributes no MESSDdt.
PNG I/O uses Pillow (OpenCV is not in this image).
"""
import argparse
import json
import os

import numpy as np
import torch

from tcvom_amd.metrics import frame_metrics


def _gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


def _alpha_of(path):
    from PIL import Image
    im = np.asarray(Image.open(path))
    return im[..., -1] if im.ndim == 3 else im


def _read_png16(path):
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


def _flow(path):
    """Optical-flow PNG of the VideoMatting108 tree -> float [H, W, 2] (x, y displacement in pixels, NaN where invalid), or
    None when the pair has no flow file.  The reference decodes it with cv2.imread(IMREAD_UNCHANGED) and takes x[..., :-1] as
    the flow and x[..., -1] as the validity mask (calc_metric.py:65-71); cv2 hands channels over in B, G, R order, so in FILE
    order (what every other reader returns) the layout is R = mask, G = y flow, B = x flow: convert first, slice second.
    A file that exists but cannot be decoded is an error (a silent None would zero MESSDdt)."""
    if not os.path.exists(path):
        return None
    x = _read_png16(path)
    if x.ndim != 3 or x.shape[-1] not in (3, 4):
        raise IOError('%s: expected a 3- or 4-channel 16-bit flow image, got shape %s' % (path, x.shape))
    x = x[..., ::-1] if x.shape[-1] == 3 else x[..., [2, 1, 0, 3]]          # file order -> cv2 order
    flow = np.float32(np.ascontiguousarray(x[..., :-1][..., :2]).astype(np.uint16).view(np.int16))
    flow[x[..., -1] == 0] = np.nan
    return flow / 100.0


def main(args):
    dev = torch.device('cuda', 0)
    with open(os.path.join(args.data, 'frame_corr.json'), 'rb') as f:
        fdict = json.load(f)
    exist = {}
    for f in sorted(fdict.keys()):
        fn = os.path.splitext(f)[0]
        exist[f] = os.path.exists(os.path.join(args.pred, fn + '_pred.png')) and os.path.exists(os.path.join(args.pred, fn + '_tri.png'))
    videos = sorted({os.path.dirname(f) for f in exist if all(ok for g, ok in exist.items() if os.path.dirname(g) == os.path.dirname(f))})
    print('Present videos:', videos)
    frames = [f for f in sorted(exist) if exist[f] and os.path.dirname(f) in videos]

    def load(f):
        fn = os.path.splitext(f)[0]
        a = torch.from_numpy(np.float32(_gray(os.path.join(args.pred, fn + '_pred.png')) / 255.0)).to(dev)
        t = torch.from_numpy(np.ascontiguousarray(_gray(os.path.join(args.pred, fn + '_tri.png')))).to(dev)
        g = torch.from_numpy(np.float32(_alpha_of(os.path.join(args.data, 'FG_done', fn + '.png')) / 255.0)).to(dev)
        return a, g, t

    results = {'avg': {}, 'all': {}}
    keys = ('mSAD', 'MSE', 'SSDA', 'dtSSD', 'MESSDdt_fix', 'MESSDdt')
    total = dict.fromkeys(keys, 0.0)
    for v in videos:
        vf = [f for f in frames if os.path.dirname(f) == v]
        per, acc = {}, dict.fromkeys(keys + ('pixel_count', 'flow_pixel_count'), 0)
        cur = load(vf[0])
        for i, f in enumerate(vf):
            nxt = load(vf[i + 1]) if i + 1 < len(vf) else None
            r = {'dtSSD': 0, 'MESSDdt_fix': 0, 'MESSDdt': 0, 'flow_pixel_count': 0}
            if nxt is not None:
                base = lambda p: os.path.splitext(os.path.basename(p))[0]
                fl = _flow(os.path.join(args.data, 'flow_png', v, 'flow_%s_%s.png' % (base(f), base(vf[i + 1]))))
                m = frame_metrics(cur[0], cur[1], cur[2], nxt[0], nxt[1], None if fl is None else torch.from_numpy(fl).to(dev))
                r['dtSSD'] = m['dtSSD']
                if fl is not None:
                    r['MESSDdt_fix'], r['MESSDdt'], r['flow_pixel_count'] = m['MESSDdt']
            else:
                m = frame_metrics(cur[0], cur[1], cur[2])
            r.update({'mSAD': m['SAD'], 'MSE': m['MSE'], 'SSDA': m['SSDA'], 'pixel_count': m['pixels']})
            per[os.path.splitext(f)[0]] = r
            for k in acc:
                acc[k] += r[k]
            cur = nxt
        for k in keys:
            acc[k] /= float(len(vf))
            total[k] += acc[k]
        results['all'][v] = {'avg': acc, 'all': per}
    for k in keys:
        total[k] /= float(max(len(videos), 1))
    results['avg'] = total
    output = args.output if args.output is not None else os.path.join(args.pred, 'metric.json')
    os.makedirs(os.path.dirname(os.path.abspath(output)), exist_ok=True)
    with open(output, 'w') as f:
        json.dump(results, f, indent=4, sort_keys=True)
    print(json.dumps(total, sort_keys=True))


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pred', required=True)
    ap.add_argument('--data', required=True)
    ap.add_argument('--output', default=None, help='/path/to/metric/json/file')
    ap.add_argument('--vis', action='store_true', help='accepted for compatibility; the visualisation is not produced')
    ap.add_argument('--n_threads', default=None, help='accepted for compatibility; frames are evaluated on the GPU')
    return ap.parse_args()


if __name__ == '__main__':
    main(parser())
