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


from tcvom_amd.data import read_flow_png as _flow, read_png16 as _read_png16      # noqa: E402  (shared with dataset.VMD's flow branch)


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
