#!/usr/bin/env python
"""The non-headline BASELINE.json configurations on one GPU (bench.py measures config 3, the headline):
    1  pred_single.py path: DIM base, single 512x512 frame, forward (the reference runs it on CPU)
    2  GCA + TAM forward-only, 3-frame 512x512 window
    5  FBA + TAM forward + backward (+ Adam), 3-frame 1088x1920 window
Each is timed with HIP events over back-to-back steps on synthetic inputs and formula weights; one line per config."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from models.model import FullModel, FullModel_VMD                         # noqa: E402
from tcvom_amd.facade import train_step_loss                             # noqa: E402
from tcvom_amd.optim import FusedAdam                                    # noqa: E402
from tcvom_amd.synthetic import formula_tensor, synthetic_window         # noqa: E402

DEV = torch.device('cuda', 0)


def timed(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def build(cls, name, **kw):
    m = cls(name, **kw)
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
    return m.to(DEV)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='1,2,5')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    args = ap.parse_args()
    for cfg in args.configs.split(','):
        if cfg == '1':
            m = build(FullModel, 'dim', dilate_kernel=12).eval()
            inp = [t.to(DEV) for t in synthetic_window(1, 1, 512, 512, seed=0)]
            with torch.no_grad():
                ms = timed(lambda: m(*inp), args.steps, args.warmup)
            print('config 1  DIM single frame 512x512 forward (+losses): %.2f ms / frame = %.1f frames/s' % (ms, 1e3 / ms))
        elif cfg == '2':
            m = build(FullModel_VMD, 'vmn_gca', agg_window=7, dilate_kernel=12).train()
            inp = [t.to(DEV) for t in synthetic_window(1, 3, 512, 512, seed=0)]
            with torch.no_grad():
                ms = timed(lambda: m(*inp), args.steps, args.warmup)
            print('config 2  GCA+TAM forward-only 3x512x512 (train-mode statistics): %.2f ms / window = %.1f windows/s' % (ms, 1e3 / ms))
        elif cfg == '5':
            m = build(FullModel_VMD, 'vmn_fba', agg_window=7, dilate_kernel=12).train()
            inp = [t.to(DEV) for t in synthetic_window(1, 3, 1088, 1920, seed=0)]
            params = [p for p in m.parameters() if p.requires_grad]
            opt = FusedAdam(params, lr=1e-5, weight_decay=1e-4)

            def step():
                loss = train_step_loss(m(*inp))
                m.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
            ms = timed(step, max(3, args.steps // 3), 1)
            print('config 5  FBA+TAM fwd+bwd+Adam 3x1088x1920: %.1f ms / window = %.2f windows/s, peak memory %.1f GiB' % (
                ms, 1e3 / ms, torch.cuda.max_memory_allocated() / 2 ** 30))
        del m
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
