#!/usr/bin/env python
"""BASELINE.json config 5 (FBA base + TAM) on one GPU: forward + backward + Adam on a synthetic 3-frame window, timed
with HIP events.  Not the headline metric (bench.py measures config 3); reports ms / window and peak memory."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from models.model import FullModel_VMD                                   # noqa: E402
from tcvom_amd.facade import train_step_loss                             # noqa: E402
from tcvom_amd.optim import FusedAdam                                    # noqa: E402
from tcvom_amd.synthetic import formula_tensor, synthetic_window         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=1088)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--forward-only', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    model = FullModel_VMD('vmn_fba', agg_window=7, dilate_kernel=12)
    model.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in model.NET.state_dict().items()})
    model = model.to(dev).train()
    a, fg, bg = (t.to(dev) for t in synthetic_window(1, 3, args.height, args.width, seed=0))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-5, weight_decay=1e-4)

    def step():
        if args.forward_only:
            with torch.no_grad():
                return train_step_loss(model(a, fg, bg))
        loss = train_step_loss(model(a, fg, bg))
        model.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    print('vmn_fba %dx%d: %.1f ms / window (%.2f windows/s), loss %.4f, peak memory %.1f GiB' % (
        args.height, args.width, ms, 1e3 / ms, float(loss), torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == '__main__':
    main()
