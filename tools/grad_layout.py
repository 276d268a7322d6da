#!/usr/bin/env python
"""How the gradients of one vmn_gca training step are laid out in memory: GradientAverager.plan splits them into
contiguous spans (all-reduced in place at N > 1) and the rest (packed into buckets).  Prints the split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build                                               # noqa: E402
from tcvom_amd.ddp import GradientAverager                            # noqa: E402
from tcvom_amd.facade import train_step_loss                          # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    model, a, fg, bg = build(dev, 256, 256, seed=0)
    params = [p for p in model.parameters() if p.requires_grad]
    train_step_loss(model(a, fg, bg)).backward()
    missing = [p for p in params if p.grad is None]
    grads = [p.grad for p in params if p.grad is not None]
    spans, rest = GradientAverager.plan(grads, GradientAverager.MIN_SPAN)
    tot = sum(g.numel() for g in grads)
    print('params %d (grad None: %d), elements %d' % (len(params), len(missing), tot))
    for run, _, first, n in spans:
        print('  span: %4d tensors, %9d elements (%.1f %%) at storage offset %d' % (len(run), n, 100.0 * n / tot, first))
    print('  rest: %4d tensors, %9d elements (%.1f %%)' % (len(rest), sum(g.numel() for g in rest),
                                                         100.0 * sum(g.numel() for g in rest) / tot))


if __name__ == '__main__':
    main()
