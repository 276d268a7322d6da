#!/usr/bin/env python
"""60 training steps at 544x960: allocated / peak / reserved memory must stay flat and the loss must fall."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tcvom_amd.facade import train_step_loss
from tcvom_amd.optim import FusedAdam
dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 544, 960, 0)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-5, weight_decay=1e-4)
for i in range(61):
    loss = train_step_loss(model(a, fg, bg))
    model.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    if i in (5, 30, 60):
        torch.cuda.synchronize()
        print(i, 'allocated %.3f GiB, peak %.3f GiB, reserved %.3f GiB, loss %.4f' % (torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, float(loss)))
