#!/usr/bin/env python
"""Which ATen / runtime ops does one training step still issue, INCLUDING the autograd thread (torch.profiler with input shapes)?
Prints the ops that launch device work, by (name, input shapes), with counts per step."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402
from tcvom_amd.optim import FusedAdam                               # noqa: E402

dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 1088, 1920, 0)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-4, weight_decay=1e-4)


def step():
    out = model(a, fg, bg)
    loss = train_step_loss(out)
    model.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile                # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::'):
        continue
    if ev.device_time_total <= 0 and ev.self_device_time_total <= 0:
        continue
    key = (ev.name, str(ev.input_shapes)[:120])
    cnt[key] += 1
    dur[key] += ev.self_device_time_total
print('%-28s %5s %9s  %s' % ('op', 'calls', 'self us', 'input shapes'))
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    if dur[key] > 0:
        print('%-28s %5d %9.1f  %s' % (key[0], n, dur[key], key[1]))
