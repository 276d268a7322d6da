#!/usr/bin/env python
"""Every ATen op of one training step that launches device work, with the innermost tcvom_amd / bench.py source line that issued it
(torch.profiler with_stack; covers the autograd thread: the Python frames of a custom Function.backward are on it)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402
from tcvom_amd.optim import FusedAdam                               # noqa: E402

dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 1088, 1920, 0, config=os.environ.get('ATEN_WHERE_CONFIG', 'gca'))
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt, dur = collections.Counter(), collections.Counter()
total = 0
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.self_device_time_total <= 0:
        continue
    loc = 'autograd engine / C++'
    for fr in ev.stack or []:
        if ('tcvom_amd' in fr or 'bench.py' in fr or 'aten_where' in fr) and 'profiler' not in fr:
            loc = fr.split('/')[-1][:70]
            break
    key = (ev.name, str(ev.input_shapes)[:70], loc)
    cnt[key] += 1
    dur[key] += ev.self_device_time_total
    total += 1
print('%d device-launching ATen ops in the step, %.1f us' % (total, sum(dur.values())))
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print('%-18s %3d %8.1f us  %-70s %s' % (key[0], n, dur[key], key[2], key[1]))
