#!/usr/bin/env python
"""Which aten ops (fills, copies, adds ...) does one training step still issue, and from where?  (torch profiler,
CPU-side op counts with source locations; used to hunt leftover framework launches on the hot path.)"""
import os
import sys
import collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tcvom_amd.facade import train_step_loss
from tcvom_amd.optim import FusedAdam

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


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
where = collections.defaultdict(collections.Counter)
for e in prof.events():
    if e.name.startswith('aten::') and e.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::clone',
                                                  'aten::contiguous', 'aten::cat', 'aten::empty', 'aten::zeros', 'aten::sum', 'aten::mul',
                                                  'aten::to', 'aten::_to_copy', 'aten::select', 'aten::slice'):
        cnt[e.name] += 1
        st = [s for s in (e.stack or []) if 'tcvom_amd' in s or 'bench.py' in s or 'models/' in s]
        where[e.name][st[0] if st else '?'] += 1
for k, v in cnt.most_common():
    print('%-18s %5d' % (k, v))
    for loc, n in where[k].most_common(6):
        print('      %4d  %s' % (n, loc))
