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

import traceback
from torch.utils._python_dispatch import TorchDispatchMode

# Python-level wrappers (these also see the calls made inside custom Function.backward on the autograd thread)
pycnt = collections.Counter()


def _wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if ('tcvom_amd' in x.filename or 'bench.py' in x.filename)]
        loc = '%s:%d %s' % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].name) if fr else '?'
        r = orig(*a, **k)
        if name in ('contiguous',) and r is a[0]:
            return r
        pycnt['%s  <- %s' % (name, loc)] += 1
        return r
    setattr(owner, name, f)


for nm in ('zeros', 'zeros_like', 'full', 'ones', 'cat', 'stack'):
    _wrap(torch, nm)
for nm in ('zero_', 'fill_', 'clone', 'contiguous', 'copy_', 'new_zeros', 'float', 'to'):
    _wrap(torch.Tensor, nm)

WATCH = ('copy_', 'fill_', 'zero_', 'add', 'add_', 'clone', 'cat', 'zeros', 'sum', 'mul', '_to_copy', 'zeros_like', 'mul_', 'sub', 'neg')
cnt = collections.Counter()
where = collections.defaultdict(collections.Counter)


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name in WATCH:
            big = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:1]
            fr = [f for f in traceback.extract_stack() if ('tcvom_amd' in f.filename or 'bench.py' in f.filename) and 'aten_ops' not in f.filename]
            loc = '%s:%d %s' % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].name) if fr else 'autograd engine'
            cnt[name] += 1
            where[name]['%s %s' % (loc, big)] += 1
        return func(*args, **(kwargs or {}))


pycnt.clear()
with Log():
    step()
torch.cuda.synchronize()
print('--- python-level calls in one step')
for k, v in pycnt.most_common(60):
    print('  %4d  %s' % (v, k))
print('--- dispatcher (forward thread)')
for k, v in cnt.most_common():
    print('%-12s %5d' % (k, v))
    for loc, n in where[k].most_common(14):
        print('      %4d  %s' % (n, loc))
