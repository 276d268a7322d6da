#!/usr/bin/env python
"""How much of a step is launch overhead?  Captures one whole training step (forward, backward, Adam) into a HIP graph with
torch.cuda.graph and times replays against eager steps on the same box.  A probe, not a product path: host-side scalars (Adam's
step count, the learning rate) are frozen into the captured kernels."""
import os
import sys
import time

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
    return loss


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for _ in range(3):
    step()
print('eager            %.3f ms/step' % timed(step))
t0 = time.time()
for _ in range(5):
    step()
host = (time.time() - t0) / 5 * 1e3
torch.cuda.synchronize()
print('host enqueue     %.3f ms/step' % host)
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step()
    print('captured; loss', float(static_loss))
    print('graph replay     %.3f ms/step' % timed(g.replay))
    print('eager again      %.3f ms/step' % timed(step))
except Exception as e:                                              # noqa: BLE001
    print('capture failed: %r' % (e,))
