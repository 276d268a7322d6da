#!/usr/bin/env python
"""Forward + backward of one window captured into a HIP graph (the optimizer stays outside): replay time against the eager
forward + backward on the same box.  Usage: graph_probe_fb.py [H W]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1088, 1920)
dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, H, W, 0)


def fb():
    loss = train_step_loss(model(a, fg, bg))
    model.zero_grad(set_to_none=True)
    loss.backward()
    return loss


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for _ in range(3):
    fb()
print('eager fwd+bwd    %.3f ms' % timed(fb), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fb()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = fb()
torch.cuda.synchronize()
g.replay()
print('captured; loss after one replay', float(static_loss), flush=True)
print('graph replay     %.3f ms' % timed(g.replay), flush=True)
print('loss after replays', float(static_loss), flush=True)
print('eager again      %.3f ms, loss %.6f' % (timed(fb), float(fb())), flush=True)
