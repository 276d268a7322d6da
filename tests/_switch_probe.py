"""One forward + backward of the GCA + TAM training window at 544 x 960 (2040 attention keys: the 256-tile GEMM paths, the fused
scores + softmax, the k-major operands, the K-split tail and the tail-only / row-range gradient paths are all active at this size)
with formula weights; prints one JSON line of output statistics and per-group gradient norms.  tests/test_gpu_switches.py runs it with
and without the A/B switches that select the alternative code paths and compares the lines."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402

dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 544, 960, 0)
out = model(a, fg, bg)
loss = train_step_loss(out)
loss.backward()
torch.cuda.synchronize()
groups = {}
for name, p in model.NET.named_parameters():
    if p.grad is None:
        continue
    key = '.'.join(name.split('.')[:2])
    g = p.grad.double()
    s, d = groups.get(key, (0.0, 0.0))
    groups[key] = (s + float((g * g).sum()), d + float(g.sum()))
alpha = out[7].detach().float().flatten()
sample = alpha[torch.linspace(0, alpha.numel() - 1, 4096, device=alpha.device).long()]       # a fixed sample of the predicted mattes
res = {'loss': float(loss), 'losses': [float(x) for x in out[:5]], 'alpha_sample': [float(x) for x in sample.cpu()],
       'grad_norm': {k: v[0] ** 0.5 for k, v in groups.items()}, 'grad_sum': {k: v[1] for k, v in groups.items()}}
print('PROBE ' + json.dumps(res))
