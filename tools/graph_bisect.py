#!/usr/bin/env python
"""Which part of the training step invalidates a HIP-graph capture?  Captures (1) forward with autograd recording, (2) + backward,
(3) + zero_grad/Adam, each under the three capture error modes of torch.cuda.graph, at a small window."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                        # noqa: E402
from tcvom_amd.facade import train_step_loss                        # noqa: E402
from tcvom_amd.optim import FusedAdam                               # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 320)
dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, H, W, 0)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-4, weight_decay=1e-4)


def stage(n):
    out = model(a, fg, bg)
    loss = train_step_loss(out)
    if n >= 2:
        model.zero_grad(set_to_none=True)
        loss.backward()
    if n >= 3:
        opt.step()
    return loss


for _ in range(3):
    stage(3)
torch.cuda.synchronize()
for mode in ('global', 'thread_local', 'relaxed'):
    for n in (1, 2, 3):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            stage(n)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode=mode):
                stage(n)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            print('%-12s stage %d: captured and replayed' % (mode, n), flush=True)
        except Exception as ex:                                     # noqa: BLE001
            print('%-12s stage %d: FAILED %s' % (mode, n, str(ex).splitlines()[0][:160]), flush=True)
            try:
                torch.cuda.synchronize()
            except Exception:                                       # noqa: BLE001
                pass
        del g
