#!/usr/bin/env python
"""How long does the CPU oracle take for one 3-frame window (fwd+bwd) on this host, by thread count?  Sizing aid for
bench.py's cpu_baseline."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle.state_spec import vmn_gca_state_spec
from tcvom_amd.synthetic import formula_tensor, synthetic_window

H, W = int(sys.argv[1]), int(sys.argv[2])
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for th in [int(x) for x in sys.argv[3:]]:
    torch.set_num_threads(th)
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
             for k, s in vmn_gca_state_spec().items()}
    for k, v in state.items():
        if v.is_floating_point() and not any(t in k for t in ('weight_u', 'weight_v', 'running_')):
            v.requires_grad_(True)
    a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
    for rep in range(2):
        t0 = time.time()
        out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True)
        t1 = time.time()
        oracle.train_step_loss(out).backward()
        t2 = time.time()
        print('threads %d rep %d: fwd %.1f s bwd %.1f s' % (th, rep, t1 - t0, t2 - t1), flush=True)
