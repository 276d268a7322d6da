#!/usr/bin/env python
"""Run-to-run spread of the north-star parity figure: the unknown-pixel alpha MSE of the HIP forward (train mode, formula weights) against the
CPU oracle at the four tested sizes, `reps` HIP runs each (the oracle once per size).  The HIP forward is not bit-reproducible (fp32 atomics in
the SpectralNorm sums / statistics): the MAX over the runs is what a bound has to hold.

    python tests/parity_repeat.py [reps=8]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                                       # noqa: E402  (checker only)
from oracle.state_spec import vmn_gca_state_spec                    # noqa: E402
from models.model import FullModel_VMD                              # noqa: E402
from tcvom_amd.synthetic import formula_tensor, synthetic_window    # noqa: E402
import tcvom_amd._lib as L                                          # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(min(os.cpu_count() or 1, 32))
state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32) for k, s in vmn_gca_state_spec().items()}
print('storage %s, %d HIP runs per size; unknown-pixel alpha MSE vs the oracle (bound 1e-4)' % (L.DTYPE_NAME, reps))
for H, W in ((256, 320), (512, 512), (544, 960), (1088, 1920)):
    a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
    with torch.no_grad():
        ro, _ = oracle.window_forward({k: v.clone() for k, v in state.items()}, a, fg, bg, window=7, dilate_kernel=12, training=True)
    um = ro[6].isclose(torch.tensor(128.0 / 255.0))
    vals, mx = [], []
    for _ in range(reps):
        m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
        m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
        m = m.to('cuda').train()
        with torch.no_grad():
            out = m(a.cuda(), fg.cuda(), bg.cuda())
        d = out[7].float().cpu() - ro[7]
        vals.append(float((d[um] ** 2).mean()))
        mx.append(float(d.abs().max()))
        del m, out
    print('%4dx%-4d  min %.3e  mean %.3e  MAX %.3e   max |d| over runs %.3e   runs: %s'
          % (H, W, min(vals), sum(vals) / len(vals), max(vals), max(mx), ' '.join('%.2e' % v for v in vals)))
