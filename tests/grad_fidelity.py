#!/usr/bin/env python
"""Gradient fidelity of the HIP path: cosine similarity of every parameter gradient against the fp32 CPU oracle, and
run-to-run repeatability of the HIP gradients (two identical runs), for one train-mode window."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from oracle.state_spec import vmn_gca_state_spec
from tcvom_amd.facade import train_step_loss
from tcvom_amd.synthetic import formula_tensor, synthetic_window
from models.model import FullModel_VMD

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 320)


def hip_grads():
    m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
    m = m.cuda().train()
    a, fg, bg = (t.cuda() for t in synthetic_window(1, 3, H, W, seed=0))
    train_step_loss(m(a, fg, bg)).backward()
    torch.cuda.synchronize()
    return {k: p.grad.double().cpu() for k, p in m.NET.named_parameters() if p.grad is not None}


def oracle_grads():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
             for k, s in vmn_gca_state_spec().items()}
    for k, v in state.items():
        if v.is_floating_point() and not any(t in k for t in ('weight_u', 'weight_v', 'running_')):
            v.requires_grad_(True)
    a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
    out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True)
    oracle.train_step_loss(out).backward()
    return {k: v.grad.double() for k, v in state.items() if getattr(v, 'grad', None) is not None}


def cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


g0, g1 = hip_grads(), hip_grads()
go = oracle_grads()
groups = {}
for k in g0:
    top = '.'.join(k.split('.')[:2])
    groups.setdefault(top, []).append(k)
print('%-28s %8s %10s %10s %10s' % ('group', 'tensors', 'cos(oracle)', 'min cos', 'cos(rerun)'))
for top, ks in groups.items():
    co = [cos(g0[k], go[k]) for k in ks if k in go]
    cr = [cos(g0[k], g1[k]) for k in ks]
    # norm-weighted mean
    wts = [float(go[k].norm()) for k in ks if k in go]
    mean = sum(c * w for c, w in zip(co, wts)) / (sum(wts) + 1e-300)
    print('%-28s %8d %10.4f %10.4f %10.4f' % (top, len(ks), mean, min(co), min(cr)))
