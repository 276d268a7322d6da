#!/usr/bin/env python
"""Crash / finiteness sweep of one training step (fwd + bwd) of vmn_gca over window shapes that neither the tests nor bench.py use:
sizes where the shape-dependent kernel choices flip (widths below a 32-pixel tile at os2 / os4, heights that are not tile multiples,
S = 4 / 5 frames, B = 2 clips), each compared with the same step under the round-4 kernels switched off (subprocess)."""
import json
import os
import subprocess
import sys

CASES = [(1, 3, 64, 64), (1, 3, 96, 160), (1, 5, 128, 224), (2, 3, 160, 96), (1, 4, 192, 320), (1, 3, 352, 480), (1, 3, 544, 992),
         (2, 5, 256, 256), (1, 3, 736, 1280)]

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tcvom_amd.facade import FullModel_VMD, train_step_loss
    from tcvom_amd.synthetic import formula_tensor, synthetic_window
    dev = torch.device('cuda:0')
    res = {}
    if os.environ.get('SWEEP_CONFIG') == 'fba':            # FBA + TAM (config 5): model and window from bench.build, B = 1, S = 3
        import bench
        for H, W in [(160, 224), (352, 480), (256, 384), (544, 960), (736, 1280)]:
            m, a, fg, bg = bench.build(dev, H, W, 0, config='fba')
            out = m(a, fg, bg)
            loss = train_step_loss(out)
            loss.backward()
            torch.cuda.synchronize()
            gn = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
            res['fba_%dx%d' % (H, W)] = (float(loss), gn, float(out[7].float().mean()))
            del m
        print('SWEEP ' + json.dumps(res))
        sys.exit(0)
    m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
    m = m.to(dev).train()
    for B, S, H, W in CASES:
        a, fg, bg = (t.to(dev) for t in synthetic_window(B, S, H, W, seed=3))
        m.zero_grad(set_to_none=True)
        out = m(a, fg, bg)
        loss = train_step_loss(out)
        loss.backward()
        torch.cuda.synchronize()
        gn = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
        res['%dx%dx%dx%d' % (B, S, H, W)] = (float(loss), gn, float(out[7].float().mean()))
    print('SWEEP ' + json.dumps(res))
    sys.exit(0)


def run(extra):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, **extra), capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('SWEEP ')][-1][6:])


new = run({})
old = run({'TCVOM_NO_SCONV': '1', 'TCVOM_TT_NARROW': '1', 'TCVOM_WGRADWS_NO_DIL': '1'})
bad = 0
for k in new:
    l1, g1, a1 = new[k]
    l0, g0, a0 = old[k]
    ok = all(x == x and abs(x) < 1e30 for x in (l1, g1, a1)) and abs(l1 - l0) <= 2e-3 * max(abs(l0), 1e-3) and abs(g1 / g0 - 1) < 0.1
    bad += not ok
    print('%-16s loss %.6f / %.6f   grad norm %.5g / %.5g   mean alpha %.5f / %.5f   %s' % (k, l1, l0, g1, g0, a1, a0, 'ok' if ok else 'MISMATCH'))
sys.exit(1 if bad else 0)
