#!/usr/bin/env python
"""Parity report at realistic sizes: alpha MSE (whole frame and unknown pixels, calc_metric.py:25 convention) and
a dtSSD-style delta of the HIP path vs the CPU oracle on the synthetic window, formula weights, train mode.
Runs on the GPU box (oracle on the host CPU, so sizes are bounded).   python tools/parity_report.py 512 512"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402  (build helper)


def main(H, W):
    from tcvom_amd.synthetic import formula_tensor, synthetic_window
    from tcvom_amd.facade import train_step_loss
    model, a, fg, bg = bench.build(torch.device('cuda', 0), H, W, 0)
    out = model(a, fg, bg)
    train_step_loss(out).backward()
    torch.cuda.synchronize()
    # checker: imported here on purpose (this is a test/report tool, not the product path)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import importlib
    oracle = importlib.import_module('oracle')
    spec = importlib.import_module('oracle.state_spec').vmn_gca_state_spec()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32) for k, s in spec.items()}
    t0 = time.time()
    with torch.no_grad():
        ro, _ = oracle.window_forward(state, a.cpu(), fg.cpu(), bg.cpu(), window=7, dilate_kernel=12, training=True)
    al, rl = out[7].float().cpu(), ro[7]
    um = ro[6].isclose(torch.tensor(128.0 / 255.0))
    d = al - rl
    print('%dx%d: alpha MSE whole %.3e, unknown-only %.3e, max |d| %.3e; losses hip %s oracle %s (oracle fwd %.1f s)' % (
        H, W, float((d ** 2).mean()), float((d[um] ** 2).mean()), float(d.abs().max()),
        ['%.5f' % float(x) for x in out[:5]], ['%.5f' % float(x) for x in ro[:5]], time.time() - t0))


if __name__ == '__main__':
    main(int(sys.argv[1]), int(sys.argv[2]))
