#!/usr/bin/env python
"""Micro-benchmark of the streaming BatchNorm kernels (bn_apply, bn_bwd_reduce, bn_bwd_apply) on the layer shapes of a
1088x1920 vmn_gca window, 3 frames per launch as the product issues them: us per launch and GB/s of algorithmic traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402

DEV = 'cuda'
SHAPES = [('os1  32', 1088, 1920, 32), ('os2  32', 544, 960, 32), ('os4  64', 272, 480, 64), ('os8 128', 136, 240, 128),
          ('os16 256', 68, 120, 256), ('os32 512', 34, 60, 512)]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    st = L.stream_ptr()
    nf = 3
    print('%-10s %22s %22s %22s' % ('shape', 'apply us (GB/s)', 'bwd_reduce us (GB/s)', 'bwd_apply us (GB/s)'))
    for name, H, W, K in SHAPES:
        P = H * W
        y = torch.randn(nf, H, W, K, device=DEV).to(torch.bfloat16)
        r1 = torch.randn_like(y)
        dz = torch.randn_like(y)
        z = torch.empty_like(y)
        dy = torch.empty_like(y)
        dres = torch.empty_like(y)
        ss = torch.randn(nf, 2 * K, device=DEV)
        saved = torch.rand(nf, 2 * K, device=DEV) + 0.5
        coef = torch.randn(nf, 3 * K, device=DEV)
        groups = L.call('tcvom_bn_bwd_groups', P, K)
        partial = torch.empty(nf * groups * 2 * K, device=DEV)
        nbytes = y.numel() * 2
        for res in (None, r1):
            t1 = timeit(lambda: L.call('tcvom_bn_apply', L.ptr(y), L.ptr(ss), L.ptr(res), None, L.ptr(z), P, K, 1, 0, nf, 2 * K, st))
            t2 = timeit(lambda: L.call('tcvom_bn_bwd_reduce', L.ptr(dz), None, L.ptr(y), L.ptr(res), L.ptr(ss), L.ptr(saved), L.ptr(partial),
                                       P, K, 1, 0, nf, 2 * K, st))
            t3 = timeit(lambda: L.call('tcvom_bn_bwd_apply', L.ptr(dz), None, L.ptr(y), L.ptr(res), L.ptr(ss), L.ptr(saved), L.ptr(coef),
                                       L.ptr(dy), L.ptr(dres) if res is not None else None, P, K, 1, 1, 0, 0, nf, 2 * K, st))
            n = 1 if res is not None else 0
            print('%-10s %12.1f (%6.0f) %12.1f (%6.0f) %12.1f (%6.0f)   %s' % (
                name, t1, (2 + n) * nbytes / t1 / 1e3, t2, (2 + n) * nbytes / t2 / 1e3, t3, (3 + 2 * n) * nbytes / t3 / 1e3,
                'with residual' if res is not None else ''))


if __name__ == '__main__':
    main()
