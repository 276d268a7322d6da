#!/usr/bin/env python
"""Weight gradients of the small-channel layers of vmn_gca at 1088x1920 as the product launches them (tcvom_wgrad_igemm_batched, the
frames of a window in one launch): HIP-event time per launch, kernel variant, algorithmic GB/s (dy + x read once).  For A/B work on the
tile rule of igemm_tt (TCVOM_TT_NARROW, TCVOM_TT_OCC)."""
import ctypes as C
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import ConvGeometry                         # noqa: E402
from tcvom_amd.ops import _phase_array                               # noqa: E402
from tcvom_amd.weights import ConvSpec, WeightBank                   # noqa: E402

DEV = 'cuda'
SHAPES = [  # name, cin, cout, k, stride, pad, transposed, H, W, frames
    ('os1   6->32 3x3', 6, 32, 3, 1, 1, False, 1088, 1920, 1),
    ('os1   6->32 3x3 s2', 6, 32, 3, 2, 1, False, 1088, 1920, 3),
    ('os1   3->16 3x3 s2 p0', 3, 16, 3, 2, 0, False, 1090, 1922, 3),
    ('os2  16->32 3x3 s2 p0', 16, 32, 3, 2, 0, False, 546, 962, 3),
    ('os2  64->32 3x3', 64, 32, 3, 1, 1, False, 544, 960, 1),
    ('os2  32->64 3x3', 32, 64, 3, 1, 1, False, 544, 960, 1),
    ('os2  32->64 3x3 s2', 32, 64, 3, 2, 1, False, 544, 960, 3),
    ('os2->1 convT 32', 32, 32, 4, 2, 1, True, 544, 960, 1),
    ('os4->2 convT 64', 64, 64, 4, 2, 1, True, 272, 480, 1),
]


def timeit(fn, iters=20):
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
    dt = torch.float16 if L.DTYPE_NAME == 'fp16' else torch.bfloat16
    print('%-24s %-28s %9s %9s' % ('layer', 'kernel', 'us', 'GB/s'))
    total = 0.0
    for name, cin, cout, k, stride, pad, tr, H, W, S in SHAPES:
        shape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        w = nn.Parameter(torch.randn(shape, device=DEV) * 0.05)
        bank = WeightBank()
        spec = ConvSpec(name, w, None, None, None, tr, stride, pad, 'frame', needs_dgrad=False)
        bank.register(spec)
        geo = ConvGeometry(spec, 1, H, W)
        xs = [(torch.randn(1, H, W, spec.cpad, device=DEV) * 0.5).to(dt) for _ in range(S)]
        dys = [(torch.randn(1, geo.OH, geo.OW, cout, device=DEV) * 0.5).to(dt) for _ in range(S)]
        dw = torch.zeros(S, spec.K * spec.T * spec.cpad, device=DEV)
        vp = lambda ts: C.cast((C.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), C.c_void_p)   # noqa: E731
        a_dy, a_x, a_dw = vp(dys), vp(xs), vp([dw[i] for i in range(S)])
        arr = _phase_array(geo.wgrad)

        def run():
            L.call('tcvom_wgrad_igemm_batched', a_dy, a_x, a_dw, S, arr, len(geo.wgrad), cout, st)

        us = timeit(run)
        total += us
        nbytes = sum(t.numel() * 2 for t in xs) + sum(t.numel() * 2 for t in dys)
        variant = L._FNS['tcvom_wgrad_igemm_variant'](C.byref(arr[0])).decode()
        print('%-24s %-28s %9.1f %9.0f' % (name, variant, us, nbytes / us * 1e-3))
    print('sum %.1f us' % total)


if __name__ == '__main__':
    main()
