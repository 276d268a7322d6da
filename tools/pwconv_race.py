#!/usr/bin/env python
"""Race screen of the streaming 1x1 conv (csrc/pwconv.hip): the kernel has no atomics, so two launches on the same operands must agree
BIT FOR BIT -- a tile read before its LDS-DMA landed (the counted vmcnt of the tile loop lets the previous tile's stores stay in
flight) would show up as a differing output.  Runs every shape `iters` times while a second stream streams through HBM."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import dense_desc                           # noqa: E402

DEV = 'cuda'
H16 = L.ACT_DTYPE


def main(iters=150):
    g = torch.Generator(device=DEV).manual_seed(3)
    rnd = lambda *s: (torch.rand(*s, device=DEV, generator=g) * 2 - 1).to(H16)   # noqa: E731
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(256 << 20, device=DEV, dtype=torch.uint8), torch.empty(256 << 20, device=DEV, dtype=torch.uint8)
    bad = 0
    for pix, kout, cin, B in ((32640, 1024, 256, 3), (32640, 2048, 512, 3), (130560, 256, 64, 3), (32640, 128, 128, 3), (32640, 64, 128, 3),
                              (130560, 32, 64, 1), (8160, 256, 128, 3), (32641, 512, 128, 2), (2040, 512, 256, 3)):
        x, w = rnd(B, pix, cin), rnd(kout, cin)
        d = dense_desc(pix, kout, cin, kout, batch=B, in_bstride=pix * cin, w_bstride=0, out_bstride=pix * kout)
        var = L._FNS['tcvom_conv_igemm_variant'](C.byref(d), 1).decode()
        groups = L.call('tcvom_conv_stats_groups', C.byref(d), 1)
        d.stats_bstride = groups
        st = L.stream_ptr()

        def run():
            y = torch.empty(B, pix, kout, device=DEV, dtype=H16)
            stats = torch.empty(B * groups * 2 * kout, device=DEV, dtype=torch.float32)
            L.call('tcvom_conv_igemm', L.ptr(x), L.ptr(w), L.ptr(y), None, None, None, L.ptr(stats), C.byref(d), st)
            return y, stats
        y0, s0 = run()
        ref = x[0, :4096].float() @ w.float().t()
        err = (y0[0, :4096].float() - ref).abs().max().item() / ref.abs().max().item()
        diff = 0
        for i in range(iters):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    big_b.copy_(big_a, non_blocking=True)
            y, s = run()
            if not (torch.equal(y, y0) and torch.equal(s, s0)):
                diff += 1
        torch.cuda.synchronize()
        print('%-14s %7d px %4d -> %4d x%d: max error %.2e, %d of %d launches differ from the first' % (var, pix, cin, kout, B, err, diff, iters))
        bad += diff + (err > 1e-2)
    assert bad == 0, 'race or wrong result'
    print('ok')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 150)
