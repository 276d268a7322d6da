#!/usr/bin/env python
"""Micro-benchmark of the igemm kernels on the layer shapes of vmn_gca at 1088x1920 (one frame, B=1):
forward (igemm_nt), data gradient (igemm_nt) and weight gradient (igemm_tt) of each conv class, timed with HIP
events over back-to-back launches.  Used for kernel A/B work; prints TFLOP/s per shape."""
import ctypes as C
import sys

import torch
import torch.nn as nn

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import ConvGeometry, dense_desc, dense_tt_desc   # noqa: E402
from tcvom_amd.weights import ConvSpec, WeightBank                   # noqa: E402

DEV = 'cuda'
SHAPES = [  # name, cin, cout, k, stride, pad, transposed, H, W
    ('os1  32->32 3x3', 32, 32, 3, 1, 1, False, 1088, 1920),
    ('os1   8->32 3x3', 6, 32, 3, 1, 1, False, 1088, 1920),
    ('os2  32->32 3x3', 32, 32, 3, 1, 1, False, 544, 960),
    ('os4  64->64 3x3', 64, 64, 3, 1, 1, False, 272, 480),
    ('os8 128->128 3x3', 128, 128, 3, 1, 1, False, 136, 240),
    ('os16 256->256 3x3', 256, 256, 3, 1, 1, False, 68, 120),
    ('os32 512->512 3x3', 512, 512, 3, 1, 1, False, 34, 60),
    ('os32->16 convT 512', 512, 512, 4, 2, 1, True, 34, 60),
    ('os4->2 convT 64', 64, 64, 4, 2, 1, True, 272, 480),
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
    return e0.elapsed_time(e1) / iters


def main():
    st = L.stream_ptr()
    print('%-22s %10s %10s %10s   (TFLOP/s; us)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
    for name, cin, cout, k, stride, pad, tr, H, W in SHAPES:
        shape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        w = nn.Parameter(torch.randn(shape, device=DEV) * 0.05)
        bank = WeightBank()
        spec = ConvSpec(name, w, None, None, None, tr, stride, pad, 'frame', needs_dgrad=cin >= 16)
        bank.register(spec)
        bank.prepare(1, True)
        geo = ConvGeometry(spec, 1, H, W)
        x = torch.randn(1, H, W, spec.cpad, device=DEV).to(torch.bfloat16)
        y = torch.empty(1, geo.OH, geo.OW, cout, device=DEV, dtype=torch.bfloat16)
        dy = torch.randn_like(y)
        dx = torch.empty(1, H, W, max(cin, 8), device=DEV, dtype=torch.bfloat16)
        flop = 2.0 * geo.out_pixels * cout * cin * k * k / (4 if tr else 1)

        from tcvom_amd.ops import _launch_conv, _phase_array
        # weight gradients as the product issues them: the 3 frames of a window in one batched launch (time / 3 shown)
        xb = [torch.randn_like(x) for _ in range(3)]
        dyb = [torch.randn_like(dy) for _ in range(3)]
        dwb = torch.zeros(3, spec.K * spec.T * spec.cpad, device=DEV)
        dys3 = (C.c_void_p * 3)(*[t.data_ptr() for t in dyb])
        xs3 = (C.c_void_p * 3)(*[t.data_ptr() for t in xb])
        dws3 = (C.c_void_p * 3)(*[dwb[i].data_ptr() for i in range(3)])

        def fwd():
            _launch_conv(geo.fwd, x, bank.fwd_ptr(spec, 0), y, None, None, 0, st)

        def dgrad():
            _launch_conv(geo.dgrad, dy, bank.bwd_ptr(spec, 0), dx, None, None, 0, st)

        def wgrad():
            L.call('tcvom_wgrad_igemm_batched', C.cast(dys3, C.c_void_p), C.cast(xs3, C.c_void_p), C.cast(dws3, C.c_void_p), 3,
                   _phase_array(geo.wgrad), len(geo.wgrad), cout, st)

        # the same layer as the product runs it: the 3 frames of a window in ONE frame-batched launch (time / 3 shown)
        x3 = torch.randn(3, H, W, spec.cpad, device=DEV).to(torch.bfloat16)
        y3 = torch.empty(3, geo.OH, geo.OW, cout, device=DEV, dtype=torch.bfloat16)
        dy3 = torch.randn_like(y3)
        dx3 = torch.empty(3, H, W, max(cin, 8), device=DEV, dtype=torch.bfloat16)

        def fwd3():
            _launch_conv(geo.fwd, x3, bank.fwd_ptr(spec, 0), y3, None, None, 0, st, 3, 0)

        def dgrad3():
            _launch_conv(geo.dgrad, dy3, bank.bwd_ptr(spec, 0), dx3, None, None, 0, st, 3, 0)

        # eligible shapes: 24 problems (8 layers x 3 frames) of the accumulator-stationary kernel in one launch
        tm = float('nan')
        if L._FNS['tcvom_wgrad_igemm_variant'](C.byref(_phase_array(geo.wgrad)[0])).startswith(b'wgrad_ws'):
            NP = 24
            dwm = torch.zeros(NP, spec.K * spec.T * spec.cpad, device=DEV)
            dysm = (C.c_void_p * NP)(*[dyb[i % 3].data_ptr() for i in range(NP)])
            xsm = (C.c_void_p * NP)(*[xb[i % 3].data_ptr() for i in range(NP)])
            dwsm = (C.c_void_p * NP)(*[dwm[i].data_ptr() for i in range(NP)])
            tm = timeit(lambda: L.call('tcvom_wgrad_ws_multi', C.cast(dysm, C.c_void_p), C.cast(xsm, C.c_void_p),
                                       C.cast(dwsm, C.c_void_p), NP, _phase_array(geo.wgrad), cout, st)) / NP

        tf = timeit(fwd)
        td = timeit(dgrad) if spec.needs_dgrad else float('nan')
        tw = timeit(wgrad) / 3
        tf3 = timeit(fwd3) / 3
        td3 = timeit(dgrad3) / 3 if spec.needs_dgrad else float('nan')
        print('%-22s %5.0f %4.0f %5.0f %4.0f %5.0f %4.0f (x24: %4.0f %4.0f) | 3-frame launch: fwd %5.0f %4.0f dgrad %5.0f %4.0f  [%s]' % (
            name, flop / tf / 1e9, tf * 1e3, flop / td / 1e9, td * 1e3, flop / tw / 1e9, tw * 1e3, flop / tm / 1e9, tm * 1e3,
            flop / tf3 / 1e9, tf3 * 1e3, flop / td3 / 1e9, td3 * 1e3,
            L._FNS['tcvom_conv_igemm_variant'](C.byref(_phase_array(geo.fwd)[0]), len(geo.fwd)).decode()))
    # GCA GEMMs at 1080p: N = 8160
    N, D, DV = 8160, 576, 2048
    ld = (N + 63) // 64 * 64
    G = torch.randn(N, D, device=DEV).to(torch.bfloat16)
    S = torch.empty(N, ld, device=DEV)
    P = torch.randn(N, ld, device=DEV).to(torch.bfloat16)
    Vt = torch.randn(DV, ld, device=DEV).to(torch.bfloat16)
    O = torch.empty(N, DV, device=DEV, dtype=torch.bfloat16)
    dV = torch.zeros(N, DV, device=DEV)
    d1 = dense_desc(N, N, D, ld, out_fp32=True)
    d2 = dense_desc(N, DV, ld, DV)
    d3 = dense_tt_desc(N, N, DV)
    t1 = timeit(lambda: L.call('tcvom_conv_igemm', L.ptr(G), L.ptr(G), L.ptr(S), None, None, None, None, C.byref(d1), st))
    t2 = timeit(lambda: L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d2), st))
    t3 = timeit(lambda: L.call('tcvom_wgrad_igemm', L.ptr(P), L.ptr(O), L.ptr(dV), C.byref(d3), ld, st))
    # the softmax backward: unfused (fp32 dP GEMM + row pass) vs fused epilogue
    V = torch.randn(N, DV, device=DEV).to(torch.bfloat16)
    dO = torch.randn(N, DV, device=DEV).to(torch.bfloat16)
    dP = torch.empty(N, ld, device=DEV)
    cvec = torch.rand(N, device=DEV) + 0.5
    delta = torch.randn(N, device=DEV)
    T = torch.empty(N, ld, device=DEV, dtype=torch.bfloat16)
    d4 = dense_desc(N, N, DV, ld, out_fp32=True)
    t4 = timeit(lambda: L.call('tcvom_conv_igemm', L.ptr(dO), L.ptr(V), L.ptr(dP), None, None, None, None, C.byref(d4), st))
    t5 = timeit(lambda: L.call('tcvom_row_softmax_bwd', L.ptr(P), L.ptr(dP), L.ptr(cvec), L.ptr(T), N, N, ld, ld, N, st))
    t6 = timeit(lambda: L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T), None, None, N, DV, ld, 1, st))
    Ttb = torch.empty(ld, ld, device=DEV, dtype=torch.bfloat16)
    Ptb = torch.empty(ld, ld, device=DEV, dtype=torch.bfloat16)
    t7 = timeit(lambda: L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T), L.ptr(Ttb), L.ptr(Ptb), N, DV, ld, 1, st))
    t8 = timeit(lambda: L.call('tcvom_transpose_bf16', L.ptr(P), L.ptr(Ptb), N, ld, ld, ld, 1, N * ld, ld * ld, st))
    print('GCA dP fp32 %5.0f us + softmax bwd %5.0f us | fused %5.0f us | fused + T^T, P^T %5.0f us (one N x N transpose pass: %5.0f us)' % (t4 * 1e3, t5 * 1e3, t6 * 1e3, t7 * 1e3, t8 * 1e3))
    print('GCA S=GG^T  %5.0f TF %5.0f us | O=PV %5.0f TF %5.0f us | dV=P^T dO %5.0f TF %5.0f us' % (
        2.0 * N * N * D / t1 / 1e9, t1 * 1e3, 2.0 * N * N * DV / t2 / 1e9, t2 * 1e3, 2.0 * N * N * DV / t3 / 1e9, t3 * 1e3))


if __name__ == '__main__':
    main()
