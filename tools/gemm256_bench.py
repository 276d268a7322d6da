#!/usr/bin/env python
"""Times the five GEMM launches of GuidedCxtAtten at 1080p (N = 8160 patches, 3 frames per launch) on gemm_nt256, with HIP events
over back-to-back launches, and checks one of them against a float reference.  For kernel A/B work: TCVOM_LIB selects a study
build of the library."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import dense_desc                           # noqa: E402

DEV = 'cuda'
BF = L.ACT_DTYPE


def timeit(fn, iters=10):
    for _ in range(2):
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
    B, N, D, DV = 3, 8160, 576, 2048
    ld = (N + 255) // 256 * 256
    st = L.stream_ptr()
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=DEV, generator=g) * 2 - 1).to(BF)   # noqa: E731
    G, Gt = rnd(B, N, D), rnd(B, D, ld)
    P, Pt = rnd(B, N, ld), rnd(B, ld, ld)
    Vt, V, dO = rnd(B, DV, ld), rnd(B, N, DV), rnd(B, N, DV)
    for t in (P, Pt, Vt, Gt):
        t[..., N:] = 0
    S = torch.empty(B, N, ld, device=DEV)
    O = torch.empty(B, N, DV, device=DEV)
    dW = torch.empty(B, N, D, device=DEV)
    T = torch.empty(B, N, ld, device=DEV, dtype=BF)
    Tt, Pt2 = torch.empty(B, ld, ld, device=DEV, dtype=BF), torch.empty(B, ld, ld, device=DEV, dtype=BF)
    cvec, delta = torch.rand(B, N, device=DEV) + 0.5, torch.randn(B, N, device=DEV)
    d1 = dense_desc(N, N, D, ld, batch=B, in_bstride=N * D, w_bstride=N * D, out_bstride=N * ld, vec_bstride=N, out_fp32=True)
    d2 = dense_desc(N, DV, ld, DV, batch=B, in_bstride=N * ld, w_bstride=DV * ld, out_bstride=N * DV, out_fp32=True)
    d3 = dense_desc(N, D, ld, D, batch=B, in_bstride=N * ld, w_bstride=D * ld, out_bstride=N * D, out_fp32=True)
    rows = [
        ('S = G G^T      (K = 576)', 2.0 * B * N * N * D,
         lambda: L.call('tcvom_conv_igemm', L.ptr(G), L.ptr(G), L.ptr(S), None, None, None, None, C.byref(d1), st)),
        ('O = P V        (K = 8192)', 2.0 * B * N * N * DV,
         lambda: L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d2), st)),
        ('dWq = T G      (M = 576)', 2.0 * B * N * N * D,
         lambda: L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Gt), L.ptr(dW), None, None, None, None, C.byref(d3), st)),
        ('T = softmax bwd (fused)', 2.0 * B * N * N * DV,
         lambda: L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T),
                        L.ptr(Tt), L.ptr(Pt2), N, DV, ld, B, st)),
    ]
    # the 1x1 convs of the FBA bottlenecks (bf16 output, 3 frames per launch): rows = pixels, weights [K][C]
    for pix, kout, cin in ((32640, 2048, 512), (32640, 1024, 256), (130560, 256, 64), (32640, 512, 2048)):
        x = rnd(B, pix, cin)
        w = rnd(kout, cin)
        y = torch.empty(B, pix, kout, device=DEV, dtype=BF)
        dd = dense_desc(pix, kout, cin, kout, batch=B, in_bstride=pix * cin, w_bstride=0, out_bstride=pix * kout)
        rows.append(('1x1 conv %d px %d -> %d' % (pix, cin, kout), 2.0 * B * pix * kout * cin,
                     (lambda x=x, w=w, y=y, dd=dd: L.call('tcvom_conv_igemm', L.ptr(x), L.ptr(w), L.ptr(y), None, None, None, None, C.byref(dd), st))))
    d2one = dense_desc(N, DV, ld, DV, out_fp32=True)
    rows.insert(2, ('O = P V, one frame', 2.0 * N * N * DV,
                    lambda: L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d2one), st)))
    dW2 = torch.empty(B, N, D, device=DEV)
    rows.append(('dWq, M\' in one launch', 4.0 * B * N * N * D,
                 lambda: L.call('tcvom_gemm_pair', L.ptr(P), L.ptr(Pt), L.ptr(Gt), L.ptr(dW), L.ptr(dW2), C.byref(d3), ld * ld, st)))
    # the launches the product actually makes (ops._GcaAttention): fused scores + softmax numerators, O = P V and dV with k-major
    # operands, dq / dk with their K-split tails
    stats = torch.empty(B, N, ld // 256, 2, device=DEV)
    dvec = torch.rand(B, N, device=DEV)
    Pn = torch.empty(B, N, ld, device=DEV, dtype=BF)
    dV = torch.empty(B, N, DV, device=DEV)
    Mp = torch.empty(B, N, D, device=DEV)
    rows += [
        ('P~ = exp(S - tile max) EPI3', 2.0 * B * N * N * D,
         lambda: L.call('tcvom_gca_scores_exp', L.ptr(G), L.ptr(cvec), L.ptr(dvec), L.ptr(Pn), L.ptr(stats), N, D, ld, B, st)),
        ('softmax rescale pass', 0.0, lambda: L.call('tcvom_gca_softmax_rescale', L.ptr(Pn), L.ptr(stats), N, ld, B, st)),
        ('O = P V  (V k-major)', 2.0 * B * N * N * DV, lambda: L.call('tcvom_gca_pv', L.ptr(P), L.ptr(V), L.ptr(O), N, DV, ld, B, st)),
        ('T = softmax bwd (T alone)', 2.0 * B * N * N * DV,
         lambda: L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T), None, None,
                        N, DV, ld, B, st)),
        ('dV = P^T dO (k-major)', 2.0 * B * N * N * DV, lambda: L.call('tcvom_gca_dv', L.ptr(P), L.ptr(dO), L.ptr(dV), N, DV, ld, B, st)),
        ('dq + dk (two launches)', 4.0 * B * N * N * D, lambda: L.call('tcvom_gca_dq_dk', L.ptr(P), L.ptr(Gt), L.ptr(dW), L.ptr(Mp), N, D, ld, B, st)),
    ]
    for name, flop, fn in rows:
        t = timeit(fn)
        print('%-30s %7.1f us  %6.0f TFLOP/s' % (name, t * 1e3, flop / t / 1e9))
    # numerics of the fused scores + softmax (pass 1 + rescale) on the first frame, rows 0..255
    G = (G.float() * 0.15).to(BF)                     # |S| of a few units: rows with many comparable entries, not one-hot ones
    L.call('tcvom_gca_scores_exp', L.ptr(G), L.ptr(cvec), L.ptr(dvec), L.ptr(Pn), L.ptr(stats), N, D, ld, B, st)
    L.call('tcvom_gca_softmax_rescale', L.ptr(Pn), L.ptr(stats), N, ld, B, st)
    Sr = (G[0, :256].float() @ G[0].float().t()) * cvec[0][None, :]
    Sr[torch.arange(256), torch.arange(256)] -= dvec[0, :256]
    Pr = torch.softmax(Sr, dim=1)
    perr = (Pn[0, :256, :N].float() - Pr).abs().max().item() / Pr.max().item()
    print('fused softmax max error / max P = %.2e ; padding columns zero: %s' % (perr, bool((Pn[0, :256, N:] == 0).all())))
    assert perr < 2e-2
    # numerics of O = P V on the first frame, rows 0..511
    L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d2), st)
    ref = P[0, :512].float() @ Vt[0].float().t()
    err = (O[0, :512] - ref).abs().max().item() / ref.abs().max().item()
    print('O = P V max error / max |O| = %.2e' % err)
    assert err < 2e-3


if __name__ == '__main__':
    main()
