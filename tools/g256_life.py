#!/usr/bin/env python
"""Life cycle of a gemm_nt256 workgroup on the short-reduction products (scores + softmax numerators, EPI 3; the plain fp32 S product):
100 MHz time stamps at entry / first K-tile landed / main loop done / epilogue phases / stores issued / stores retired, for a workgroup
of the first round and one of a late round.  Needs a library built with -DG256_LIFE:
    make -C tcvom_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -DG256_LIFE" LIB=../lib/libtcvom_hip_life.so ../lib/libtcvom_hip_life.so
    TCVOM_LIB=$PWD/tcvom_amd/lib/libtcvom_hip_life.so python tools/g256_life.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import dense_desc                           # noqa: E402

B, N, D, DV = 3, 8160, 576, 2048
ld = (N + 255) // 256 * 256
BF = L.ACT_DTYPE
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s: (torch.rand(*s, device='cuda', generator=g) * 2 - 1).to(BF)      # noqa: E731
G = rnd(B, N, D)
cvec, dvec = torch.rand(B, N, device='cuda') + 0.5, torch.rand(B, N, device='cuda')
Pn = torch.empty(B, N, ld, device='cuda', dtype=BF)
stats = torch.empty(B, N, ld // 256, 2, device='cuda')
S = torch.empty(B, N, ld, device='cuda')
P, V = rnd(B, N, ld), rnd(B, N, DV)
O = torch.empty(B, N, DV, device='cuda')
d1 = dense_desc(N, N, D, ld, batch=B, in_bstride=N * D, w_bstride=N * D, out_bstride=N * ld, vec_bstride=N, out_fp32=True)
st = L.stream_ptr()
fn = L._lib.tcvom_life256_read
fn.argtypes = [C.c_void_p]
dO, Vv = rnd(B, N, DV), rnd(B, N, DV)
Pp = (torch.rand(B, N, ld, device='cuda', generator=g) * 1e-3).to(BF)
delta, T = torch.rand(B, N, device='cuda'), torch.empty(B, N, ld, device='cuda', dtype=BF)
names = ['entry', 'K-tile 0 landed', 'main loop done', 'max pass', 'exp pass', 'stores issued', 'stores retired']
for what, call in (('scores + softmax numerators (EPI 3, K = 576)', lambda: L.call('tcvom_gca_scores_exp', L.ptr(G), L.ptr(cvec), L.ptr(dvec), L.ptr(Pn), L.ptr(stats), N, D, ld, B, st)),
                   ('S = G G^T fp32 (EPI 0, K = 576)', lambda: L.call('tcvom_conv_igemm', L.ptr(G), L.ptr(G), L.ptr(S), None, None, None, None, C.byref(d1), st)),
                   ('T = P (dO V^T - delta) c (EPI 2, K = 2048; stamps: P tile landed / T computed in LDS)',
                    lambda: L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(Vv), L.ptr(Pp), L.ptr(delta), L.ptr(cvec), L.ptr(T), None, None, N, DV, ld, B, st)),
                   ('O = P V (K = 8192)', lambda: L.call('tcvom_gca_pv', L.ptr(P), L.ptr(V), L.ptr(O), N, DV, ld, B, st))):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    assert fn(C.cast(buf, C.c_void_p)) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(4, 16)
    print('%s: launch %.1f us' % (what, e0.elapsed_time(e1) * 1e3))
    for slot, nm in ((0, 'first-round workgroup'), (1, 'late-round workgroup')):
        t = a[slot, :7]
        rel = (t - t[0]) / 100.0
        print('  %-22s' % nm, '  '.join('%s %.2f' % (n, r) for n, r in zip(names, rel) if r >= 0), ' (us since entry)')
