#!/usr/bin/env python
"""Cycle stamps at every workgroup barrier of gemm_nt256 for one workgroup (waves 0 and 4 = the two staggered groups), on
the GCA P.V shape: per section of the K-tile (G256_PHASES = 2 or 4, as the library was built), the wave's own work and its wait at the closing barrier.
Needs a library built with -DG256_TRACE:
    make -C tcvom_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -DG256_TRACE"
    TCVOM_LIB=<that .so> python tools/g256_trace.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import dense_desc                           # noqa: E402

N, DV = 8160, 2048
ld = (N + 63) // 64 * 64
P = torch.randn(N, ld, device='cuda').to(torch.bfloat16)
Vt = torch.randn(DV, ld, device='cuda').to(torch.bfloat16)
O = torch.empty(N, DV, device='cuda', dtype=torch.bfloat16)
d = dense_desc(N, DV, ld, DV)
st = L.stream_ptr()
for _ in range(int(os.environ.get('G256_LAUNCHES', '40'))):           # enough back-to-back launches to reach the steady clock
    L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d), st)
torch.cuda.synchronize()
buf = (C.c_uint64 * 512)()
fn = L._lib.tcvom_trace256_read
fn.argtypes = [C.c_void_p]
assert fn(C.cast(buf, C.c_void_p)) == 0
a = np.array(buf[:], dtype=np.int64)
names = {0: ['M1*', 'L2', 'M2*', 'L3', 'M3', 'L4', 'M4w', 'L1'], 1: ['M1*', 'L2', 'M2*', 'L3', 'M3', 'L4w', 'M4', 'L1']}
print('main loop of the traced workgroup: %d shader cycles in %.1f us -> %.2f GHz' % (a[254], a[255] / 100.0, a[254] / (a[255] * 10.0)))
NS = int(os.environ.get('G256_PHASES', '2')) * 2                 # sections per K-tile; must match the library's G256_PHASES
if NS == 4:
    names = {0: ['MA*', 'LB', 'MBw', 'LA'], 1: ['MA*', 'LBv', 'MBw', 'LA']}
for grp in range(2):
    t = a[grp * 256:grp * 256 + 250].reshape(-1, 2)            # (arrive, leave) per barrier
    arrive, leave = t[:, 0], t[:, 1]
    first = 1 + grp                                              # barrier index after which L1 of K-tile 0 starts
    work = arrive[first + 1:] - leave[first:-1]                  # own section between two barriers
    wait = leave[first + 1:] - arrive[first + 1:]                # waiting for the others at the barrier that ends it
    nt = (len(work) // NS) - 1
    W = work[:NS * nt].reshape(nt, NS)[3:]
    Q = wait[:NS * nt].reshape(nt, NS)[3:]
    print('group %d   %s' % (grp, '  '.join('%5s' % n for n in names[grp])))
    print('  work    ', '  '.join('%5.0f' % v for v in W.mean(0)), '  sum %.0f' % W.mean(0).sum())
    print('  wait    ', '  '.join('%5.0f' % v for v in Q.mean(0)), '  sum %.0f' % Q.mean(0).sum())
