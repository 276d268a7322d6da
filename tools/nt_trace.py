#!/usr/bin/env python
"""One-off: per-k-step cycle stamps of one workgroup of the dense NT GEMM (library built with -DNT_TRACE)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L
from tcvom_amd.conv_plan import dense_desc
N, DV = 8160, 2048
ld = 8192
P = torch.randn(N, ld, device='cuda').to(torch.bfloat16)
Vt = torch.randn(DV, ld, device='cuda').to(torch.bfloat16)
O = torch.empty(N, DV, device='cuda', dtype=torch.bfloat16)
d2 = dense_desc(N, DV, ld, DV)
st = L.stream_ptr()
for _ in range(3):
    L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d2), st)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
L._lib.tcvom_trace_read(buf, 2048)
a = np.array(buf[:], dtype=np.int64).reshape(2, 256, 4)
for w in range(2):
    t = a[w]
    n = 128
    print('wave', w, 'k-step period (cycles) median', np.median(np.diff(t[:n, 0])))
    print('  wait vmcnt  median', np.median(t[:n, 1] - t[:n, 0]), ' barrier', np.median(t[:n, 2] - t[:n, 1]),
          ' dma issue', np.median(t[:n, 3] - t[:n, 2]), ' reads+mfma (to next top)', np.median(t[1:n, 0] - t[:n - 1, 3]))
    print('  first 12 steps:', (t[:12, :] - t[0, 0]).tolist())
