#!/usr/bin/env python
"""One-off: entry / loop / epilogue cycle stamps of one workgroup of a conv igemm (library built with -DNT_TRACE)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L
from tcvom_amd.conv_plan import ConvGeometry
from tcvom_amd.weights import ConvSpec, WeightBank
from tcvom_amd.ops import _launch_conv
SHAPES = [('os8 128', 128, 128, 136, 240), ('os16 256', 256, 256, 68, 120), ('os32 512', 512, 512, 34, 60), ('os4 64', 64, 64, 272, 480),
          ('os2 32', 32, 32, 544, 960)]
st = L.stream_ptr()
for name, cin, cout, H, W in SHAPES:
    w = nn.Parameter(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05)
    bank = WeightBank()
    spec = ConvSpec(name, w, None, None, None, False, 1, 1, 'frame')
    bank.register(spec)
    bank.prepare(1, True)
    geo = ConvGeometry(spec, 1, H, W)
    x = torch.randn(1, H, W, cin, device='cuda').to(torch.bfloat16)
    y = torch.empty(1, H, W, cout, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        _launch_conv(geo.fwd, x, bank.fwd_ptr(spec, 0), y, None, None, 0, st)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8192)()
    L._lib.tcvom_trace_read(buf, 8192)
    a = np.array(buf[:], dtype=np.int64)
    for wv in range(2):
        e = a[4096 + wv * 8: 4096 + wv * 8 + 4]
        steps = a[wv * 1024: wv * 1024 + 1024].reshape(256, 4)
        n = (cin * 9) // 64
        per = np.diff(steps[:n, 0])
        print('%s wave%d: prologue %d  loop %d (%d steps, median period %d)  epilogue %d  total %d cycles' % (
            name, wv, e[1] - e[0], e[2] - e[1], n, np.median(per) if len(per) else 0, e[3] - e[2], e[3] - e[0]))
        if n > 2:
            t = steps[:n]
            print('    per step: wait vmcnt %d  barrier %d  dma issue %d  reads + mfma %d' % (
                np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(t[1:, 0] - t[:-1, 3])))
