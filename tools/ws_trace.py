#!/usr/bin/env python
"""Cycle stamps of one workgroup of the weight-stationary conv (csrc/wsconv.hip), env TCVOM_CONV_TRACE=1:
prologue (halo issue, weight load) and per tile barrier wait / halo issue / MFMA loop / DMA wait / epilogue."""
import ctypes as C
import os
import sys
import numpy as np
import torch
import torch.nn as nn
os.environ.setdefault('TCVOM_CONV_TRACE', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                       # noqa: E402
from tcvom_amd.conv_plan import ConvGeometry          # noqa: E402
from tcvom_amd.weights import ConvSpec, WeightBank    # noqa: E402
from tcvom_amd.ops import _launch_conv                # noqa: E402

st = L.stream_ptr()
for name, c, H, W in (('os8 128', 128, 136, 240), ('os4 64', 64, 272, 480)):
    for nf in (1, 3):
        w = nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05)
        bank = WeightBank()
        spec = ConvSpec(name, w, None, None, None, False, 1, 1, 'frame')
        bank.register(spec)
        bank.prepare(1, True)
        geo = ConvGeometry(spec, 1, H, W)
        x = torch.randn(nf, H, W, c, device='cuda').to(torch.bfloat16)
        y = torch.empty(nf, H, W, c, device='cuda', dtype=torch.bfloat16)
        for _ in range(3):
            _launch_conv(geo.fwd, x, bank.fwd_ptr(spec, 0), y, None, None, 0, st, nf, 0)
        torch.cuda.synchronize()
        buf = (C.c_uint64 * 64)()
        L.call('tcvom_conv_trace_read', C.cast(buf, C.c_void_p), 64)
        a = np.array(buf[:], dtype=np.int64)
        t0 = a[0]
        print('%s nf=%d: halo issue %d, weights loaded %d, end %d cycles after entry' % (name, nf, a[1] - t0, a[2] - t0, a[3] - t0))
        prev = a[2]
        for k in range(15):
            s4 = a[4 + 4 * k: 8 + 4 * k]
            if s4[0] == 0:
                break
            last = s4[2] == 0
            print('   tile %d: barrier +%d  k-steps (+ DMA issue + previous epilogue) +%d' % (k, s4[0] - prev, s4[1] - s4[0]) +
                  ('' if last else '  hand-off +%d  vmcnt wait +%d' % (s4[2] - s4[1], s4[3] - s4[2])))
            prev = s4[1] if last else s4[3]
        print('   last epilogue +%d' % (a[3] - prev))
