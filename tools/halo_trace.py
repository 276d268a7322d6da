#!/usr/bin/env python
"""Cycle stamps of one workgroup (wave 0) of the halo conv kernel on a 32 -> 32 channel 3x3 layer at 1088x1920, 3 frames: per tile
the wait for the halo DMA + barrier, the issue of the next tile's DMA, the MFMA section and the epilogue.  Needs a library built
with -DHALO_TRACE (TCVOM_LIB=<that .so>)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcvom_amd import _lib as L                                      # noqa: E402
from tcvom_amd.conv_plan import ConvGeometry                         # noqa: E402
from tcvom_amd.weights import ConvSpec, WeightBank                   # noqa: E402
from tcvom_amd.ops import _launch_conv, _stats_groups                # noqa: E402

H, W = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (1088, 1920)))
w = nn.Parameter(torch.randn(32, 32, 3, 3, device='cuda') * 0.05)
bank = WeightBank()
spec = ConvSpec('t', w, None, None, None, False, 1, 1, 'frame')
bank.register(spec)
bank.prepare(3, True)
geo = ConvGeometry(spec, 1, H, W)
x = torch.randn(3, H, W, 32, device='cuda').to(torch.bfloat16)
y = torch.empty(3, H, W, 32, device='cuda', dtype=torch.bfloat16)
stats = torch.empty(3 * _stats_groups(geo.fwd, 3) * 2 * 32, device='cuda')
st = L.stream_ptr()
for _ in range(3):
    _launch_conv(geo.fwd, x, bank.fwd_ptr(spec, 0), y, None, stats, 1, st, 3, bank.fwd_stride)
torch.cuda.synchronize()
buf = (C.c_uint64 * 1024)()
fn = L._lib.tcvom_halo_trace_read
fn.argtypes = [C.c_void_p]
assert fn(C.cast(buf, C.c_void_p)) == 0
a = np.array(buf[:1000], dtype=np.int64).reshape(200, 5)
a = a[(a[:, 4] > a[:, 0]) & (a[:, 0] > 0)][2:-1]
d = np.diff(a, axis=1)
print('%d tiles: wait+barrier %.0f  dma issue %.0f  mfma %.0f  epilogue %.0f  | tile period %.0f cycles' % (
    len(a), d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), np.diff(a[:, 0]).mean()))
