"""Host logic of the implicit-GEMM descriptors: a numpy emulation of exactly what the kernel computes from a
tcvom_conv_desc (gather taps, multiply by packed weights, scatter to the phase grid) must reproduce
F.conv2d / F.conv_transpose2d and their data gradients for every conv flavour of the network."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tcvom_amd.conv_plan import ConvGeometry
from tcvom_amd.weights import ConvSpec


def emulate(descs, x, wpack, out_shape):
    """x [N,H,W,C]; wpack [K][wt][C]; returns out [N,OH,OW,K] following the kernel's index arithmetic."""
    out = np.zeros(out_shape, dtype=np.float64)
    written = np.zeros(out_shape[:3], dtype=np.int32)
    for d in descs:
        for n in range(d.N):
            for i in range(d.PH):
                for j in range(d.PW):
                    acc = np.zeros(d.K)
                    for t in range(d.ntaps):
                        ws = d.tap_w[t]
                        if ws < 0:
                            continue
                        ih, iw = i * d.in_step + d.tap_dh[t], j * d.in_step + d.tap_dw[t]
                        if 0 <= ih < d.H and 0 <= iw < d.W:
                            acc += wpack[:, ws, :] @ x[n, ih, iw, :]
                    oh, ow = i * d.out_step + d.out_off_h, j * d.out_step + d.out_off_w
                    out[n, oh, ow, :] = acc
                    written[n, oh, ow] += 1
        assert ((d.ntaps * d.C + 63) // 64 * 64 - 1) // d.C < 32        # the last 64-deep step stays inside the tap table
    assert (written == 1).all(), 'every output pixel must be produced by exactly one phase'
    return out


CASES = [('c3s1', 16, 8, 3, 1, 1, False), ('c3s2', 8, 16, 3, 2, 1, False), ('c1', 16, 8, 1, 1, 0, False),
         ('c3s2p0', 16, 8, 3, 2, 0, False), ('convT', 8, 16, 4, 2, 1, True)]


@pytest.mark.parametrize('name,cin,cout,k,stride,pad,transposed', CASES, ids=[c[0] for c in CASES])
def test_geometry_matches_torch(name, cin, cout, k, stride, pad, transposed):
    torch.manual_seed(0)
    N, H, W = 1, (7 if pad == 0 else 6), (9 if pad == 0 else 8)
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    w = torch.randn(shape, dtype=torch.float64)
    spec = ConvSpec(name, w, None, None, None, transposed, stride, pad, 'frame')
    geo = ConvGeometry(spec, N, H, W)
    x = torch.randn(N, cin, H, W, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride, pad) if transposed else F.conv2d(x, w, None, stride, pad)
    # packed weights exactly as sn_pack_kernel writes them
    T = k * k
    if transposed:
        fwd = w.permute(1, 2, 3, 0).reshape(cout, T, cin)           # [K][T][C] = W[c][k][t]
        bwd = w.permute(0, 2, 3, 1).reshape(cin, T, cout)           # [C][T][K]
    else:
        fwd = w.permute(0, 2, 3, 1).reshape(cout, T, cin)
        bwd = w.permute(1, 2, 3, 0).reshape(cin, T, cout)
    xn = x.detach().permute(0, 2, 3, 1).numpy()
    out = emulate(geo.fwd, xn, fwd.numpy(), (N, geo.OH, geo.OW, cout))
    np.testing.assert_allclose(out, y.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    gy = torch.randn_like(y)
    y.backward(gy)
    dx = emulate(geo.dgrad, gy.permute(0, 2, 3, 1).numpy(), bwd.numpy(), (N, H, W, cin))
    np.testing.assert_allclose(dx, x.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)


def test_small_channel_inputs_are_padded_to_eight():
    w = torch.zeros(32, 6, 3, 3)
    spec = ConvSpec('stem', w, None, None, None, False, 2, 1, 'frame', needs_dgrad=False)
    geo = ConvGeometry(spec, 1, 8, 8)
    d = geo.fwd[0]
    # 9 taps x 8 channels = 72 elements = two 64-deep steps; the second ends in the zero-tap slots 9 .. 15
    assert spec.cpad == 8 and d.C == 8 and d.ntaps == 9 and list(d.tap_w)[9:16] == [-1] * 7
    assert not geo.dgrad
