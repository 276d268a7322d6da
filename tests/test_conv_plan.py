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


# ------------------------------------------------------------------------------------------------ round 6: host-side planners (no GPU)
def _geo(cin, cout, k, stride, transposed, N, H, W, f16=False):
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    spec = ConvSpec('p', torch.zeros(shape), None, None, None, transposed, stride, 1 if k in (3, 4) else 0, 'frame', needs_dgrad=False)
    spec.f16 = f16
    return spec, ConvGeometry(spec, N, H, W)


def test_hetero_wgrad_plan_covers_every_pixel_tile_and_phase_exactly_once():
    """tcvom_wgrad_igemm_hetero_plan (csrc/igemm.hip, host code): for a launch of problems with different descriptors the work list must
    hold, for every (problem, phase, column tile, row tile), pixel chunks that tile the phase's pixels without gap or overlap; the
    descriptor index of an item is the problem's first descriptor + its phase; ldy and the chunk length ride along."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.weights import _wgrad_tt_tile
    probs = [_geo(64, 128, 3, 2, False, 1, 272, 480), _geo(128, 128, 1, 1, False, 3, 136, 240), _geo(512, 512, 4, 2, True, 1, 34, 60),
             _geo(256, 128, 1, 1, False, 1, 68, 120), _geo(128, 256, 3, 2, False, 2, 40, 36)]
    assert {_wgrad_tt_tile(g) for _, g in probs} == {(128, 128)}
    tm = tn = 128
    descs = [d for _, g in probs for d in g.wgrad]
    arr = (L.ConvDesc * len(descs))(*descs)
    n = len(probs)
    nph = (C.c_int32 * n)(*[len(g.wgrad) for _, g in probs])
    ldy = (C.c_int32 * n)(*[s.K for s, _ in probs])
    nwork = L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldy, C.c_void_p), n, tm, tn, None, 0)
    assert nwork > 0
    work = (C.c_int32 * (8 * nwork))()
    assert L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldy, C.c_void_p), n, tm, tn, C.cast(work, C.c_void_p), nwork) == nwork
    # a too small table is refused
    assert L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldy, C.c_void_p), n, tm, tn, C.cast(work, C.c_void_p), nwork - 1) < 0
    w = np.ctypeslib.as_array(work).reshape(nwork, 8)
    first = np.cumsum([0] + [len(g.wgrad) for _, g in probs])
    seen = {}
    for prob_desc, chunk, tiles, ld, pchunk, *_ in w.tolist():
        prob, di = prob_desc & 0xff, prob_desc >> 8
        ph = di - first[prob]
        spec, geo = probs[prob]
        assert 0 <= ph < len(geo.wgrad) and ld == spec.K and pchunk % 64 == 0 and pchunk >= 512
        d = geo.wgrad[ph]
        nt, mt = -(-d.ntaps * d.C // tn), -(-d.K // tm)
        y, z = tiles & 0xffff, tiles >> 16
        assert y < nt and z < mt
        seen.setdefault((prob, ph, y, z), []).append((chunk, pchunk))
    for prob, (spec, geo) in enumerate(probs):
        for ph, d in enumerate(geo.wgrad):
            P = d.N * d.PH * d.PW
            for y in range(-(-d.ntaps * d.C // tn)):
                for z in range(-(-d.K // tm)):
                    items = sorted(seen.pop((prob, ph, y, z)))
                    pc = items[0][1]
                    assert [c for c, _ in items] == list(range(len(items))) and all(p == pc for _, p in items)
                    assert (len(items) - 1) * pc < P <= len(items) * pc, 'chunks must tile the %d pixels of problem %d phase %d' % (P, prob, ph)
    assert not seen
    # problems of another tile shape in the same launch are refused
    mixed = probs + [_geo(6, 32, 3, 2, False, 1, 64, 64)]
    descs = [d for _, g in mixed for d in g.wgrad]
    arr = (L.ConvDesc * len(descs))(*descs)
    nph = (C.c_int32 * (n + 1))(*[len(g.wgrad) for _, g in mixed])
    ldy = (C.c_int32 * (n + 1))(*[s.K for s, _ in mixed])
    assert L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldy, C.c_void_p), n + 1, tm, tn, None, 0) < 0


@pytest.mark.parametrize('H,W', [(64, 64), (64, 96), (128, 160), (256, 320), (512, 512), (544, 960), (1088, 1920)])
def test_fp16_island_descriptors_select_kernels_that_serve_them(H, W):
    """The forward descriptors of the fp16 island (in_f16 = 1, out_fp32 = 2; gca_net.ISLAND_LAYERS) at every tested window size, 1 and 3
    frames per launch: the planner names one of the three kernels with an IEEE fp16 operand mode (halo_conv, wsconv, igemm_nt below the
    256-row tile) -- never pwconv / sconv / gemm_nt256 -- and the weight-gradient descriptors of the same layers carry no such flag."""
    import ctypes as C
    from tcvom_amd import _lib as L
    layers = [(6, 32, 3, 2, 1), (32, 32, 3, 1, 2), (32, 64, 3, 2, 2), (64, 64, 3, 1, 4), (64, 128, 3, 2, 4), (128, 128, 3, 1, 8), (64, 128, 1, 1, 8)]
    for cin, cout, k, stride, ds in layers:
        spec, geo = _geo(cin, cout, k, stride, False, 1, H // ds, W // ds, f16=True)
        assert all(d.in_f16 == 1 and d.out_fp32 == 2 for d in geo.fwd) and all(d.in_f16 == 0 and d.out_fp32 == 0 for d in geo.wgrad)
        for nf in (1, 3):
            arr = (L.ConvDesc * 1)(*geo.fwd)
            d = arr[0]
            d.batch = nf
            if nf > 1:
                d.in_bstride, d.out_bstride, d.w_bstride, d.vec_bstride = d.N * d.H * d.W * d.C, d.N * d.OH * d.OW * d.ldo, d.K * d.wt * d.C, 0
            var = L._FNS['tcvom_conv_igemm_variant'](C.byref(d), 1).decode()
            assert var.startswith(('halo_conv', 'wsconv', 'igemm_nt<')) and not var.startswith('igemm_nt<256'), (cin, cout, k, stride, H, W, nf, var)
            assert L.call('tcvom_conv_stats_groups', C.byref(d), 1) > 0
            # the same layer without the flags may go elsewhere (the 1 x 1 downsample conv: pwconv at the large sizes) -- with them it may not
            d.in_f16, d.out_fp32 = 0, 0
            plain = L._FNS['tcvom_conv_igemm_variant'](C.byref(d), 1).decode()
            if plain.startswith(('pwconv', 'sconv', 'gemm_nt256')):
                assert not var.startswith(plain.split('<')[0])
