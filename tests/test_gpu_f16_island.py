"""The fp16 island of the bf16 build (tcvom_amd/ops.py: F16_ISLAND): encoder stem, layer1 and layer2 of the GCA network
(/root/reference models/GCA/encoders/resnet_enc.py:70-84,130-138) run their FORWARD on IEEE fp16 weights, activations and conv outputs
whatever the build stores -- tcvom_conv_desc.in_f16 = 1, out_fp32 = 2, tcvom_bn_apply_f16, tcvom_avgpool2_f16, the fp16 forward pack of
tcvom_sn_pack (kind bit 64).  Every kernel that serves an island layer at ANY of the tested window sizes is checked here against plain fp32
PyTorch on the same fp16-rounded operands (tight: only the fp16 rounding of the output and the summation order differ), then a BasicBlock
through the ops (both formats of z, mask, gradients), then the whole encoder's routing.  Skipped in the fp16 build (it is one island)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tcvom_amd._lib import ACT_DTYPE as H16
from helpers import hu, Checker
from tcvom_amd.synthetic import formula_tensor

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(H16 != torch.bfloat16, reason='the fp16 build has no island: every layer is IEEE fp16')]
DEV = 'cuda'


def f16(t):
    return t.to(torch.float16).float()


def bf(t):
    return t.to(torch.bfloat16).float()


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(DEV)


def nchw(t):
    return t.detach().float().cpu().permute(0, 3, 1, 2)


def rel_err(got, want):
    got, want = got.double(), want.double()
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


# name, cin, cout, k, stride, N, H, W, expected kernel prefix (None: whatever the planner picks, named in the output)
ISLAND_CONVS = [
    ('stem_conv1_halo8', 6, 32, 3, 2, 1, 64, 128, 'halo_conv<8>'),
    ('stem_conv1_igemm', 6, 32, 3, 2, 2, 24, 40, 'igemm_nt'),
    ('stem_conv2_halo32', 32, 32, 3, 1, 1, 32, 64, 'halo_conv<32>'),
    ('stem_conv2_igemm', 32, 32, 3, 1, 1, 20, 28, 'igemm_nt'),
    ('stem_conv3_igemm', 32, 64, 3, 2, 2, 24, 40, 'igemm_nt'),
    ('layer1_wsconv64', 64, 64, 3, 1, 2, 20, 44, 'wsconv<64>'),
    ('layer1_wsconv64_ragged', 64, 64, 3, 1, 1, 13, 37, 'wsconv<64>'),
    ('layer2_first_s2', 64, 128, 3, 2, 1, 32, 48, 'igemm_nt'),
    ('layer2_wsconv128', 128, 128, 3, 1, 1, 24, 40, 'wsconv<128>'),
    ('layer2_wsconv128_ragged', 128, 128, 3, 1, 2, 9, 21, 'wsconv<128>'),
    ('layer2_down_1x1', 64, 128, 1, 1, 1, 16, 24, 'igemm_nt'),
    ('layer2_down_1x1_big', 64, 128, 1, 1, 1, 64, 160, 'igemm_nt'),      # (pwconv's shape without the island: it must decline)
]


def _island_layer(name, cin, cout, k, stride, spectral=True):
    from tcvom_amd.weights import ConvSpec, WeightBank
    shape = (cout, cin, k, k)
    w = nn.Parameter((formula_tensor('isl.%s.weight' % name, shape) * 0.5).to(DEV))
    u = v = None
    if spectral:
        u = nn.Parameter(formula_tensor('isl.%s.u' % name, (cout,)).to(DEV), requires_grad=False)
        v = nn.Parameter(formula_tensor('isl.%s.v' % name, (cin * k * k,)).to(DEV), requires_grad=False)
    bank = WeightBank()
    spec = ConvSpec('isl.' + name, w, u, v, None, False, stride, (k - 1) // 2, 'frame')
    spec.f16 = True
    bank.register(spec)
    return bank, spec


def _sigma_weight(spec, training=True):
    """fp32 SpectralNorm of the layer as the reference computes it (one power iteration from the stored u, v: models/GCA/ops.py:25-45)."""
    w = spec.weight.detach().cpu().double()
    if spec.u is None:
        return w.float()
    h = w.shape[0]
    wm = w.view(h, -1)
    u, v = spec.u.detach().cpu().double(), spec.v.detach().cpu().double()
    if training:
        v = wm.t() @ u
        v = v / (v.norm() + 1e-12)
        u = wm @ v
        u = u / (u.norm() + 1e-12)
    sigma = u @ (wm @ v)
    return (w / sigma).float()


@pytest.mark.parametrize('case', ISLAND_CONVS, ids=[c[0] for c in ISLAND_CONVS])
@pytest.mark.parametrize('nf', [1, 3])
def test_island_conv_kernels(case, nf):
    """Raw forward launch of an island layer: IEEE fp16 input x IEEE fp16 packed weight -> IEEE fp16 output + BatchNorm partial sums,
    for nf frames in one launch (each with its own SpectralNorm'd weight copy)."""
    from tcvom_amd import _lib as L
    from tcvom_amd import ops
    from tcvom_amd.conv_plan import ConvGeometry
    name, cin, cout, k, stride, N, H, W, want = case
    bank, spec = _island_layer(name, cin, cout, k, stride)
    snaps = []
    w_calls = []
    u0, v0 = spec.u.detach().clone(), spec.v.detach().clone()
    # the reference weights of the nf calls: chained power iterations
    with torch.no_grad():
        for f in range(nf):
            w = spec.weight.detach().cpu().double().view(cout, -1)
            u, v = (u0 if f == 0 else snaps[-1][0]).cpu().double(), (v0 if f == 0 else snaps[-1][1]).cpu().double()
            v = w.t() @ u
            v = v / (v.norm() + 1e-12)
            u = w @ v
            u = u / (u.norm() + 1e-12)
            snaps.append((u, v))
            w_calls.append((spec.weight.detach().cpu().double() / (u @ (w @ v))).float())
    bank.prepare(nf, True)
    geo = ConvGeometry(spec, N, H, W)
    d = geo.fwd[0]
    assert d.in_f16 == 1 and d.out_fp32 == 2 and geo.wgrad[0].in_f16 == 0
    arr = ops._phase_array(geo.fwd)
    ops._set_frames(arr, 1, nf, bank.fwd_stride if nf > 1 else 0)
    var = L._FNS['tcvom_conv_igemm_variant'](C.byref(arr[0]), 1).decode()
    assert var.startswith(want), (var, want)
    x = hu('isl.x.' + name, (nf * N, cin, H, W)) * 2 - 1
    cp = spec.cpad
    x16 = torch.zeros(nf * N, H, W, cp, dtype=torch.float16, device=DEV)
    x16[..., :cin] = nhwc16(x)
    OH, OW = geo.OH, geo.OW
    y = torch.full((nf * N, OH, OW, cout), float('nan'), dtype=torch.float16, device=DEV)
    groups = ops._stats_groups(geo.fwd, nf)
    stats = torch.full((nf * groups * 2 * cout,), float('nan'), dtype=torch.float32, device=DEV)
    call, wsf, _ = bank.next_calls(spec, nf)
    ops._launch_conv(geo.fwd, x16, bank.fwd_ptr(spec, call), y, None, stats, 0, L.stream_ptr(), nf, wsf)
    torch.cuda.synchronize()
    worst = 0.0
    for f in range(nf):
        yref = F.conv2d(f16(x[f * N:(f + 1) * N]), f16(w_calls[f]), None, stride, (k - 1) // 2)
        got = nchw(y[f * N:(f + 1) * N])
        assert torch.isfinite(got).all()
        e = rel_err(got, yref)
        worst = max(worst, e)
        assert e < 1.2e-3, '%s frame %d: %.3e (%s)' % (name, f, e, var)        # fp16 output rounding: 2^-11 = 4.9e-4 of |y|
        sums = stats.view(nf, groups, 2, cout)[f].double().sum(0).cpu()
        assert torch.isfinite(sums).all()
        assert rel_err(sums[0], yref.double().sum((0, 2, 3))) < 2e-4 and rel_err(sums[1], (yref.double() ** 2).sum((0, 2, 3))) < 2e-4
    print('%s nf=%d %s: max rel err %.2e' % (name, nf, var, worst))


def test_fp16_forward_pack_keeps_the_backward_pack_in_bf16():
    """tcvom_sn_pack kind bit 64: the forward pack is IEEE fp16, the data-gradient pack the build's bf16 -- both from one LDS image, in
    the tiled form and in the one-thread-per-element form (plain and fragment-major layouts)."""
    from tcvom_amd import weights as Wm
    for name, cin, cout, k in (('pk64', 64, 64, 3), ('pk1x1', 64, 128, 1), ('pk32', 32, 64, 3), ('pk8', 6, 32, 3)):
        for tiled in (True, False):
            old = Wm.TILED_PACK
            Wm.TILED_PACK = tiled
            try:
                bank, spec = _island_layer(name, cin, cout, k, 1)
                bank.prepare(1, False)          # eval: sigma from the stored u, v
                torch.cuda.synchronize()
            finally:
                Wm.TILED_PACK = old
            wn = _sigma_weight(spec, training=False)
            T, cp = k * k, spec.cpad
            fwd = bank.fwd_arena[spec.fwd_off:spec.fwd_off + cout * T * cp].view(torch.float16).float().cpu()
            bwd = bank.bwd_arena[spec.bwd_off:spec.bwd_off + cin * T * cout].float().cpu()
            want_f = torch.zeros(cout, T, cp)
            want_f[:, :, :cin] = f16(wn).view(cout, cin, T).permute(0, 2, 1)
            want_b = bf(wn).view(cout, cin, T).permute(1, 2, 0).contiguous()
            if spec.frag:
                idx = lambda row, slot, col, TT, ncols: ((((row >> 5) * TT + slot) * (ncols >> 4) + (col >> 4)) * 64 + ((col >> 3) & 1) * 32 + (row & 31)) * 8 + (col & 7)
                kk, tt, cc = torch.meshgrid(torch.arange(cout), torch.arange(T), torch.arange(cp), indexing='ij')
                got_f = fwd[idx(kk, tt, cc, T, cp)]
                c2, t2, k2 = torch.meshgrid(torch.arange(cin), torch.arange(T), torch.arange(cout), indexing='ij')
                got_b = bwd[idx(c2, t2, k2, T, cout)]
            else:
                got_f, got_b = fwd.view(cout, T, cp), bwd.view(cin, T, cout)
            # sigma is computed on the device in fp32: compare after rounding with one unit of slack
            assert (got_f - want_f).abs().max() <= 2.0 ** -10 * want_f.abs().max(), (name, tiled)
            assert (got_b - want_b).abs().max() <= 2.0 ** -7 * want_b.abs().max(), (name, tiled)
            # the forward pack really is fp16-valued (not bf16 bits): most entries are not representable in bf16
            assert float((bf(got_f) != got_f).float().mean()) > 0.5


def _bn(Cc, tag):
    bn = nn.BatchNorm2d(Cc).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(hu(tag + '.g', (Cc,)) * 0.4 + 0.8)
        bn.bias.copy_(hu(tag + '.b', (Cc,)) * 0.4 - 0.2)
    return bn


@pytest.mark.parametrize('Cc,N,H,W,down', [(64, 2, 16, 24, False), (128, 1, 24, 40, False), (64, 1, 32, 48, True)])
def test_island_basic_block(Cc, N, H, W, down):
    """An encoder BasicBlock inside the island (resnet_enc.py:17-49; `down`: the first block of layer2 -- stride-2 conv1, AvgPool2d +
    1x1 conv + BatchNorm on the identity path): z comes back in bf16 (autograd's tensor) WITH an IEEE fp16 twin that the next op reads;
    the twin agrees with an fp32 evaluation to fp16 precision (the bf16 tensor only to bf16 precision); gradients against fp32 autograd
    with the storage points of the HIP path (fp16 in the forward, bf16 in the backward)."""
    from tcvom_amd import ops
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    assert ops.F16_ISLAND
    Co = 2 * Cc if down else Cc
    bank = WeightBank()

    def layer(tag, ci, co, k, stride):
        w = nn.Parameter((formula_tensor('blk.%s' % tag, (co, ci, k, k)) * (0.6 if k == 3 else 1.5)).to(DEV))
        spec = ConvSpec('blk.' + tag, w, None, None, None, False, stride, (k - 1) // 2, 'frame')
        spec.f16 = True
        bank.register(spec)
        return w, spec
    w0, s0 = layer('pre%d' % Cc, Cc, Cc, 3, 1)
    w1, s1 = layer('c1%d%d' % (Cc, down), Cc, Co, 3, 2 if down else 1)
    w2, s2 = layer('c2%d%d' % (Cc, down), Co, Co, 3, 1)
    bn0, bn1, bn2 = _bn(Cc, 'blk0'), _bn(Co, 'blk1'), _bn(Co, 'blk2')
    cfg0 = ops.ConvCfg(bank, s0, bn=bn0, act=ops.ACT_RELU)
    cfg1 = ops.ConvCfg(bank, s1, bn=bn1, act=ops.ACT_RELU)
    cfg2 = ops.ConvCfg(bank, s2, bn=bn2, act=ops.ACT_RELU)
    if down:
        wd, sd = layer('dn%d' % Cc, Cc, Co, 1, 1)
        bnd = _bn(Co, 'blkd')
        cfgd = ops.ConvCfg(bank, sd, bn=bnd, act=ops.ACT_NONE)
    x = hu('blk.x%d%d' % (Cc, down), (N, Cc, H, W)) * 2 - 1
    xg = x.permute(0, 2, 3, 1).contiguous().to(H16).to(DEV).requires_grad_(True)
    ops.set_f16_twin(xg, nhwc16(x))
    token = bank_token(bank, 1, True)
    x0 = ops.conv_bn_act(cfg0, xg, token, True)                 # an island producer: x0 has a twin
    assert ops.f16_twin(x0) is not None and ops.f16_twin(x0).dtype == torch.float16 and x0.dtype == H16
    idt = x0
    if down:
        pooled = ops.avgpool2(x0)
        assert ops.f16_twin(pooled) is not None
        idt = ops.conv_bn_act(cfgd, pooled, token, True)
    o = ops.conv_bn_act(cfg1, x0, token, True)
    z = ops.conv_bn_act(cfg2, o, token, True, res1=idt)
    z16 = ops.f16_twin(z)
    assert z16 is not None
    gz = hu('blk.gz%d%d' % (Cc, down), tuple(nchw(z).shape)) - 0.5
    (z.float() * gz.permute(0, 2, 3, 1).contiguous().to(H16).to(DEV).float()).sum().backward()
    bank.flush_bn_counters()

    # ---- reference: fp32 autograd; forward storage points fp16 (straight-through), none in the backward beyond bf16 of dz
    def r16(t):
        return t + (f16(t.detach()) - t.detach())
    xr = f16(x).requires_grad_(True)
    ws = [w.detach().cpu() for w in ((w0, w1, w2, wd) if down else (w0, w1, w2))]
    wr = [f16(w).requires_grad_(True) for w in ws]
    bnp = lambda b: (b.weight.detach().cpu(), b.bias.detach().cpu(), b.eps)
    norm = lambda t, b: F.batch_norm(t, None, None, bnp(b)[0], bnp(b)[1], True, 0.1, bnp(b)[2])
    x0r = r16(F.relu(norm(r16(F.conv2d(xr, wr[0], None, 1, 1)), bn0)))
    idr = x0r
    if down:
        idr = r16(norm(r16(F.conv2d(r16(F.avg_pool2d(x0r, 2, 2)), wr[3], None, 1, 0)), bnd))
    orr = r16(F.relu(norm(r16(F.conv2d(x0r, wr[1], None, 2 if down else 1, 1)), bn1)))
    zr = F.relu(norm(r16(F.conv2d(orr, wr[2], None, 1, 1)), bn2) + idr)
    (zr * bf(gz)).sum().backward()
    ck = Checker()
    ck.rel('z16 (fp16 twin)', nchw(z16), zr, 4e-3)
    ck.rel('z (bf16)', nchw(z), zr, 1.2e-2)
    e16, e = rel_err(nchw(z16), zr.detach()), rel_err(nchw(z), zr.detach())
    ck.l2('z16', nchw(z16), zr, 1e-3)
    ck.l2('dx', nchw(xg.grad), xr.grad, 3e-2)
    ck.l2('dw0', w0.grad, wr[0].grad, 3e-2)
    ck.l2('dw1', w1.grad, wr[1].grad, 3e-2)
    ck.l2('dw2', w2.grad, wr[2].grad, 3e-2)
    if down:
        ck.l2('dwd', wd.grad, wr[3].grad, 3e-2)
    ck.done()
    assert e16 < e, 'the fp16 twin must be closer to fp32 than the bf16 tensor (%.2e vs %.2e)' % (e16, e)


def test_encoder_routes_the_island_and_only_the_island():
    """The GCA encoder marks conv1 / conv2 / conv3 / layer1 / layer2 (incl. layer2's downsample conv) as island layers and nothing else;
    every island activation carries a twin, the encoder's outputs to GCA / layer3 / the shortcuts are plain bf16 tensors."""
    from tcvom_amd import ops
    from tcvom_amd.vmn import build_vmn_gca
    net = build_vmn_gca(agg_window=7).to(DEV)
    isl = sorted(s.name for s in net._bank.specs if getattr(s, 'f16', False))
    want = ['encoder.conv1', 'encoder.conv2', 'encoder.conv3'] + ['encoder.layer1.%d.conv%d' % (b, c) for b in range(3) for c in (1, 2)] + \
           ['encoder.layer2.%d.conv%d' % (b, c) for b in range(4) for c in (1, 2)] + ['encoder.layer2.0.downsample.1']
    assert isl == sorted(want)
    assert not any(hasattr(s, 'hp') or hasattr(s, 'y16') for s in net._bank.specs)        # (the round-5 doubled-tap scheme is gone)
