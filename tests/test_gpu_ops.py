"""GPU parity tests, op level: every HIP kernel family against the CPU oracle / plain fp32 PyTorch
math on identical (bf16-rounded) inputs, called through the C ABI (ctypes).

Tolerances: activations are stored in bf16 (8-bit mantissa) with fp32 accumulation, so a single op
is expected within ~2^-8 relative of an fp32 evaluation of the same bf16 inputs; tolerances are
stated per test.  Integer/mask outputs must be bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from tcvom_amd._lib import ACT_DTYPE as H16      # the 16-bit storage type of the loaded build (bf16 / fp16)
import torch.nn as nn
import torch.nn.functional as F

from helpers import hu, golden, tam_mask, gca_unknown, TAM_CASES, GCA_CASES, assert_close, Checker, tol
from tcvom_amd.synthetic import formula_tensor, synthetic_window

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def bf(t):
    return t.to(H16).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(H16).to(DEV)


def nchw(t):
    return t.detach().float().cpu().permute(0, 3, 1, 2)


def rel_err(got, want):
    got, want = got.double(), want.double()
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


# --------------------------------------------------------------------------------------------- conv engine
CONV_CASES = [
    # name, cin, cout, k, stride, pad, transposed, N, H, W
    ('c3x3_s1_32_64', 32, 64, 3, 1, 1, False, 2, 12, 20),
    ('c3x3_s2_64_128', 64, 128, 3, 2, 1, False, 1, 16, 24),
    ('c3x3_s1_128_256', 128, 256, 3, 1, 1, False, 1, 10, 14),
    ('c1x1_256_128', 256, 128, 1, 1, 0, False, 2, 6, 10),
    ('c3x3_s2_p0_16_32', 16, 32, 3, 2, 0, False, 1, 18, 26),
    ('c3x3_s1_32_32_big', 32, 32, 3, 1, 1, False, 1, 40, 72),
    ('convT_64_64', 64, 64, 4, 2, 1, True, 1, 9, 13),
    ('convT_512_512', 512, 512, 4, 2, 1, True, 1, 4, 6),
    ('c3x3_s1_512_256', 512, 256, 3, 1, 1, False, 1, 6, 8),
]


def _mini_bank(cin, cout, k, stride, pad, transposed, spectral, bias=False, tag='t'):
    from tcvom_amd.weights import WeightBank, ConvSpec
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    w = nn.Parameter(formula_tensor('conv.%s.weight' % tag, shape).to(DEV))
    u = v = None
    if spectral:
        u = nn.Parameter(formula_tensor('conv.%s.weight_u' % tag, (shape[0],)).to(DEV), requires_grad=False)
        v = nn.Parameter(formula_tensor('conv.%s.weight_v' % tag, (int(np.prod(shape[1:])),)).to(DEV), requires_grad=False)
    b = nn.Parameter(formula_tensor('conv.%s.bias' % tag, (cout,)).to(DEV)) if bias else None
    bank = WeightBank()
    spec = ConvSpec(tag, w, u, v, b, transposed, stride, pad, 'frame')
    bank.register(spec)
    return bank, spec


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(case):
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    name, cin, cout, k, stride, pad, transposed, N, H, W = case
    bank, spec = _mini_bank(cin, cout, k, stride, pad, transposed, spectral=False, bias=True, tag=name)
    cfg = ops.ConvCfg(bank, spec)
    x = hu('x.' + name, (N, cin, H, W))
    xg = nhwc(x).requires_grad_(True)
    token = bank_token(bank, 1, True)
    y = ops.conv_bn_act(cfg, xg, token, True)
    # reference on the same bf16-rounded operands
    xr = bf(x).requires_grad_(True)
    wr = bf(spec.weight.detach().cpu()).requires_grad_(True)
    br = spec.bias.detach().cpu().clone().requires_grad_(True)
    if transposed:
        yr = F.conv_transpose2d(xr, wr, br, stride, pad)
    else:
        yr = F.conv2d(xr, wr, br, stride, pad)
    assert tuple(nchw(y).shape) == tuple(yr.shape)
    assert rel_err(nchw(y), yr) < 1.5e-2, 'fwd'                 # bf16 output rounding: 2^-8 = 3.9e-3 of |y|
    gy = hu('gy.' + name, tuple(yr.shape))
    (y.float() * nhwc(gy).float()).sum().backward()
    (yr * bf(gy)).sum().backward()
    assert rel_err(nchw(xg.grad), xr.grad) < 1.5e-2, 'dgrad'
    assert rel_err(spec.weight.grad.cpu(), wr.grad) < 1e-2, 'wgrad'
    assert rel_err(spec.bias.grad.cpu(), br.grad) < 1e-2, 'bias grad'


def test_residual_gradient_goes_through_the_producers_batchnorm_backward():
    """A BasicBlock-shaped graph: z0 = act(BN(conv0(x))) + skip;  o = act(BN(conv1(z0)));  z1 = act(BN(conv2(o)) + z0).  z0 feeds conv1
    AND the residual add: the second op deposits d(res1) on z0's list instead of returning it, and the op that produced z0 reads it as
    the second addend `dz2` of its BatchNorm backward (ops.conv_bn_act).  Every gradient -- x, skip (which must see the WHOLE gradient
    of z0), the three weights -- against fp32 PyTorch autograd on the same bf16-rounded operands."""
    from tcvom_amd import ops
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    Cc, N, H, W = 64, 2, 16, 24
    bank = WeightBank()
    ws, bns, cfgs = [], [], []
    for i in range(3):
        w = nn.Parameter((formula_tensor('dep.w%d' % i, (Cc, Cc, 3, 3))).to(DEV))
        bn = nn.BatchNorm2d(Cc).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(hu('dep.g%d' % i, (Cc,)) * 0.3 + 1.0)
            bn.bias.copy_(hu('dep.b%d' % i, (Cc,)) * 0.3)
        spec = ConvSpec('dep%d' % i, w, None, None, None, False, 1, 1, 'frame')
        bank.register(spec)
        ws.append(w); bns.append(bn); cfgs.append(ops.ConvCfg(bank, spec, bn=bn, act=ops.ACT_RELU))
    x, skip, gz = hu('dep.x', (N, Cc, H, W)), hu('dep.s', (N, Cc, H, W)), hu('dep.gz', (N, Cc, H, W))
    xg, sg = nhwc(x).requires_grad_(True), nhwc(skip).requires_grad_(True)
    token = bank_token(bank, 1, True)
    z0 = ops.conv_bn_act(cfgs[0], xg, token, True, res2=sg)
    assert getattr(z0, '_tcvom_grad_stash', None) is not None
    o = ops.conv_bn_act(cfgs[1], z0, token, True)
    z1 = ops.conv_bn_act(cfgs[2], o, token, True, res1=z0)
    (z1.float() * nhwc(gz).float()).sum().backward()
    assert len(z0._tcvom_grad_stash) == 0                          # deposited by op 2, consumed by op 0

    def rb(t):                                                     # bf16 storage points of the HIP path, straight-through gradient
        return t + (bf(t.detach()) - t.detach())
    xr, sr = bf(x).requires_grad_(True), bf(skip).requires_grad_(True)
    wr = [bf(w.detach().cpu()).requires_grad_(True) for w in ws]
    bnp = [(b.weight.detach().cpu(), b.bias.detach().cpu(), b.eps) for b in bns]
    conv = lambda t, i: rb(F.conv2d(t, wr[i], None, 1, 1))
    norm = lambda t, i: F.batch_norm(t, None, None, bnp[i][0], bnp[i][1], True, 0.1, bnp[i][2])
    z0r = rb(F.relu(norm(conv(xr, 0), 0)) + sr)
    orf = rb(F.relu(norm(conv(z0r, 1), 1)))
    z1r = F.relu(norm(conv(orf, 2), 2) + z0r)
    (z1r * bf(gz)).sum().backward()
    l2 = lambda got, want: float((got.double() - want.double()).norm() / (want.double().norm() + 1e-12))
    assert l2(nchw(z1), z1r.detach()) < 1e-2, 'forward'
    # (bf16 activations flip a few ReLU masks against the fp32 graph: percent-level L2 differences; a lost deposit would be ~50 %)
    assert l2(nchw(sg.grad), sr.grad) < 6e-2, 'skip gradient (autograd part + deposited part)'
    assert l2(nchw(xg.grad), xr.grad) < 6e-2, 'input gradient'
    for i in range(3):
        assert l2(ws[i].grad.cpu(), wr[i].grad) < 6e-2, 'weight gradient %d' % i


def test_conv_small_cin_padded_channels():
    """First-layer convs read the 8-channel packed input with 6 (or 3) real channels."""
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    for cin, stride, pad in ((6, 2, 1), (6, 1, 1), (3, 2, 0)):
        bank, spec = _mini_bank(cin, 32 if cin == 6 else 16, 3, stride, pad, False, spectral=False, tag='small%d%d' % (cin, stride))
        spec.needs_dgrad = False
        cfg = ops.ConvCfg(bank, spec)
        x = hu('x.small', (1, 8, 20, 28))
        token = bank_token(bank, 1, True)
        y = ops.conv_bn_act(cfg, nhwc(x), token, True)
        wr = bf(spec.weight.detach().cpu()).requires_grad_(True)
        yr = F.conv2d(bf(x)[:, :cin], wr, None, stride, pad)
        assert rel_err(nchw(y), yr) < 1.5e-2
        gy = hu('gy.small', tuple(yr.shape))
        (y.float() * nhwc(gy).float()).sum().backward()
        (yr * bf(gy)).sum().backward()
        assert rel_err(spec.weight.grad.cpu(), wr.grad) < 1e-2


@pytest.mark.parametrize('act,pre_relu,res', [(1, False, 'res1'), (2, False, 'both'), (0, True, None), (0, False, None)])
def test_conv_bn_act_train(act, pre_relu, res):
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    cin, cout, N, H, W = 32, 64, 2, 14, 18
    bank, spec = _mini_bank(cin, cout, 3, 1, 1, False, spectral=False, tag='bn%d%d' % (act, pre_relu))
    bn = nn.BatchNorm2d(cout).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(formula_tensor('bn.weight', (cout,)))
        bn.bias.copy_(formula_tensor('bn.bias', (cout,)))
    cfg = ops.ConvCfg(bank, spec, bn=bn, act=act, pre_relu=pre_relu)
    x = hu('x.bn', (N, cin, H, W))
    r1 = hu('r1.bn', (N, cout, H, W)) if res in ('res1', 'both') else None
    r2 = hu('r2.bn', (N, cout, H, W)) if res == 'both' else None
    xg = nhwc(x).requires_grad_(True)
    r1g = nhwc(r1).requires_grad_(True) if r1 is not None else None
    r2g = nhwc(r2).requires_grad_(True) if r2 is not None else None
    token = bank_token(bank, 1, True)
    z = ops.conv_bn_act(cfg, xg, token, True, res1=r1g, res2=r2g)
    bank.flush_bn_counters()            # running statistics are applied after the window (deferred, in call order)

    xr = bf(x).requires_grad_(True)
    wr = bf(spec.weight.detach().cpu()).requires_grad_(True)
    gam = bn.weight.detach().cpu().clone().requires_grad_(True)
    bet = bn.bias.detach().cpu().clone().requires_grad_(True)
    rm, rv = torch.zeros(cout), torch.ones(cout)
    yr = F.conv2d(xr, wr, None, 1, 1)
    if pre_relu:
        yr = F.relu(yr)
    # The kernel takes the statistics from the fp32 accumulators but normalises the bf16-stored conv
    # output; mirror that (straight-through rounding) so that activation masks are decided on identical
    # values — otherwise a handful of sign flips at pre-activations ~0 dominate a max-error metric.
    mean = yr.mean((0, 2, 3), keepdim=True)
    var = yr.var((0, 2, 3), unbiased=False, keepdim=True)
    n_el = yr.numel() // cout
    rm = 0.9 * rm + 0.1 * mean.detach().flatten()
    rv = 0.9 * rv + 0.1 * var.detach().flatten() * n_el / (n_el - 1)
    yq = yr + (bf(yr) - yr).detach()
    zr = (yq - mean) / torch.sqrt(var + 1e-5) * gam.view(1, -1, 1, 1) + bet.view(1, -1, 1, 1)
    r1r = bf(r1).requires_grad_(True) if r1 is not None else None
    r2r = bf(r2).requires_grad_(True) if r2 is not None else None
    if r1r is not None:
        zr = zr + r1r
    zr = F.relu(zr) if act == 1 else (F.leaky_relu(zr, 0.2) if act == 2 else zr)
    if r2r is not None:
        zr = zr + r2r
    ck = Checker()
    ck.rel('fwd', nchw(z), zr, 2e-2)
    ck.rel('running_mean', bn.running_mean, rm, 2e-2)
    ck.rel('running_var', bn.running_var, rv, 2e-2)
    gz = hu('gz.bn', tuple(zr.shape))
    (z.float() * nhwc(gz).float()).sum().backward()
    (zr * bf(gz)).sum().backward()
    ck.rel('dx', nchw(xg.grad), xr.grad, 4e-2)
    ck.rel('dgamma', bn.weight.grad, gam.grad, 2e-2)
    ck.rel('dbeta', bn.bias.grad, bet.grad, 2e-2)
    ck.rel('dw', spec.weight.grad, wr.grad, 3e-2)
    if r1r is not None:
        ck.rel('dres1', nchw(r1g.grad), r1r.grad, 2e-2)
    if r2r is not None:
        ck.rel('dres2', nchw(r2g.grad), r2r.grad, 2e-2)
    ck.done()


@pytest.mark.parametrize('cin,cout,k,stride,transposed,H,W', [
    (64, 128, 3, 1, False, 256, 264),      # >= 512 tiles of 128x128: the 8-wave 128x128 configuration WITH statistics
    (128, 128, 4, 2, True, 136, 120),      # 4 phases in one launch, 128x128 tiles, statistics per phase
    (128, 128, 3, 1, False, 136, 240),     # the os8 class at 1080p: 128x64 tiles
    (32, 32, 3, 1, False, 272, 480),       # the os2/os1 class: 32x256 tiles
    (256, 256, 3, 1, False, 68, 120),      # the os16 class: 64x64 tiles, 4-slot ring
    (256, 256, 3, 1, False, 136, 180),     # 24480 pixels (= 3 os16 frames): 128x96 tiles, 510 workgroups instead of 384 of 128x128
    (256, 256, 3, 1, False, 408, 240),     # 97920 pixels (= 3 os8 frames of FBA's 256-plane layers): 256x192 tiles, 510 workgroups instead of 383 of 256x256 (round 6)
    (512, 1024, 1, 1, False, 96, 130),     # 1 x 1 expand conv of the FBA trunk class: gemm_nt256 WITH the statistics epilogue (196 tiles, ragged tail)
    (64, 256, 1, 1, False, 200, 260),      # the os4 expand conv: one K-tile
])
def test_conv_bn_large_tile_configs(cin, cout, k, stride, transposed, H, W):
    """Tile configurations are chosen from the problem size, so the production (1080p) configurations need their
    own cases: forward + batch statistics vs fp32 PyTorch on the same bf16 operands, and gradient sanity."""
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    tag = 'big%d_%d_%d_%d' % (cin, cout, k, H)
    pad = 0 if k == 1 else 1
    bank, spec = _mini_bank(cin, cout, k, stride, pad, transposed, spectral=False, tag=tag)
    bn = nn.BatchNorm2d(cout).to(DEV)
    cfg = ops.ConvCfg(bank, spec, bn=bn, act=1)
    x = hu('x.' + tag, (1, cin, H, W))
    xg = nhwc(x).requires_grad_(True)
    token = bank_token(bank, 1, True)
    if (cin, H) == (256, 408):
        import ctypes as C
        from tcvom_amd import _lib as L
        assert L._FNS['tcvom_conv_igemm_variant'](C.byref(cfg.geometry(1, H, W).fwd[0]), 1).decode() == 'igemm_nt<256,192,64,96,2>'
    z = ops.conv_bn_act(cfg, xg, token, True)
    bank.flush_bn_counters()
    wr = bf(spec.weight.detach().cpu())
    yr = F.conv_transpose2d(bf(x), wr, None, stride, pad) if transposed else F.conv2d(bf(x), wr, None, stride, pad)
    mean, var = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    zr = F.relu((bf(yr) - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5))
    ck = Checker()
    ck.rel('z', nchw(z), zr, 2e-2)
    ck.rel('running_mean', bn.running_mean, 0.1 * mean, 2e-2)
    n_el = yr.numel() // cout
    ck.rel('running_var', bn.running_var, 0.9 + 0.1 * var * n_el / (n_el - 1), 2e-2)
    z.float().sum().backward()
    assert torch.isfinite(xg.grad.float()).all() and torch.isfinite(spec.weight.grad).all()
    ck.done()


@pytest.mark.parametrize('cin,cout,k,stride,transposed,H,W,f16', [(64, 128, 3, 2, False, 24, 40, False), (128, 64, 4, 2, True, 12, 20, False),
                                                                  (32, 32, 3, 1, False, 16, 64, True), (64, 64, 3, 1, False, 20, 44, True), (64, 128, 3, 2, False, 24, 40, True), (256, 256, 1, 1, False, 10, 12, False),
                                                                  (128, 128, 3, 1, False, 20, 36, False), (64, 64, 3, 1, False, 36, 40, False),
                                                                  # csrc/sconv.hip: four phases x three frames with statistics; two channel blocks
                                                                  (64, 64, 4, 2, True, 16, 40, False), (32, 32, 4, 2, True, 24, 64, False),
                                                                  (32, 64, 3, 1, False, 16, 48, False), (64, 32, 3, 1, False, 24, 33, False),
                                                                  # the 3-frame launch runs on gemm_nt256 with statistics (396 tiles), the
                                                                  # frame-by-frame ones (132 tiles each) on igemm_nt: same groups, same sums
                                                                  (512, 1024, 1, 1, False, 48, 86, False)])
def test_frame_batched_conv_bn_equals_frame_by_frame(cin, cout, k, stride, transposed, H, W, f16):
    """Three frames through a SpectralNorm'd conv + BatchNorm + ReLU as ONE frame-batched op (bank.frames_per_op = 3: per-frame
    weight slot, per-frame batch statistics, batched data gradient, deferred batched weight gradient) must equal three
    frame-by-frame ops on the same bank state: outputs and input gradients bit for bit (same kernels, same tiles may differ
    -> compared to bf16 precision), parameter gradients and running statistics to fp32 summation order."""
    from tcvom_amd import ops
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    S, B = 3, 2
    pad = 0 if k == 1 else 1

    def run(batched):
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        tag = 'fb%d_%d_%d' % (cin, cout, k)
        w = nn.Parameter(formula_tensor('conv.%s.weight' % tag, shape).to(DEV))
        u = nn.Parameter(formula_tensor('conv.%s.weight_u' % tag, (shape[0],)).to(DEV), requires_grad=False)
        v = nn.Parameter(formula_tensor('conv.%s.weight_v' % tag, (int(np.prod(shape[1:])),)).to(DEV), requires_grad=False)
        bank = WeightBank()
        spec = ConvSpec(tag, w, u, v, None, transposed, stride, pad, 'frame')
        spec.f16 = bool(f16) and H16 == torch.bfloat16     # a layer of the fp16 island of the bf16 build (forward in IEEE fp16)
        bank.register(spec)
        bn = nn.BatchNorm2d(cout).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(formula_tensor('bn.weight', (cout,)))
            bn.bias.copy_(formula_tensor('bn.bias', (cout,)))
        cfg = ops.ConvCfg(bank, spec, bn=bn, act=1)
        xs = [nhwc(hu('x%d.%s' % (f, tag), (B, cin, H, W)) * (1.0 + 0.5 * f)) for f in range(S)]
        token = bank_token(bank, S, True)
        if batched:
            xg = torch.cat(xs, 0).requires_grad_(True)
            bank.frames_per_op = S
            z = ops.conv_bn_act(cfg, xg, token, True)
            bank.frames_per_op = 1
            zs = [z[f * B:(f + 1) * B] for f in range(S)]
        else:
            xl = [t.clone().requires_grad_(True) for t in xs]
            zs = [ops.conv_bn_act(cfg, t, token, True) for t in xl]
        bank.flush_bn_counters()
        gz = [nhwc(hu('gz%d.%s' % (f, tag), (B, cout, zs[0].shape[1], zs[0].shape[2]))) for f in range(S)]
        sum((a.float() * g.float()).sum() for a, g in zip(zs, gz)).backward()
        torch.cuda.synchronize()
        dx = [xg.grad[f * B:(f + 1) * B] for f in range(S)] if batched else [t.grad for t in xl]
        return ([t.detach().float().cpu() for t in zs], [t.float().cpu() for t in dx], w.grad.cpu(), bn.weight.grad.cpu(),
                bn.bias.grad.cpu(), bn.running_mean.cpu(), bn.running_var.cpu(), u.detach().cpu().clone(), int(bn.num_batches_tracked))

    a, b = run(True), run(False)
    ck = Checker()
    for f in range(S):
        ck.rel('z[%d]' % f, a[0][f], b[0][f], 1e-2)
        ck.rel('dx[%d]' % f, a[1][f], b[1][f], 5e-2)
    # the power iteration reduces with fp32 atomics: sigma can differ by an ulp between the two runs, which flips the
    # bf16 rounding of a few packed weights and, through them, a few ReLU masks -- 1-2 % on the summed gradients
    ck.rel('dw', a[2], b[2], 3e-2)
    ck.rel('dgamma', a[3], b[3], 3e-2)
    ck.rel('dbeta', a[4], b[4], 3e-2)
    ck.rel('running_mean', a[5], b[5], 2e-4)          # different tile -> different fp32 order of the partial sums
    ck.rel('running_var', a[6], b[6], 2e-4)
    ck.rel('u', a[7], b[7], 1e-6)
    ck.done()
    assert a[8] == b[8] == S
    assert not torch.allclose(a[0][0], a[0][1])


def test_row_range_gradient_is_added_inside_the_batchnorm_backward():
    """A frame-batched conv + BatchNorm op whose output is read by one consumer for all three frames and by another for the
    centre frame only (ops.frame_slice: the decoder tail's shortcut inputs, VMN_model.py:107-110): the centre-frame gradient is
    deposited with the producer and added inside tcvom_bn_bwd_reduce_ranged / tcvom_bn_bwd_apply_ranged.  Must equal the plain
    slice, whose gradient autograd pads with zeros and adds."""
    from tcvom_amd import ops
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    S, B, cin, cout, H, W = 3, 2, 64, 64, 20, 28

    def run(deposit):
        w = nn.Parameter(formula_tensor('conv.rr.weight', (cout, cin, 3, 3)).to(DEV))
        bank = WeightBank()
        spec = ConvSpec('rr', w, None, None, None, False, 1, 1, 'frame')
        bank.register(spec)
        bn = nn.BatchNorm2d(cout).to(DEV)
        cfg = ops.ConvCfg(bank, spec, bn=bn, act=1)
        xg = torch.cat([nhwc(hu('x%d.rr' % f, (B, cin, H, W))) for f in range(S)], 0).requires_grad_(True)
        token = bank_token(bank, S, True)
        bank.frames_per_op = S
        z = ops.conv_bn_act(cfg, xg, token, True)
        bank.frames_per_op = 1
        assert getattr(z, '_tcvom_grad_stash', None) is not None
        zc = ops.frame_slice(z, B, 2 * B) if deposit else z[B:2 * B]
        assert (zc.grad_fn.__class__.__name__ == '_FrameSliceBackward') == deposit
        g_all = nhwc(hu('g_all.rr', (S * B, cout, H, W)))
        g_c = nhwc(hu('g_c.rr', (B, cout, H, W)))
        ((z.float() * g_all.float()).sum() + 3.0 * (zc.float() * g_c.float()).sum()).backward()
        bank.flush_bn_counters()
        torch.cuda.synchronize()
        return xg.grad.float().cpu(), w.grad.cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu()

    a, b = run(True), run(False)
    ck = Checker()
    for f in range(S):
        ck.rel('dx[%d]' % f, a[0][f * B:(f + 1) * B], b[0][f * B:(f + 1) * B], 1e-2)
    ck.rel('dw', a[1], b[1], 1e-2)
    ck.rel('dgamma', a[2], b[2], 1e-2)
    ck.rel('dbeta', a[3], b[3], 1e-2)
    ck.done()


@pytest.mark.parametrize('residual', [False, True, 'res2'])
def test_three_gradient_addends_inside_the_batchnorm_backward(residual, monkeypatch):
    """The output of a frame-batched conv + BatchNorm op with THREE consumers -- autograd (all frames), a second conv + BatchNorm op (its
    data gradient is deposited with the producer) and ops.frame_slice of the centre frame (a row-range deposit): the encoder stage
    outputs of the window (next stage's conv1, its down-sampling branch, the shortcut branch).  tcvom_bn_bwd_reduce3 / _apply3 add all
    three in fp32; must equal the two-addend path, where the third gradient is zero-padded and added by element-wise passes.
    residual: the producer is a residual site (activation mask instead of res1 in its backward).  'res2': a residual added AFTER the
    activation (the decoder blocks) and two consumers -- its gradient, dz + dz2 itself, is written by tcvom_bn_bwd_apply3 (dsum)."""
    from tcvom_amd import ops
    from tcvom_amd import _lib as L
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    S, B, cin, cout, H, W = 3, 1, 64, 64, 24, 40
    calls = []
    real = L.call
    monkeypatch.setattr(L, 'call', lambda name, *a: (calls.append(name), real(name, *a))[1])

    def run(three):
        monkeypatch.setattr(ops, 'THREE_ADDENDS', three)
        del calls[:]
        bank = WeightBank()
        specs, cfgs = [], []
        for i in range(2):
            w = nn.Parameter(formula_tensor('conv.t3%d.weight' % i, (cout, cin, 3, 3)).to(DEV))
            spec = ConvSpec('t3%d' % i, w, None, None, None, False, 1, 1, 'frame')
            bank.register(spec)
            specs.append(spec)
            cfgs.append(ops.ConvCfg(bank, spec, bn=nn.BatchNorm2d(cout).to(DEV), act=1))
        xg = torch.cat([nhwc(hu('x%d.t3' % f, (B, cin, H, W))) for f in range(S)], 0).requires_grad_(True)
        res = torch.cat([nhwc(hu('r%d.t3' % f, (B, cout, H, W))) for f in range(S)], 0).requires_grad_(True) if residual else None
        token = bank_token(bank, S, True)
        bank.frames_per_op = S
        z = ops.conv_bn_act(cfgs[0], xg, token, True, res2=res) if residual == 'res2' else ops.conv_bn_act(cfgs[0], xg, token, True, res1=res)
        z2 = ops.conv_bn_act(cfgs[1], z, token, True)                  # consumer 2: deposits its data gradient with z's producer
        bank.frames_per_op = 1
        zc = z[B:2 * B] * 0.0 if residual == 'res2' else ops.frame_slice(z, B, 2 * B)      # consumer 3: the centre frame only
        g1, g2 = nhwc(hu('g1.t3', (S * B, cout, H, W))), nhwc(hu('g2.t3', (S * B, cout, H, W)))
        g3 = nhwc(hu('g3.t3', (B, cout, H, W)))
        ((z.float() * g1.float()).sum() + 0.5 * (z2.float() * g2.float()).sum() + 2.0 * (zc.float() * g3.float()).sum()).backward()
        bank.flush_bn_counters()
        torch.cuda.synchronize()
        return (xg.grad.float().cpu(), specs[0].weight.grad.cpu(), cfgs[0].bn.weight.grad.cpu(), cfgs[0].bn.bias.grad.cpu(),
                res.grad.float().cpu() if residual else None, list(calls))

    a, b = run(True), run(False)
    assert 'tcvom_bn_bwd_apply3' in a[5] and 'tcvom_bn_bwd_apply3' not in b[5]
    assert residual == 'res2' or ('tcvom_bn_bwd_reduce3' in a[5] and 'tcvom_bn_bwd_reduce3' not in b[5])
    ck = Checker()
    ck.rel('dx', a[0], b[0], 1e-2)
    ck.rel('dw', a[1], b[1], 1e-2)
    ck.rel('dgamma', a[2], b[2], 1e-2)
    ck.rel('dbeta', a[3], b[3], 1e-2)
    if residual:
        ck.rel('dres1', a[4], b[4], 1e-2)
    ck.done()


@pytest.mark.parametrize('cin,cout,k,stride,transposed,H,W', [(128, 128, 3, 1, False, 40, 64), (64, 64, 4, 2, True, 20, 24),
                                                               (32, 64, 3, 2, False, 48, 64), (32, 32, 3, 1, False, 48, 64)])
def test_batched_weight_gradient_launch(cin, cout, k, stride, transposed, H, W):
    """tcvom_wgrad_igemm_batched (the S calls of a layer as one launch, what WeightBank.run_deferred_wgrads issues) must
    equal S separate tcvom_wgrad_igemm_phases launches up to the fp32 order of the split pixel reduction."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    tag = 'bw%d_%d_%d' % (cin, cout, k)
    bank, spec = _mini_bank(cin, cout, k, stride, 1, transposed, spectral=False, tag=tag)
    geo = ConvGeometry(spec, 1, H, W)
    arr = _phase_array(geo.wgrad)
    S = 3
    xs = [(hu('x%d.%s' % (i, tag), (1, H, W, spec.cpad))).to(DEV).to(H16) for i in range(S)]
    dys = [(hu('dy%d.%s' % (i, tag), (1, geo.OH, geo.OW, cout))).to(DEV).to(H16) for i in range(S)]
    n = spec.K * spec.T * spec.cpad
    single = torch.zeros(S, n, device=DEV)
    batched = torch.zeros(S, n, device=DEV)
    st = L.stream_ptr()
    for i in range(S):
        L.call('tcvom_wgrad_igemm_phases', L.ptr(dys[i]), L.ptr(xs[i]), L.ptr(single[i]), arr, len(geo.wgrad), cout, st)
    vp = lambda ts: C.cast((C.c_void_p * S)(*[t.data_ptr() for t in ts]), C.c_void_p)
    L.call('tcvom_wgrad_igemm_batched', vp(dys), vp(xs), vp([batched[i] for i in range(S)]), S, arr, len(geo.wgrad), cout, st)
    torch.cuda.synchronize()
    assert float(single.abs().max()) > 0
    assert rel_err(batched.cpu(), single.cpu()) < 2e-5
    assert not torch.equal(single[0], single[1])


@pytest.mark.parametrize('cin,cout,N,H,W,S', [(64, 64, 2, 20, 40, 1), (64, 64, 1, 33, 70, 3), (128, 128, 1, 17, 30, 3),
                                              (128, 128, 2, 40, 64, 1), (256, 128, 1, 9, 16, 2), (128, 256, 1, 24, 20, 1),
                                              (512, 512, 1, 6, 10, 3), (64, 128, 1, 16, 32, 1), (128, 128, 1, 17, 30, 20),
                                              (64, 64, 1, 16, 32, 40), (256, 256, 1, 9, 12, 96),
                                              # 64-channel windows of a C that is not a multiple of 128; K = 32 (half a k group)
                                              (320, 64, 1, 20, 36, 1), (192, 128, 2, 9, 33, 2), (64, 32, 2, 24, 40, 3),
                                              (128, 32, 1, 17, 30, 1), (192, 32, 1, 16, 64, 9)])
def test_wgrad_ws_kernel(cin, cout, N, H, W, S):
    """Accumulator-stationary weight gradient (csrc/wgradws.hip; stride-1 3x3, C a multiple of 64, K = 32 or a multiple
    of 64): ragged tiles, several samples, several problems per launch, every (k, c) block split -- against
    torch.nn.grad.conv2d_weight in fp32 on the same bf16 operands."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    tag = 'wgws%d_%d_%d' % (cin, cout, H)
    assert S <= L.call('tcvom_wgrad_ws_max_problems')    # (96: the largest case fills a launch)
    bank, spec = _mini_bank(cin, cout, 3, 1, 1, False, spectral=False, tag=tag)
    geo = ConvGeometry(spec, N, H, W)
    assert L._FNS['tcvom_wgrad_igemm_variant'](C.byref(_phase_array(geo.wgrad)[0])).decode().startswith('wgrad_ws')
    xs = [hu('x%d.%s' % (i, tag), (N, H, W, cin)).to(DEV).to(H16) for i in range(S)]
    dys = [hu('dy%d.%s' % (i, tag), (N, H, W, cout)).to(DEV).to(H16) for i in range(S)]
    dw = torch.zeros(S, cout, 9, cin, device=DEV)
    vp = lambda ts: C.cast((C.c_void_p * S)(*[t.data_ptr() for t in ts]), C.c_void_p)
    for _ in range(2):                                   # accumulates: two launches = twice the gradient
        if S <= 8:
            L.call('tcvom_wgrad_igemm_batched', vp(dys), vp(xs), vp([dw[i] for i in range(S)]), S, _phase_array(geo.wgrad),
                   len(geo.wgrad), cout, L.stream_ptr())
        else:                                            # many problems of one geometry: WeightBank.run_deferred_wgrads
            L.call('tcvom_wgrad_ws_multi', vp(dys), vp(xs), vp([dw[i] for i in range(S)]), S, _phase_array(geo.wgrad),
                   cout, L.stream_ptr())
    torch.cuda.synchronize()
    for i in range(S):
        ref = torch.nn.grad.conv2d_weight(xs[i].float().cpu().permute(0, 3, 1, 2), (cout, cin, 3, 3),
                                          dys[i].float().cpu().permute(0, 3, 1, 2), padding=1)
        got = dw[i].cpu().view(cout, 3, 3, cin).permute(0, 3, 1, 2) / 2
        assert rel_err(got, ref) < 1e-5, 'problem %d' % i


HALO_WGRAD_CASES = [
    # cin, cout, stride, pad, N, H, W (input), problems, variant
    (32, 32, 1, 1, 2, 16, 64, 3, 'halo_wgrad<32>'),
    (6, 32, 1, 1, 1, 24, 96, 1, 'halo_wgrad<8,1,32>'),           # os1 shortcut (8 -> 32)
    (8, 32, 1, 1, 2, 16, 64, 3, 'halo_wgrad<8,1,32>'),
    (6, 32, 2, 1, 1, 48, 192, 3, 'halo_wgrad<8,2,32>'),          # encoder conv1
    (3, 16, 2, 0, 2, 34, 130, 2, 'halo_wgrad<8,2,16>'),          # guidance head conv1 on a reflection-padded input (taps 0..2)
    (3, 16, 2, 1, 1, 32, 64, 1, 'halo_wgrad<8,2,16>'),
    (16, 32, 2, 0, 1, 50, 130, 3, 'halo_wgrad<16,2,32>'),        # guidance head conv2 on a reflection-padded input
    (16, 32, 2, 1, 2, 16, 64, 1, 'halo_wgrad<16,2,32>'),
    (6, 32, 1, 1, 1, 136, 320, 1, 'halo_wgrad<8,1,32>'),         # 170 tiles: several per workgroup, both tile buffers in use
    (6, 32, 2, 1, 3, 272, 640, 2, 'halo_wgrad<8,2,32>'),         # 1020 tiles over 3 samples and 2 problems
]


@pytest.mark.parametrize('cin,cout,stride,pad,N,H,W,S,variant', HALO_WGRAD_CASES)
def test_halo_wgrad_kernel(cin, cout, stride, pad, N, H, W, S, variant):
    """Halo-form weight gradient (csrc/halo.hip: halo_wgrad_kernel<C, S, K>; 3x3, <= 32 channels in and out, stride 1 or 2, the dy grid
    a multiple of 8 x 32): every instantiated shape, several samples and problems per launch, inputs with their own padding ring --
    against torch.nn.grad.conv2d_weight in fp32 on the same 16-bit operands."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    tag = 'hwg%d_%d_%d_%d_%d' % (cin, cout, stride, pad, H)
    bank, spec = _mini_bank(cin, cout, 3, stride, pad, False, spectral=False, tag=tag)
    geo = ConvGeometry(spec, N, H, W)
    arr = _phase_array(geo.wgrad)
    assert L._FNS['tcvom_wgrad_igemm_variant'](C.byref(arr[0])).decode() == variant
    cp = spec.cpad
    xs = [hu('x%d.%s' % (i, tag), (N, H, W, cp)).to(DEV).to(H16) for i in range(S)]
    for x in xs:
        x[..., cin:] = 0
    dys = [hu('dy%d.%s' % (i, tag), (N, geo.OH, geo.OW, cout)).to(DEV).to(H16) for i in range(S)]
    dw = torch.zeros(S, cout, 9, cp, device=DEV)
    vp = lambda ts: C.cast((C.c_void_p * S)(*[t.data_ptr() for t in ts]), C.c_void_p)
    for _ in range(2):                                   # accumulates: two launches = twice the gradient
        L.call('tcvom_wgrad_igemm_batched', vp(dys), vp(xs), vp([dw[i] for i in range(S)]), S, arr, len(geo.wgrad), cout, L.stream_ptr())
    torch.cuda.synchronize()
    for i in range(S):
        ref = torch.nn.grad.conv2d_weight(xs[i][..., :cin].double().cpu().permute(0, 3, 1, 2), (cout, cin, 3, 3),
                                          dys[i].double().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad)
        got = dw[i].cpu().view(cout, 3, 3, cp).permute(0, 3, 1, 2) / 2
        assert rel_err(got[:, :cin], ref) < 1e-5, 'problem %d' % i
        assert float(got[:, cin:].abs().max()) == 0 if cp > cin else True


@pytest.mark.parametrize('cin,cout,N,H,W', [(32, 32, 2, 16, 64), (6, 32, 1, 16, 160), (32, 32, 1, 72, 96), (64, 64, 2, 16, 64),
                                           (64, 32, 1, 16, 96), (32, 64, 1, 40, 64)])
def test_halo_conv_kernel(cin, cout, N, H, W):
    _halo_conv_case(cin, cout, N, H, W, 1)


@pytest.mark.parametrize('cin,cout,N,H,W', [(6, 32, 2, 32, 128), (3, 16, 1, 16, 64)])
def test_halo_conv_kernel_stride2(cin, cout, N, H, W):
    """The stride-2 3x3 convs on the 8-channel full-resolution inputs (encoder conv1, guidance head; out % (8, 32) == 0) take the
    halo kernel too: forward with fused ReLU + batch statistics and the weight gradient against fp32 PyTorch."""
    _halo_conv_case(cin, cout, N, H, W, 2)


def test_halo_conv_kernel_stride2_on_a_reflection_padded_input():
    """ReflectionPad2d(1) + Conv2d(3 -> 16, stride 2, padding 0) of the guidance head (res_gca_enc.py:20-28): taps 0..2 on an
    input that carries its own ring."""
    _halo_conv_case(3, 16, 2, 34, 130, 2, pad=0)


def _halo_conv_case(cin, cout, N, H, W, stride, pad=1):
    """Shapes served by the halo-tile direct conv (stride-1 3x3, <= 32 channels, H % 8 == 0, W % 32 == 0): forward
    with fused ReLU + batch statistics (64 output channels = two workgroups per tile), data gradient (also through the halo kernel) and weight gradient, against
    fp32 PyTorch on the same 16-bit operands.  (The IEEE fp16 operand mode of the kernel -- the fp16 island of the bf16 build -- is covered by
    tests/test_gpu_f16_island.py.)"""
    from tcvom_amd import ops
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    tag = 'halo%d_%d_%d_%d_s%d_p%d' % (cin, cout, 0, H, stride, pad)
    w = nn.Parameter((formula_tensor('conv.%s.weight' % tag, (cout, cin, 3, 3)) * 0.2).to(DEV))
    bank = WeightBank()
    spec = ConvSpec(tag, w, None, None, None, False, stride, pad, 'frame', needs_dgrad=cin >= 16)
    bank.register(spec)
    bn = nn.BatchNorm2d(cout).to(DEV)
    cfg = ops.ConvCfg(bank, spec, bn=bn, act=0, pre_relu=True)
    x = hu('x.' + tag, (N, cin, H, W)) - 0.5
    xp = F.pad(x, (0, 0, 0, 0, 0, spec.cpad - cin))
    xg = nhwc(xp).requires_grad_(True)
    token = bank_token(bank, 1, True)
    if stride == 2:
        import ctypes as C
        from tcvom_amd import _lib as L
        from tcvom_amd.conv_plan import ConvGeometry
        geo = ConvGeometry(spec, N, H, W)
        assert L._FNS['tcvom_conv_igemm_variant'](C.byref(geo.fwd[0]), len(geo.fwd)).decode() == 'halo_conv<8>'
    z = ops.conv_bn_act(cfg, xg, token, True)
    bank.flush_bn_counters()
    xr = bf(x).requires_grad_(True)
    wf = spec.weight.detach().cpu()
    wr = bf(wf).clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, wr, None, stride, pad))
    mean, var = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    yq = yr + (bf(yr) - yr).detach()
    zr = (yq - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
    ck = Checker()
    ck.rel('z', nchw(z), zr, 2e-2)
    ck.rel('running_mean', bn.running_mean, 0.1 * mean.detach(), 1e-2)
    n_el = yr.numel() // cout
    ck.rel('running_var', bn.running_var, 0.9 + 0.1 * var.detach() * n_el / (n_el - 1), 1e-2)
    gz = hu('gz.' + tag, tuple(zr.shape)) - 0.5
    (z.float() * nhwc(gz).float()).sum().backward()
    (zr * bf(gz)).sum().backward()
    if cin >= 16:
        ck.rel('dx', nchw(xg.grad)[:, :cin], xr.grad, 4e-2)
    ck.rel('dw', spec.weight.grad, wr.grad, 3e-2)
    ck.done()


PWCONV_CASES = [
    # cin, cout, N, H, W, bias      (pixels per frame >= 1024; sizes that are not multiples of the 32 .. 256-pixel tiles mask the tail)
    (64, 128, 1, 40, 72, False), (128, 64, 2, 24, 33, True), (128, 128, 1, 37, 53, False), (32, 64, 1, 64, 96, True),
    (64, 32, 1, 50, 70, False), (256, 128, 1, 34, 60, False), (128, 256, 2, 20, 31, True), (256, 512, 1, 34, 60, False),
    (512, 256, 1, 34, 60, False), (64, 256, 1, 48, 80, False), (256, 1024, 1, 36, 44, True), (512, 2048, 1, 32, 40, False),
    (32, 32, 1, 40, 40, False), (256, 64, 1, 33, 47, False),
]


@pytest.mark.parametrize('cin,cout,N,H,W,bias', PWCONV_CASES)
def test_pwconv_kernel(cin, cout, N, H, W, bias):
    """Shapes served by the weight-stationary streaming 1x1 conv (csrc/pwconv.hip: one tap, stride 1, C in {32 .. 512}): forward with
    fused ReLU + batch statistics (+ bias), data gradient (the transposed 1x1 shape, through the same kernel where it qualifies) and
    weight gradient, against fp32 PyTorch on the same 16-bit operands; then the raw kernel, tight, with its statistics groups."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd import ops
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    tag = 'pw%d_%d_%d_%d_%d' % (cin, cout, N, H, W)
    w = nn.Parameter((formula_tensor('conv.%s.weight' % tag, (cout, cin, 1, 1)) * 0.3).to(DEV))
    b = nn.Parameter(formula_tensor('conv.%s.bias' % tag, (cout,)).to(DEV)) if bias else None
    bank = WeightBank()
    spec = ConvSpec(tag, w, None, None, b, False, 1, 0, 'frame')
    bank.register(spec)
    geo = ConvGeometry(spec, N, H, W)
    assert L._FNS['tcvom_conv_igemm_variant'](C.byref(geo.fwd[0]), 1).decode().startswith('pwconv<%d,' % cin)
    bn = nn.BatchNorm2d(cout).to(DEV)
    cfg = ops.ConvCfg(bank, spec, bn=bn, act=0, pre_relu=True)
    x = hu('x.' + tag, (N, cin, H, W)) - 0.5
    xg = nhwc(x).requires_grad_(True)
    token = bank_token(bank, 1, True)
    z = ops.conv_bn_act(cfg, xg, token, True)
    bank.flush_bn_counters()
    xr = bf(x).requires_grad_(True)
    wr = bf(spec.weight.detach().cpu()).clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, wr, b.detach().cpu() if bias else None, 1, 0))
    mean, var = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    yq = yr + (bf(yr) - yr).detach()
    zr = (yq - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
    ck = Checker()
    ck.rel('z', nchw(z), zr, 2e-2)
    ck.rel('running_mean', bn.running_mean, 0.1 * mean.detach(), 1e-2)
    n_el = yr.numel() // cout
    ck.rel('running_var', bn.running_var, 0.9 + 0.1 * var.detach() * n_el / (n_el - 1), 1e-2)
    gz = hu('gz.' + tag, tuple(zr.shape)) - 0.5
    (z.float() * nhwc(gz).float()).sum().backward()
    (zr * bf(gz)).sum().backward()
    ck.rel('dx', nchw(xg.grad), xr.grad, 4e-2)
    ck.rel('dw', spec.weight.grad, wr.grad, 3e-2)
    ck.done()
    # raw kernel: tight against fp32 on the same 16-bit operands (only the output rounding and the summation order differ), the
    # statistics groups against the unrounded results
    st = L.stream_ptr()
    yref = F.conv2d(bf(x), bf(spec.weight.detach().cpu()), b.detach().cpu() if bias else None, 1, 0)
    groups = ops._stats_groups(geo.fwd, 1)
    y16 = torch.empty(N, H, W, cout, device=DEV, dtype=H16)
    stats = torch.full((groups * 2 * cout,), float('nan'), device=DEV, dtype=torch.float32)
    ops._launch_conv(geo.fwd, xg.detach(), bank.fwd_ptr(spec, 0), y16, b, stats, 0, st)
    assert rel_err(nchw(y16), yref) < 6e-3
    sums = stats.view(groups, 2, cout).double().sum(0).cpu()
    assert torch.isfinite(sums).all()
    assert rel_err(sums[0], yref.double().sum((0, 2, 3))) < 1e-4 and rel_err(sums[1], (yref.double() ** 2).sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize('cin,cout,dil,N,H,W,S', [(256, 256, 2, 1, 16, 24, 3), (512, 512, 4, 1, 16, 32, 1), (64, 64, 2, 2, 10, 18, 2),
                                                  (128, 32, 4, 1, 32, 64, 1), (128, 64, 2, 2, 34, 44, 1), (64, 128, 3, 1, 27, 30, 4)])
def test_wgrad_ws_kernel_dilated(cin, cout, dil, N, H, W, S):
    """Dilated 3x3 layers (ResnetDilated of the FBA base, models/FBA/models.py:203-217: dilation 2 and 4, padding = dilation) through
    the accumulator-stationary weight gradient: the N * dil^2 sub-grids (pixels of one residue class mod dil) are dilation-1
    problems on strided pixels.  Against torch.nn.grad.conv2d_weight(dilation=dil) in fp32 on the same 16-bit operands."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    from tcvom_amd.weights import ConvSpec, WeightBank
    tag = 'wgwsd%d_%d_%d_%d' % (cin, cout, dil, H)
    w = nn.Parameter(formula_tensor('conv.%s.weight' % tag, (cout, cin, 3, 3)).to(DEV))
    bank = WeightBank()
    spec = ConvSpec(tag, w, None, None, None, False, 1, dil, 'frame', dilation=dil)
    bank.register(spec)
    geo = ConvGeometry(spec, N, H, W)
    assert L._FNS['tcvom_wgrad_igemm_variant'](C.byref(_phase_array(geo.wgrad)[0])).decode().startswith('wgrad_ws')
    xs = [(hu('x%d.%s' % (i, tag), (N, H, W, cin)) - 0.5).to(DEV).to(H16) for i in range(S)]
    dys = [(hu('dy%d.%s' % (i, tag), (N, H, W, cout)) - 0.5).to(DEV).to(H16) for i in range(S)]
    dw = torch.zeros(S, cout * 9 * cin, device=DEV)
    vp = lambda ts: C.cast((C.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), C.c_void_p)     # noqa: E731
    L.call('tcvom_wgrad_ws_multi', vp(dys), vp(xs), vp([dw[i] for i in range(S)]), S, _phase_array(geo.wgrad), cout, L.stream_ptr())
    torch.cuda.synchronize()
    for i in range(S):
        ref = torch.nn.grad.conv2d_weight(xs[i].float().cpu().permute(0, 3, 1, 2), (cout, cin, 3, 3),
                                          dys[i].float().cpu().permute(0, 3, 1, 2), padding=dil, dilation=dil)
        got = dw[i].cpu().view(cout, 3, 3, cin).permute(0, 3, 1, 2)
        assert rel_err(got, ref) < 1e-5, 'problem %d' % i


@pytest.mark.parametrize('cw', [128, 64])
def test_wgrad_ws_hetero_launch(cw):
    """Problems of DIFFERENT geometries (image size, channel counts, dilation, samples) in one launch of the accumulator-stationary
    weight gradient (csrc/wgradws.hip: tcvom_wgrad_ws_hetero) -- the channel-changing convs of a trunk riding with the big groups: a
    workgroup's run crosses problem and geometry borders (DMA map rebuilt, accumulators flushed), runs of one tile included.
    Against torch.nn.grad.conv2d_weight in fp32 on the same 16-bit operands, and against one launch per geometry."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.weights import ConvSpec, WeightBank
    if cw == 128:
        shapes = [(128, 128, 1, 1, 24, 40, 3), (256, 128, 1, 1, 13, 21, 2), (128, 64, 1, 2, 17, 33, 1), (256, 256, 2, 1, 16, 24, 2),
                  (128, 32, 1, 1, 9, 50, 1), (512, 128, 1, 1, 8, 16, 1)]
    else:
        shapes = [(64, 64, 1, 1, 20, 44, 3), (192, 64, 1, 1, 11, 37, 1), (64, 128, 2, 1, 12, 36, 2), (320, 64, 1, 2, 9, 33, 1), (64, 32, 1, 1, 40, 64, 1)]
    bank = WeightBank()
    descs, dys, xs, dws, gidx, refs = [], [], [], [], [], []
    for gi, (cin, cout, dil, N, H, W, S) in enumerate(shapes):
        tag = 'wgh%d_%d_%d_%d_%d' % (cw, gi, cin, cout, dil)
        w = nn.Parameter(formula_tensor('conv.%s.weight' % tag, (cout, cin, 3, 3)).to(DEV))
        spec = ConvSpec(tag, w, None, None, None, False, 1, dil, 'frame', dilation=dil)
        bank.register(spec)
        geo = ConvGeometry(spec, N, H, W)
        assert len(geo.wgrad) == 1 and L._FNS['tcvom_wgrad_igemm_variant'](C.byref(geo.wgrad[0])).decode() == 'wgrad_ws<%d>' % cw
        descs.append(geo.wgrad[0])
        for i in range(S):
            x = (hu('x%d.%s' % (i, tag), (N, H, W, cin)) - 0.5).to(DEV).to(H16)
            dy = (hu('dy%d.%s' % (i, tag), (N, H, W, cout)) - 0.5).to(DEV).to(H16)
            xs.append(x); dys.append(dy); gidx.append(gi)
            dws.append(torch.zeros(cout * 9 * cin, device=DEV))
            refs.append((torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2), (cout, cin, 3, 3),
                                                     dy.float().cpu().permute(0, 3, 1, 2), padding=dil, dilation=dil), cout, cin))
    n = len(xs)
    vp = lambda ts: C.cast((C.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), C.c_void_p)     # noqa: E731
    arr = (type(descs[0]) * len(descs))(*descs)
    gi_arr = (C.c_int32 * n)(*gidx)
    assert n <= L.call('tcvom_wgrad_ws_max_problems') and len(descs) <= L.call('tcvom_wgrad_ws_max_geometries')
    L.call('tcvom_wgrad_ws_hetero', vp(dys), vp(xs), vp(dws), n, arr, len(descs), C.cast(gi_arr, C.c_void_p), L.stream_ptr())
    torch.cuda.synchronize()
    for i in range(n):
        ref, cout, cin = refs[i]
        got = dws[i].cpu().view(cout, 3, 3, cin).permute(0, 3, 1, 2)
        assert rel_err(got, ref) < 1e-5, 'problem %d (geometry %d)' % (i, gidx[i])
    # a mixed channel window is refused, not mis-launched
    if cw == 128:
        w64 = nn.Parameter(formula_tensor('conv.wgh_mix.weight', (64, 64, 3, 3)).to(DEV))
        s64 = ConvSpec('wgh_mix', w64, None, None, None, False, 1, 1, 'frame')
        bank.register(s64)
        g64 = ConvGeometry(s64, 1, 16, 32)
        arr2 = (type(descs[0]) * 2)(descs[0], g64.wgrad[0])
        with pytest.raises(L.TcvomError):
            L.call('tcvom_wgrad_ws_hetero', vp(dys[:1]), vp(xs[:1]), vp(dws[:1]), 1, arr2, 2, C.cast((C.c_int32 * 1)(0), C.c_void_p), L.stream_ptr())


TT_HETERO_SETS = {
    # (cin, cout, k, stride, transposed, N, H, W) of every problem; all of one tile shape of igemm_tt
    '128x128': [(64, 128, 3, 2, False, 1, 32, 48), (128, 128, 1, 1, False, 2, 16, 24), (128, 256, 3, 2, False, 1, 16, 24),
                (128, 128, 4, 2, True, 1, 12, 20), (256, 128, 1, 1, False, 1, 10, 14), (128, 256, 1, 1, False, 1, 17, 9)],
    # (dy grids that are no multiple of 8 x 32: the halo-form weight gradient takes those, test_halo_wgrad_kernel)
    '32x128': [(6, 32, 3, 2, False, 1, 32, 48), (6, 32, 3, 1, False, 1, 24, 40), (32, 32, 4, 2, True, 1, 12, 36), (3, 16, 3, 2, False, 2, 18, 34)],
    '64x128': [(32, 64, 3, 2, False, 1, 32, 64), (128, 64, 1, 1, False, 1, 20, 28), (16, 64, 3, 2, False, 1, 26, 30)],
}


@pytest.mark.parametrize('name', list(TT_HETERO_SETS))
def test_wgrad_igemm_hetero_launch(name):
    """csrc/igemm.hip: igemm_tt_hetero_kernel -- the weight gradients of convs with DIFFERENT descriptors (1 x 1, stride-2 3 x 3,
    4-phase ConvTranspose, padded 6 -> 8 channel inputs; two calls per layer) in ONE launch, work list and descriptor table in device
    memory (tcvom_wgrad_igemm_hetero_plan), against one uniform `tcvom_wgrad_igemm_batched` launch per problem: the same kernel body on
    the same operands -- equal up to the order of the fp32 atomics -- and against torch.nn.grad in fp32."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    from tcvom_amd.weights import ConvSpec, WeightBank, _wgrad_tt_tile
    st = L.stream_ptr()
    bank = WeightBank()
    probs = []
    for i, (cin, cout, k, stride, tr, N, H, W) in enumerate(TT_HETERO_SETS[name]):
        tag = 'tth_%s_%d' % (name, i)
        shape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        w = nn.Parameter(formula_tensor('conv.%s.weight' % tag, shape).to(DEV))
        pad = 1 if k in (3, 4) else 0
        if (cin, k, stride) == (3, 3, 2):
            pad = 0
        spec = ConvSpec(tag, w, None, None, None, tr, stride, pad, 'frame', needs_dgrad=False)
        bank.register(spec)
        geo = ConvGeometry(spec, N, H, W)
        for call in range(2):
            x = torch.zeros(N, H, W, spec.cpad, dtype=H16, device=DEV)
            x[..., :cin] = nhwc(hu('x%d.%s' % (call, tag), (N, cin, H, W)) - 0.5)
            dy = nhwc(hu('dy%d.%s' % (call, tag), (N, cout, geo.OH, geo.OW)) - 0.5)
            probs.append((spec, geo, x, dy))
    tiles = {_wgrad_tt_tile(g) for _, g, _, _ in probs}
    assert len(tiles) == 1 and None not in tiles, tiles
    tm, tn = tiles.pop()
    assert '%dx%d' % (tm, tn) == name
    n = len(probs)
    vp = lambda ts: C.cast((C.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), C.c_void_p)
    # reference: one uniform launch per problem
    refs = []
    for spec, geo, x, dy in probs:
        dw = torch.zeros(spec.K * spec.T * spec.cpad, dtype=torch.float32, device=DEV)
        L.call('tcvom_wgrad_igemm_batched', vp([dy]), vp([x]), vp([dw]), 1, _phase_array(geo.wgrad), len(geo.wgrad), spec.K, st)
        refs.append(dw)
    # hetero: plan on the host, tables on the device, one launch
    descs = [d for _, g, _, _ in probs for d in g.wgrad]
    arr = (L.ConvDesc * len(descs))(*descs)
    nph = (C.c_int32 * n)(*[len(g.wgrad) for _, g, _, _ in probs])
    ldys = (C.c_int32 * n)(*[s.K for s, _, _, _ in probs])
    nwork = L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldys, C.c_void_p), n, tm, tn, None, 0)
    assert nwork >= n
    work = (C.c_int32 * (8 * nwork))()
    assert L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldys, C.c_void_p), n, tm, tn, C.cast(work, C.c_void_p), nwork) == nwork
    dtab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    wtab = torch.frombuffer(bytearray(bytes(work)), dtype=torch.int32).to(DEV)
    dws = [torch.zeros_like(r) for r in refs]
    L.call('tcvom_wgrad_igemm_hetero', vp([p[3] for p in probs]), vp([p[2] for p in probs]), vp(dws), n, L.ptr(dtab), L.ptr(wtab), nwork, tm, tn, st)
    torch.cuda.synchronize()
    for i, (spec, geo, x, dy) in enumerate(probs):
        assert rel_err(dws[i].cpu(), refs[i].cpu()) < 1e-5, 'problem %d vs its uniform launch' % i
        cin, cout, k, stride, tr, N, H, W = TT_HETERO_SETS[name][i // 2]
        xr = x[..., :cin].float().cpu().permute(0, 3, 1, 2)
        dyr = dy.float().cpu().permute(0, 3, 1, 2)
        pad = spec.pad
        if tr:
            want = torch.nn.grad.conv2d_weight(dyr, (cin, cout, k, k), xr, stride=stride, padding=pad)        # ConvTranspose: roles swapped
            got = dws[i].cpu().view(cout, k * k, spec.cpad)[:, :, :cin].permute(2, 0, 1).reshape(cin, cout, k, k)
        else:
            want = torch.nn.grad.conv2d_weight(xr, (cout, cin, k, k), dyr, stride=stride, padding=pad)
            got = dws[i].cpu().view(cout, k * k, spec.cpad)[:, :, :cin].permute(0, 2, 1).reshape(cout, cin, k, k)
        assert rel_err(got, want) < 2e-5, 'problem %d vs torch.nn.grad' % i
    # a problem of another tile shape is refused by the plan, not mis-launched
    other = (C.c_int32 * 1)(probs[0][0].K)
    with pytest.raises(L.TcvomError):
        rc = L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldys, C.c_void_p), n, 32 if tm != 32 else 128, 32 if tm != 32 else 128, None, 0)
        if rc < 0:
            raise L.TcvomError(L.last_error())


SCONV_CASES = [
    # cin, cout, k, stride, pad, transposed, N, H, W, which launches go to the kernel
    (32, 32, 4, 2, 1, True, 2, 16, 64, 'fwd'),        # decoder conv1 (resnet_dec.py:23-41): 4 phases x 4 taps, C = 32, one channel block
    (64, 64, 4, 2, 1, True, 1, 24, 40, 'fwd'),        # decoder layer4's ConvTranspose: two channel blocks, W not a multiple of the tile
    (64, 32, 3, 1, 1, False, 2, 16, 96, 'fwd+dgrad'), # 64 -> 32 at os2; its data gradient is the 32 -> 64 shape
    (32, 64, 3, 1, 1, False, 1, 40, 70, 'fwd+dgrad'),
    (32, 64, 3, 2, 1, False, 1, 32, 128, 'dgrad'),    # encoder conv3: the stride-2 data gradient = phases of 1 / 2 / 2 / 4 taps on dy
    (16, 32, 3, 2, 1, False, 2, 48, 64, 'dgrad'),     # 32-channel dy
    (16, 32, 3, 2, 0, False, 2, 38, 134, 'dgrad'),    # guidance head (res_gca_enc.py:20-28): stride 2, padding 0 on a reflection-padded
                                                      # input -> the phase grid (19 x 67) is one larger than dy (18 x 66), not a tile multiple
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,transposed,N,H,W,which', SCONV_CASES)
@pytest.mark.parametrize('bias', [False, True])
def test_sconv_kernel(cin, cout, k, stride, pad, transposed, N, H, W, which, bias):
    """Shapes served by the halo-tile direct conv of csrc/sconv.hip (K <= 64, C in {32, 64}: transposed 4x4 stride-2 forwards,
    stride-2 data gradients, 3x3 between 32 and 64 channels): conv (+ bias) + ReLU + batch statistics, data gradient and weight
    gradient against fp32 PyTorch on the same 16-bit operands."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd import ops
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    from tcvom_amd.weights import bank_token
    tag = 'sc%d_%d_%d_%d_%d_%d' % (cin, cout, k, stride, H, int(bias))
    bank, spec = _mini_bank(cin, cout, k, stride, pad, transposed, spectral=False, bias=bias, tag=tag)
    with torch.no_grad():
        spec.weight.mul_(0.3)
    geo = ConvGeometry(spec, N, H, W)
    variant = L._FNS['tcvom_conv_igemm_variant']
    if 'fwd' in which:
        assert variant(C.byref(_phase_array(geo.fwd)[0]), len(geo.fwd)).decode().startswith('sconv')
    if 'dgrad' in which:
        assert variant(C.byref(_phase_array(geo.dgrad)[0]), len(geo.dgrad)).decode().startswith('sconv')
    bn = nn.BatchNorm2d(cout).to(DEV)
    cfg = ops.ConvCfg(bank, spec, bn=bn, act=0, pre_relu=True)
    x = hu('x.' + tag, (N, cin, H, W)) - 0.5
    xg = nhwc(x).requires_grad_(True)
    token = bank_token(bank, 1, True)
    z = ops.conv_bn_act(cfg, xg, token, True)
    bank.flush_bn_counters()
    xr = bf(x).requires_grad_(True)
    wr = bf(spec.weight.detach().cpu()).requires_grad_(True)
    br = spec.bias.detach().cpu().clone().requires_grad_(True) if bias else None
    yr = F.relu(F.conv_transpose2d(xr, wr, br, stride, pad) if transposed else F.conv2d(xr, wr, br, stride, pad))
    mean, var = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    yq = yr + (bf(yr) - yr).detach()
    zr = (yq - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
    ck = Checker()
    ck.rel('z', nchw(z), zr, 2e-2)
    ck.rel('running_mean', bn.running_mean, 0.1 * mean.detach(), 1e-2)
    n_el = yr.numel() // cout
    ck.rel('running_var', bn.running_var, 0.9 + 0.1 * var.detach() * n_el / (n_el - 1), 1e-2)
    gz = hu('gz.' + tag, tuple(zr.shape)) - 0.5
    (z.float() * nhwc(gz).float()).sum().backward()
    (zr * bf(gz)).sum().backward()
    ck.rel('dx', nchw(xg.grad)[:, :cin], xr.grad, 4e-2)
    ck.rel('dw', spec.weight.grad, wr.grad, 3e-2)
    if bias:
        ck.rel('dbias', spec.bias.grad, br.grad, 3e-2)
    ck.done()


@pytest.mark.parametrize('cin,N,H,W,bias', [(64, 2, 20, 44, False), (64, 1, 48, 64, True), (128, 1, 13, 21, True),
                                             (128, 2, 24, 32, False), (128, 1, 40, 72, False)])
def test_wsconv_kernel(cin, N, H, W, bias):
    """Shapes served by the weight-stationary conv (csrc/wsconv.hip: stride-1 3x3, C == K in {64, 128}; sizes that are not
    multiples of the 16x32 / 8x16 pixel tiles exercise the masked halo, stores and statistics): forward with fused ReLU +
    batch statistics (+ bias), data gradient through the same kernel, against fp32 PyTorch on the same bf16 operands, and
    against the implicit-GEMM kernel the shape used before."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd import ops
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    tag = 'ws%d_%d_%d_%d' % (cin, N, H, W)
    w = nn.Parameter((formula_tensor('conv.%s.weight' % tag, (cin, cin, 3, 3)) * 0.2).to(DEV))
    b = nn.Parameter(formula_tensor('conv.%s.bias' % tag, (cin,)).to(DEV)) if bias else None
    bank = WeightBank()
    spec = ConvSpec(tag, w, None, None, b, False, 1, 1, 'frame')
    bank.register(spec)
    geo = ConvGeometry(spec, N, H, W)
    assert L._FNS['tcvom_conv_igemm_variant'](C.byref(geo.fwd[0]), 1).decode().startswith('wsconv')
    assert L._FNS['tcvom_conv_igemm_variant'](C.byref(geo.dgrad[0]), 1).decode().startswith('wsconv')
    bn = nn.BatchNorm2d(cin).to(DEV)
    cfg = ops.ConvCfg(bank, spec, bn=bn, act=0, pre_relu=True)
    x = hu('x.' + tag, (N, cin, H, W)) - 0.5
    xg = nhwc(x).requires_grad_(True)
    token = bank_token(bank, 1, True)
    z = ops.conv_bn_act(cfg, xg, token, True)
    bank.flush_bn_counters()
    xr = bf(x).requires_grad_(True)
    wr = bf(spec.weight.detach().cpu()).clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, wr, b.detach().cpu() if bias else None, 1, 1))
    mean, var = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    yq = yr + (bf(yr) - yr).detach()
    zr = (yq - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
    ck = Checker()
    ck.rel('z', nchw(z), zr, 2e-2)
    ck.rel('running_mean', bn.running_mean, 0.1 * mean.detach(), 1e-2)
    n_el = yr.numel() // cin
    ck.rel('running_var', bn.running_var, 0.9 + 0.1 * var.detach() * n_el / (n_el - 1), 1e-2)
    gz = hu('gz.' + tag, tuple(zr.shape)) - 0.5
    (z.float() * nhwc(gz).float()).sum().backward()
    (zr * bf(gz)).sum().backward()
    ck.rel('dx', nchw(xg.grad), xr.grad, 4e-2)
    ck.rel('dw', spec.weight.grad, wr.grad, 3e-2)
    ck.done()
    # raw kernel against the implicit GEMM on the same operands: same products, different fp32 summation order
    st = L.stream_ptr()
    y_ws = torch.empty(N, H, W, cin, device=DEV, dtype=H16)
    ops._launch_conv(geo.fwd, xg.detach(), bank.fwd_ptr(spec, 0), y_ws, b, None, 1, st)
    d = geo.fwd[0]
    taps = [(d.tap_dh[t], d.tap_dw[t]) for t in range(9)]
    yref = F.relu(F.conv2d(bf(x), bf(spec.weight.detach().cpu()), b.detach().cpu() if bias else None, 1, 1))
    assert len(taps) == 9 and rel_err(nchw(y_ws), yref) < 1e-2
    # (IEEE fp16 results come with IEEE fp16 operands only -- the fp16 island of the bf16 build: tests/test_gpu_f16_island.py)
    if H16 == torch.bfloat16:
        y_h = torch.empty(N, H, W, cin, device=DEV, dtype=torch.float16)
        with pytest.raises(L.TcvomError):
            ops._launch_conv(geo.fwd, xg.detach(), bank.fwd_ptr(spec, 0), y_h, b, None, 1, st)


# --------------------------------------------------------------------------------------------- spectral norm
def test_tiled_weight_pack_equals_elementwise_pack(monkeypatch):
    """The tiled pack (32 x 64 x T tiles through LDS, csrc/spectral.hip sn_pack_tile) must write the same bytes as the
    one-thread-per-element pack for every layer kind: plain / SpectralNorm / ConvTranspose [cin][cout] order / IEEE fp16 forward pack
    (the fp16 island of the bf16 build) / weight-standardised, padded channels (6 -> 8, 72 -> 128), K below and off the tile size, no data-gradient pack."""
    from tcvom_amd import weights as Wm
    from tcvom_amd.weights import WeightBank, ConvSpec, bank_token

    def build():
        bank = WeightBank()
        specs = []

        def add(tag, shape, transposed=False, spectral=False, **kw):
            w = nn.Parameter(formula_tensor('pk.%s.weight' % tag, shape).to(DEV))
            u = v = None
            if spectral:
                u = nn.Parameter(formula_tensor('pk.%s.u' % tag, (shape[0],)).to(DEV), requires_grad=False)
                v = nn.Parameter(formula_tensor('pk.%s.v' % tag, (int(np.prod(shape[1:])),)).to(DEV), requires_grad=False)
            f16 = kw.pop('f16', False)
            s = ConvSpec(tag, w, u, v, None, transposed, 2 if transposed else 1, 1, 'frame', **kw)
            s.f16 = f16
            bank.register(s)
            specs.append(s)
        add('plain', (48, 32, 3, 3))
        add('sn', (64, 64, 3, 3), spectral=True)
        add('sn_small_k', (20, 6, 3, 3), spectral=True, needs_dgrad=False)
        add('convT', (96, 40, 4, 4), transposed=True, spectral=True)
        add('f16', (32, 32, 3, 3), spectral=True, f16=True)
        add('f16_frag', (64, 64, 3, 3), spectral=True, f16=True)
        add('f16_1x1', (128, 64, 1, 1), spectral=True, f16=True)
        add('ws', (64, 128, 1, 1), ws=True)
        add('ws3', (36, 72, 3, 3), ws=True, cpad=128)
        bank_token(bank, 1, True)
        torch.cuda.synchronize()
        return bank, specs

    # ONE bank, one set of sigmas (the power iteration's reductions are not bit-reproducible from run to run): pack it with
    # the element-wise work list, keep the bytes, then re-pack the same call slot with the tiled work list
    import ctypes as C
    from tcvom_amd import _lib as L
    monkeypatch.setattr(Wm, 'TILED_PACK', False)
    bank, specs = build()
    assert int(bank.work_pack_all.view(-1, 3)[:, 1].max()) <= 1
    ref_f, ref_b = bank.fwd_arena.clone(), bank.bwd_arena.clone()
    assert float(ref_f.float().abs().sum()) > 0 and float(ref_b.float().abs().sum()) > 0
    monkeypatch.setattr(Wm, 'TILED_PACK', True)
    rows = torch.tensor(WeightBank._pack_rows(specs), dtype=torch.int32).reshape(-1).to(DEV)
    assert int(rows.view(-1, 3)[:, 1].min()) >= 2
    bank.fwd_arena.fill_(7.0)
    bank.bwd_arena.fill_(7.0)
    L.call('tcvom_sn_pack', L.ptr(bank.table), C.byref(bank.scratch), L.ptr(rows), rows.numel() // 3, 0, L.ptr(bank.fwd_arena),
           L.ptr(bank.bwd_arena), bank.fwd_stride, bank.bwd_stride, L.stream_ptr())
    torch.cuda.synchronize()
    # the layers' own ranges (alignment gaps between layers are never written and keep the fill value)
    for s in specs:
        nf = s.K * s.T * s.cpad
        assert torch.equal(bank.fwd_arena[s.fwd_off:s.fwd_off + nf].view(torch.int16), ref_f[s.fwd_off:s.fwd_off + nf].view(torch.int16)), s.name
        if s.needs_dgrad:
            nb = s.C * s.T * s.K
            assert torch.equal(bank.bwd_arena[s.bwd_off:s.bwd_off + nb].view(torch.int16), ref_b[s.bwd_off:s.bwd_off + nb].view(torch.int16)), s.name


@pytest.mark.parametrize('R,Cc,ldi,ldo,batch', [(200, 130, 136, 256, 2), (70, 64, 64, 72, 1), (33, 50, 50, 35, 3), (4100, 576, 576, 4160, 1)])
def test_transpose_bf16(R, Cc, ldi, ldo, batch):
    """in [batch][R][ldi] (first Cc columns) -> out [batch][Cc][ldo] (columns >= R zero): the 16-byte kernel (aligned strides)
    and the scalar fallback (odd strides), bit exact."""
    from tcvom_amd import _lib as L
    x = hu('tr.x', (batch, R, ldi)).to(DEV).to(H16)
    out = torch.full((batch, Cc, ldo), 7.0, dtype=H16, device=DEV)
    L.call('tcvom_transpose_bf16', L.ptr(x), L.ptr(out), R, Cc, ldi, ldo, batch, R * ldi, Cc * ldo, L.stream_ptr())
    ref = torch.zeros((batch, Cc, ldo), dtype=H16, device=DEV)
    ref[:, :, :R] = x[:, :, :Cc].transpose(1, 2)
    assert torch.equal(out, ref)


@pytest.mark.parametrize('B,N,DV', [(2, 80, 256), (1, 1020, 2048), (3, 264, 512), (2, 500, 256)])
def test_gca_fused_softmax_backward_gemm(B, N, DV):
    """tcvom_gca_dp_softmax_bwd (dP GEMM with the softmax backward in its epilogue, row sums from <dO, O>) against the
    unfused formula in fp32 on the same bf16 operands; padding columns N..ld must come out zero."""
    from tcvom_amd import _lib as L
    ld = (N + 63) // 64 * 64
    tag = 'fsb%d_%d' % (N, DV)
    st = L.stream_ptr()
    Pf = torch.softmax(hu('p.' + tag, (B, N, N)) * 8, dim=2)
    P = torch.zeros(B, N, ld)
    P[:, :, :N] = Pf
    P = P.to(H16).to(DEV)
    V = (hu('v.' + tag, (B, N, DV)) - 0.5).to(H16).to(DEV)
    dO = (hu('do.' + tag, (B, N, DV)) - 0.5).to(H16).to(DEV)
    cvec = (hu('c.' + tag, (B, N)) + 0.5).to(DEV)
    Pq, Vq, dOq = P[:, :, :N].float(), V.float(), dO.float()
    O = torch.bmm(Pq, Vq).contiguous()                            # what the forward saves (fp32)
    delta = torch.empty(B, N, device=DEV)
    L.call('tcvom_rowdot_bf16', L.ptr(dO), L.ptr(O), 1, L.ptr(delta), B * N, DV, st)
    assert rel_err(delta.cpu(), (dOq * O).sum(2).cpu()) < 1e-5
    Ob = O.to(H16)
    L.call('tcvom_rowdot_bf16', L.ptr(dO), L.ptr(Ob), 0, L.ptr(delta), B * N, DV, st)
    assert rel_err(delta.cpu(), (dOq * Ob.float()).sum(2).cpu()) < 1e-5
    L.call('tcvom_rowdot_bf16', L.ptr(dO), L.ptr(O), 1, L.ptr(delta), B * N, DV, st)
    T = torch.full((B, N, ld), float('nan'), dtype=H16, device=DEV)
    fused_t = ld % 256 == 0                                  # (1020 -> 1024, 500 -> 512: the epilogue also writes T^T and P^T)
    Tt = torch.full((B, ld, ld), float('nan'), dtype=H16, device=DEV) if fused_t else None
    Pt = torch.full((B, ld, ld), float('nan'), dtype=H16, device=DEV) if fused_t else None
    L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T), L.ptr(Tt), L.ptr(Pt),
           N, DV, ld, B, st)
    dP = torch.bmm(dOq, Vq.transpose(1, 2))
    ref = Pq * (dP - (Pq * dP).sum(2, keepdim=True)) * cvec[:, None, :]
    assert rel_err(T[:, :, :N].float().cpu(), ref.cpu()) < 1.5e-2
    if ld > N:
        assert float(T[:, :, N:].float().abs().max()) == 0.0
    if fused_t:
        want_t = torch.zeros(B, ld, ld, dtype=H16, device=DEV)
        want_t[:, :, :N] = T.transpose(1, 2)
        want_p = torch.zeros(B, ld, ld, dtype=H16, device=DEV)
        want_p[:, :, :N] = P.transpose(1, 2)
        assert torch.equal(Tt, want_t) and torch.equal(Pt, want_p)          # exact transposes, zero padding rows / columns


@pytest.mark.parametrize('B,N,DV,D', [(2, 1000, 512, 192), (3, 2040, 2048, 576), (1, 8160, 256, 576)])
def test_gca_row_contractions_read_the_matrix_k_major(B, N, DV, D):
    """tcvom_gca_dv (dV = P^T dO), tcvom_gca_pv (O = P V) and tcvom_gca_dq_dk (dWq = T G, M' = T^T G): the N x N matrix is read
    as it lies in memory by the products that contract its ROW index (k-major B operand), dO / V likewise (k-major A operand),
    through transposing LDS reads -- against fp32 matmuls on the same 16-bit operands.  Ragged N (rows N .. ld of a batch entry's slot belong to the NEXT entry and must
    count as zeros), the K-split tail of the 192-row tiles (3 x 2040 and 1 x 8160: 1.125 rounds of tiles per product)."""
    from tcvom_amd import _lib as L
    ld = (N + 255) // 256 * 256
    tag = 'km%d_%d' % (N, DV)
    st = L.stream_ptr()
    P = torch.zeros((B, N, ld), dtype=H16, device=DEV)
    P[:, :, :N] = (hu('p.' + tag, (B, N, N)) * (hu('pm.' + tag, (B, N, N)) > 0.7)).to(H16).to(DEV)      # zeros in the padding columns
    dO = (hu('do.' + tag, (B, N, DV)) - 0.5).to(H16).to(DEV)
    dV = torch.full((B, N, DV), 7.0, device=DEV)
    L.call('tcvom_gca_dv', L.ptr(P), L.ptr(dO), L.ptr(dV), N, DV, ld, B, st)
    ref = torch.bmm(P[:, :, :N].float().transpose(1, 2), dO.float())
    assert rel_err(dV.cpu(), ref.cpu()) < 1e-5
    # O = P V with V k-major (the A operand)
    L.call('tcvom_gca_pv', L.ptr(P), L.ptr(dO), L.ptr(dV), N, DV, ld, B, st)
    ref = torch.bmm(P[:, :, :N].float(), dO.float())
    assert rel_err(dV.cpu(), ref.cpu()) < 1e-5
    del dV, ref
    G = (hu('g.' + tag, (B, N, D)) - 0.5).to(H16).to(DEV)
    Gt = torch.zeros((B, D, ld), dtype=H16, device=DEV)
    Gt[:, :, :N] = G.transpose(1, 2)
    T = P                                                              # (any N x N matrix with zero padding columns)
    dWq = torch.full((B, N, D), 7.0, device=DEV)
    Mp = torch.full((B, N, D), 7.0, device=DEV)
    L.call('tcvom_gca_dq_dk', L.ptr(T), L.ptr(Gt), L.ptr(dWq), L.ptr(Mp), N, D, ld, B, st)
    Tf = T[:, :, :N].float()
    assert rel_err(dWq.cpu(), torch.bmm(Tf, G.float()).cpu()) < 1e-5
    assert rel_err(Mp.cpu(), torch.bmm(Tf.transpose(1, 2), G.float()).cpu()) < 1e-5


@pytest.mark.parametrize('B,N,D', [(2, 1016, 192), (1, 2040, 576), (3, 512, 64)])
def test_gca_scores_softmax_without_the_score_matrix(B, N, D):
    """tcvom_gca_scores_softmax (score GEMM whose epilogue writes exp(S' - tile row max) + per-tile row statistics, then an
    in-place rescale) against softmax(c_j <G_i, G_j> - d_j [i == j]) in fp32 on the same 16-bit operands; padding columns zero;
    rows sum to one; the -1e4 self-mask of unknown keys (ops.py:186-188) must survive as an exact zero probability."""
    from tcvom_amd import _lib as L
    ld = (N + 255) // 256 * 256
    tag = 'gss%d_%d' % (N, D)
    st = L.stream_ptr()
    G = ((hu('g.' + tag, (B, N, D)) - 0.5) * 1.5).to(H16).to(DEV)
    cvec = (hu('c.' + tag, (B, N)) * 2 + 0.2).to(DEV)
    dvec = torch.where(hu('d.' + tag, (B, N)) > 0.5, torch.full((B, N), 1e4), torch.zeros(B, N)).to(DEV)
    assert L.call('tcvom_gca_scores_softmax_ok', N, D, ld, B) == 1
    P = torch.full((B, N, ld), float('nan'), dtype=H16, device=DEV)
    stats = torch.empty((B, N, ld // 256, 2), dtype=torch.float32, device=DEV)
    L.call('tcvom_gca_scores_softmax', L.ptr(G), L.ptr(cvec), L.ptr(dvec), L.ptr(P), L.ptr(stats), N, D, ld, B, st)
    Gf = G.float()
    S = torch.bmm(Gf, Gf.transpose(1, 2)) * cvec[:, None, :]
    S = S - torch.diag_embed(dvec)
    ref = torch.softmax(S, dim=2)
    got = P[:, :, :N].float()
    assert bool(torch.isfinite(P.float()).all())
    if ld > N:
        assert float(P[:, :, N:].float().abs().max()) == 0.0
    # (peaked rows: one probability near 1 rounded twice -- numerator, then rescaled value -- to 8 / 11 significant bits)
    assert float((got.sum(2) - 1).abs().max()) < tol(5e-3, 1.5e-3)
    assert rel_err(got.cpu(), ref.cpu()) < tol(1.5e-2, 2e-3)
    masked = torch.diag_embed(dvec > 0)
    assert float(got[masked].max()) == 0.0                   # exp(-1e4 - max) underflows to an exact zero, as in the reference


@pytest.mark.parametrize('ncols', [8160, 2040, 3000, 250])
def test_row_softmax_kernels(ncols):
    """GCA attention softmax forward / backward rows at the 1080p length (8160 keys: 4 register chunks) and shorter / ragged
    ones, against fp32 PyTorch."""
    from tcvom_amd import _lib as L
    rows = 37
    ld = (ncols + 63) // 64 * 64
    S = (hu('sm.s%d' % ncols, (rows, ld)) * 6).to(DEV)
    P = torch.empty(rows, ld, dtype=H16, device=DEV)
    st = L.stream_ptr()
    L.call('tcvom_row_softmax', L.ptr(S), L.ptr(P), rows, ncols, ld, ld, st)
    ref = torch.softmax(S[:, :ncols].float(), dim=1)
    assert rel_err(P[:, :ncols].float().cpu(), ref.cpu()) < 5e-3
    assert float(P[:, ncols:].float().abs().max()) == 0.0 if ld > ncols else True
    dP = hu('sm.dp%d' % ncols, (rows, ld)).to(DEV)
    cvec = (hu('sm.c%d' % ncols, (ncols,)) + 1.5).to(DEV)
    T = torch.empty(rows, ld, dtype=H16, device=DEV)
    L.call('tcvom_row_softmax_bwd', L.ptr(P), L.ptr(dP), L.ptr(cvec), L.ptr(T), rows, ncols, ld, ld, rows, st)
    Pf = P[:, :ncols].float()
    a = (Pf * dP[:, :ncols]).sum(1, keepdim=True)
    tref = Pf * (dP[:, :ncols] - a) * cvec[None, :]
    assert rel_err(T[:, :ncols].float().cpu(), tref.cpu()) < 1e-2
    assert float(T[:, ncols:].float().abs().max()) == 0.0 if ld > ncols else True


@pytest.mark.parametrize('rows_b,rows_a,kred,fp32', [(4100, 4096, 192, True), (4000, 4608, 320, False)])
def test_dense_gemm_256_tile_config(rows_b, rows_a, kred, fp32):
    """Dense NT GEMM at sizes that select the 256x256 tile (the GCA score / P.V GEMMs at 1080p), ragged pixel edge,
    with the per-row scale / diagonal epilogue of the score GEMM."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import dense_desc
    B = (hu('g256.b', (rows_b, kred)) - 0.5).to(DEV).to(H16)
    A = (hu('g256.a', (rows_a, kred)) - 0.5).to(DEV).to(H16)
    sc = (hu('g256.s', (rows_a,)) + 0.5).to(DEV)
    ld = (rows_a + 63) // 64 * 64
    out = torch.zeros(rows_b, ld, device=DEV, dtype=torch.float32 if fp32 else H16)
    d = dense_desc(rows_b, rows_a, kred, ld, out_fp32=fp32)
    L.call('tcvom_conv_igemm', L.ptr(B), L.ptr(A), L.ptr(out), None, L.ptr(sc), None, None, C.byref(d), L.stream_ptr())
    ref = (B.float() @ A.float().t()) * sc[None, :]
    got = out[:, :rows_a].float()
    assert rel_err(got.cpu(), ref.cpu()) < (1e-5 if fp32 else 6e-3)
    assert float(out[:, rows_a:].abs().max()) == 0.0 if ld > rows_a else True


@pytest.mark.parametrize('fp32', [True, False])
def test_dense_gemm_256_staggered_kernel_epilogues(fp32):
    """gemm_nt256 (csrc/gemm256.hip) on a batch of 2 problems whose row counts are not multiples of the 256-row
    tile on either side, with every epilogue term: out = relu(acc * scale[m] + bias[m] - [m == n] diag[m])."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import dense_desc
    nb, rows_b, rows_a, kred = 2, 2590, 2700, 192
    ld = (rows_a + 63) // 64 * 64
    B = (hu('g2s.b', (nb, rows_b, kred)) - 0.5).to(DEV).to(H16)
    A = (hu('g2s.a', (nb, rows_a, kred)) - 0.5).to(DEV).to(H16)
    sc = (hu('g2s.s', (nb, rows_a)) + 0.5).to(DEV)
    bias = (hu('g2s.bias', (nb, rows_a)) - 0.5).to(DEV)
    dg = (hu('g2s.d', (nb, rows_a)) * 3).to(DEV)
    out = torch.zeros(nb, rows_b, ld, device=DEV, dtype=torch.float32 if fp32 else H16)
    d = dense_desc(rows_b, rows_a, kred, ld, batch=nb, in_bstride=rows_b * kred, w_bstride=rows_a * kred,
                   out_bstride=rows_b * ld, vec_bstride=rows_a, out_fp32=fp32)
    d.act = 1
    assert L._FNS['tcvom_conv_igemm_variant'](C.byref(d), 1).decode() == 'gemm_nt256'
    L.call('tcvom_conv_igemm', L.ptr(B), L.ptr(A), L.ptr(out), L.ptr(bias), L.ptr(sc), L.ptr(dg), None, C.byref(d), L.stream_ptr())
    ref = torch.bmm(B.float(), A.float().transpose(1, 2)) * sc[:, None, :] + bias[:, None, :]
    n = min(rows_a, rows_b)
    idx = torch.arange(n, device=DEV)
    ref[:, idx, idx] -= dg[:, :n]
    ref = torch.relu(ref)
    assert rel_err(out[:, :, :rows_a].float().cpu(), ref.cpu()) < (1e-5 if fp32 else 6e-3)
    assert float(out[:, :, rows_a:].abs().max()) == 0.0


@pytest.mark.parametrize('rows_a,rows_b,kred', [(576, 5700, 512), (64, 300, 128), (576, 8000, 2048)])
def test_dense_gemm_pair_one_launch(rows_a, rows_b, kred):
    """tcvom_gemm_pair: two products against the same weight operand (the d(query) / d(key) GEMMs of the attention scores,
    576 rows on 192-row tiles), with different batch strides of the two inputs, against float matmuls; the small case takes
    the two-launch fallback; the third has 576 tiles = two rounds of 256 + 64 tiles whose reduction is split over 4 workgroups
    each (fp32 atomics into the zeroed tail of the last output slice)."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import dense_desc
    nb = 3
    pad = rows_b + 10                                         # rows of in2's per-batch allocation
    W_ = (hu('gp.w', (nb, rows_a, kred)) - 0.5).to(DEV).to(H16)
    X1 = (hu('gp.x1', (nb, rows_b, kred)) - 0.5).to(DEV).to(H16)
    X2 = (hu('gp.x2', (nb, pad, kred)) - 0.5).to(DEV).to(H16)
    o1 = torch.full((nb, rows_b, rows_a), 7.0, device=DEV)
    o2 = torch.full((nb, rows_b, rows_a), 7.0, device=DEV)
    d = dense_desc(rows_b, rows_a, kred, rows_a, batch=nb, in_bstride=rows_b * kred, w_bstride=rows_a * kred,
                   out_bstride=rows_b * rows_a, out_fp32=True)
    assert (L._FNS['tcvom_conv_igemm_variant'](C.byref(d), 1).decode() == 'gemm_nt256') == (rows_a == 576)
    L.call('tcvom_gemm_pair', L.ptr(X1), L.ptr(X2), L.ptr(W_), L.ptr(o1), L.ptr(o2), C.byref(d), pad * kred, L.stream_ptr())
    r1 = torch.bmm(X1.float(), W_.float().transpose(1, 2))
    r2 = torch.bmm(X2[:, :rows_b].float(), W_.float().transpose(1, 2))
    assert rel_err(o1.cpu(), r1.cpu()) < 1e-5
    assert rel_err(o2.cpu(), r2.cpu()) < 1e-5


@pytest.mark.parametrize('training', [True, False])
def test_spectral_norm_inner_product_from_the_batchnorm_backward(training, monkeypatch):
    """<dW~, weight_bar> of a SpectralNorm'd conv in front of a BatchNorm (models/GCA/ops.py:25-45 under autograd) equals
    sigma <dy, y>, which the BatchNorm-backward finalize derives from its two per-channel sums (tcvom_sn_dot) -- against the pass
    over the weight gradient (sn_bwd_inner_kernel, itself checked against the oracle in test_spectral_norm_bank): the scalar per
    (call, layer) and the weight_bar gradient.  eps = 0.5 in training mode makes the term large (it is proportional to eps:
    batch-statistics BatchNorm is scale-invariant up to eps); eval mode uses running statistics, where it is not small at all."""
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    S, B, cin, cout, H, W = 3, 1, 32, 64, 20, 28

    def run(dot):
        monkeypatch.setattr(ops, 'SN_DOT', dot)
        bank, spec = _mini_bank(cin, cout, 3, 1, 1, False, spectral=True, tag='snd')
        bn = nn.BatchNorm2d(cout, eps=0.5 if training else 1e-5).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(formula_tensor('bn.weight', (cout,)))
            bn.bias.copy_(formula_tensor('bn.bias', (cout,)))
            bn.running_mean.copy_(0.1 * formula_tensor('bn.rm', (cout,)))
            bn.running_var.copy_(0.5 + formula_tensor('bn.rv', (cout,)).abs())
        cfg = ops.ConvCfg(bank, spec, bn=bn, act=1)
        assert bool(spec.sn_dot) == dot
        xg = torch.cat([nhwc(hu('x%d.snd' % f, (B, cin, H, W)) * (1.0 + 0.5 * f)) for f in range(S)], 0).requires_grad_(True)
        token = bank_token(bank, S, training)
        bank.frames_per_op = S
        z = ops.conv_bn_act(cfg, xg, token, training)
        bank.frames_per_op = 1
        g = nhwc(hu('g.snd', (S * B, cout, H, W)) - 0.3)
        (z.float() * g.float()).sum().backward()
        bank.flush_bn_counters()
        torch.cuda.synchronize()
        nl, calls = len(bank.specs), (S if training else 1)
        sig = bank.sigma.view(-1, nl)[:calls, spec.layer_id].cpu()
        inner = (sig * bank.sn_dots.view(-1, nl)[:calls, spec.layer_id].cpu()) if dot else bank.inner.view(-1, nl)[:calls, spec.layer_id].cpu()
        return spec.weight.grad.cpu(), inner, xg.grad.float().cpu()

    a, b = run(True), run(False)
    assert float(b[1].abs().min()) > 0
    # (the pass over dW~ sums products of 16-bit-rounded dy; the BatchNorm sums are taken before that rounding)
    assert rel_err(a[1], b[1]) < 2e-2, (a[1], b[1])
    assert rel_err(a[0], b[0]) < 1e-3
    assert rel_err(a[2], b[2]) < 5e-3                        # (the power iteration's atomics: sigma may differ by an ulp between runs)
    # the term matters in this test: without it the gradient would be visibly different
    nsig = float((b[1].abs() / 1.0).max())
    assert nsig > 1e-3 * float(b[0].abs().max())


@pytest.mark.parametrize('transposed', [False, True])
def test_spectral_norm_bank(transposed):
    """Chained per-call power iterations, packed weights and the weight_bar gradient vs the oracle."""
    import oracle
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    cin, cout, k = (32, 64, 4) if transposed else (32, 64, 3)
    bank, spec = _mini_bank(cin, cout, k, 2 if transposed else 1, 1, transposed, spectral=True, tag='sn%d' % transposed)
    cfg = ops.ConvCfg(bank, spec)
    state = {'m.weight_bar': spec.weight.detach().cpu().clone().requires_grad_(True),
             'm.weight_u': spec.u.detach().cpu().clone(), 'm.weight_v': spec.v.detach().cpu().clone()}
    calls = 3
    token = bank_token(bank, calls, True)
    xs = [hu('x.sn%d' % c, (1, cin, 8, 10)) for c in range(calls)]
    gys, ys, yrs = [], [], []
    for c in range(calls):
        y = ops.conv_bn_act(cfg, nhwc(xs[c]), token, True)
        wn = oracle.spectral_weight(state, 'm', True)
        yr = F.conv_transpose2d(bf(xs[c]), wn, None, 2, 1) if transposed else F.conv2d(bf(xs[c]), wn, None, 1, 1)
        assert rel_err(nchw(y), yr) < 1.5e-2, 'call %d' % c
        ys.append(y)
        yrs.append(yr)
    assert_close(spec.u.cpu(), state['m.weight_u'], 1e-4, 1e-5, 'u after %d calls' % calls)
    assert_close(spec.v.cpu(), state['m.weight_v'], 1e-4, 1e-5, 'v after %d calls' % calls)
    loss = 0
    lossr = 0
    for c in range(calls):
        gy = hu('gy.sn%d' % c, tuple(yrs[c].shape))
        loss = loss + (ys[c].float() * nhwc(gy).float()).sum()
        lossr = lossr + (yrs[c] * bf(gy)).sum()
    loss.backward()
    lossr.backward()
    assert rel_err(spec.weight.grad.cpu(), state['m.weight_bar'].grad) < 2e-2, 'weight_bar grad'


# --------------------------------------------------------------------------------------------- resampling
def test_pool_upsample_reflect():
    from tcvom_amd import ops
    x = hu('x.pool', (2, 16, 12, 20))
    xg = nhwc(x).requires_grad_(True)
    xr = bf(x).requires_grad_(True)
    for fn, ref in ((ops.avgpool2, lambda t: F.avg_pool2d(t, 2, 2)),
                    (ops.upsample2, lambda t: F.interpolate(t, scale_factor=2, mode='nearest')),
                    (ops.reflect_pad1, lambda t: F.pad(t, (1, 1, 1, 1), mode='reflect'))):
        xg.grad = None
        xr.grad = None
        y, yr = fn(xg), ref(xr)
        assert rel_err(nchw(y), yr) < 1e-2
        g = hu('g.pool', tuple(yr.shape))
        (y.float() * nhwc(g).float()).sum().backward()
        (yr * bf(g)).sum().backward()
        assert rel_err(nchw(xg.grad), xr.grad) < 1e-2


def test_head_conv():
    from tcvom_amd import ops
    x = hu('x.head', (2, 32, 12, 20))
    w = nn.Parameter(formula_tensor('decoder.conv2.weight', (1, 32, 3, 3)).to(DEV))
    b = nn.Parameter(formula_tensor('decoder.conv2.bias', (1,)).to(DEV))
    xg = nhwc(x).requires_grad_(True)
    a = ops.head_conv(xg, w, b)
    xr = bf(x).requires_grad_(True)
    wr, br = w.detach().cpu().clone().requires_grad_(True), b.detach().cpu().clone().requires_grad_(True)
    ar = (torch.tanh(F.conv2d(xr, wr, br, 1, 1)) + 1) / 2
    assert_close(a.cpu(), ar, 0, 1e-5, 'alpha')
    g = hu('g.head', tuple(ar.shape))
    (a * g.to(DEV)).sum().backward()
    (ar * g).sum().backward()
    assert rel_err(nchw(xg.grad), xr.grad) < 1e-2
    assert rel_err(w.grad.cpu(), wr.grad) < 1e-3
    assert rel_err(b.grad.cpu(), br.grad) < 1e-3


# --------------------------------------------------------------------------------------------- TAM
@pytest.mark.parametrize('name', list(TAM_CASES))
def test_tam_module_vs_reference_golden(name):
    """FeatureAggregationModule through its reference signature vs vectors captured from the reference."""
    from models.VMN.VMN_model import FeatureAggregationModule
    B, C, H, W, win, kind = TAM_CASES[name]
    g = golden(name)
    fam = FeatureAggregationModule(C, 1, win)
    fam.load_state_dict({k: formula_tensor('decoder.fam.' + k, v.shape) for k, v in fam.state_dict().items()})
    fam.to(DEV)
    x, b, f = (hu('tam.' + t, (B, C, H, W)).to(DEV).requires_grad_(True) for t in 'xbf')
    out, attb, attf, small = fam(x, b, f, tam_mask(kind, B, H, W).to(DEV))
    assert np.array_equal(small.cpu().numpy().astype(np.uint8), g['small'])
    scale = float(np.abs(g['out']).max())
    assert_close(out.cpu(), g['out'], 0, 2e-2 * scale, 'out')        # bf16 features: 2 % of the range
    lscale = max(float(np.abs(g['attb']).max()), 1e-3)
    assert_close(attb.cpu(), g['attb'], 0, 2e-2 * lscale, 'attb')
    assert_close(attf.cpu(), g['attf'], 0, 2e-2 * lscale, 'attf')
    ((out * hu('tam.gout', out.shape).to(DEV)).sum() + (attb * hu('tam.gattb', attb.shape).to(DEV)).sum()
     + (attf * hu('tam.gattf', attf.shape).to(DEV)).sum()).backward()
    for got, key in ((x.grad, 'gx'), (b.grad, 'gb'), (f.grad, 'gf'), (fam.key_conv.weight.grad, 'gkw'),
                     (fam.query_conv.weight.grad, 'gqw'), (fam.value_conv.bias.grad, 'gvb'), (fam.key_conv.bias.grad, 'gkb')):
        want = g[key]
        assert_close(got.cpu(), want, 0, 3e-2 * float(np.abs(want).max()) + 1e-6, key)


@pytest.mark.parametrize('C', [16, 128, 256])
def test_tam_attention_kernel_tight(C):
    """The fused attention kernels alone against the oracle's dense formula on identical 16-bit inputs: C = 128 runs on the matrix
    cores (forward, backward pass A and -- round 5 -- pass B), C = 16 and C = 256 (the FBA / DIM bases) on the LDS-tiled vector kernels."""
    import oracle
    from tcvom_amd import ops
    B, H, W, win = 2, 10, 14, 7
    q, kb, kf, v = (hu('tamk.' + t, (B, C, H, W)) for t in ('q', 'kb', 'kf', 'v'))
    mask = (hu('tamk.m', (B, 1, H, W)) > -0.2)
    args = [nhwc(t).requires_grad_(True) for t in (q, kb, kf, v)]
    out, attb, attf = ops.tam_attention(*args, mask[:, 0].to(torch.uint8).to(DEV).contiguous(), win)
    refs = [bf(t).requires_grad_(True) for t in (q, kb, kf, v)]
    ab, lb = oracle.temporal_attention(refs[0], refs[1], mask, win)
    af, lf = oracle.temporal_attention(refs[0], refs[2], mask, win)
    outr = refs[3] + ab + af
    assert_close(attb.cpu(), lb, 1e-4, 1e-4, 'logits b')
    assert_close(attf.cpu(), lf, 1e-4, 1e-4, 'logits f')
    assert rel_err(nchw(out), outr) < 1e-2
    go, gb, gf = hu('tamk.go', tuple(outr.shape)), hu('tamk.gb', tuple(lb.shape)), hu('tamk.gf', tuple(lf.shape))
    ((out.float() * nhwc(go).float()).sum() + (attb * gb.to(DEV)).sum() + (attf * gf.to(DEV)).sum()).backward()
    ((outr * bf(go)).sum() + (lb * gb).sum() + (lf * gf).sum()).backward()
    for a, r, nm in zip(args, refs, 'q kb kf v'.split()):
        assert rel_err(nchw(a.grad), r.grad) < 1.5e-2, nm


# --------------------------------------------------------------------------------------------- GCA
@pytest.mark.parametrize('name', list(GCA_CASES))
def test_gca_module_vs_reference_golden(name):
    from models.GCA.ops import GuidedCxtAtten
    B, h, w = 2, 12, 16
    g = golden(name)
    mod = GuidedCxtAtten(128, 128)
    mod.load_state_dict({k: formula_tensor('encoder.gca.' + k, v.shape, v.dtype) for k, v in mod.state_dict().items()})
    mod.to(DEV).train()
    f = hu('gca.f', (B, 128, h, w)).to(DEV).requires_grad_(True)
    al = hu('gca.alpha', (B, 128, h, w)).to(DEV).requires_grad_(True)
    y, (_, scale) = mod(f, al, gca_unknown(GCA_CASES[name], B, h, w).to(DEV))
    ck = Checker()
    ck.rel('scale', scale, g['scale'], 1e-5)
    ck.rel('y', y, g['y'], 2e-2)
    (y * hu('gca.gy', y.shape).to(DEV)).sum().backward()
    # the guidance gradient goes through a sharply peaked softmax: rounding the guidance features to bf16
    # perturbs logits of magnitude ~50 by ~0.1, hence the looser bound (the kernel itself is pinned tightly
    # on identical bf16 inputs in test_gca_attention_kernel_tight)
    for got, key, tol in ((al.grad, 'galpha', 5e-2), (f.grad, 'gf', 2e-1), (mod.W[0].weight.grad, 'gW0', 5e-2),
                          (mod.guidance_conv.weight.grad, 'ggw', 2e-1)):
        if float(np.abs(g[key]).max()) < 1e-20:
            # all-known case: the softmax saturates to exact one-hot rows in fp32, the reference's guidance gradient is
            # exactly 0 (1e-33).  The fused backward forms sum_j P dP as <dO_i, O_i>, so dP[i][i] - delta_i cancels only
            # to fp32 summation-order noise: bound it in absolute terms against the O(1) value gradient
            assert float(got.abs().max()) <= 1e-4 * float(np.abs(g['galpha']).max()), key
            continue
        ck.rel(key, got, g[key], tol)
    ck.rel('running_mean', mod.W[1].running_mean, g['run_mean'], 2e-2)
    ck.done()


@pytest.mark.parametrize('B,h8,w8', [(2, 12, 16), (1, 20, 28), (1, 6, 16)])
def test_gca_attention_kernel_tight(B, h8, w8):
    """The attention core (patches, scores, softmax, PV, fold) against the oracle's dense formula on
    IDENTICAL bf16 guidance/value maps (so only accumulation order and the bf16 storage of P differ)."""
    from oracle.gca_net import _patch_matrix, gca_scales
    from tcvom_amd import ops
    CG, Ca = 64, 128
    g8 = hu('gcak.g', (B, CG, h8, w8)) * 0.35            # |logits| ~ 10: peaked but not degenerate
    al = hu('gcak.a', (B, Ca, h8, w8))
    unk = (hu('gcak.u', (B, 1, h8, w8)) > 0.2).float()
    gg, ag = nhwc(g8).requires_grad_(True), nhwc(al).requires_grad_(True)
    y, scales = ops.gca_attention(gg, ag, unk[:, 0].to(torch.uint8).to(DEV).contiguous())
    gr, ar = bf(g8).requires_grad_(True), bf(al).requires_grad_(True)
    g16, u16 = gr[:, :, ::2, ::2], unk[:, :, ::2, ::2]
    sc = gca_scales(u16)
    Wp = _patch_matrix(g16, 3, 1, 1, 1)
    Kh = Wp / torch.clamp(Wp.pow(2).sum(1, keepdim=True).sqrt(), min=1e-4)
    mm = (_patch_matrix(u16, 3, 1, 1, 1).mean(1) > 0).float()
    S = torch.bmm(Kh.transpose(1, 2), Wp) * (sc[:, 0:1] * mm + sc[:, 1:2] * (1 - mm)).unsqueeze(2)
    P = torch.softmax(S - 1e4 * torch.diag_embed(mm), dim=1)
    O = torch.bmm(_patch_matrix(ar, 4, 2, 1, 1), P)
    yr = F.fold(O, (h8, w8), 4, stride=2, padding=1) / 4.0
    ck = Checker()
    ck.rel('scales', scales, sc, 1e-5)
    ck.rel('y', nchw(y), yr, 1.5e-2)
    gy = hu('gcak.gy', tuple(yr.shape))
    (y.float() * nhwc(gy).float()).sum().backward()
    (yr * bf(gy)).sum().backward()
    ck.rel('dalpha', nchw(ag.grad), ar.grad, 2e-2)
    ck.rel('dg', nchw(gg.grad), gr.grad, 4e-2)
    ck.done()


# --------------------------------------------------------------------------------------------- facade
def test_preprocess_and_trimap_bit_exact():
    from tcvom_amd.facade import preprocess_window
    g = golden('facade')
    a, fg, bg = (t.to(DEV) for t in synthetic_window(2, 3, 48, 64, seed=3))
    for r in (0, 2, 5, 12, 20):
        p = preprocess_window(a, fg, bg, r, 0.0)
        tris = p.x8[..., 3:6].permute(0, 1, 4, 2, 3).float().cpu().numpy().astype(np.uint8)
        assert np.array_equal(tris, g['tris_r%d' % r]), 'one-hot trimap r=%d' % r
        assert np.array_equal(p.trimask.cpu().numpy().astype(np.uint8), g['trimask_r%d' % r]), 'trimask r=%d' % r
    assert_close(p.imgs.cpu(), g['scaled_imgs'], 1e-6, 1e-6, 'scaled_imgs')
    imgs = p.x8[..., 0:3].permute(0, 1, 4, 2, 3).float().cpu()
    assert_close(imgs, g['imgs'], 8e-3, 8e-3, 'normalised imgs (bf16)')
    p = preprocess_window(a, fg, bg, 2, 0.3)
    tris = p.x8[..., 3:6].permute(0, 1, 4, 2, 3).float().cpu().numpy().astype(np.uint8)
    assert np.array_equal(tris, g['tris_eps'])
    assert np.array_equal(p.trimask.cpu().numpy().astype(np.uint8), g['trimask_eps'])


def test_random_trimap_width_per_clip_bit_exact():
    """dilate_kernel=None (train_ddp.py's default): the reference draws one radius PER CLIP inside its loop over the batch
    (models/model.py:60-64).  Reference-generated golden (tests/golden/gen_golden.py: gen_facade_random): trimaps bit for
    bit, and torch's generator in the same state afterwards."""
    from models.model import FullModel, FullModel_VMD
    g = golden('facade_random')
    for tag, ctor in (('gca', lambda: FullModel_VMD('vmn_gca', agg_window=7)), ('gca6', lambda: FullModel_VMD('vmn_gca', agg_window=7)),
                      ('dim', lambda: FullModel('dim'))):
        B, S, H, W, seed = (int(v) for v in g[tag + '_shape'])
        fm = ctor().to(DEV)
        assert fm.DILATION_KERNEL is None
        a, fg, bg = (t.to(DEV) for t in synthetic_window(B, S, H, W, seed=5))
        torch.manual_seed(seed)
        _, _, _, _, tris, trimasks, _ = fm.preprocess(a, fg, bg)
        assert int(torch.randint(0, 2 ** 31 - 1, size=())) == int(g[tag + '_next_draw']), 'generator state after the draws (%s)' % tag
        scale = 255 if tag == 'dim' else 1
        got = (tris.cpu().numpy() * scale).round().astype(np.uint8)
        if tag == 'dim' and H16 == torch.bfloat16:
            # the 1-channel trimap carries alpha k/255 in 16-bit storage: bf16 holds it to +-1 of 255; the unknown band is exact
            assert np.abs(got.astype(int) - g[tag + '_tris'].astype(int)).max() <= 1
        else:
            assert np.array_equal(got, g[tag + '_tris']), 'trimap (%s)' % tag
        assert np.array_equal(trimasks.cpu().numpy().astype(np.uint8), g[tag + '_trimask']), 'trimask (%s)' % tag
        # the radii the façade hands to the kernels are the ones the reference drew
        torch.manual_seed(seed)
        assert fm._dilation(B) == g[tag + '_radii'].tolist()


def test_window_losses_vs_oracle():
    """_WindowLoss on synthetic predictions/logits vs oracle loss functions (fp32, tight)."""
    import oracle
    from tcvom_amd.facade import preprocess_window, _WindowLoss
    B, S, H, W, win = 2, 5, 32, 64, 7
    a, fg, bg = synthetic_window(B, S, H, W, seed=1)
    prep = preprocess_window(a.to(DEV), fg.to(DEV), bg.to(DEV), 3, 0.0)
    prep.unk8 = [prep.unk[:, s, ::8, ::8].contiguous() for s in range(S)]
    h, w = H // 8, W // 8
    preds = [torch.sigmoid(hu('wl.p%d' % c, (B, 1, H, W)) * 3) for c in range(1, S - 1)]
    attb = [hu('wl.b%d' % c, (B, win * win, h * w)) * 2 for c in range(1, S - 1)]
    attf = [hu('wl.f%d' % c, (B, win * win, h * w)) * 2 for c in range(1, S - 1)]
    dp = [t.to(DEV).requires_grad_(True) for t in preds + attb + attf]
    La, Ld, Lt, alphas, comps = _WindowLoss.apply(prep, win, 0.3, 0.2, S, *dp)
    (La + 0.5 * Ld + 0.25 * Lt).backward()
    # oracle
    sc, fgs, bgs, gts, tris, trimasks, imgs = oracle.preprocess(a, fg, bg, 3)
    rp = [t.clone().requires_grad_(True) for t in preds + attb + attf]
    ni = S - 2
    pr = [None] + rp[:ni] + [None]
    small = [None] + [trimasks[:, c, :, ::8, ::8].bool() for c in range(1, S - 1)] + [None]
    refine = [torch.where(trimasks[:, c].bool(), pr[c], gts[:, c]) for c in range(1, S - 1)]
    La_r = sum(oracle.l1_mask(refine[k], gts[:, k + 1], trimasks[:, k + 1]) for k in range(ni)) / ni
    al = torch.stack([torch.zeros_like(refine[0])] + refine + [torch.zeros_like(refine[0])], dim=1)
    Ld_r = oracle.dtssd_loss(al, gts, trimasks)
    Lt_r = oracle.attention_loss([None] + rp[ni:2 * ni] + [None], [None] + rp[2 * ni:] + [None], small, gts, win)
    (La_r + 0.5 * Ld_r + 0.25 * Lt_r).backward()
    assert_close(La.cpu(), La_r, 1e-4, 1e-6, 'L_alpha')
    assert_close(Ld.cpu(), Ld_r, 1e-4, 1e-6, 'L_dt')
    assert_close(Lt.cpu(), Lt_r, 1e-4, 1e-6, 'L_att')
    assert_close(alphas.cpu()[:, 1:-1], al[:, 1:-1].clamp(0, 1), 1e-6, 1e-6, 'alphas')
    for got, want in zip(dp, rp):
        assert_close(got.grad.cpu(), want.grad, 1e-3, 1e-9 + 1e-4 * float(want.grad.abs().max()), 'grad')


def test_fp16_overflow_guard_drops_the_step_and_halves_the_loss_scale():
    """the fp16 conversions saturate at +-65504, so an overflowing activation gradient was clipped silently.  The
    BatchNorm-backward reductions count workgroups that read a saturated gradient element (tcvom_overflow_sink); FusedAdam drops
    the step ON THE DEVICE (tcvom_adam_mt_guarded) and the host, one step later, halves the loss scale of the registered banks and
    takes the dropped step's counters back (ops.LossScaler; torch.cuda.amp.GradScaler's rule)."""
    import tcvom_amd._lib as L
    from tcvom_amd import ops
    from tcvom_amd.optim import FusedAdam
    from tcvom_amd.weights import WeightBank
    if L.DTYPE_NAME != 'fp16':
        pytest.skip('bf16 storage has fp32 range: no loss scale, no guard')
    sc = ops.SCALER
    assert sc.enabled
    saved = (sc.scale, sc.clean_steps, sc.skipped_steps)
    bank = WeightBank()
    sc.register(bank)
    try:
        sc._set_scale(65536.0)
        cnt = sc.counter(DEV)
        # (1) the reduction kernel counts saturated gradient elements, and only those
        Cn, P = 64, 4096
        x = hu('ovf.x', (1, P, Cn)).to(DEV).to(H16)
        dz = (hu('ovf.dz', (1, P, Cn)) * 100).to(DEV).to(H16)
        ss = torch.cat([torch.ones(Cn), torch.zeros(Cn)]).to(DEV)
        stat = torch.cat([torch.zeros(Cn), torch.ones(Cn)]).to(DEV)
        groups = L.call('tcvom_bn_bwd_groups', P, Cn)
        part = torch.empty((groups, 2, Cn), dtype=torch.float32, device=DEV)

        def reduce_pass(g):
            L.call('tcvom_bn_bwd_reduce', L.ptr(g), None, L.ptr(x), None, L.ptr(ss), L.ptr(stat), L.ptr(part), P, Cn, 0, 0, 1, 0, L.stream_ptr())
        cnt.zero_()
        reduce_pass(dz)
        assert cnt.tolist() == [0, 0]
        bad = dz.clone()
        bad[0, 1234, 7] = 65504.0
        bad[0, 77, 3] = -65504.0
        reduce_pass(bad)
        assert cnt.tolist()[0] >= 1
        # (2) the guarded Adam: nothing moves while the counter is non-zero; one step later the host reacts
        p = torch.nn.Parameter(torch.ones(5000, device=DEV))
        opt = FusedAdam([p], lr=0.1, weight_decay=0.01)
        p.grad = torch.full_like(p, 0.5)
        opt.step()                                                     # dropped: the counter still holds the saturation of (1)
        torch.cuda.synchronize()
        assert torch.equal(p.detach(), torch.ones_like(p)) and float(opt.state[p]['exp_avg'].abs().max()) == 0.0
        assert cnt.tolist() == [0, 0]                                  # read back and zeroed for the next step
        before = sc.skipped_steps
        opt.step()                                                     # clean step; the host now learns of the dropped one
        torch.cuda.synchronize()
        assert sc.skipped_steps == before + 1 and sc.scale == 32768.0 and bank.loss_scale == 32768.0
        assert opt.state[p]['step'] == 1                               # the dropped step does not count (bias correction of step 1)
        ref = torch.nn.Parameter(torch.ones(5000, device=DEV))
        ropt = torch.optim.Adam([ref], lr=0.1, weight_decay=0.01)
        ref.grad = torch.full_like(ref, 0.5)
        ropt.step()
        assert_close(p.detach().cpu(), ref.detach().cpu(), 1e-6, 1e-6, 'first real step = torch.optim.Adam step 1')
        # (3) the scale grows back after `growth_interval` clean steps
        gi, sc.growth_interval = sc.growth_interval, 3
        for _ in range(4):
            opt.step()
        torch.cuda.synchronize()
        sc.growth_interval = gi
        assert sc.scale == 65536.0 and bank.loss_scale == 65536.0
    finally:
        sc._set_scale(saved[0])
        sc.clean_steps, sc.skipped_steps = saved[1], saved[2]


@pytest.mark.parametrize('cin,cout,H,W,S', [(512, 2048, 72, 96, 1), (2048, 512, 68, 120, 3), (1024, 1024, 64, 80, 2), (520, 2040, 70, 61, 1)])
def test_dense_weight_gradient_on_the_kmajor_gemm(cin, cout, H, W, S):
    """Weight gradient of a 1 x 1 conv with K, C >= 256 as a split-K TT GEMM on the k-major operand path of gemm_nt256
    (gemm256.hip: gemm_tt256_try_launch; every tile split over the pixel reduction, fp32 atomics) against an fp32 matmul of the
    same 16-bit operands: ragged channel counts (masked 256-tiles), a pixel count that is not a multiple of 64, batched calls."""
    import ctypes as C
    from tcvom_amd import _lib as L
    from tcvom_amd.conv_plan import ConvGeometry
    from tcvom_amd.ops import _phase_array
    tag = 'tt%d_%d_%d' % (cin, cout, H)
    bank, spec = _mini_bank(cin, cout, 1, 1, 0, False, spectral=False, tag=tag)
    geo = ConvGeometry(spec, 1, H, W)
    arr = _phase_array(geo.wgrad)
    assert L._FNS['tcvom_wgrad_igemm_variant'](C.byref(arr[0])).decode() == 'gemm_tt256'
    xs = [hu('x%d.%s' % (i, tag), (1, H, W, cin)).to(DEV).to(H16) for i in range(S)]
    dys = [hu('dy%d.%s' % (i, tag), (1, H, W, cout)).to(DEV).to(H16) for i in range(S)]
    dw = torch.zeros(S, cout * cin, device=DEV)
    vp = lambda ts: C.cast((C.c_void_p * S)(*[t.data_ptr() for t in ts]), C.c_void_p)
    L.call('tcvom_wgrad_igemm_batched', vp(dys), vp(xs), vp([dw[i] for i in range(S)]), S, arr, len(geo.wgrad), cout, L.stream_ptr())
    torch.cuda.synchronize()
    refs = [dys[i].float().reshape(-1, cout).t() @ xs[i].float().reshape(-1, cin) for i in range(S)]           # [K][C]
    for i in range(S):
        assert rel_err(dw[i].reshape(cout, cin).cpu(), refs[i].cpu()) < 1e-4, i
    if S > 1:
        # (a) the problems at UNIFORM strides (frames of one tensor) go out as one launch; (b) ... into ONE shared gradient (a layer
        # whose weight does not change between the frames)
        xcat, dycat = torch.stack(xs), torch.stack(dys)
        dwu = torch.zeros(S, cout * cin, device=DEV)
        L.call('tcvom_wgrad_igemm_batched', vp([dycat[i] for i in range(S)]), vp([xcat[i] for i in range(S)]), vp([dwu[i] for i in range(S)]),
               S, arr, len(geo.wgrad), cout, L.stream_ptr())
        one = torch.zeros(cout * cin, device=DEV)
        L.call('tcvom_wgrad_igemm_batched', vp([dycat[i] for i in range(S)]), vp([xcat[i] for i in range(S)]), vp([one] * S),
               S, arr, len(geo.wgrad), cout, L.stream_ptr())
        torch.cuda.synchronize()
        assert rel_err(dwu.cpu(), dw.cpu()) < 1e-5
        assert rel_err(one.reshape(cout, cin).cpu(), sum(refs).cpu()) < 1e-4
