"""GPU parity tests of the DIM base (BASELINE.json config 1) on the HIP path: pooling / unpooling / loss kernels vs
PyTorch, and FullModel('dim') vs vectors captured from the reference (tests/golden/gen_golden.py:gen_dim)."""
import numpy as np
import pytest
import torch

from tcvom_amd._lib import ACT_DTYPE as H16      # the 16-bit storage type of the loaded build (bf16 / fp16)
import torch.nn.functional as F

from helpers import hu, golden, assert_close, Checker
from tcvom_amd.synthetic import formula_tensor, synthetic_window

pytestmark = pytest.mark.gpu
DEV = 'cuda'
DIM_CASES = {'dim_s1_64x64': (2, 1, 64, 64, 3), 'dim_s3_96x128': (1, 3, 96, 128, 5), 'dim_s1_512x512': (1, 1, 512, 512, 12)}
DIM_FULL_GRADS = ('conv11.weight', 'bn33.weight', 'dconv1.bias', 'alpha_pred.weight', 'dconv6.bias')


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(H16).to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2).float().cpu()


def test_maxpool_indices_and_unpool_bit_exact():
    """MaxPool2d(2, return_indices) / MaxUnpool2d(2) and both gradients: selections are exact operations on bf16 values,
    so everything must be bit-identical to PyTorch on the same bf16 data (ties broken towards the first position)."""
    from tcvom_amd import ops
    x = (hu('dim.pool.x', (2, 16, 12, 20)) * 4).round() / 4            # coarse values: plenty of exact ties
    xb = x.to(H16).float()
    xg = nhwc(x).requires_grad_(True)
    y, idx = ops.maxpool2_idx(xg)
    yr, ir = F.max_pool2d(xb, 2, 2, return_indices=True)
    assert torch.equal(nchw(y), yr)
    W = 20
    pos = ((ir // W) % 2) * 2 + (ir % W) % 2                            # flat input index -> position inside the window
    assert torch.equal(idx.permute(0, 3, 1, 2).cpu().long(), pos)
    gy = hu('dim.pool.gy', tuple(yr.shape)).to(H16).float()
    y.backward(nhwc(gy))
    xr = xb.clone().requires_grad_(True)
    F.max_pool2d(xr, 2, 2).backward(gy)
    assert torch.equal(nchw(xg.grad), xr.grad)
    z = hu('dim.unpool.z', tuple(yr.shape)).to(H16).float()
    zg = nhwc(z).requires_grad_(True)
    up = ops.unpool2(zg, idx)
    zr = z.clone().requires_grad_(True)
    upr = F.max_unpool2d(zr, ir, 2, 2)
    assert torch.equal(nchw(up), upr)
    gu = hu('dim.unpool.g', tuple(upr.shape)).to(H16).float()
    up.backward(nhwc(gu))
    upr.backward(gu)
    assert torch.equal(nchw(zg.grad), zr.grad)


def test_head_conv_5x5_clamp():
    from tcvom_amd import ops
    Cc, N, H, W = 64, 2, 12, 20
    x = hu('dim.head.x', (N, Cc, H, W))                  # hu: uniform in [-1, 1)
    w = hu('dim.head.w', (1, Cc, 5, 5)) * 0.04
    b = torch.tensor([0.45])
    xg = nhwc(x).requires_grad_(True)
    wg, bg_ = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    a = ops.head_conv(xg, wg, bg_, 5, 1)
    xr = x.to(H16).float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ar = F.conv2d(xr, wr, br, 1, 2).clamp(0, 1)
    assert 0.05 < float(((ar > 0) & (ar < 1)).float().mean()) < 0.95          # both clamp branches are exercised
    ck = Checker()
    ck.rel('alpha', a.cpu(), ar, 1e-4)
    g = hu('dim.head.g', tuple(ar.shape))
    (a * g.to(DEV)).sum().backward()
    (ar * g).sum().backward()
    ck.rel('dx', nchw(xg.grad), xr.grad, 2e-2)
    ck.rel('dw', wg.grad.cpu(), wr.grad, 1e-3)
    ck.rel('db', bg_.grad.cpu(), br.grad, 1e-3)
    ck.done()


def test_conv6_as_unfold_dense():
    """7x7 conv + bias + ReLU through unfold -> dense GEMM, forward and all three gradients."""
    import torch.nn as nn
    from tcvom_amd import ops
    from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
    cin, cout, N, H, W = 64, 128, 1, 6, 10
    conv = nn.Conv2d(cin, cout, 7, padding=3).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(formula_tensor('dim.c6.weight', conv.weight.shape))
        conv.bias.copy_(formula_tensor('dim.c6.bias', conv.bias.shape))
    bank = WeightBank()
    spec = ConvSpec('c6', conv.weight, None, None, conv.bias, False, 1, 3, 'frame')
    bank.register(spec)
    cfg = ops.ConvCfg(bank, spec, pre_relu=True)
    x = hu('dim.c6.x', (N, cin, H, W))
    xg = nhwc(x).requires_grad_(True)
    token = bank_token(bank, 1, True)
    y = ops.conv_unfold_dense(cfg, xg, token)
    xr = x.to(H16).float().requires_grad_(True)
    wr = conv.weight.detach().cpu().to(H16).float().requires_grad_(True)
    br = conv.bias.detach().cpu().clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, wr, br, 1, 3))
    ck = Checker()
    ck.rel('y', nchw(y), yr, 1e-2)
    g = hu('dim.c6.g', tuple(yr.shape))
    (y.float() * nhwc(g).float()).sum().backward()
    (yr * g.to(H16).float()).sum().backward()
    ck.rel('dx', nchw(xg.grad), xr.grad, 3e-2)
    ck.rel('dw', conv.weight.grad.cpu(), wr.grad, 2e-2)
    ck.rel('db', conv.bias.grad.cpu(), br.grad, 2e-2)
    ck.done()


def test_dim_losses_vs_oracle():
    """L_alpha / L_comp / L_grad of FullModel.single_image_loss and their gradient w.r.t. the prediction."""
    from oracle import dim_net
    from oracle.window import l1_mask
    from tcvom_amd.facade import _SingleImageLoss, preprocess_window
    B, S, H, W, dil = 2, 3, 32, 64, 3
    a, fg, bg = synthetic_window(B, S, H, W, seed=4)
    c = S // 2
    pred = hu('dim.loss.pred', (B, 1, H, W)) * 0.6 + 0.5
    prep = preprocess_window(a.to(DEV), fg.to(DEV), bg.to(DEV), dil, 0.0, 1)
    pg = pred.to(DEV).requires_grad_(True)
    la, lc, lg, alphas, comps = _SingleImageLoss.apply(prep, c, S, pg)
    (la + 0.7 * lc + 1.3 * lg).backward()
    gts = a / 255.0
    fgs, bgs = fg.flip([2]) / 255.0, bg.flip([2]) / 255.0
    simgs = fgs * gts + bgs * (1 - gts)
    tris, tm = dim_net.make_trimap1(gts, dil)
    pr = pred.clone().requires_grad_(True)
    m = tm[:, c].float()
    refine = torch.where(m.bool(), pr, gts[:, c])
    comp = fgs[:, c] * refine + bgs[:, c] * (1 - refine)
    ra, rc, rg = l1_mask(refine, gts[:, c], m), l1_mask(comp, simgs[:, c], m), dim_net.l1_grad(refine, gts[:, c], m)
    (ra + 0.7 * rc + 1.3 * rg).backward()
    assert_close(torch.stack([la, lc, lg]).detach().cpu(), torch.stack([ra, rc, rg]).detach(), 1e-4, 1e-6, 'losses')
    assert_close(pg.grad.cpu(), pr.grad, 1e-3, 1e-7, 'dpred')
    assert_close(alphas[:, c].cpu(), refine.detach().clamp(0, 1), 1e-6, 1e-6, 'alphas')
    assert_close(comps[:, c].cpu(), comp.detach().clamp(0, 1), 1e-5, 1e-5, 'comps')
    assert_close(prep.x8[..., 3].float().cpu(), tris[:, :, 0].to(H16).float(), 0, 0, 'trimap channel')
    assert float(alphas[:, 0].abs().max()) == 0.0 and float(comps[:, -1].abs().max()) == 0.0


def _dim_model(dil):
    from models.model import FullModel
    m = FullModel('dim', dilate_kernel=dil)
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
    return m.to(DEV)


@pytest.mark.parametrize('name', list(DIM_CASES))
def test_dim_full_model_vs_reference_golden(name):
    B, S, H, W, dil = DIM_CASES[name]
    g = golden(name)
    m = _dim_model(dil).train()
    a, fg, bg = (t.to(DEV) for t in synthetic_window(B, S, H, W, seed=1))
    out = m(a, fg, bg)
    (out[0] + out[1] + out[2]).backward()
    torch.cuda.synchronize()
    losses = torch.stack([o.detach().float().cpu() for o in out[:3]])
    alphas = out[5].float().cpu()
    if H > 128:
        ref = torch.from_numpy(g['alphas'])
        got = F.avg_pool2d(alphas[:, S // 2], 8)
    else:
        ref, got = torch.from_numpy(g['alphas']), alphas
    mse = float(((got - ref) ** 2).mean())
    print('%s: alpha MSE %.3e, losses %s vs %s' % (name, mse, losses.tolist(), g['losses'].tolist()))
    assert mse <= 1e-4, 'alpha MSE vs the reference'
    assert_close(losses, g['losses'], 2e-2, 1e-3, 'losses')
    assert_close(out[4].sum().cpu(), g['tris_sum'], 1e-5, 1e-2, 'trimap')
    assert_close(out[6].sum().cpu(), g['comps_sum'], 1e-2, 1.0, 'comps')
    params = dict(m.NET.named_parameters())
    names = [str(n) for n in g['grad_names']]
    got_n = np.array([float(params[n].grad.double().norm()) for n in names])
    ratio = got_n / np.maximum(g['grad_norms'], 1e-12)
    big = g['grad_norms'] > 1e-3 * g['grad_norms'].max()
    print('   grad-norm ratio over %d significant tensors: min %.3f max %.3f' % (big.sum(), ratio[big].min(), ratio[big].max()))
    # per-tensor norms of a train-mode-BatchNorm backward move by several % between identical runs (atomic summation
    # order, amplified): the bulk must agree, single tensors get margin
    assert abs(np.median(ratio[big]) - 1) < 0.1 and 0.7 < ratio[big].min() and ratio[big].max() < 1.5
    sd = m.NET.state_dict()
    assert_close(sd['bn11.running_mean'].cpu(), g['state:bn11.running_mean'], 2e-2, 1e-3, 'bn11.running_mean')
    assert int(sd['bn11.num_batches_tracked']) == int(g['state:bn11.num_batches_tracked'])


# --------------------------------------------------------------------------------------------- vmn_dim (DIM base + TAM)
@pytest.mark.parametrize('name', ['vmn_dim_s3_64x64', 'vmn_dim_s5_64x96'])
def test_vmn_dim_window_vs_reference_golden(name):
    """FullModel_VMD('vmn_dim') (models/VMN/VMN_DIM.py): 5 losses (L_alpha, L_comp, L_grad, L_dt, L_att), alphas, gradient
    norms and BatchNorm state of one training window against the reference."""
    from helpers import VMN_DIM_CASES
    from tcvom_amd.facade import FullModel_VMD
    B, S, H, W, dil = VMN_DIM_CASES[name]
    g = golden(name)
    keys = golden('vmn_dim_state_keys')
    m = FullModel_VMD('vmn_dim', agg_window=7, dilate_kernel=dil)
    sd = m.NET.state_dict()
    assert list(sd.keys()) == [str(k) for k in keys['keys']]
    assert [','.join(str(int(d)) for d in v.shape) for v in sd.values()] == [str(s) for s in keys['shapes']]
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in sd.items()})
    m = m.to(DEV).train()
    a, fg, bg = (t.to(DEV) for t in synthetic_window(B, S, H, W, seed=4))
    out = m(a, fg, bg)
    (out[0] + out[1] + out[2] + 0.5 * out[3] + 0.25 * out[4]).backward()
    torch.cuda.synchronize()
    losses = torch.stack([o.detach().float().cpu() for o in out[:5]])
    mse = float(((out[7].float().cpu() - torch.from_numpy(g['alphas'])) ** 2).mean())
    print('%s: alpha MSE %.3e, losses %s vs %s' % (name, mse, losses.tolist(), g['losses'].tolist()))
    assert mse <= 1e-4, 'alpha MSE vs the reference'
    assert_close(losses, g['losses'], 2e-2, 1e-3, 'losses')
    assert_close(out[8].double().sum().cpu(), g['comps_sum'], 1e-2, 1.0, 'comps')
    params = dict(m.NET.named_parameters())
    names = [str(n) for n in g['grad_names']]
    assert all(params[n].grad is not None for n in names)
    got_n = np.array([float(params[n].grad.double().norm()) for n in names])
    ratio = got_n / np.maximum(g['grad_norms'], 1e-12)
    big = g['grad_norms'] > 1e-3 * g['grad_norms'].max()
    print('   grad-norm ratio over %d significant tensors: min %.3f max %.3f' % (big.sum(), ratio[big].min(), ratio[big].max()))
    # per-tensor norms of a train-mode-BatchNorm backward move by several % between identical runs (atomic summation
    # order, amplified): the bulk must agree, single tensors get margin
    assert abs(np.median(ratio[big]) - 1) < 0.1 and 0.7 < ratio[big].min() and ratio[big].max() < 1.5
    post = m.NET.state_dict()
    assert_close(post['encoder.bn11.running_mean'].cpu(), g['state:encoder.bn11.running_mean'], 2e-2, 1e-3, 'bn11.running_mean')
    assert int(post['encoder.bn11.num_batches_tracked']) == int(g['state:encoder.bn11.num_batches_tracked'])
