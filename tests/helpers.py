"""Shared test helpers: hash-generated inputs (same formulas as tests/golden/gen_golden.py)."""
import os
import numpy as np
import torch

from tcvom_amd.synthetic import hash_uniform, formula_tensor

def tol(bf16, fp16):
    """A bound that depends on the 16-bit storage type of the loaded build (TCVOM_DTYPE): fp16 stores 3 more mantissa bits than
    bf16, its measured errors are ~25x smaller and the bounds follow (each ~4x above the measured value)."""
    from tcvom_amd._lib import DTYPE_NAME
    return fp16 if DTYPE_NAME == 'fp16' else bf16


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def hu(tag, shape, scale=1.0):
    return torch.from_numpy(hash_uniform(tag, int(np.prod(shape))).reshape(shape)).float() * scale


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def tam_mask(kind, B, H, W):
    m = torch.zeros(B, 1, H * 8, W * 8)
    if kind == 'random':
        small = (hu('tam.mask', (B, 1, H, W)) > 0.1).float()
    elif kind == 'full':
        small = torch.ones(B, 1, H, W)
    elif kind == 'single':
        small = torch.zeros(B, 1, H, W)
        small[0, 0, 4, 7] = 1
    else:
        small = torch.zeros(B, 1, H, W)
    m[:, :, ::8, ::8] = small
    m[:, :, 1::8, 3::8] = 1 - small
    return m


def gca_unknown(kind, B, h, w):
    if kind == 'random':
        return (hu('gca.unknown', (B, 1, h, w)) > 0.3).float()
    return torch.zeros(B, 1, h, w) if kind == 'zeros' else torch.ones(B, 1, h, w)


TAM_CASES = {
    'tam_w7_random': (2, 16, 9, 11, 7, 'random'),
    'tam_w1_random': (2, 16, 9, 11, 1, 'random'),
    'tam_w7_empty': (1, 16, 9, 11, 7, 'empty'),
    'tam_w7_single': (1, 16, 9, 11, 7, 'single'),
    'tam_w7_full_c128': (1, 128, 6, 8, 7, 'full'),
}
GCA_CASES = {'gca_random': 'random', 'gca_all_known': 'zeros', 'gca_all_unknown': 'ones'}
WINDOW_CASES = {
    'window_s3_64x64': (2, 3, 64, 64, 3, 7),
    'window_s5_64x96': (1, 5, 64, 96, 4, 7),
    'window_s3_128x160': (1, 3, 128, 160, 12, 7),
    'window_s3_64x96_w5': (1, 3, 64, 96, 4, 5),      # agg_window 5 at model level (models/VMN/VMN_model.py:10-16)
}
FULL_GRADS = ('decoder.fam.key_conv.bias', 'decoder.fam.query_conv.bias', 'encoder.bn1.weight',
              'decoder.conv2.weight', 'encoder.gca.W.1.weight', 'decoder.layer3.0.bn1.bias')


def assert_close(got, want, rtol, atol, what=''):
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    want = torch.as_tensor(np.asarray(want.detach().cpu() if torch.is_tensor(want) else want)).double()
    assert got.shape == want.shape, '%s: shape %s vs %s' % (what, tuple(got.shape), tuple(want.shape))
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bool(bad.any()), '%s: %d/%d off, max err %.3e (max |want| %.3e)' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(want.abs().max()))


def vmn_gca_template():
    """key -> (shape, dtype) of FullModel_VMD('vmn_gca').NET.state_dict() — from the product's own module."""
    from tcvom_amd.vmn import build_vmn_gca
    net = build_vmn_gca(agg_window=7)
    return net.state_dict()


class Checker(object):
    """Collects named max-relative-error checks so that one run reports every quantity."""

    def __init__(self):
        self.rows = []

    def rel(self, name, got, want, tol):
        got = torch.as_tensor(np.asarray(got.detach().float().cpu() if torch.is_tensor(got) else got)).double()
        want = torch.as_tensor(np.asarray(want.detach().float().cpu() if torch.is_tensor(want) else want)).double()
        assert got.shape == want.shape, '%s: shape %s vs %s' % (name, tuple(got.shape), tuple(want.shape))
        err = float((got - want).abs().max() / (want.abs().max() + 1e-12))
        self.rows.append((name, err, tol, err <= tol))

    def l2(self, name, got, want, tol):
        """relative L2 error: for element-wise quantities behind an activation mask, where a single rounding-induced
        mask flip is a large LOCAL error but a negligible part of the tensor"""
        got = torch.as_tensor(np.asarray(got.detach().float().cpu() if torch.is_tensor(got) else got)).double()
        want = torch.as_tensor(np.asarray(want.detach().float().cpu() if torch.is_tensor(want) else want)).double()
        assert got.shape == want.shape, '%s: shape %s vs %s' % (name, tuple(got.shape), tuple(want.shape))
        err = float((got - want).norm() / (want.norm() + 1e-12))
        self.rows.append((name + '(L2)', err, tol, err <= tol))

    def done(self):
        msg = ', '.join('%s=%.2e%s' % (n, e, '' if ok else ' (> %.1e!)' % t) for n, e, t, ok in self.rows)
        print(msg)
        assert all(ok for _, _, _, ok in self.rows), msg


def dim_template():
    """key -> tensor (shape / dtype only) of FullModel('dim').NET.state_dict() — from the product's own module
    (its key list is checked against the reference in tests/test_oracle_golden.py::test_dim_state_dict_layout)."""
    from tcvom_amd.dim_net import DIM_VGG
    return DIM_VGG().state_dict()


FBA_CASES = {
    # name: (B, S, H, W, dilate_kernel)  -- tests/golden/gen_golden.py:FBA_CASES
    'fba_s3_64x64': (1, 3, 64, 64, 3),
    'fba_s5_64x96': (1, 5, 64, 96, 5),
}
FBA_FULL_GRADS = ('decoder.conv_up4.4.weight', 'decoder.conv_up4.4.bias', 'decoder.fam.query_conv.bias', 'encoder.bn1.weight',
                  'decoder.conv_up3.1.weight', 'decoder.ppm.0.1.bias')


def fba_formula_state(requires_grad=True):
    """Formula-initialised state of FullModel_VMD('vmn_fba').NET: keys / shapes from the reference's own state_dict
    (tests/golden/fba_state_keys.npz)."""
    from tcvom_amd.synthetic import formula_tensor
    g = golden('fba_state_keys')
    state = {}
    for k, shp in zip(g['keys'], g['shapes']):
        shape = tuple(int(d) for d in str(shp).split(',')) if str(shp) else ()
        state[str(k)] = formula_tensor(str(k), shape).requires_grad_(requires_grad)
    return state


VMN_DIM_CASES = {'vmn_dim_s3_64x64': (1, 3, 64, 64, 3), 'vmn_dim_s5_64x96': (1, 5, 64, 96, 5)}
VMN_DIM_FULL_GRADS = ('encoder.conv11.weight', 'encoder.bn33.weight', 'decoder.dconv1.bias', 'decoder.alpha_pred.weight',
                      'decoder.fam.key_conv.bias')


VMN_INDEX_CASES = {'vmn_index_s3_64x96': (2, 3, 64, 96, 3), 'vmn_index_s3_128x128': (2, 3, 128, 128, 5)}
VMN_INDEX_FULL_GRADS = ('encoder.layer0.0.weight', 'encoder.layer3.1.conv.3.weight', 'encoder.index2.indexnet3.3.weight',
                        'encoder.dconv_pp.aspp3.atrous_conv.0.weight', 'decoder.decoder_layer2.dconv.0.weight',
                        'decoder.pred.1.weight', 'decoder.fam.key_conv.bias')


def golden_formula_state(name, requires_grad=True):
    """Formula-initialised state from a key / shape list captured from the reference (tests/golden/<name>.npz)."""
    from tcvom_amd.synthetic import formula_tensor
    g = golden(name)
    state = {}
    for k, shp in zip(g['keys'], g['shapes']):
        shape = tuple(int(d) for d in str(shp).split(',')) if str(shp) else ()
        t = formula_tensor(str(k), shape)
        buf = str(k).rsplit('.', 1)[-1] in ('running_mean', 'running_var', 'num_batches_tracked')
        state[str(k)] = t.requires_grad_(requires_grad) if (t.dtype.is_floating_point and not buf) else t
    return state


def metric_inputs(H=48, W=64):
    """tests/golden/gen_golden.py:metric_inputs."""
    a = (hu('metric.a', (H, W)) * 0.5 + 0.5).numpy().astype(np.float32)
    g = np.clip(a + hu('metric.g', (H, W)).numpy() * 0.1, 0, 1).astype(np.float32)
    ha = np.clip(a + hu('metric.ha', (H, W)).numpy() * 0.2, 0, 1).astype(np.float32)
    hg = np.clip(g + hu('metric.hg', (H, W)).numpy() * 0.2, 0, 1).astype(np.float32)
    u = hu('metric.tri', (H, W)).numpy()
    tri = np.where(u < -0.3, 0, np.where(u > 0.4, 255, 128)).astype(np.uint8)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    flow = np.stack([3.0 * np.sin(ys / 7.0) + 2.5, -2.0 * np.cos(xs / 9.0) - 1.25], -1).astype(np.float32)
    flow[5:12, 20:31] = np.nan
    flow[0, :] = np.array([-4.5, -3.25], dtype=np.float32)
    return a, g, tri, ha, hg, flow
