"""Why the 64-pixel-high reference goldens are compared at 1e-3 and not at the north-star 1e-4:
a pipeline that STORES activations and weights in bf16 has an intrinsic alpha error that the tiny windows amplify (8..24
element BatchNorms at os32).  Emulating bf16 storage inside the fp32 CPU oracle -- rounding weights, conv outputs,
activations and BatchNorm outputs where a bf16 pipeline stores them -- gives that floor without any GPU code involved:
it is above 1e-4 at 64 x 64 and falls with the window size, which is why tests/test_gpu_window.py asserts 1e-4 from
128 x 160 upwards (and at 544 x 960 / 1088 x 1920) and 1e-3 on the two smallest goldens."""
import torch
import torch.nn.functional as TF

import oracle
import oracle.gca_net as G
import oracle.tam as T
from oracle.state_spec import vmn_gca_state_spec
from tcvom_amd.synthetic import formula_tensor, synthetic_window


STORAGE = [torch.bfloat16]          # the 16-bit type the emulated pipeline stores in


def _bf(t):
    return t.to(STORAGE[0]).float()


class _Bf16Storage(object):
    """torch.nn.functional with bf16 rounding at the points where a bf16 pipeline stores a tensor."""

    def __init__(self, on):
        self.on = on

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv2d(self, x, w, b=None, *a, **k):
        y = TF.conv2d(x, _bf(w) if self.on else w, b, *a, **k)
        return _bf(y) if self.on else y

    def conv_transpose2d(self, x, w, b=None, *a, **k):
        y = TF.conv_transpose2d(x, _bf(w) if self.on else w, b, *a, **k)
        return _bf(y) if self.on else y

    def relu(self, x):
        y = TF.relu(x)
        return _bf(y) if self.on else y

    def leaky_relu(self, x, a):
        y = TF.leaky_relu(x, a)
        return _bf(y) if self.on else y

    def batch_norm(self, *a, **k):
        y = TF.batch_norm(*a, **k)
        return _bf(y) if self.on else y

    def relu6(self, x):
        y = TF.relu6(x)
        return _bf(y) if self.on else y


def _alpha(B, S, H, W, dil, storage):
    old = (G.F, T.F)
    G.F = T.F = _Bf16Storage(storage)
    try:
        state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
                 for k, s in vmn_gca_state_spec().items()}
        a, fg, bg = synthetic_window(B, S, H, W, seed=0)
        with torch.no_grad():
            out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=dil, training=True)
        return out[7], out[6].isclose(torch.tensor(128.0 / 255.0))
    finally:
        G.F, T.F = old


def test_bf16_storage_noise_floor_shrinks_with_window_size():
    torch.set_num_threads(8)
    floors = {}
    for name, (B, S, H, W, dil) in {'64x64': (2, 3, 64, 64, 3), '128x160': (1, 3, 128, 160, 12)}.items():
        ref, um = _alpha(B, S, H, W, dil, False)
        emu, _ = _alpha(B, S, H, W, dil, True)
        floors[name] = float(((emu - ref)[um] ** 2).mean())
    print('unknown-pixel alpha MSE of an all-bf16-storage pipeline vs fp32:', floors)
    assert floors['64x64'] > 1e-4, 'the 64 x 64 golden cannot be matched to 1e-4 by any bf16-storage pipeline'
    assert floors['128x160'] < floors['64x64']


def _alpha_index(H, W, storage):
    import oracle.index_net as I
    from helpers import golden_formula_state
    old = (I.F, G.F, T.F)
    I.F = G.F = T.F = _Bf16Storage(storage)
    try:
        state = golden_formula_state('vmn_index_state_keys', requires_grad=False)
        a, fg, bg = synthetic_window(2, 3, H, W, seed=11)
        with torch.no_grad():
            out, _ = I.vmn_index_window_forward(state, a, fg, bg, window=7, dilate_kernel=8, training=True)
        return out[7][:, 1:2], (out[6][:, 1:2] == 128.0 / 255.0)
    finally:
        I.F, G.F, T.F = old


def test_16_bit_storage_noise_floors_bf16_vs_fp16():
    """What 16-bit activation / weight STORAGE costs, measured without any GPU code by rounding inside the fp32 oracle at the
    points where the HIP pipeline stores a tensor (unknown-pixel alpha MSE, calc_metric.py:25):
      * IndexNet + TAM at 256 x 320 (the size tests/test_gpu_index.py compares at): bf16 storage alone costs ~5e-4 -- the
        HIP path's measured 5.9e-4 sits AT that floor, no kernel change can bring a bf16 pipeline to the north-star 1e-4 there
        (a 100-layer ReLU6 / train-mode BatchNorm net with random weights amplifies the 2^-9 rounding) -- which is what the
        1e-3 bound of the IndexNet tests rests on;
      * fp16 storage (3 more mantissa bits, same MFMA rate) lowers both floors ~35x: IndexNet 1.5e-5, GCA 3.5e-6 (bf16:
        1.5e-4, and 8.5e-5 with the doubled-tap high-precision stem the bf16 engine needs to get under 1e-4)."""
    torch.set_num_threads(8)
    floors = {}
    try:
        ref, um = _alpha_index(256, 320, False)
        gref, gum = _alpha(1, 3, 256, 320, 12, False)
        for dt in (torch.bfloat16, torch.float16):
            STORAGE[0] = dt
            emu, _ = _alpha_index(256, 320, True)
            floors['index', dt] = float(((emu - ref)[um] ** 2).mean())
            gemu, _ = _alpha(1, 3, 256, 320, 12, True)
            floors['gca', dt] = float(((gemu - gref)[gum] ** 2).mean())
    finally:
        STORAGE[0] = torch.bfloat16
    print('unknown-pixel alpha MSE of an all-16-bit-storage pipeline vs fp32 at 256x320:', floors)
    assert floors['index', torch.bfloat16] > 3e-4, 'the 1e-3 bound of the IndexNet window tests rests on this floor'
    assert floors['gca', torch.bfloat16] > 1e-4, 'plain bf16 storage misses the north-star bound: hence the high-precision stem'
    assert floors['index', torch.float16] < 5e-5 and floors['gca', torch.float16] < 1e-5
