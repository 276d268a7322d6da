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


def _bf(t):
    return t.to(torch.bfloat16).float()


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
