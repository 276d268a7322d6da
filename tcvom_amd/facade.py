"""Model façade: FullModel / FullModel_VMD / EvalModel with the reference's constructor and
forward() signatures (models/model.py:15-453), running on libtcvom_hip.so.

    FullModel_VMD(model='vmn_gca', agg_window=7, dilate_kernel=None, eps=0, att_thres=0.3, label_smooth=0.2)
    .forward(a[B,S,1,H,W], fg[B,S,3,H,W], bg[B,S,3,H,W])      float32 0..255, BGR, H % 32 == W % 32 == 0
        -> [L_alpha, L_comp, L_grad, L_dt, L_att, scaled_imgs, tris_vis, alphas, comps, scaled_gts, Fs, Bs]
    .NET                                                      state_dict == the reference checkpoint layout
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from ._lib import ACT_DTYPE as H16
from . import vmn as VMN
from .dim_net import DIM_VGG
from .index_net import IndexMatting
from .fba_net import FBA
from .gca_net import GCA

TAM_OS = 8


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


class _Prep(object):
    """Outputs of the fused preprocessing kernels for one window."""
    pass


def preprocess_window(a, fg, bg, dilate_kernel, eps, tri_channels=3):
    """FullModel.preprocess + make_trimap (models/model.py:54-92) for the 3-channel one-hot trimap."""
    if not a.is_cuda:
        raise RuntimeError('tcvom_amd runs on the GPU through libtcvom_hip.so only (no CPU fallback)')
    B, S, _, H, W = a.shape
    dev = a.device
    a, fg = a.float().contiguous(), fg.float().contiguous()
    bg = bg.float().contiguous() if bg is not None else None      # None: EvalModel (a = user trimap, fg = frame)
    p = _Prep()
    p.gts = _f32((B, S, 1, H, W), dev)
    p.fgs, p.imgs = _f32((B, S, 3, H, W), dev), _f32((B, S, 3, H, W), dev)
    p.bgs = _f32((B, S, 3, H, W), dev) if bg is not None else None
    u8 = lambda: torch.empty((B, S, H, W), dtype=torch.uint8, device=dev)
    p.unk_raw, tmp, p.unk = u8(), u8(), u8()
    p.x8 = torch.empty((B, S, H, W, 8), dtype=H16, device=dev)
    # bf16 build, GCA input: the same tensor a second time as IEEE fp16 -- the first conv of the encoder's fp16 island reads it
    # (gca_net.py: F16_ISLAND; ops.f16_twin)
    p.x8_f16 = torch.empty((B, S, H, W, 8), dtype=torch.float16, device=dev) if (ops.F16_ISLAND and tri_channels == 3) else None
    p.trimask = _f32((B, S, 1, H, W), dev)
    p.tris_vis = _f32((B, S, 1, H, W), dev)
    # one radius per clip (models/model.py:60-64): an int serves every clip, a sequence names them in clip order
    radii = [int(r) for r in dilate_kernel] if hasattr(dilate_kernel, '__len__') else [int(dilate_kernel)] * B
    if len(radii) != B:
        raise ValueError('preprocess_window: %d dilation radii for %d clips' % (len(radii), B))
    p.radii = radii
    L.call('tcvom_preprocess_clips_f16', L.ptr(a), L.ptr(fg), L.ptr(bg), L.ptr(p.gts), L.ptr(p.fgs), L.ptr(p.bgs), L.ptr(p.imgs),
           L.ptr(p.unk_raw), L.ptr(tmp), L.ptr(p.unk), L.ptr(p.x8), L.ptr(p.x8_f16), L.ptr(p.trimask), L.ptr(p.tris_vis), B, S, H, W,
           (ctypes.c_int32 * B)(*radii), float(eps), int(tri_channels), L.stream_ptr())
    return p


def x8_frame(prep, s):
    """Frame s of the network input [B, H, W, 8] (a view for B = 1), with its IEEE fp16 twin attached where the window has one."""
    t = prep.x8[:, s].contiguous()
    if getattr(prep, 'x8_f16', None) is not None:
        ops.set_f16_twin(t, prep.x8_f16[:, s].contiguous())
    return t


def unk8_frames(prep, S):
    """The unknown mask at the TAM's stride (VMN_model.py:22: nearest, every 8th pixel) per frame, [B, h, w] uint8 each: ONE strided copy
    into a frame-major buffer (the frames are consecutive slices of it: vmn._stack_frames hands the batched path a view)."""
    u = prep.unk.transpose(0, 1)[:, :, ::TAM_OS, ::TAM_OS].contiguous()          # [S, B, h, w]
    return [u[s] for s in range(S)]


class _WindowLoss(torch.autograd.Function):
    """L_alpha (masked L1 on interior frames), L_dt (temporal, S >= 5) and L_att (attention BCE), plus the
    visualisation tensors `alphas` / `comps` — models/model.py:94-127,285-345."""

    @staticmethod
    def forward(ctx, prep, window, att_thres, label_smooth, S, *tensors):
        ctx.set_materialize_grads(False)         # no zero tensors for the visualisation outputs / unused losses in backward
        ni = S - 2
        preds, attb, attf = tensors[:ni], tensors[ni:2 * ni], tensors[2 * ni:3 * ni]
        dev = preds[0].device
        st = L.stream_ptr()
        B, _, H, W = preds[0].shape
        HW = H * W
        gts, tm = prep.gts, prep.trimask
        alphas = torch.zeros((B, S, 1, H, W), dtype=torch.float32, device=dev)
        comps = torch.zeros((B, S, 3, H, W), dtype=torch.float32, device=dev)
        losses = torch.zeros(3, dtype=torch.float32, device=dev)          # L_alpha, L_dt, L_att
        acc = torch.zeros((3, max(S, 1), 2), dtype=torch.float32, device=dev)
        preds = [p.contiguous() for p in preds]
        for k in range(ni):
            c = k + 1
            L.call('tcvom_masked_l1_fwd', L.ptr(preds[k]), L.ptr(gts[:, c]), L.ptr(tm[:, c]), None, None, None,
                   L.ptr(prep.fgs[:, c]), L.ptr(prep.bgs[:, c]), L.ptr(alphas[:, c]), L.ptr(comps[:, c]), L.ptr(acc[0, c]),
                   B, HW, HW, S * HW, S * 3 * HW, st)
            L.call('tcvom_loss_finalize', L.ptr(acc[0, c]), L.ptr(losses[0:]), 1.0 / ni, 0, float(B * HW), window, 1, st)
        ndt = S - 3 if S >= 5 else 0
        for k in range(ndt):
            c = k + 1
            L.call('tcvom_masked_l1_fwd', L.ptr(preds[k]), L.ptr(gts[:, c]), L.ptr(tm[:, c]), L.ptr(preds[k + 1]),
                   L.ptr(gts[:, c + 1]), L.ptr(tm[:, c + 1]), None, None, None, None, L.ptr(acc[1, c]),
                   B, HW, HW, S * HW, S * 3 * HW, st)
            L.call('tcvom_loss_finalize', L.ptr(acc[1, c]), L.ptr(losses[1:]), 1.0 / ndt, 0, float(B * HW), window, 1, st)
        h, w = H // TAM_OS, W // TAM_OS
        pooled = _f32((B, S, h, w), dev)
        L.call('tcvom_avgpool8', L.ptr(gts), L.ptr(pooled), B * S, H, W, st)
        dlog = []
        for k in range(ni):
            c = k + 1
            mask = prep.unk8[c]
            db, df = torch.empty_like(attb[k]), torch.empty_like(attf[k])
            L.call('tcvom_att_bce', L.ptr(attb[k].contiguous()), L.ptr(pooled[:, c]), L.ptr(pooled[:, c - 1]), L.ptr(mask),
                   L.ptr(db), L.ptr(acc[2, c]), B, h, w, window, att_thres, label_smooth, S * h * w, 1, st)
            L.call('tcvom_att_bce', L.ptr(attf[k].contiguous()), L.ptr(pooled[:, c]), L.ptr(pooled[:, c + 1]), L.ptr(mask),
                   L.ptr(df), L.ptr(acc[2, c]), B, h, w, window, att_thres, label_smooth, S * h * w, 0, st)
            L.call('tcvom_loss_finalize', L.ptr(acc[2, c]), L.ptr(losses[2:]), 1.0 / ni, 1, 0.0, window, 1, st)
            dlog.append((db, df))
        ctx.prep, ctx.S, ctx.window, ctx.ndt = prep, S, window, ndt
        ctx.preds, ctx.acc, ctx.dlog = preds, acc, dlog
        ctx.mark_non_differentiable(alphas, comps)
        return losses[0], losses[1], losses[2], alphas, comps

    @staticmethod
    def backward(ctx, g_alpha, g_dt, g_att, _ga, _gc):
        prep, S, window, acc = ctx.prep, ctx.S, ctx.window, ctx.acc
        ni = S - 2
        st = L.stream_ptr()
        preds = ctx.preds
        B, _, H, W = preds[0].shape
        HW = H * W
        gts, tm = prep.gts, prep.trimask
        dev = preds[0].device
        one = lambda g: (g if g is not None else torch.zeros((), device=dev)).reshape(1).float().contiguous()
        g_alpha, g_dt, g_att = one(g_alpha), one(g_dt), one(g_att)
        dpreds = [torch.empty_like(p) for p in preds]
        for k in range(ni):
            c = k + 1
            L.call('tcvom_masked_l1_bwd', L.ptr(preds[k]), L.ptr(gts[:, c]), L.ptr(tm[:, c]), None, None, None,
                   L.ptr(acc[0, c]), L.ptr(g_alpha), 1.0 / ni, L.ptr(dpreds[k]), None, 0, B, HW, HW, S * HW, st)
        for k in range(ctx.ndt):
            c = k + 1
            L.call('tcvom_masked_l1_bwd', L.ptr(preds[k]), L.ptr(gts[:, c]), L.ptr(tm[:, c]), L.ptr(preds[k + 1]),
                   L.ptr(gts[:, c + 1]), L.ptr(tm[:, c + 1]), L.ptr(acc[1, c]), L.ptr(g_dt), 1.0 / ctx.ndt,
                   L.ptr(dpreds[k]), L.ptr(dpreds[k + 1]), 1, B, HW, HW, S * HW, st)
        datt_b, datt_f = [], []
        for k in range(ni):
            c = k + 1
            db, df = ctx.dlog[k]
            ob, of = torch.empty_like(db), torch.empty_like(df)
            L.call('tcvom_att_bce_bwd', L.ptr(db), L.ptr(acc[2, c]), L.ptr(g_att), 1.0 / ni, L.ptr(ob), db.numel(), window, st)
            L.call('tcvom_att_bce_bwd', L.ptr(df), L.ptr(acc[2, c]), L.ptr(g_att), 1.0 / ni, L.ptr(of), df.numel(), window, st)
            datt_b.append(ob)
            datt_f.append(of)
        return (None, None, None, None, None) + tuple(dpreds) + tuple(datt_b) + tuple(datt_f)


class _SingleImageLoss(torch.autograd.Function):
    """FullModel.single_image_loss for the DIM base on ONE predicted frame c (models/model.py:94-127): L_alpha (masked
    L1), L_comp (masked L1 of the re-composited image) and L_grad (masked L1 of the gradient magnitude,
    utils/loss_func.py:42-59), plus the clamped `alphas` / `comps` visualisation tensors."""

    @staticmethod
    def forward(ctx, prep, c, S, pred):
        ctx.set_materialize_grads(False)
        st = L.stream_ptr()
        pred = pred.contiguous()
        B, _, H, W = pred.shape
        HW = H * W
        dev = pred.device
        gts, tm = prep.gts, prep.trimask
        alphas = torch.zeros((B, S, 1, H, W), dtype=torch.float32, device=dev)
        comps = torch.zeros((B, S, 3, H, W), dtype=torch.float32, device=dev)
        losses = torch.zeros(3, dtype=torch.float32, device=dev)
        acc = torch.zeros(6, dtype=torch.float32, device=dev)          # {L1 sum, count | comp sum, count, grad sum, count}
        L.call('tcvom_masked_l1_fwd', L.ptr(pred), L.ptr(gts[:, c]), L.ptr(tm[:, c]), None, None, None,
               L.ptr(prep.fgs[:, c]), L.ptr(prep.bgs[:, c]), L.ptr(alphas[:, c]), L.ptr(comps[:, c]), L.ptr(acc[0:]),
               B, HW, HW, S * HW, S * 3 * HW, st)
        L.call('tcvom_loss_finalize', L.ptr(acc[0:]), L.ptr(losses[0:]), 1.0, 0, float(B * HW), 1, 0, st)
        L.call('tcvom_dim_losses_fwd', L.ptr(pred), L.ptr(gts[:, c]), L.ptr(tm[:, c]), L.ptr(prep.fgs[:, c]), L.ptr(prep.bgs[:, c]),
               L.ptr(prep.imgs[:, c]), None, L.ptr(acc[2:]), B, H, W, HW, S * HW, S * 3 * HW, st)
        L.call('tcvom_loss_finalize', L.ptr(acc[2:]), L.ptr(losses[1:]), 1.0, 0, float(B * 3 * HW), 1, 0, st)
        L.call('tcvom_loss_finalize', L.ptr(acc[4:]), L.ptr(losses[2:]), 1.0, 0, float(B * HW), 1, 0, st)
        ctx.prep, ctx.c, ctx.S, ctx.pred, ctx.acc = prep, c, S, pred, acc
        ctx.mark_non_differentiable(alphas, comps)
        return losses[0], losses[1], losses[2], alphas, comps

    @staticmethod
    def backward(ctx, g_alpha, g_comp, g_grad, _ga, _gc):
        prep, c, S, pred, acc = ctx.prep, ctx.c, ctx.S, ctx.pred, ctx.acc
        st = L.stream_ptr()
        B, _, H, W = pred.shape
        HW = H * W
        dev = pred.device
        one = lambda g: (g if g is not None else torch.zeros((), device=dev)).reshape(1).float().contiguous()
        g_alpha, g_comp, g_grad = one(g_alpha), one(g_comp), one(g_grad)
        dpred = torch.empty_like(pred)
        L.call('tcvom_masked_l1_bwd', L.ptr(pred), L.ptr(prep.gts[:, c]), L.ptr(prep.trimask[:, c]), None, None, None,
               L.ptr(acc[0:]), L.ptr(g_alpha), 1.0, L.ptr(dpred), None, 0, B, HW, HW, S * HW, st)
        L.call('tcvom_dim_losses_bwd', L.ptr(pred), L.ptr(prep.gts[:, c]), L.ptr(prep.trimask[:, c]), L.ptr(prep.fgs[:, c]),
               L.ptr(prep.bgs[:, c]), L.ptr(prep.imgs[:, c]), L.ptr(acc[2:]), L.ptr(g_comp), L.ptr(g_grad), L.ptr(dpred), 1,
               B, H, W, HW, S * HW, S * 3 * HW, st)
        return None, None, None, dpred


class FullModel(nn.Module):
    """Baseline (no TAM) façade — models/model.py:15-246: the DIM base (BASELINE.json config 1) and the VMN archs."""
    ARCH_DICT = {'gca': GCA, 'dim': DIM_VGG, 'fba': FBA, 'index': IndexMatting}
    TRIMAP_CHANNEL_DICT = {'gca': 3, 'dim': 1, 'index': 1, 'fba': 8}
    FBA_LOSS_NORMALIZE = True
    FBA_L_ATT_MULTIPLIER = 1

    def __init__(self, model, dilate_kernel=None, eps=0, **kwargs):
        super().__init__()
        self.DILATION_KERNEL = dilate_kernel
        self.EPS = eps
        self.IMG_SCALE = 1. / 255
        self.register_buffer('IMG_MEAN', torch.tensor([0.485, 0.456, 0.406]).reshape([1, 1, 3, 1, 1]).float())
        self.register_buffer('IMG_STD', torch.tensor([0.229, 0.224, 0.225]).reshape([1, 1, 3, 1, 1]).float())
        self.model_name = model
        if model.startswith('vmn'):
            self.NET = VMN.get_VMN_models(arch=model, **kwargs)
            self.window = int(kwargs['agg_window'])
        else:
            if model not in self.ARCH_DICT:
                raise KeyError(model)
            if self.ARCH_DICT[model] is None:
                raise NotImplementedError('%s: this single-image base is not on the MI355X path yet (SURVEY.md §8)' % model)
            self.NET = self.ARCH_DICT[model]()
        from .ddp import banks_of
        for bank in banks_of(self.NET):                        # fp16 build: the network's backward runs under a loss scale (ops.py)
            ops.SCALER.register(bank)                              # (bank.loss_scale; halves after an overflowed backward: ops.LossScaler)
        self.method = model[model.rfind('_') + 1:]
        self.TRIMAP_CHANNEL = self.TRIMAP_CHANNEL_DICT[self.method]
        self.att_thres, self.label_smooth = 0.3, 0.2          # FullModel_VMD's defaults (its forward serves VMN archs here too)

    def _dilation(self, clips):
        """The dilation radius of every clip of the batch (models/model.py:60-64): with `dilate_kernel=None` (train_ddp.py's
        default) the reference draws `int(torch.randint(0, 26, size=()))` INSIDE its loop over the clips -- one draw per clip from
        torch's global CPU generator, in clip order: trimap widths 1..51, a different one per clip, and the generator ends in the
        state the reference leaves it in.  (The draw is a CPU tensor: no device synchronisation.)"""
        if self.DILATION_KERNEL is None:
            return [int(torch.randint(0, 26, size=())) for _ in range(clips)]
        return [int(self.DILATION_KERNEL)] * clips

    def preprocess(self, a, fg, bg):
        p = preprocess_window(a, fg, bg, self._dilation(a.shape[0]), self.EPS, self.TRIMAP_CHANNEL)
        tris = p.x8[..., 3:3 + self.TRIMAP_CHANNEL].permute(0, 1, 4, 2, 3).float()
        imgs = p.x8[..., 0:3].permute(0, 1, 4, 2, 3).float()
        return p.imgs, p.fgs, p.bgs, p.gts, tris, p.trimask, imgs

    def forward(self, a, fg, bg):
        """models/model.py:199-246 -> [L_alpha, L_comp, L_grad, scaled_imgs, tris_vis, alphas, comps, scaled_gts, Fs, Bs];
        single-image bases predict only the centre frame of the clip."""
        B, S = a.shape[:2]
        H, W = a.shape[-2:]
        assert H % 32 == 0 and W % 32 == 0, 'H and W must be multiples of 32'
        if self.model_name.startswith('vmn'):
            # a VMN architecture through the baseline façade: the single-image losses of the interior frames only (:214-218)
            out = FullModel_VMD.forward(self, a, fg, bg)
            return out[:3] + out[5:]
        c = S // 2
        prep = preprocess_window(a, fg, bg, self._dilation(a.shape[0]), self.EPS, 1 if self.TRIMAP_CHANNEL == 1 else 3)
        if self.method == 'fba':
            from . import fba_losses as FL
            x2, extras, _ = fba_network_input(prep, self.EPS)
            pred = ops.enter_backward(self.NET.run(x2[:, c].contiguous(), extras[:, c].contiguous(), prep.imgs[:, c].contiguous()), self.NET._bank)
            # the loss kernels take the window tensors and a frame index: predict "frame c of a clip whose interior is c"
            alphas = torch.zeros_like(prep.gts)
            comps, Fs, Bs = torch.zeros_like(prep.fgs), torch.zeros_like(prep.fgs), torch.zeros_like(prep.fgs)
            L1, L2, L3 = FL._FbaFrameLoss.apply(pred, prep.gts, prep.trimask, prep.fgs, prep.bgs, prep.imgs, c, alphas, comps, Fs, Bs)
            return [L1, L2, L3, prep.imgs, prep.tris_vis, alphas, comps, prep.gts, Fs, Bs]
        if self.method == 'gca':
            pred = ops.enter_backward(self.NET.run(x8_frame(prep, c), prep.unk[:, c, ::TAM_OS, ::TAM_OS].contiguous()), self.NET._bank)
            L_alpha, _lc, _lg, alphas, comps = _SingleImageLoss.apply(prep, c, S, pred)
            zero = torch.zeros_like(L_alpha)                   # GCA: alpha loss only (models/model.py:110-114)
            return [L_alpha, zero, zero.clone(), prep.imgs, prep.tris_vis, alphas, comps, prep.gts, prep.fgs, prep.bgs]
        pred = ops.enter_backward(self.NET.run(prep.x8[:, c].contiguous()), self.NET._bank)
        L_alpha, L_comp, L_grad, alphas, comps = _SingleImageLoss.apply(prep, c, S, pred)
        return [L_alpha, L_comp, L_grad, prep.imgs, prep.tris_vis, alphas, comps, prep.gts, prep.fgs, prep.bgs]


class FullModel_VMD(FullModel):
    """models/model.py:248-357."""
    TAM_OS = TAM_OS

    def __init__(self, model, att_thres=0.3, label_smooth=0.2, **kwargs):
        assert model.startswith('vmn'), 'FullModel_VMD only support VMN arch'
        super().__init__(model, **kwargs)
        self.att_thres = att_thres
        self.label_smooth = label_smooth

    def forward(self, a, fg, bg, wb=None, wf=None):
        B, S = a.shape[:2]
        assert S >= 3, 'a window needs at least 3 frames'
        H, W = a.shape[-2:]
        assert H % 32 == 0 and W % 32 == 0, 'H and W must be multiples of 32 (pred_vmn.py:90)'
        prep = preprocess_window(a, fg, bg, self._dilation(a.shape[0]), self.EPS, 1 if self.TRIMAP_CHANNEL == 1 else 3)
        if self.method == 'fba':
            return self._forward_fba(prep, B, S, H, W)
        frames = [x8_frame(prep, s) for s in range(S)]
        prep.unk8 = unk8_frames(prep, S)
        # (fp16 build: the gradients the loss kernels send back into the network are scaled here -- ops.LOSS_SCALE)
        preds, attb, attf = ops.enter_backward(self.NET.run(frames, prep.unk8), self.NET._bank)
        ni = S - 2
        tensors = [preds[c] for c in range(1, S - 1)] + [attb[c] for c in range(1, S - 1)] + [attf[c] for c in range(1, S - 1)]
        L_alpha, L_dt, L_att, alphas, comps = _WindowLoss.apply(prep, self.window, float(self.att_thres),
                                                                float(self.label_smooth), S, *tensors)
        if S < 5:
            L_dt = torch.zeros_like(L_att)
        if self.method != 'gca':
            # DIM / Index bases also train on the composition and gradient losses of the interior frames (:110-118)
            per = [_SingleImageLoss.apply(prep, c, S, preds[c]) for c in range(1, S - 1)]
            L_comp = sum(p[1] for p in per) / float(ni)
            L_grad = sum(p[2] for p in per) / float(ni)
            return [L_alpha, L_comp, L_grad, L_dt, L_att, prep.imgs, prep.tris_vis, alphas, comps, prep.gts, prep.fgs, prep.bgs]
        zero = torch.zeros_like(L_alpha)                       # GCA: L_comp = L_grad = 0 (models/model.py:112-114)
        return [L_alpha, zero, zero.clone(), L_dt, L_att,
                prep.imgs, prep.tris_vis, alphas, comps, prep.gts, prep.fgs, prep.bgs]


    def _forward_fba(self, prep, B, S, H, W):
        """FBA base (config 5): 8-channel trimap with the distance-transform click maps (models/model.py:71-77), network
        output (alpha, F, B) per interior frame, fba_single_image_loss + L_att (+ L_dt on alpha, F and B for S >= 5)."""
        from . import fba_losses as FL
        x2, extras, _tris = fba_network_input(prep, self.EPS)
        unk_small = prep.unk[:, :, ::TAM_OS, ::TAM_OS].contiguous()
        pred, attb, attf = ops.enter_backward(self.NET.run(x2, extras, prep.imgs, unk_small), self.NET._bank)
        norm = self.FBA_LOSS_NORMALIZE
        L1, L2, L3, alphas, comps, Fs, Bs = FL.fba_single_image_loss(pred, prep.trimask, prep.gts, prep.fgs, prep.bgs, prep.imgs, norm)
        L_att = FL.attention_loss(attb, attf, unk_small, prep.gts, self.window, float(self.att_thres), float(self.label_smooth), TAM_OS)
        L_att = L_att * self.FBA_L_ATT_MULTIPLIER
        if S >= 5:
            # L_dt needs alphas / Fs / Bs as differentiable functions of the prediction (the loss kernels write them as
            # plain outputs): rebuild the interior frames with the same selection
            m = prep.trimask[:, 1:S - 1] > 0
            ends = lambda t: torch.cat([torch.zeros_like(t[:, :1]), t, torch.zeros_like(t[:, :1])], dim=1)
            al = ends(torch.where(m, pred[:, :, :1], prep.gts[:, 1:S - 1]))
            cF = ends(torch.where(m, pred[:, :, 1:4], prep.fgs[:, 1:S - 1]))
            cB = ends(torch.where(m, pred[:, :, 4:7], prep.bgs[:, 1:S - 1]))
            L_dt = FL.dtssd(al, prep.gts, prep.trimask, norm) + 0.25 * (FL.dtssd(cF, prep.fgs, prep.trimask, norm) +
                                                                       FL.dtssd(cB, prep.bgs, prep.trimask, norm))
        else:
            L_dt = torch.zeros_like(L_att)
        return [L1, L2, L3, L_dt, L_att, prep.imgs, prep.tris_vis, alphas, comps, prep.gts, Fs, Bs]


def fba_network_input(prep, eps, want_tris=False, use_dilated=True):
    """tcvom_fba_input on the outputs of preprocess_window: (x2 bf16 [B,S,H/2,W/2,64] space-to-depth network input, extras
    bf16 [B,S,H,W,8], tris fp32 [B,S,8,H,W] or None).  use_dilated=False: EvalModel, where the bg / fg classes are read
    from the user trimap itself (models/model.py:380-385) and the dilation only widens the blend / TAM mask."""
    B, S, _, H, W = prep.gts.shape
    dev = prep.gts.device
    x2 = torch.empty((B, S, H // 2, W // 2, 64), dtype=H16, device=dev)
    extras = torch.empty((B, S, H, W, 8), dtype=H16, device=dev)
    tris = torch.empty((B, S, 8, H, W), dtype=torch.float32, device=dev) if want_tris else None
    scratch = torch.empty((B * S, 2, H, W), dtype=torch.float32, device=dev)
    L.call('tcvom_fba_input', L.ptr(prep.gts), L.ptr(prep.unk) if use_dilated else None, L.ptr(prep.imgs), L.ptr(x2), L.ptr(extras), L.ptr(tris),
           L.ptr(scratch), B * S, H, W, float(eps), L.stream_ptr())
    return x2, extras, tris


class EvalModel(FullModel):
    """Inference on frames + user trimaps (models/model.py:359-453)."""

    def __init__(self, model, agg_window, dilate_kernel):
        super().__init__(model, dilate_kernel=dilate_kernel, agg_window=agg_window)

    def forward(self, imgs, tris):
        """imgs [B,S,3,H,W] BGR 0..255, tris [B,S,1,H,W] in {0, 128, 255} -> alphas [B,S,1,H,W]: the prediction inside the
        (optionally dilated) unknown region and the trimap value elsewhere for the interior frames, zeros for the
        first and last frame (models/model.py:388-424; callers pred_test.py:90-109)."""
        B, S = imgs.shape[:2]
        H, W = imgs.shape[-2:]
        assert self.model_name.startswith('vmn'), 'only the VMN architectures run on the HIP path'
        assert S >= 3 and H % 32 == 0 and W % 32 == 0, 'pad the frames to multiples of 32 (pred_test.py:47-66)'
        with torch.no_grad():
            dil = self.DILATION_KERNEL if self.DILATION_KERNEL is not None else 0
            prep = preprocess_window(tris, imgs, None, dil, 0.0, 1 if self.TRIMAP_CHANNEL == 1 else 3)
            if self.method == 'fba':
                return self._forward_fba_eval(prep, B, S, H, W)
            frames = [x8_frame(prep, s) for s in range(S)]
            prep.unk8 = unk8_frames(prep, S)
            preds, _attb, _attf = self.NET.run(frames, prep.unk8)
            alphas = torch.zeros((B, S, 1, H, W), dtype=torch.float32, device=imgs.device)
            for c in range(1, S - 1):
                alphas[:, c] = torch.where(prep.trimask[:, c] > 0, preds[c].float(), prep.gts[:, c])
        return alphas


    def forward_video(self, imgs, tris, chunk=4):
        """A whole clip at once: imgs [T,3,H,W] BGR 0..255, tris [T,1,H,W] in {0,128,255} -> alphas
        [T,1,H,W], frame c predicted from (c-1, c, c+1) with the clip ends mirrored exactly as pred_test.py:27-41 builds
        its samples.  In eval mode the encoder + decoder-front output of a frame does not depend on its window
        (BatchNorm running statistics, stored SpectralNorm u / v), so it is computed ONCE per frame (`chunk` frames per
        launch) and shared by the three windows that contain the frame: a third of the encoder work of per-sample calls."""
        from .weights import bank_token
        assert self.method == 'gca' and not self.training, 'feature caching: vmn_gca in eval mode'
        T, _, H, W = imgs.shape
        assert T >= 2 and chunk >= 2 and H % 32 == 0 and W % 32 == 0
        net, bank = self.NET, self.NET._bank
        dev = self.IMG_MEAN.device                                  # the clip may stay on the host: chunks are uploaded
        dil = self.DILATION_KERNEL if self.DILATION_KERNEL is not None else 0
        alphas = torch.zeros((T, 1, H, W), dtype=torch.float32, device=dev)
        feats, mids, unk8, trimask, gts = {}, {}, {}, {}, {}
        with torch.no_grad():
            for c in range(T):
                p = c + 1 if c == 0 else c - 1
                n = c - 1 if c == T - 1 else c + 1
                if max(p, n) not in feats:                              # next chunk of frames through encoder + front
                    lo = max(feats) + 1 if feats else 0
                    hi = min(T, lo + chunk)
                    prep = preprocess_window(tris[lo:hi].unsqueeze(0).to(dev), imgs[lo:hi].unsqueeze(0).to(dev), None, dil, 0.0)
                    X = prep.x8[0].contiguous()
                    if getattr(prep, 'x8_f16', None) is not None:
                        ops.set_f16_twin(X, prep.x8_f16[0].contiguous())
                    U = prep.unk[0, :, ::TAM_OS, ::TAM_OS].contiguous()
                    token = bank_token(bank, hi - lo, False, net)
                    emb, mid = net.encoder.run(X, U, token, False)
                    feat = net.decoder.run_front(emb, mid, token, False)
                    for k, i in enumerate(range(lo, hi)):
                        feats[i] = feat[k:k + 1]
                        mids[i] = {key: (tuple(t[k:k + 1] for t in v) if isinstance(v, (tuple, list)) else v[k:k + 1])
                                   for key, v in mid.items() if key != 'unknown'}
                        mids[i]['unknown'] = U[k:k + 1]
                        unk8[i], trimask[i], gts[i] = U[k:k + 1], prep.trimask[0, k], prep.gts[0, k]
                token = bank_token(bank, 1, False, net)
                pred, _ab, _af = net.decoder.run_tail(feats[c], feats[p], feats[n], unk8[c], mids[c], token, False)
                alphas[c] = torch.where(trimask[c] > 0, pred[0].float(), gts[c])
                for i in [i for i in feats if i < c - 1]:               # frames no later window needs
                    for d in (feats, mids, unk8, trimask, gts):
                        d.pop(i, None)
        return alphas

    def _forward_fba_eval(self, prep, B, S, H, W):
        """FBA: (alphas, Fs, Bs) -- prediction inside the unknown region, trimap value / image elsewhere (:425-453)."""
        x2, extras, _ = fba_network_input(prep, 0.0, use_dilated=False)
        unk_small = prep.unk[:, :, ::TAM_OS, ::TAM_OS].contiguous()
        pred, _ab, _af = self.NET.run(x2, extras, prep.imgs, unk_small)
        dev = pred.device
        alphas = torch.zeros((B, S, 1, H, W), dtype=torch.float32, device=dev)
        Fs = torch.zeros((B, S, 3, H, W), dtype=torch.float32, device=dev)
        Bs = torch.zeros((B, S, 3, H, W), dtype=torch.float32, device=dev)
        for c in range(1, S - 1):
            m = prep.trimask[:, c] > 0
            p = pred[:, c - 1]
            alphas[:, c] = torch.where(m, p[:, :1], prep.gts[:, c])
            Fs[:, c] = torch.where(m, p[:, 1:4], prep.imgs[:, c])
            Bs[:, c] = torch.where(m, p[:, 4:7], prep.imgs[:, c])
        return alphas, Fs, Bs


_LOSS_WEIGHTS = {}


def train_step_loss(out):
    """train_ddp.py:56-61: L_alpha + L_comp + L_grad + 0.5 L_dt + 0.25 L_att (each `.mean()` of a 0-d loss) -- as one stack and one
    weighted sum instead of a dozen scalar kernels."""
    terms = [o if o.dim() == 0 else o.mean() for o in out[:5]]
    dev = terms[0].device
    w = _LOSS_WEIGHTS.get(dev)
    if w is None:
        w = _LOSS_WEIGHTS[dev] = torch.tensor([1.0, 1.0, 1.0, 0.5, 0.25], dtype=torch.float32).to(dev)
    return (torch.stack([t.float() for t in terms]) * w).sum()
