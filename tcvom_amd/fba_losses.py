"""Losses of the FBA base (models/model.py:129-197,285-345; utils/loss_func.py:9-158) on the GPU.

`fba_single_image_loss` runs one `_FbaFrameLoss` per interior frame: the L1 family + L1_grad in one kernel, the exclusion
loss as 3 levels x (reduction, per-sample terms), the Laplacian loss of alpha, F and B as ONE 7-channel pyramid of the
difference image (the pyramid is linear) -- csrc/fba_loss.hip; ~40 launches forward, ~30 backward per frame, no host
sync.  The handful of scalars in between (means, fourth roots, weights) are tiny device-side tensor expressions.
`attention_loss` (L_att) runs on the facade's `avgpool8` / `att_bce` kernels; `dtssd` (L_dt) is a small tensor expression on masked tensors.
"""
import ctypes as C

import torch

from . import _lib as L

EPS = 1.001e-5
LAP_LEVELS, EXCL_LEVELS = 5, 3


class _FbaFrameLoss(torch.autograd.Function):
    """fba_single_image_loss for ONE interior frame c (models/model.py:142-175, normalize=True):
    (pred [B,7,H,W]) -> L_alpha_comp, L_lap, L_grad; writes the alphas / comps / Fs / Bs slices of the window tensors."""

    @staticmethod
    def forward(ctx, pred, gts, trimask, fgs, bgs, imgs, c, alphas, comps, Fs, Bs):
        B, S, _, H, W = gts.shape
        assert H % 32 == 0 and W % 32 == 0, 'the 5-level Laplacian pyramid needs H, W multiples of 32'
        dev = pred.device
        st = L.stream_ptr()
        HW = H * W
        pred = pred.contiguous()
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        d0, fb = f32(B, 7, H, W), f32(B, 6, H, W)
        acc = torch.zeros(6 + LAP_LEVELS * 7 + EXCL_LEVELS * (4 + 2 * B), dtype=torch.float32, device=dev)
        frame = (L.ptr(pred), L.ptr(gts[:, c]), L.ptr(trimask[:, c]), L.ptr(fgs[:, c]), L.ptr(bgs[:, c]), L.ptr(imgs[:, c]),
                 7 * HW, S * HW, S * 3 * HW)
        L.call('tcvom_fba_point_fwd', *frame, L.ptr(d0), L.ptr(fb), L.ptr(alphas[:, c]), L.ptr(comps[:, c]), L.ptr(Fs[:, c]),
               L.ptr(Bs[:, c]), L.ptr(acc), B, H, W, st)
        # exclusion loss: 3 levels of (F, B)
        lv, ex = [fb], []
        off = 6 + LAP_LEVELS * 7
        for l in range(EXCL_LEVELS):
            h, w = H >> l, W >> l
            sums, terms = acc[off:off + 4], acc[off + 4:off + 4 + 2 * B]
            off += 4 + 2 * B
            L.call('tcvom_excl_abs', L.ptr(lv[l]), L.ptr(sums), B, h, w, st)
            L.call('tcvom_excl_terms', L.ptr(lv[l]), L.ptr(sums), None, L.ptr(terms), 0, B, h, w, st)
            ex.append((sums, terms.view(B, 2), 3.0 * h * w))
            if l + 1 < EXCL_LEVELS:
                nxt = f32(B, 6, h // 2, w // 2)
                L.call('tcvom_avgpool2_f32', L.ptr(lv[l]), L.ptr(nxt), B * 6, h, w, st)
                lv.append(nxt)
        # Laplacian loss: one pyramid of d0 = (refine - gt, F - fg, B - bg)
        cur, sgns = d0, []
        for l in range(LAP_LEVELS):
            h, w = H >> l, W >> l
            down = f32(B, 7, h // 2, w // 2)
            sg = torch.empty((B, 7, h, w), dtype=torch.int8, device=dev)
            L.call('tcvom_lap_down', L.ptr(cur), L.ptr(down), B * 7, h, w, st)
            L.call('tcvom_lap_resid', L.ptr(cur), L.ptr(down), L.ptr(sg), L.ptr(acc[6 + 7 * l:]), B * 7, h, w, st)
            sgns.append(sg)
            cur = down
        n1, n3 = float(B * HW), float(3 * B * HW)
        # the scalar arithmetic on the accumulators (fourth roots of the exclusion terms, level weights of the pyramid) in ONE launch
        out3 = f32(3)
        ctx.excl_n = tuple(float(n) for _, _, n in ex)
        L.call('tcvom_fba_loss_finish', L.ptr(acc), B, n1, n3, ctx.excl_n[0], ctx.excl_n[1], ctx.excl_n[2], L.ptr(out3), st)
        ctx.frame_tensors = (pred, gts, trimask, fgs, bgs, imgs)
        ctx.c, ctx.lv, ctx.ex, ctx.sgns, ctx.acc = c, lv, ex, sgns, acc
        ctx.mark_non_differentiable(alphas, comps, Fs, Bs)
        return out3[0], out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_ac, g_lap, g_grad):
        pred, gts, trimask, fgs, bgs, imgs = ctx.frame_tensors
        c = ctx.c
        B, S, _, H, W = gts.shape
        HW = H * W
        dev = pred.device
        st = L.stream_ptr()
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        n1, n3 = float(B * HW), float(3 * B * HW)
        gp = lambda g: None if g is None else g.float().contiguous()
        g_ac, g_lap, g_grad = gp(g_ac), gp(g_lap), gp(g_grad)
        cbuf = f32(6 + 6 * B + 7 * LAP_LEVELS)            # coef[6] | exclusion d L / d terms [3][B][2] | Laplacian level weights [5][7]
        L.call('tcvom_fba_loss_coefs', L.ptr(ctx.acc), L.ptr(g_ac), L.ptr(g_lap), L.ptr(g_grad), B, n1, n3, ctx.excl_n[0], ctx.excl_n[1],
               ctx.excl_n[2], L.ptr(cbuf), st)
        coef = cbuf[:6]
        # exclusion: d L / d terms, then level by level from the coarsest
        dl = None
        for l in reversed(range(EXCL_LEVELS)):
            sums, terms, n = ctx.ex[l]
            h, w = H >> l, W >> l
            wts = cbuf[6 + 2 * B * l:6 + 2 * B * (l + 1)]
            dsum = torch.zeros(2, dtype=torch.float32, device=dev)
            L.call('tcvom_excl_terms', L.ptr(ctx.lv[l]), L.ptr(sums), L.ptr(wts), L.ptr(dsum), 1, B, h, w, st)
            cur = f32(B, 6, h, w)
            L.call('tcvom_excl_bwd', L.ptr(ctx.lv[l]), L.ptr(sums), L.ptr(wts), L.ptr(dsum), L.ptr(dl), L.ptr(cur), B, h, w, st)
            dl = cur
        # Laplacian: g_l = s_l + D^T (g_{l+1} - U^T s_l), from the top level down
        g = None
        for l in reversed(range(LAP_LEVELS)):
            h, w = H >> l, W >> l
            cf = cbuf[6 + 6 * B + 7 * l:6 + 6 * B + 7 * (l + 1)]
            r = f32(B, 7, h // 2, w // 2)
            L.call('tcvom_lap_bwd_coarse', L.ptr(ctx.sgns[l]), L.ptr(cf), L.ptr(g), L.ptr(r), B * 7, h, w, st)
            g = f32(B, 7, h, w)
            L.call('tcvom_lap_bwd_fine', L.ptr(ctx.sgns[l]), L.ptr(cf), L.ptr(r), L.ptr(g), B * 7, h, w, st)
        dpred = torch.empty_like(pred)
        L.call('tcvom_fba_point_bwd', L.ptr(pred), L.ptr(gts[:, c]), L.ptr(trimask[:, c]), L.ptr(fgs[:, c]), L.ptr(bgs[:, c]),
               L.ptr(imgs[:, c]), 7 * HW, S * HW, S * 3 * HW, L.ptr(coef), L.ptr(g), L.ptr(dl), L.ptr(dpred), B, H, W, st)
        return (dpred,) + (None,) * 10


def fba_single_image_loss(preds, trimasks, gts, fgs, bgs, imgs, normalize=True):
    """models/model.py:129-197 for the interior frames.  preds [B,S-2,7,H,W] (interior frames only); the other tensors
    [B,S,*,H,W].  -> L_alpha_comp, L_lap, L_grad, alphas, comps, Fs, Bs ([B,S,*,H,W], zeros at the ends)."""
    assert normalize, 'FullModel.FBA_LOSS_NORMALIZE is True in the reference (models/model.py:28); the summed variant is not built'
    B, S = gts.shape[:2]
    alphas = torch.zeros_like(gts)
    comps, Fs, Bs = torch.zeros_like(fgs), torch.zeros_like(fgs), torch.zeros_like(fgs)
    La, Ll, Lg = [], [], []
    for c in range(1, S - 1):
        a, l, g = _FbaFrameLoss.apply(preds[:, c - 1], gts, trimasks, fgs, bgs, imgs, c, alphas, comps, Fs, Bs)
        La.append(a)
        Ll.append(l)
        Lg.append(g)
    n = float(len(La))
    return sum(La) / n, sum(Ll) / n, sum(Lg) / n, alphas, comps, Fs, Bs


class _AttLoss(torch.autograd.Function):
    """L_att (models/model.py:285-323) on the kernels the GCA facade uses (csrc/facade.hip): `avgpool8` pools the ground truth to the
    TAM grid, `att_bce` takes the 49 neighbour targets of every unknown os8 pixel from the pooled adjacent frame in place (no unfold
    temporary) and accumulates the BCE sum + the unknown count of both directions, `loss_finalize` divides ON THE DEVICE (no host
    read of the count), `att_bce_bwd` scales the saved sigmoid(x) - t."""

    @staticmethod
    def forward(ctx, gts, unk_small, window, att_thres, label_smooth, S, *logits):
        ctx.set_materialize_grads(False)
        ni = S - 2
        attb, attf = logits[:ni], logits[ni:]
        B, _, _, H, W = gts.shape
        h, w = unk_small.shape[-2:]
        dev = gts.device
        st = L.stream_ptr()
        pooled = torch.empty((B, S, h, w), dtype=torch.float32, device=dev)
        L.call('tcvom_avgpool8', L.ptr(gts.contiguous()), L.ptr(pooled), B * S, H, W, st)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        acc = torch.zeros((S, 2), dtype=torch.float32, device=dev)
        dlog = []
        for k in range(ni):
            c = k + 1
            mask = unk_small[:, c].contiguous()
            db, df = torch.empty_like(attb[k]), torch.empty_like(attf[k])
            L.call('tcvom_att_bce', L.ptr(attb[k].contiguous()), L.ptr(pooled[:, c]), L.ptr(pooled[:, c - 1]), L.ptr(mask),
                   L.ptr(db), L.ptr(acc[c]), B, h, w, window, att_thres, label_smooth, S * h * w, 1, st)
            L.call('tcvom_att_bce', L.ptr(attf[k].contiguous()), L.ptr(pooled[:, c]), L.ptr(pooled[:, c + 1]), L.ptr(mask),
                   L.ptr(df), L.ptr(acc[c]), B, h, w, window, att_thres, label_smooth, S * h * w, 0, st)
            L.call('tcvom_loss_finalize', L.ptr(acc[c]), L.ptr(loss), 1.0 / ni, 1, 0.0, window, 1, st)
            dlog.append((db, df))
        ctx.acc, ctx.dlog, ctx.window, ctx.ni = acc, dlog, window, ni
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ni, acc = ctx.ni, ctx.acc
        if g is None:
            return (None,) * (6 + 2 * ni)
        st = L.stream_ptr()
        g = g.reshape(1).float().contiguous()
        outs_b, outs_f = [], []
        for k in range(ni):
            db, df = ctx.dlog[k]
            ob, of = torch.empty_like(db), torch.empty_like(df)
            L.call('tcvom_att_bce_bwd', L.ptr(db), L.ptr(acc[k + 1]), L.ptr(g), 1.0 / ni, L.ptr(ob), db.numel(), ctx.window, st)
            L.call('tcvom_att_bce_bwd', L.ptr(df), L.ptr(acc[k + 1]), L.ptr(g), 1.0 / ni, L.ptr(of), df.numel(), ctx.window, st)
            outs_b.append(ob)
            outs_f.append(of)
        return (None,) * 6 + tuple(outs_b) + tuple(outs_f)


def attention_loss(attb, attf, unk_small, gts, window, att_thres, label_smooth, os=8):
    """L_att (models/model.py:285-323): attb / attf lists of S ([B,w*w,h*w] logits, None at the ends), unk_small
    uint8 [B,S,h,w].  Zero for a frame without unknown os8 pixels."""
    assert os == 8, 'the pooling kernel is the 8 x 8 average of the TAM grid'
    B, S = gts.shape[:2]
    assert gts.shape[-2] // os == unk_small.shape[-2] and gts.shape[-1] // os == unk_small.shape[-1]
    return _AttLoss.apply(gts, unk_small, int(window), float(att_thres), float(label_smooth), S,
                          *([attb[c] for c in range(1, S - 1)] + [attf[c] for c in range(1, S - 1)]))


def dtssd(pred, gt, trimasks, normalize=True, epsilon=1.001e-5):
    """_dtSSD (models/model.py:326-333) with the masked L1_mask."""
    S = pred.shape[1]
    terms = []
    for c in range(1, S - 2):
        m = trimasks[:, c]
        res = ((pred[:, c] - pred[:, c + 1]) - (gt[:, c] - gt[:, c + 1])).abs() * m
        if normalize:
            terms.append(res.sum() / (m > epsilon).float().sum().clamp(epsilon, float(gt[:, c].numel() + 1)))
        else:
            terms.append(res.sum())
    return sum(terms) / float(len(terms))
