"""Losses of the FBA base (models/model.py:129-197,285-345; utils/loss_func.py:9-158) on the GPU.

First version: these are fp32 tensor expressions on device tensors (ATen element-wise / reduction kernels with
autograd), not yet fused HIP kernels like the GCA / DIM losses of tcvom_amd.facade — about 1.5 k small launches
per window.  DESIGN.md lists them as the next FBA item to move into libtcvom_hip.so.  The 5x5 Gaussian of the
Laplacian pyramid is applied as two separable 5-tap passes over shifted views (no library convolution).
"""
import torch
import torch.nn.functional as F

_G5 = (1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16)


def _l1(x, y, normalize):
    d = (x - y).abs()
    return d.mean() if normalize else d.sum()


def _gradient(im):
    dy = F.pad(im[:, :, 1:, :] - im[:, :, :-1, :], (0, 0, 0, 1))
    dx = F.pad(im[:, :, :, 1:] - im[:, :, :, :-1], (0, 1, 0, 0))
    return dx, dy


def l1_grad(pred, gt, normalize, epsilon=1.001e-5):
    """utils/loss_func.py:49-58 (no mask)."""
    fx, fy = _gradient(pred)
    tx, ty = _gradient(gt)
    return _l1(torch.sqrt(fx * fx + fy * fy + epsilon), torch.sqrt(tx * tx + ty * ty + epsilon), normalize)


def exclusion_loss(img1, img2, level=3, epsilon=1.001e-5, normalize=True):
    """utils/loss_func.py:63-90."""
    lx, ly = [], []
    for _ in range(level):
        gx1, gy1 = _gradient(img1)
        gx2, gy2 = _gradient(img2)
        ax = 2.0 * gx1.abs().mean() / (gx2.abs().mean() + epsilon)
        ay = 2.0 * gy1.abs().mean() / (gy2.abs().mean() + epsilon)
        sx1, sy1 = torch.sigmoid(gx1) * 2 - 1, torch.sigmoid(gy1) * 2 - 1
        sx2, sy2 = torch.sigmoid(gx2 * ax) * 2 - 1, torch.sigmoid(gy2 * ay) * 2 - 1
        lx.append((((sx1 * sx1) * (sx2 * sx2)).mean(dim=(1, 2, 3)) + epsilon) ** 0.25)
        ly.append((((sy1 * sy1) * (sy2 * sy2)).mean(dim=(1, 2, 3)) + epsilon) ** 0.25)
        img1 = F.avg_pool2d(img1, 2, 2)
        img2 = F.avg_pool2d(img2, 2, 2)
    red = torch.mean if normalize else torch.sum
    return red(sum(lx) / float(level)) + red(sum(ly) / float(level))


def _gauss5(img, scale=1.0):
    """5x5 binomial filter with reflect padding (LapLoss.conv_gauss), separable."""
    x = F.pad(img, (2, 2, 2, 2), mode='reflect')
    H, W = img.shape[-2:]
    v = sum(_G5[k] * x[:, :, k:k + H, :] for k in range(5))
    return sum((_G5[k] * scale) * v[:, :, :, k:k + W] for k in range(5))


def laplacian_pyramid(img, levels=5):
    """utils/loss_func.py:114-147."""
    cur, pyr = img, []
    for _ in range(levels):
        down = _gauss5(cur)[:, :, ::2, ::2]
        up = torch.zeros((down.shape[0], down.shape[1], down.shape[2] * 2, down.shape[3] * 2), dtype=img.dtype, device=img.device)
        up[:, :, ::2, ::2] = down
        pyr.append(cur - _gauss5(up, 4.0))
        cur = down
    return pyr


def lap_loss(img, tgt, normalize):
    """LapLoss.forward (utils/loss_func.py:149-158, no mask)."""
    with torch.no_grad():
        pt = laplacian_pyramid(tgt)
    loss = sum((2 ** lvl) * (a - b).abs().sum() for lvl, (a, b) in enumerate(zip(laplacian_pyramid(img), pt)))
    return loss / float(tgt.numel()) if normalize else loss


def fba_single_image_loss(preds, trimasks, gts, fgs, bgs, imgs, normalize=True):
    """models/model.py:129-197 for the interior frames.  preds [B,S-2,7,H,W] (interior frames only); the other tensors
    [B,S,*,H,W].  -> L_alpha_comp, L_lap, L_grad, alphas, comps, Fs, Bs ([B,S,*,H,W], zeros at the ends)."""
    B, S = gts.shape[:2]
    La, Ll, Lg = [], [], []
    zero1, zero3 = torch.zeros_like(gts[:, 0]), torch.zeros_like(fgs[:, 0])
    alphas, comps, Fs, Bs = [zero1] * S, [zero3] * S, [zero3] * S, [zero3] * S
    for c in range(1, S - 1):
        gt, img, fg, bg = gts[:, c], imgs[:, c], fgs[:, c], bgs[:, c]
        m = trimasks[:, c] > 0
        p = preds[:, c - 1]
        refine = torch.where(m, p[:, :1], gt)
        cF = torch.where(m, p[:, 1:4], fg)
        cB = torch.where(m, p[:, 4:7], bg)
        alphas[c], Fs[c], Bs[c] = refine, cF, cB
        comps[c] = cF * refine + cB * (1.0 - refine)
        L_a1 = _l1(refine, gt, normalize)
        L_ac = _l1(cF * gt + cB * (1.0 - gt), img, normalize)
        L_FBc = _l1(fg * refine + bg * (1.0 - refine), img, normalize)
        L_FB1 = _l1(cF, fg, normalize) + _l1(cB, bg, normalize)
        La.append(L_a1 + L_ac + 0.25 * (L_FBc + L_FB1))
        Lg.append(l1_grad(refine, gt, normalize) + 0.25 * exclusion_loss(cF, cB, 3, normalize=normalize))
        Ll.append(lap_loss(refine, gt, normalize) + 0.25 * (lap_loss(cF, fg, normalize) + lap_loss(cB, bg, normalize)))
    n = float(len(La))
    st = lambda lst: torch.stack(lst, dim=1)
    return sum(La) / n, sum(Ll) / n, sum(Lg) / n, st(alphas), st(comps), st(Fs), st(Bs)


def attention_loss(attb, attf, unk_small, gts, window, att_thres, label_smooth, os=8):
    """L_att (models/model.py:285-323): attb / attf lists of S ([B,w*w,h*w] logits, None at the ends), unk_small
    uint8 [B,S,h,w].  Zero for a frame without unknown os8 pixels."""
    B, S = gts.shape[:2]
    h, w = gts.shape[-2] // os, gts.shape[-1] // os
    w2 = window * window
    pooled = F.avg_pool2d(gts.reshape(B * S, 1, gts.shape[-2], gts.shape[-1]), os, os).reshape(B, S, 1, h, w)
    terms = []
    for c in range(1, S - 1):
        m = (unk_small[:, c] != 0).reshape(B, 1, h * w).float()
        cnt = m.sum()
        cgt = pooled[:, c].reshape(B, 1, h * w)
        tot = 0.0
        for logits, adj in ((attb[c], pooled[:, c - 1]), (attf[c], pooled[:, c + 1])):
            nb = F.unfold(adj, window, padding=window // 2)
            tgt = ((cgt - nb).abs() < att_thres).float() * (1.0 - label_smooth)
            bce = F.binary_cross_entropy_with_logits(logits, tgt, reduction='none')
            tot = tot + (bce * m).sum() / (cnt * w2).clamp(min=1.0)
        terms.append(tot / 2.0)
    return sum(terms) / float(len(terms))


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
