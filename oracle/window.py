"""Oracle: the 3..5-frame window forward — façade preprocessing, the per-frame VMN
loop, and the loss assembly (reference `FullModel_VMD.forward`).

Reference sources followed (under /root/reference):
  * FullModel.preprocess / make_trimap ... models/model.py:54-92
  * single_image_loss ..................... models/model.py:94-127
  * L1_mask ............................... utils/loss_func.py:9-22
  * FullModel_VMD.forward (L_att, L_dt) ... models/model.py:258-357
  * VMN.forward ........................... models/VMN/VMN_model.py:83-113
  * loss weights .......................... train_ddp.py:56-61

Only the `vmn_gca` arch (3-channel one-hot trimap, alpha-only image loss).
TEST INFRASTRUCTURE — never imported by the product path.
"""
import torch
import torch.nn.functional as F

from .gca_net import encoder_frame, decoder_front, decoder_tail
from .tam import tam_forward

IMG_MEAN = (0.485, 0.456, 0.406)
IMG_STD = (0.229, 0.224, 0.225)
TAM_OS = 8


def make_trimap(alpha, dilate_kernel, eps=0.0):
    """alpha [B,S,1,H,W] in 0..1 -> (one-hot trimap [B,S,3,H,W] {bg,unk,fg}, dilated unknown mask [B,S,1,H,W]).
    models/model.py:54-80.  dilate_kernel None: one radius PER CLIP from torch's global generator, `int(torch.randint(0, 26, ()))`
    in clip order (models/model.py:60-64); an int: that radius; a sequence: the given per-clip radii."""
    alpha = torch.where(alpha < eps, torch.zeros_like(alpha), alpha)
    alpha = torch.where(alpha > 1 - eps, torch.ones_like(alpha), alpha)
    unk = ((alpha > 0) & (alpha < 1)).float()
    B, S, _, H, W = unk.shape
    if dilate_kernel is None:
        radii = [int(torch.randint(0, 26, size=())) for _ in range(B)]
    elif isinstance(dilate_kernel, (list, tuple)):
        radii = [int(r) for r in dilate_kernel]
    else:
        radii = [int(dilate_kernel)] * B
    assert len(radii) == B
    dil = torch.stack([F.max_pool2d(unk[i], 2 * r + 1, 1, r) for i, r in enumerate(radii)])
    cls = torch.where(dil > 0.5, torch.ones_like(alpha), 2 * alpha).long()   # 0 bg, 1 unknown, 2 fg
    onehot = F.one_hot(cls.squeeze(2), 3).permute(0, 1, 4, 2, 3).float()
    return onehot, dil


def preprocess(a, fg, bg, dilate_kernel, eps=0.0):
    """a [B,S,1,H,W], fg/bg [B,S,3,H,W]; float 0..255, BGR (models/model.py:82-92)."""
    with torch.no_grad():
        gts = a / 255.0
        fgs = fg.flip([2]) / 255.0
        bgs = bg.flip([2]) / 255.0
        scaled = fgs * gts + bgs * (1.0 - gts)
        tris, trimasks = make_trimap(gts, dilate_kernel, eps)
        mean = torch.tensor(IMG_MEAN).reshape(1, 1, 3, 1, 1).to(a)
        std = torch.tensor(IMG_STD).reshape(1, 1, 3, 1, 1).to(a)
        imgs = (scaled - mean) / std
    return scaled, fgs, bgs, gts, tris, trimasks, imgs


def l1_mask(x, y, mask, epsilon=1.001e-5):
    """utils/loss_func.py:9-22 (mask given, normalize=True)."""
    res = (x - y).abs() * mask
    denom = (mask > epsilon).float().sum().clamp(epsilon, float(y.numel() + 1))
    return res.sum() / denom


def vmn_forward(state, frames, masks, window, training):
    """VMN.forward (VMN_model.py:83-113).  frames: list of S tensors [B,6,H,W];
    masks: list of S tensors [B,1,H,W].  The encoder and decoder-front run once
    PER FRAME (so train-mode BN statistics and the SpectralNorm power iteration
    are per frame call), then the TAM + decoder tail run for interior frames."""
    S = len(frames)
    enc = [None] * S
    feats = [None] * S
    for i in range(S):
        emb, mid = encoder_frame(state, frames[i], training)
        enc[i] = mid
        feats[i] = decoder_front(state, emb, mid, training)
    preds, attb, attf, small = [None] * S, [None] * S, [None] * S, [None] * S
    for i in range(1, S - 1):
        x, attb[i], attf[i], small[i] = tam_forward(state, 'decoder.fam', feats[i], feats[i - 1],
                                                    feats[i + 1], masks[i], window)
        preds[i] = decoder_tail(state, x, enc[i], training)
    preds[0] = torch.zeros_like(preds[1])
    preds[-1] = torch.zeros_like(preds[-2])
    return preds, attb, attf, small, feats


def attention_loss(attb, attf, small, gts, window, att_thres=0.3, label_smooth=0.2):
    """L_att (models/model.py:285-323): BCE-with-logits between the TAM logits at
    unknown os8 pixels and 0.8*[|avgpool8(gt_c)(u) - avgpool8(gt_adj)(u+d)| < 0.3]."""
    B, S = gts.shape[:2]
    H, W = gts.shape[-2] // TAM_OS, gts.shape[-1] // TAM_OS
    w2 = window * window
    terms = []
    for c in range(1, S - 1):
        m = small[c].reshape(B, 1, H * W).float()                     # [B,1,N]
        n_unknown = m.sum()
        if n_unknown == 0:
            terms.append(torch.zeros((), dtype=gts.dtype, device=gts.device))
            continue
        pool = lambda t: F.avg_pool2d(t, TAM_OS, TAM_OS)
        cgt = pool(gts[:, c]).reshape(B, 1, H * W)
        tot = 0.0
        for logits, adj in ((attb[c], gts[:, c - 1]), (attf[c], gts[:, c + 1])):
            nb = F.unfold(pool(adj), window, padding=window // 2)      # [B,w2,N], zero pad
            tgt = ((cgt - nb).abs() < att_thres).float() * (1 - label_smooth)
            bce = F.binary_cross_entropy_with_logits(logits, tgt, reduction='none')
            tot = tot + (bce * m).sum() / (n_unknown * w2)             # mean over [w2, #unknown]
        terms.append(tot / 2.0)
    return sum(terms) / float(len(terms))


def dtssd_loss(alphas, gts, trimasks):
    """L_dt (models/model.py:326-345); identically zero when S < 5."""
    S = alphas.shape[1]
    if S < 5:
        return torch.zeros((), dtype=alphas.dtype, device=alphas.device)
    terms = []
    for c in range(1, S - 2):
        terms.append(l1_mask(alphas[:, c] - alphas[:, c + 1], gts[:, c] - gts[:, c + 1], trimasks[:, c]))
    return sum(terms) / float(len(terms))


def window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True,
                   att_thres=0.3, label_smooth=0.2, eps=0.0):
    """FullModel_VMD('vmn_gca').forward(a, fg, bg) -> the same 12-item list
    (models/model.py:258-357), plus a dict of intermediates for the tests."""
    scaled_imgs, fgs, bgs, gts, tris, trimasks, imgs = preprocess(a, fg, bg, dilate_kernel, eps)
    B, S = a.shape[:2]
    frames = [torch.cat([imgs[:, s], tris[:, s]], dim=1) for s in range(S)]
    masks = [trimasks[:, s] for s in range(S)]
    preds, attb, attf, small, feats = vmn_forward(state, frames, masks, window, training)
    preds = torch.stack(preds, dim=1)

    L_alpha = []
    alphas = [None] * S
    comps = [None] * S
    for c in range(1, S - 1):                                           # single_image_loss, start=1, end=S-1
        unk = trimasks[:, c]
        refine = torch.where(unk.bool(), preds[:, c], gts[:, c])
        alphas[c] = refine
        comps[c] = fgs[:, c] * refine + bgs[:, c] * (1.0 - refine)
        L_alpha.append(l1_mask(refine, gts[:, c], unk))
    L_alpha = sum(L_alpha) / float(len(L_alpha))
    zero = torch.zeros_like(L_alpha)                                    # GCA: L_comp = L_grad = 0 (:112-114)
    alphas[0] = alphas[-1] = torch.zeros_like(alphas[1])
    comps[0] = comps[-1] = torch.zeros_like(comps[1])
    alphas = torch.stack(alphas, dim=1).clamp(0, 1)
    comps = torch.stack(comps, dim=1).clamp(0, 1)

    L_att = attention_loss(attb, attf, small, gts, window, att_thres, label_smooth)
    L_dt = dtssd_loss(alphas, gts, trimasks)
    with torch.no_grad():
        tris_vis = torch.where(trimasks.bool(), torch.full_like(gts, 128.0 / 255.0), gts)
    out = [L_alpha, zero, zero.clone(), L_dt, L_att, scaled_imgs, tris_vis, alphas, comps, gts, fgs, bgs]
    extra = {'preds': preds, 'attb': attb, 'attf': attf, 'small_mask': small, 'features': feats,
             'trimasks': trimasks, 'tris': tris, 'imgs': imgs}
    return out, extra


def train_step_loss(out):
    """train_ddp.py:56-61."""
    return out[0].mean() + out[1].mean() + out[2].mean() + 0.5 * out[3].mean() + 0.25 * out[4].mean()


def single_gca_forward(state, a, fg, bg, dilate_kernel=12, training=True, eps=0.0):
    """FullModel('gca').forward (models/model.py:199-246 with models/GCA/generators.py): the GCA base WITHOUT the temporal
    module on the centre frame -> [L_alpha, 0, 0, imgs, tris_vis, alphas, comps, gts, fgs, bgs]."""
    scaled_imgs, fgs, bgs, gts, tris, trimasks, imgs = preprocess(a, fg, bg, dilate_kernel, eps)
    S = a.shape[1]
    c = S // 2
    emb, mid = encoder_frame(state, torch.cat([imgs[:, c], tris[:, c]], dim=1), training)
    pred = decoder_tail(state, decoder_front(state, emb, mid, training), mid, training)
    unk = trimasks[:, c]
    refine = torch.where(unk.bool(), pred, gts[:, c])
    L_alpha = l1_mask(refine, gts[:, c], unk)
    alphas, comps = torch.zeros_like(gts), torch.zeros_like(fgs)
    alphas[:, c] = refine.detach().clamp(0, 1)
    comps[:, c] = (fgs[:, c] * refine + bgs[:, c] * (1.0 - refine)).detach().clamp(0, 1)
    zero = torch.zeros_like(L_alpha)
    with torch.no_grad():
        tris_vis = torch.where(trimasks.bool(), torch.full_like(gts, 128.0 / 255.0), gts)
    return [L_alpha, zero, zero.clone(), scaled_imgs, tris_vis, alphas, comps, gts, fgs, bgs]
