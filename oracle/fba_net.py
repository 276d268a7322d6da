"""TEST INFRASTRUCTURE — CPU restatement of the reference's FBA base with the Temporal Attention Module
(`FullModel_VMD('vmn_fba')`, BASELINE config 5) and its losses.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path
(tcvom_amd/) never does.  Plain fp32 PyTorch over a flat `state` dict with the reference's state_dict keys
(203 tensors, 36 463 271 parameters).

Follows (under /root/reference):
  * weight-standardised conv, GroupNorm(32) ........ models/FBA/layers_WS.py:6-27
  * Bottleneck / ResNet-50 [3,4,6,3] ................ models/FBA/resnet_GN_WS.py:50-137
  * ResnetDilated (layer3 dilate 2, layer4 dilate 4) . models/FBA/models.py:183-236
  * 11-channel stem ................................. models/FBA/models.py:43-66
  * fba_decoder modules, fba_fusion ................. models/FBA/models.py:246-324
  * vmn_fba_decoder.forward ......................... models/VMN/VMN_FBA.py:19-59
  * trimap_transform / dt ........................... utils/utils.py:12-39
  * make_trimap (8 channels), preprocess ............ models/model.py:54-92
  * fba_single_image_loss ........................... models/model.py:129-197
  * L1_mask / L1_grad / exclusion_loss / LapLoss ..... utils/loss_func.py:9-22,42-58,63-90,101-158
  * FullModel_VMD.forward ........................... models/model.py:258-357
Pinned by tests/golden/fba_*.npz (generated from the reference itself by tests/golden/gen_golden.py; OpenCV is
not installed here, so the generator gives the reference scipy's exact Euclidean distance transform in place of
cv2.distanceTransform(DIST_L2, maskSize 0) — the same quantity up to fp32 rounding).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .tam import tam_forward
from .window import IMG_MEAN, IMG_STD, attention_loss

LAYERS = (('layer1', 3, 64), ('layer2', 4, 128), ('layer3', 6, 256), ('layer4', 3, 512))
PPM_SCALES = (1, 2, 3, 6)


# ----------------------------------------------------------------------------- layers
def ws_conv(state, name, x, stride=1, padding=0, dilation=1):
    """layers_WS.py:13-23: per-output-channel (w - mean) / (sqrt(var_unbiased + 1e-12) + 1e-5), then conv2d."""
    w = state[name + '.weight']
    w = w - w.mean(dim=(1, 2, 3), keepdim=True)
    std = torch.sqrt(torch.var(w.reshape(w.shape[0], -1), dim=1) + 1e-12).reshape(-1, 1, 1, 1) + 1e-5
    return F.conv2d(x, w / std, state.get(name + '.bias'), stride, padding, dilation)


def group_norm(state, name, x):
    return F.group_norm(x, 32, state[name + '.weight'], state[name + '.bias'], 1e-5)


def bottleneck(state, p, x, stride, dilation, has_down):
    """resnet_GN_WS.py:50-92; `stride` and `dilation` are those of conv2 AFTER ResnetDilated._nostride_dilate."""
    out = F.relu(group_norm(state, p + '.bn1', ws_conv(state, p + '.conv1', x)))
    out = F.relu(group_norm(state, p + '.bn2', ws_conv(state, p + '.conv2', out, stride, dilation, dilation)))
    out = group_norm(state, p + '.bn3', ws_conv(state, p + '.conv3', out))
    if has_down:
        x = group_norm(state, p + '.downsample.1', ws_conv(state, p + '.downsample.0', x, stride))
    return F.relu(out + x)


def block_geometry(layer, block):
    """(stride, dilation) of conv2 (and stride of the downsample conv) of `layer`.`block` in the os8 network:
    layer2.0 keeps its stride 2; the strided convs of layer3.0 / layer4.0 become stride 1 with dilation
    dilate // 2, every other 3x3 of layer3 / layer4 gets dilation 2 / 4 (models.py:203-217)."""
    if layer == 'layer1':
        return 1, 1
    if layer == 'layer2':
        return (2 if block == 0 else 1), 1
    dilate = 2 if layer == 'layer3' else 4
    return 1, (dilate // 2 if block == 0 else dilate)


def encoder(state, x):
    """ResnetDilated.forward (models.py:219-236): x [B,11,H,W] -> conv_out list of 6 (os1, os2, os4, os8 x3)."""
    conv_out = [x]
    x = F.relu(group_norm(state, 'encoder.bn1', ws_conv(state, 'encoder.conv1', x, 2, 3)))
    conv_out.append(x)
    x = F.max_pool2d(x, 3, 2, 1)
    for layer, blocks, _ in LAYERS:
        for b in range(blocks):
            stride, dil = block_geometry(layer, b)
            x = bottleneck(state, 'encoder.%s.%d' % (layer, b), x, stride, dil, b == 0)
        conv_out.append(x)
    return conv_out


def _ws_gn_lrelu(state, conv, norm, x, padding):
    return F.leaky_relu(group_norm(state, norm, ws_conv(state, conv, x, 1, padding)), 0.01)


def decoder_feature(state, conv_out):
    """vmn_fba_decoder.forward(extract_feature=True) (VMN_FBA.py:21-33): pyramid pooling + conv_up1, os8, 256 ch."""
    conv5 = conv_out[-1]
    size = conv5.shape[2:]
    ppm = [conv5]
    for i, s in enumerate(PPM_SCALES):
        y = _ws_gn_lrelu(state, 'decoder.ppm.%d.1' % i, 'decoder.ppm.%d.2' % i, F.adaptive_avg_pool2d(conv5, s), 0)
        ppm.append(F.interpolate(y, size, mode='bilinear', align_corners=False))
    x = _ws_gn_lrelu(state, 'decoder.conv_up1.0', 'decoder.conv_up1.1', torch.cat(ppm, 1), 1)
    return _ws_gn_lrelu(state, 'decoder.conv_up1.3', 'decoder.conv_up1.4', x, 1)


def fba_fusion(alpha, img, Fg, Bg):
    """models.py:246-255."""
    Fg = alpha * img + (1 - alpha ** 2) * Fg - alpha * (1 - alpha) * Bg
    Bg = (1 - alpha) * img + (2 * alpha - alpha ** 2) * Bg - alpha * (1 - alpha) * Fg       # uses the UPDATED F
    Fg = Fg.clamp(0, 1)
    Bg = Bg.clamp(0, 1)
    la = 0.1
    alpha = (alpha * la + ((img - Bg) * (Fg - Bg)).sum(1, keepdim=True)) / (((Fg - Bg) ** 2).sum(1, keepdim=True) + la)
    return alpha.clamp(0, 1), Fg, Bg


def decoder_tail(state, x, conv_out, img, two_chan_trimap):
    """vmn_fba_decoder.forward(extract_feature=False) after the TAM (VMN_FBA.py:36-59): -> [B,7,H,W]."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False)
    x = _ws_gn_lrelu(state, 'decoder.conv_up2.0', 'decoder.conv_up2.1', torch.cat((up(x), conv_out[-4]), 1), 1)
    x = _ws_gn_lrelu(state, 'decoder.conv_up3.0', 'decoder.conv_up3.1', torch.cat((up(x), conv_out[-5]), 1), 1)
    x = torch.cat((up(x), conv_out[-6][:, :3], img, two_chan_trimap), 1)
    conv = lambda n, t, p: F.conv2d(t, state['decoder.conv_up4.%d.weight' % n], state['decoder.conv_up4.%d.bias' % n], 1, p)
    out = conv(4, F.leaky_relu(conv(2, F.leaky_relu(conv(0, x, 1), 0.01), 1), 0.01), 0)
    alpha, Fg, Bg = fba_fusion(out[:, :1].clamp(0, 1), img, torch.sigmoid(out[:, 1:4]), torch.sigmoid(out[:, 4:7]))
    return torch.cat((alpha, Fg, Bg), 1)


# ----------------------------------------------------------------------------- façade: trimap channels
def distance_to_zero(x):
    """utils/utils.py:12-23 `dt`: per-pixel Euclidean distance to the nearest ZERO pixel of the uint8 image
    (x * 255).astype(uint8) (0 where the pixel itself is zero).  x [..., H, W]."""
    import scipy.ndimage as ndi
    arr = (x.detach().cpu().numpy() * 255).astype(np.uint8)
    flat = arr.reshape((-1,) + arr.shape[-2:])
    out = np.stack([ndi.distance_transform_edt(f != 0).astype(np.float32) if (f == 0).any()
                    else _all_nonzero_distance(f.shape) for f in flat])
    return torch.from_numpy(out.reshape(arr.shape)).float()


def _all_nonzero_distance(shape):
    # no zero pixel at all: scipy (the generator's stand-in) measures to an imaginary zero outside the image
    import scipy.ndimage as ndi
    return ndi.distance_transform_edt(np.ones(shape, dtype=np.uint8)).astype(np.float32)


def trimap_transform(trimap2):
    """utils/utils.py:25-39: trimap2 [B,S,2,H,W] (bg, fg indicator) -> 6 Gaussian click maps of the distance to the
    class, sigma = 0.02 / 0.08 / 0.16 * 320; a class that is absent from the whole tensor keeps zeros."""
    B, S, _, H, W = trimap2.shape
    clicks = torch.zeros(B, S, 6, H, W)
    for k in range(2):
        tk = trimap2[:, :, k]
        if (tk != 0).sum() > 0:
            d2 = -distance_to_zero(1.0 - tk) ** 2
            for j, s in enumerate((0.02, 0.08, 0.16)):
                clicks[:, :, 3 * k + j] = torch.exp(d2 / (2 * (s * 320) ** 2))
    return clicks


def make_trimap8(alpha, dilate_kernel, eps=0.0):
    """models/model.py:54-80 with TRIMAP_CHANNEL == 8 -> (tris [B,S,8,H,W] = 6 click maps + (bg, fg), dilated
    unknown mask [B,S,1,H,W])."""
    alpha = torch.where(alpha < eps, torch.zeros_like(alpha), alpha)
    alpha = torch.where(alpha > 1 - eps, torch.ones_like(alpha), alpha)
    unk = ((alpha > 0) & (alpha < 1)).float()
    B, S, _, H, W = unk.shape
    r = int(dilate_kernel)
    dil = F.max_pool2d(unk.reshape(B * S, 1, H, W), 2 * r + 1, 1, r).reshape(B, S, 1, H, W)
    tri1 = torch.where(dil > 0.5, torch.full_like(alpha, 255.0), alpha)
    tri2 = torch.cat([(tri1 == 0).float(), (tri1 == 1).float()], dim=2)
    return torch.cat([trimap_transform(tri2), tri2], dim=2), dil


# ----------------------------------------------------------------------------- losses
def l1(x, y, normalize):
    """L1_mask without a mask (utils/loss_func.py:19-22)."""
    return (x - y).abs().mean() if normalize else (x - y).abs().sum()


def _gradient(im):
    dy = F.pad(im[:, :, 1:, :] - im[:, :, :-1, :], (0, 0, 0, 1))
    dx = F.pad(im[:, :, :, 1:] - im[:, :, :, :-1], (0, 1, 0, 0))
    return dx, dy


def l1_grad(pred, gt, normalize, epsilon=1.001e-5):
    """utils/loss_func.py:49-58 without a mask."""
    fx, fy = _gradient(pred)
    tx, ty = _gradient(gt)
    return l1(torch.sqrt(fx ** 2 + fy ** 2 + epsilon), torch.sqrt(tx ** 2 + ty ** 2 + epsilon), normalize)


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
        lx.append((((sx1 ** 2) * (sx2 ** 2)).mean(dim=(1, 2, 3)) + epsilon) ** 0.25)
        ly.append((((sy1 ** 2) * (sy2 ** 2)).mean(dim=(1, 2, 3)) + epsilon) ** 0.25)
        img1 = F.avg_pool2d(img1, 2, 2)
        img2 = F.avg_pool2d(img2, 2, 2)
    red = torch.mean if normalize else torch.sum
    return red(sum(lx) / float(level)) + red(sum(ly) / float(level))


_GAUSS = torch.tensor([1., 4., 6., 4., 1.])
_GAUSS2D = (_GAUSS[:, None] * _GAUSS[None, :]) / 256.0


def _conv_gauss(img, scale=1.0):
    C = img.shape[1]
    k = (_GAUSS2D * scale).to(img).repeat(C, 1, 1, 1)
    return F.conv2d(F.pad(img, (2, 2, 2, 2), mode='reflect'), k, groups=C)


def laplacian_pyramid(img, levels=5):
    """utils/loss_func.py:114-147: Gaussian 5x5 (reflect) -> drop odd rows/cols -> zero-interleave -> 4x Gaussian."""
    cur, pyr = img, []
    for _ in range(levels):
        down = _conv_gauss(cur)[:, :, ::2, ::2]
        up = torch.zeros(down.shape[0], down.shape[1], down.shape[2] * 2, down.shape[3] * 2, dtype=img.dtype)
        up[:, :, ::2, ::2] = down
        pyr.append(cur - _conv_gauss(up, 4.0))
        cur = down
    return pyr


def lap_loss(img, tgt, normalize):
    """LapLoss.forward without a mask (utils/loss_func.py:149-158)."""
    loss = sum((2 ** lvl) * (a - b).abs().sum() for lvl, (a, b) in enumerate(zip(laplacian_pyramid(img), laplacian_pyramid(tgt))))
    return loss / float(tgt.numel()) if normalize else loss


def fba_single_image_loss(preds, trimasks, gts, fgs, bgs, imgs, normalize=True):
    """models/model.py:129-197 (start = 1, end = S - 1).  -> L_alpha_comp, L_lap, L_grad, alphas, comps, Fs, Bs."""
    S = preds.shape[1]
    La, Ll, Lg = [], [], []
    alphas, comps, Fs, Bs = [None] * S, [None] * S, [None] * S, [None] * S
    for c in range(1, S - 1):
        gt, img, fg, bg = gts[:, c], imgs[:, c], fgs[:, c], bgs[:, c]
        m = trimasks[:, c].bool()
        m3 = m.repeat(1, 3, 1, 1)
        refine = torch.where(m, preds[:, c, :1], gt)
        cF = torch.where(m3, preds[:, c, 1:4], fg)
        cB = torch.where(m3, preds[:, c, 4:7], bg)
        alphas[c], Fs[c], Bs[c] = refine, cF, cB
        comps[c] = cF * refine + cB * (1.0 - refine)
        L_a1 = l1(refine, gt, normalize)
        L_ac = l1(cF * gt + cB * (1.0 - gt), img, normalize)
        L_FBc = l1(fg * refine + bg * (1.0 - refine), img, normalize)
        L_FB1 = l1(cF, fg, normalize) + l1(cB, bg, normalize)
        La.append(L_a1 + L_ac + 0.25 * (L_FBc + L_FB1))
        Lg.append(l1_grad(refine, gt, normalize) + 0.25 * exclusion_loss(cF, cB, 3, normalize=normalize))
        Ll.append(lap_loss(refine, gt, normalize) + 0.25 * (lap_loss(cF, fg, normalize) + lap_loss(cB, bg, normalize)))
    n = float(len(La))
    for lst in (alphas, comps, Fs, Bs):
        lst[0] = lst[-1] = torch.zeros_like(lst[1])
    st = lambda lst: torch.stack(lst, dim=1)
    return sum(La) / n, sum(Ll) / n, sum(Lg) / n, st(alphas), st(comps), st(Fs), st(Bs)


def dtssd(pred, gt, trimasks, normalize=True, epsilon=1.001e-5):
    """_dtSSD of models/model.py:326-333 (the masked L1_mask)."""
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


# ----------------------------------------------------------------------------- the window
def fba_window_forward(state, a, fg, bg, window=7, dilate_kernel=12, reduction=1, att_thres=0.3, label_smooth=0.2,
                       eps=0.0, normalize=True):
    """FullModel_VMD('vmn_fba').forward(a, fg, bg) -> the reference's 12-item list + intermediates.
    GroupNorm has no running statistics and there is no SpectralNorm: train and eval mode compute the same."""
    B, S = a.shape[:2]
    with torch.no_grad():
        gts = a / 255.0
        fgs, bgs = fg.flip([2]) / 255.0, bg.flip([2]) / 255.0
        simgs = fgs * gts + bgs * (1.0 - gts)
        tris, trimasks = make_trimap8(gts, dilate_kernel, eps)
        mean = torch.tensor(IMG_MEAN).reshape(1, 1, 3, 1, 1)
        std = torch.tensor(IMG_STD).reshape(1, 1, 3, 1, 1)
        imgs = (simgs - mean) / std
    conv_outs, feats = [None] * S, [None] * S
    for s in range(S):
        conv_outs[s] = encoder(state, torch.cat([imgs[:, s], tris[:, s]], dim=1))
        feats[s] = decoder_feature(state, conv_outs[s])
    preds, attb, attf, small = [None] * S, [None] * S, [None] * S, [None] * S
    for s in range(1, S - 1):
        x, attb[s], attf[s], small[s] = tam_forward(state, 'decoder.fam', feats[s], feats[s - 1], feats[s + 1],
                                                    trimasks[:, s], window)
        preds[s] = decoder_tail(state, x, conv_outs[s], simgs[:, s], tris[:, s, -2:])
    preds[0] = preds[-1] = torch.zeros_like(preds[1])
    preds = torch.stack(preds, dim=1)
    L1, L2, L3, alphas, comps, Fs, Bs = fba_single_image_loss(preds, trimasks, gts, fgs, bgs, simgs, normalize)
    L_att = attention_loss(attb, attf, small, gts, window, att_thres, label_smooth)
    if S >= 5:
        L_dt = dtssd(alphas, gts, trimasks, normalize) + 0.25 * (dtssd(Fs, fgs, trimasks, normalize) +
                                                                dtssd(Bs, bgs, trimasks, normalize))
    else:
        L_dt = torch.zeros_like(L_att)
    with torch.no_grad():
        tris_vis = torch.where(trimasks.bool(), torch.full_like(gts, 128.0 / 255.0), gts)
    out = [L1, L2, L3, L_dt, L_att, simgs, tris_vis, alphas, comps, gts, Fs, Bs]
    return out, {'preds': preds, 'attb': attb, 'attf': attf, 'small_mask': small, 'features': feats, 'tris': tris,
                 'trimasks': trimasks, 'imgs': imgs, 'conv_outs': conv_outs}


def fba_single_forward(state, a, fg, bg, dilate_kernel=12, eps=0.0, normalize=True):
    """FullModel('fba').forward (models/model.py:199-246 with MattingModule, models/FBA/models.py:18-32): the FBA base WITHOUT
    the temporal module on the centre frame -> [L_alpha_comp, L_lap, L_grad, imgs, tris_vis, alphas, comps, gts, Fs, Bs]."""
    B, S = a.shape[:2]
    c = S // 2
    with torch.no_grad():
        gts = a / 255.0
        fgs, bgs = fg.flip([2]) / 255.0, bg.flip([2]) / 255.0
        simgs = fgs * gts + bgs * (1.0 - gts)
        tris, trimasks = make_trimap8(gts, dilate_kernel, eps)
        mean = torch.tensor(IMG_MEAN).reshape(1, 1, 3, 1, 1)
        std = torch.tensor(IMG_STD).reshape(1, 1, 3, 1, 1)
        imgs = (simgs - mean) / std
    conv_out = encoder(state, torch.cat([imgs[:, c], tris[:, c]], dim=1))
    pred = decoder_tail(state, decoder_feature(state, conv_out), conv_out, simgs[:, c], tris[:, c, -2:])
    preds = [torch.zeros_like(pred)] * S
    preds[c] = pred
    # fba_single_image_loss with start = c, end = c + 1: the same per-frame terms as the window loss on frame c only
    lo = torch.stack([torch.zeros_like(pred), pred, torch.zeros_like(pred)], dim=1)
    pick = lambda t: torch.stack([t[:, c]] * 3, dim=1)
    L1, L2, L3, al, co, Fs_, Bs_ = fba_single_image_loss(lo, pick(trimasks), pick(gts), pick(fgs), pick(bgs), pick(simgs), normalize)
    alphas, comps, Fs, Bs = torch.zeros_like(gts), torch.zeros_like(fgs), torch.zeros_like(fgs), torch.zeros_like(fgs)
    alphas[:, c], comps[:, c], Fs[:, c], Bs[:, c] = al[:, 1].detach(), co[:, 1].detach(), Fs_[:, 1].detach(), Bs_[:, 1].detach()
    with torch.no_grad():
        tris_vis = torch.where(trimasks.bool(), torch.full_like(gts, 128.0 / 255.0), gts)
    return [L1, L2, L3, simgs, tris_vis, alphas, comps, gts, Fs, Bs]
