"""TEST INFRASTRUCTURE — CPU restatement of the reference's IndexNet base (MobileNetV2 encoder with learned index blocks) and
of `vmn_index` (the same network split around the Temporal Attention Module at output stride 8).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path
(tcvom_amd/) never does.  Plain fp32 PyTorch over a flat `state` dict with the reference's state_dict keys.

Follows:  models/Index/net.py:25-83 (InvertedResidual: 1x1 expand, depthwise 3x3 behind `fixed_padding`, 1x1 project, BN, ReLU6),
          net.py:85-249 (IndexMattingEncoder(output_stride=32): every stride of the MobileNetV2 settings is moved into the
              index blocks -- `idx_en * l`, then 4 * avg_pool2d(2) -- so all convs run at stride 1, all dilations are 1),
          models/Index/hlindex.py:120-168 (DepthwiseM2OIndexBlock(use_nonlinear, use_context): four 4x4 stride-2 conv + BN +
              ReLU6 + 1x1 conv branches, sigmoid, softmax over the four, pixel shuffle),
          models/Index/hlaspp.py:36-142 (ASPP, output_stride 32: dilations 1, 2, 4, 8, image-pooling branch, dropout 0.5),
          models/Index/hldecoder.py:115-135 (IndexedUpsamlping: indices * nearest x2, concat, 5x5 conv + BN + ReLU6),
          models/Index/hlconv.py:36-41, net.py:16-22 (conv_bn, pred), net.py:252-280 (IndexMattingDecoder),
          models/VMN/VMN_Index.py:7-28 (decoder layers 6, 5, 4 -> 32-channel os8 feature -> TAM -> layers 3, 2, 1, 0, pred),
          models/model.py:54-69,94-127,258-357 (one-channel trimap, single_image_loss, FullModel_VMD.forward).
Pinned by tests/golden/vmn_index_*.npz (generated from the reference itself by tests/golden/gen_golden.py: gen_vmn_index).
The ASPP dropout draws from torch's global RNG in train mode; the goldens are generated with that one module in eval mode
(`dropout=False` here), everything else in train mode.
"""
import torch
import torch.nn.functional as F

from .dim_net import l1_grad, make_trimap1
from .gca_net import batch_norm
from .window import l1_mask

# expand_ratio, input channels, output channels, blocks (net.py:108-116; strides and dilations all end up 1, see above)
SETTINGS = ((1, 32, 16, 1), (6, 16, 24, 2), (6, 24, 32, 3), (6, 32, 64, 4), (6, 64, 96, 3), (6, 96, 160, 3), (6, 160, 320, 1))
ASPP_DILATIONS = (1, 2, 4, 8)


def _bn6(state, name, x, training):
    return F.relu6(batch_norm(state, name, x, training))


def inverted_residual(state, p, x, inp, oup, expand, training):
    """net.py:25-83 with stride 1, dilation 1: fixed_padding pads by 1 and the depthwise conv runs unpadded."""
    hidden = round(inp * expand)
    t = F.pad(x, (1, 1, 1, 1))
    if expand == 1:
        t = _bn6(state, p + '.conv.1', F.conv2d(t, state[p + '.conv.0.weight'], None, 1, 0, 1, hidden), training)
        t = batch_norm(state, p + '.conv.4', F.conv2d(t, state[p + '.conv.3.weight']), training)
    else:
        t = _bn6(state, p + '.conv.1', F.conv2d(t, state[p + '.conv.0.weight']), training)
        t = _bn6(state, p + '.conv.4', F.conv2d(t, state[p + '.conv.3.weight'], None, 1, 0, 1, hidden), training)
        t = batch_norm(state, p + '.conv.7', F.conv2d(t, state[p + '.conv.6.weight']), training)
    # (the 1x1 expand conv sees the padded image: its border outputs are BN(0) = f(beta), and the unpadded depthwise conv
    #  consumes them -- this is NOT the same as zero padding in front of the depthwise conv)
    return x + t if inp == oup else t


def layer(state, p, x, setting, training):
    expand, inp, oup, n = setting
    for i in range(n):
        x = inverted_residual(state, '%s.%d' % (p, i), x, inp if i == 0 else oup, oup, expand, training)
    return x


def index_block(state, p, x, training):
    """hlindex.py:120-168 -> (idx_en, idx_de), both [B, C, H, W]."""
    B, C, H, W = x.shape
    ys = []
    for k in range(1, 5):
        q = '%s.indexnet%d' % (p, k)
        t = F.conv2d(x, state[q + '.0.weight'], None, 2, 1)
        t = _bn6(state, q + '.1', t, training)
        ys.append(F.conv2d(t, state[q + '.3.weight']).unsqueeze(2))
    y = torch.sigmoid(torch.cat(ys, dim=2))                        # [B, C, 4, H/2, W/2]
    z = F.softmax(y, dim=2)
    idx_en = F.pixel_shuffle(z.reshape(B, C * 4, H // 2, W // 2), 2)
    idx_de = F.pixel_shuffle(y.reshape(B, C * 4, H // 2, W // 2), 2)
    return idx_en, idx_de


def aspp(state, p, x, training, dropout):
    """hlaspp.py:87-142 (output_stride 32)."""
    outs = [_bn6(state, p + '.aspp1.atrous_conv.1', F.conv2d(x, state[p + '.aspp1.atrous_conv.0.weight']), training)]
    for k, d in zip((2, 3, 4), ASPP_DILATIONS[1:]):
        q = '%s.aspp%d.atrous_conv' % (p, k)
        t = _bn6(state, q + '.1', F.conv2d(x, state[q + '.0.weight'], None, 1, d, d, x.shape[1]), training)
        outs.append(_bn6(state, q + '.4', F.conv2d(t, state[q + '.3.weight']), training))
    g = F.adaptive_avg_pool2d(x, (1, 1))
    g = _bn6(state, p + '.global_avg_pool.2', F.conv2d(g, state[p + '.global_avg_pool.1.weight']), training)
    outs.append(F.interpolate(g, size=x.shape[2:], mode='nearest'))
    t = _bn6(state, p + '.bottleneck_conv.1', F.conv2d(torch.cat(outs, dim=1), state[p + '.bottleneck_conv.0.weight']), training)
    return F.dropout(t, 0.5, training and dropout)


def encoder(state, x, training, dropout=False, p='encoder'):
    """net.py:200-236: x [B, 4, H, W] -> [l, l6, idx6_de, l5, l4, idx4_de, l3, idx3_de, l2, idx2_de, l1, l0, idx0_de]."""
    def pooled(t, name):
        en, de = index_block(state, '%s.%s' % (p, name), t, training)
        t = en * t
        return t, 4 * F.avg_pool2d(t, (2, 2), stride=2), de
    l0 = _bn6(state, p + '.layer0.1', F.conv2d(x, state[p + '.layer0.0.weight'], None, 1, 1), training)
    l0, l0p, idx0 = pooled(l0, 'index0')
    l1 = layer(state, p + '.layer1', l0p, SETTINGS[0], training)
    l2 = layer(state, p + '.layer2', l1, SETTINGS[1], training)
    l2, l2p, idx2 = pooled(l2, 'index2')
    l3 = layer(state, p + '.layer3', l2p, SETTINGS[2], training)
    l3, l3p, idx3 = pooled(l3, 'index3')
    l4 = layer(state, p + '.layer4', l3p, SETTINGS[3], training)
    l4, l4p, idx4 = pooled(l4, 'index4')
    l5 = layer(state, p + '.layer5', l4p, SETTINGS[4], training)
    l6 = layer(state, p + '.layer6', l5, SETTINGS[5], training)
    l6, l6p, idx6 = pooled(l6, 'index6')
    l7 = layer(state, p + '.layer7', l6p, SETTINGS[6], training)
    l = aspp(state, p + '.dconv_pp', l7, training, dropout)
    return [l, l6, idx6, l5, l4, idx4, l3, idx3, l2, idx2, l1, l0, idx0]


def indexed_upsampling(state, p, l_encode, l_low, indices, training):
    """hldecoder.py:128-135."""
    if indices is not None:
        l_encode = indices * F.interpolate(l_encode, size=l_low.shape[2:], mode='nearest')
    t = F.conv2d(torch.cat((l_encode, l_low), dim=1), state[p + '.dconv.0.weight'], None, 1, 2)
    return _bn6(state, p + '.dconv.1', t, training)


def decoder_front(state, inputs, training, p='decoder'):
    """VMN_Index.py:15-21: os32 -> os16 -> os16 -> os8 (32 channels)."""
    l, l6, idx6, l5, l4, idx4 = inputs[:6]
    l = indexed_upsampling(state, p + '.decoder_layer6', l, l6, idx6, training)
    l = indexed_upsampling(state, p + '.decoder_layer5', l, l5, None, training)
    return indexed_upsampling(state, p + '.decoder_layer4', l, l4, idx4, training)


def decoder_tail(state, x, inputs, training, p='decoder'):
    """VMN_Index.py:23-28 after the TAM: os8 -> os4 -> os2 -> os2 -> os1 -> pred (conv 5x5 + BN + ReLU6, conv 5x5; no clamp)."""
    l3, idx3, l2, idx2, l1, l0, idx0 = inputs[6:13]
    l = indexed_upsampling(state, p + '.decoder_layer3', x, l3, idx3, training)
    l = indexed_upsampling(state, p + '.decoder_layer2', l, l2, idx2, training)
    l = indexed_upsampling(state, p + '.decoder_layer1', l, l1, None, training)
    l = indexed_upsampling(state, p + '.decoder_layer0', l, l0, idx0, training)
    l = _bn6(state, p + '.pred.0.1', F.conv2d(l, state[p + '.pred.0.0.weight'], None, 1, 2), training)
    return F.conv2d(l, state[p + '.pred.1.weight'], None, 1, 2)


def vmn_index_window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True, att_thres=0.3, label_smooth=0.2,
                             eps=0.0, dropout=False):
    """FullModel_VMD('vmn_index').forward -> the reference's 12-item list (models/model.py:258-357, non-GCA branch of
    single_image_loss), plus the raw predictions of the interior frames."""
    from .tam import tam_forward
    from .window import attention_loss, dtssd_loss
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(1, 1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(1, 1, 3, 1, 1)
    B, S = a.shape[:2]
    gts = a / 255.0
    fgs, bgs = fg.flip([2]) / 255.0, bg.flip([2]) / 255.0
    simgs = fgs * gts + bgs * (1.0 - gts)
    tris, trimasks = make_trimap1(gts, dilate_kernel, eps)
    x = torch.cat([(simgs - mean) / std, tris], dim=2)
    inputs, feats = [None] * S, [None] * S
    for s in range(S):
        inputs[s] = encoder(state, x[:, s], training, dropout)
        feats[s] = decoder_front(state, inputs[s], training)
    preds, attb, attf, small = [None] * S, [None] * S, [None] * S, [None] * S
    for s in range(1, S - 1):
        t, attb[s], attf[s], small[s] = tam_forward(state, 'decoder.fam', feats[s], feats[s - 1], feats[s + 1], trimasks[:, s], window)
        preds[s] = decoder_tail(state, t, inputs[s], training)
    La, Lc, Lg = [], [], []
    alphas, comps = [None] * S, [None] * S
    for c in range(1, S - 1):
        m = trimasks[:, c].float()
        refine = torch.where(m.bool(), preds[c], gts[:, c])
        comp = fgs[:, c] * refine + bgs[:, c] * (1.0 - refine)
        alphas[c], comps[c] = refine, comp
        La.append(l1_mask(refine, gts[:, c], m))
        Lc.append(l1_mask(comp, simgs[:, c], m))
        Lg.append(l1_grad(refine, gts[:, c], m))
    n = float(len(La))
    alphas[0] = alphas[-1] = torch.zeros_like(alphas[1])
    comps[0] = comps[-1] = torch.zeros_like(comps[1])
    alphas = torch.stack(alphas, dim=1).clamp(0, 1)
    comps = torch.stack(comps, dim=1).clamp(0, 1)
    L_att = attention_loss(attb, attf, small, gts, window, att_thres, label_smooth)
    L_dt = dtssd_loss(alphas, gts, trimasks)
    return [sum(La) / n, sum(Lc) / n, sum(Lg) / n, L_dt, L_att, simgs, tris, alphas, comps, gts, fgs, bgs], preds


def index_single_forward(state, a, fg, bg, dilate_kernel=12, training=True, eps=0.0, dropout=False):
    """FullModel('index').forward (models/model.py:199-246; IndexMatting of net.py:284-293 = encoder + IndexMattingDecoder,
    net.py:252-280, no TAM): the centre frame of the clip -> the reference's 10-item list."""
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(1, 1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(1, 1, 3, 1, 1)
    S = a.shape[1]
    c = S // 2
    gts = a / 255.0
    fgs, bgs = fg.flip([2]) / 255.0, bg.flip([2]) / 255.0
    simgs = fgs * gts + bgs * (1.0 - gts)
    tris, trimasks = make_trimap1(gts, dilate_kernel, eps)
    x = torch.cat([(simgs - mean) / std, tris], dim=2)[:, c]
    inputs = encoder(state, x, training, dropout)
    pred = decoder_tail(state, decoder_front(state, inputs, training), inputs, training)
    m = trimasks[:, c].float()
    refine = torch.where(m.bool(), pred, gts[:, c])
    comp = fgs[:, c] * refine + bgs[:, c] * (1.0 - refine)
    L_alpha = l1_mask(refine, gts[:, c], m)
    L_comp = l1_mask(comp, simgs[:, c], m)
    L_grad = l1_grad(refine, gts[:, c], m)
    alphas = torch.zeros_like(gts)
    comps = torch.zeros_like(fgs)
    alphas[:, c] = refine.detach().clamp(0, 1)
    comps[:, c] = comp.detach().clamp(0, 1)
    return [L_alpha, L_comp, L_grad, simgs, tris, alphas, comps, gts, fgs, bgs], pred
