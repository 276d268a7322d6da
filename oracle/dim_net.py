"""TEST INFRASTRUCTURE — CPU restatement of the reference's DIM base and its single-image losses.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path
(tcvom_amd/) never does.  Plain fp32 PyTorch over a flat `state` dict with the reference's state_dict keys.

Follows:  models/DIM/vggnet.py:10-128 (DeepMatting.forward: VGG16-BN encoder with max-pool indices, conv6 7x7,
          dconv6 1x1, five max-unpool + 5x5 conv stages, alpha_pred 5x5 + clamp(0, 1))
          models/model.py:54-92,94-127,199-246 (make_trimap with one channel, single_image_loss, FullModel.forward)
          utils/loss_func.py:9-22,42-59 (L1_mask, get_gradient, L1_grad)
          models/VMN/VMN_DIM.py:6-136 (vmn_dim: the same network split around the TAM at os8)
Pinned by tests/golden/dim_*.npz and vmn_dim_*.npz (generated from the reference itself by tests/golden/gen_golden.py).
"""
import torch
import torch.nn.functional as F

from .gca_net import batch_norm
from .window import l1_mask

_STAGES = (('11', '12'), ('21', '22'), ('31', '32', '33'), ('41', '42', '43'), ('51', '52', '53'))
_DEC = ('dconv5', 'dconv4', 'dconv3', 'dconv2', 'dconv1')


def dim_forward(state, x, training):
    """x [B,4,H,W] -> alpha [B,1,H,W] (vggnet.py:78-128)."""
    idx = []
    for stage in _STAGES:
        for tag in stage:
            x = F.conv2d(x, state['conv%s.weight' % tag], state['conv%s.bias' % tag], 1, 1)
            x = F.relu(batch_norm(state, 'bn' + tag, x, training))
        x, i = F.max_pool2d(x, (2, 2), stride=2, return_indices=True)
        idx.append(i)
    x = F.relu(F.conv2d(x, state['conv6.weight'], state['conv6.bias'], 1, 3))
    x = F.relu(F.conv2d(x, state['dconv6.weight'], state['dconv6.bias'], 1, 0))
    for name, i in zip(_DEC, reversed(idx)):
        x = F.max_unpool2d(x, i, (2, 2), stride=2)
        x = F.relu(F.conv2d(x, state[name + '.weight'], state[name + '.bias'], 1, 2))
    return F.conv2d(x, state['alpha_pred.weight'], state['alpha_pred.bias'], 1, 2).clamp(0, 1)


def make_trimap1(alpha, dilate_kernel, eps=0.0):
    """models/model.py:54-69 with TRIMAP_CHANNEL == 1: (128/255 inside the dilated unknown region, alpha elsewhere)."""
    alpha = torch.where(alpha < eps, torch.zeros_like(alpha), alpha)
    alpha = torch.where(alpha > 1 - eps, torch.ones_like(alpha), alpha)
    masks = ((alpha > 0) & (alpha < 1.)).float()
    B = alpha.shape[0]
    r = int(dilate_kernel)
    tri = torch.stack([F.max_pool2d(masks[b], kernel_size=2 * r + 1, stride=1, padding=r) for b in range(B)])
    return torch.where(tri > 0.5, torch.full_like(alpha, 128.0 / 255.0), alpha), tri


def l1_grad(pred, gt, mask, epsilon=1.001e-5):
    """utils/loss_func.py:42-59."""
    def grad(im):
        dy = F.pad(im[:, :, 1:, :] - im[:, :, :-1, :], (0, 0, 0, 1))
        dx = F.pad(im[:, :, :, 1:] - im[:, :, :, :-1], (0, 1, 0, 0))
        return dx, dy
    fx, fy = grad(pred)
    tx, ty = grad(gt)
    return l1_mask(torch.sqrt(fx ** 2 + fy ** 2 + epsilon), torch.sqrt(tx ** 2 + ty ** 2 + epsilon), mask, epsilon)


def full_model_dim_forward(state, a, fg, bg, dilate_kernel=12, training=True, eps=0.0):
    """FullModel('dim').forward (models/model.py:199-246) -> the reference's 10-item list."""
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(1, 1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(1, 1, 3, 1, 1)
    S = a.shape[1]
    c = S // 2
    gts = a / 255.0
    fgs, bgs = fg.flip([2]) / 255.0, bg.flip([2]) / 255.0
    simgs = fgs * gts + bgs * (1.0 - gts)
    tris, trimasks = make_trimap1(gts, dilate_kernel, eps)
    imgs = (simgs - mean) / std
    pred = dim_forward(state, torch.cat([imgs, tris], dim=2)[:, c], training)
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


# ----------------------------------------------------------------------------- vmn_dim (models/VMN/VMN_DIM.py)
def vmn_dim_window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True, att_thres=0.3, label_smooth=0.2, eps=0.0):
    """FullModel_VMD('vmn_dim').forward -> the reference's 12-item list (models/model.py:258-357 with the non-GCA branch
    of single_image_loss: L_alpha, L_comp, L_grad on the interior frames)."""
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
    conv = lambda name, t, pad: F.conv2d(t, state[name + '.weight'], state[name + '.bias'], 1, pad)
    idxs, feats = [None] * S, [None] * S
    for s in range(S):                                               # DIMEncoder + DIMDecoder(extract_feature=True)
        t, idx = x[:, s], []
        for stage in _STAGES:
            for tag in stage:
                t = F.relu(batch_norm(state, 'encoder.bn' + tag, conv('encoder.conv' + tag, t, 1), training))
            t, i = F.max_pool2d(t, (2, 2), stride=2, return_indices=True)
            idx.append(i)
        t = F.relu(conv('encoder.conv6', t, 3))
        t = F.relu(conv('decoder.dconv6', t, 0))
        t = F.relu(conv('decoder.dconv5', F.max_unpool2d(t, idx[4], (2, 2), stride=2), 2))
        feats[s] = F.relu(conv('decoder.dconv4', F.max_unpool2d(t, idx[3], (2, 2), stride=2), 2))
        idxs[s] = idx
    preds, attb, attf, small = [None] * S, [None] * S, [None] * S, [None] * S
    for s in range(1, S - 1):
        t, attb[s], attf[s], small[s] = tam_forward(state, 'decoder.fam', feats[s], feats[s - 1], feats[s + 1], trimasks[:, s], window)
        for name, i in (('dconv3', 2), ('dconv2', 1), ('dconv1', 0)):
            t = F.relu(conv('decoder.' + name, F.max_unpool2d(t, idxs[s][i], (2, 2), stride=2), 2))
        preds[s] = conv('decoder.alpha_pred', t, 2).clamp(0, 1)
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
    return [sum(La) / n, sum(Lc) / n, sum(Lg) / n, L_dt, L_att, simgs, tris, alphas, comps, gts, fgs, bgs]
