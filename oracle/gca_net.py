"""Oracle: GCA-Matting encoder / decoder halves as used by VMN (`vmn_gca`).

Functional restatement over a flat ``state`` dict (reference state_dict keys).
Reference sources followed (all paths under /root/reference):
  * SpectralNorm ................. models/GCA/ops.py:12-80
  * GuidedCxtAtten ............... models/GCA/ops.py:83-259
  * ResNet_D / BasicBlock (enc) .. models/GCA/encoders/resnet_enc.py:17-145
  * ResGuidedCxtAtten ............ models/GCA/encoders/res_gca_enc.py:8-90
  * ResNet_D_Dec / BasicBlock .... models/GCA/decoders/resnet_dec.py:23-144
  * ResGuidedCxtAtten_FAM_Dec .... models/VMN/VMN_GCA.py:8-48

TEST INFRASTRUCTURE — never imported by the product path.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- spectral norm
def _unit(x, eps=1e-12):
    # models/GCA/ops.py:8-9
    return x / (x.norm() + eps)


def spectral_weight(state, prefix, training):
    """W_bar / sigma with one power iteration per *call* in train mode
    (models/GCA/ops.py:25-45,74-80).  Matrix height is weight.shape[0] — i.e.
    in_channels for ConvTranspose2d weights (ops.py:30).  u, v are re-bound
    in ``state`` and do not take part in autograd."""
    w = state[prefix + '.weight_bar']
    u = state[prefix + '.weight_u']
    v = state[prefix + '.weight_v']
    mat = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v_new = _unit(mat.t().mv(u))
            u_new = _unit(mat.mv(v_new))
        # the reference re-binds .data (ops.py:32-33): earlier calls' graphs keep their own u, v
        state[prefix + '.weight_u'] = u = u_new
        state[prefix + '.weight_v'] = v = v_new
    sigma = u.detach().dot(mat.mv(v.detach()))
    return w / sigma


# --------------------------------------------------------------------------- batch norm
def batch_norm(state, prefix, x, training):
    """nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1, unbiased running var."""
    if training:
        state[prefix + '.num_batches_tracked'] += 1
    return F.batch_norm(x, state[prefix + '.running_mean'], state[prefix + '.running_var'],
                        state[prefix + '.weight'], state[prefix + '.bias'],
                        training, BN_MOMENTUM, BN_EPS)


def _sn_conv(state, prefix, x, training, stride=1, padding=1):
    return F.conv2d(x, spectral_weight(state, prefix + '.module', training), None, stride, padding)


def _sn_convT(state, prefix, x, training):
    # ConvTranspose2d(k=4, s=2, p=1) (resnet_dec.py:33-34,77)
    return F.conv_transpose2d(x, spectral_weight(state, prefix + '.module', training), None, 2, 1)


# --------------------------------------------------------------------------- guided contextual attention
def _patch_matrix(x, kernel, stride, pad_lo, pad_hi):
    """Reflect-pad then im2col: [B, C*kernel*kernel, L] with rows ordered (c, ky, kx)."""
    xp = F.pad(x, (pad_lo, pad_hi, pad_lo, pad_hi), mode='reflect')
    return F.unfold(xp, kernel_size=kernel, stride=stride)


def gca_scales(unknown16):
    """softmax_scale = [clamp(sqrt(u/k), .1, 10), clamp(sqrt(k/u), .1, 10)]  (ops.py:139-143)."""
    um = unknown16.mean(dim=[2, 3])
    km = 1 - um
    return torch.cat([torch.clamp(torch.sqrt(um / km), 0.1, 10),
                      torch.clamp(torch.sqrt(km / um), 0.1, 10)], dim=1)


def guided_context_attention(state, prefix, f, alpha, unknown, training, rate=2, return_probs=False):
    """GuidedCxtAtten.forward (models/GCA/ops.py:106-229), dense matrix form.

    f       [B,128,h8,w8]  image guidance feature
    alpha   [B,128,h8,w8]  feature to propagate
    unknown [B,1,h8,w8]    unknown-region indicator at os8
    returns y [B,128,h8,w8], softmax_scale [B,2]

    With g = nearest_down2(guidance_conv(f)), w_i the raw reflect-padded 3x3x64
    patch of g at os16 position i, n_j = max(||w_j||, 1e-4), mm_j = [mean 3x3
    reflect patch of unknown16 at j > 0]:
        S[j,i] = <w_i, w_j>/n_j * (mm_j ? s0 : s1) - 1e4*[i==j]*mm_j
        P[:,i] = softmax_j S[j,i]
        y      = fold_{4x4,s2,p1}( V @ P ) / 4 , V_j = 4x4 stride-2 reflect patch of alpha
        out    = BN(conv1x1(y)) + alpha
    """
    B, C, h8, w8 = alpha.shape
    g = F.conv2d(f, state[prefix + '.guidance_conv.weight'], state[prefix + '.guidance_conv.bias'])
    g = g[:, :, ::rate, ::rate]                        # nearest, scale 1/2 (ops.py:121)
    unk = unknown[:, :, ::rate, ::rate]                # ops.py:137
    h, w = g.shape[2:]
    N = h * w
    scale = gca_scales(unk).to(alpha)                  # [B,2]

    Wp = _patch_matrix(g, 3, 1, 1, 1)                  # [B, 576, N]   raw patches (queries AND keys)
    nrm = Wp.pow(2).sum(dim=1, keepdim=True).sqrt()    # [B,1,N]
    Kh = Wp / torch.clamp(nrm, min=1e-4)               # ops.py:174-175
    mm = (_patch_matrix(unk, 3, 1, 1, 1).mean(dim=1) > 0).to(alpha)       # [B,N] key flags (ops.py:148-156)

    S = torch.bmm(Kh.transpose(1, 2), Wp)              # [B, N(j keys), N(i queries)]
    per_key = scale[:, 0:1] * mm + scale[:, 1:2] * (1 - mm)               # ops.py:186
    S = S * per_key.unsqueeze(2)
    S = S - 1e4 * torch.diag_embed(mm)                 # ops.py:159-161,188
    P = torch.softmax(S, dim=1)                        # over keys j

    V = _patch_matrix(alpha, 2 * rate, rate, 1, 1)     # [B, 128*16, N]  (ops.py:115-119, extract_patches 231-238)
    O = torch.bmm(V, P)                                # [B, 2048, N(i)]
    y = F.fold(O, output_size=(h8, w8), kernel_size=2 * rate, stride=rate, padding=1) / 4.   # ops.py:204

    y = F.conv2d(y, state[prefix + '.W.0.weight'])
    y = batch_norm(state, prefix + '.W.1', y, training)
    out = y + alpha                                    # ops.py:227
    if return_probs:
        return out, scale, P
    return out, scale


# --------------------------------------------------------------------------- encoder
def _enc_block(state, p, x, training, stride, has_down):
    """Encoder BasicBlock (resnet_enc.py:33-49): SN3x3-BN-ReLU-SN3x3-BN, +identity, ReLU."""
    out = _sn_conv(state, p + '.conv1', x, training, stride=stride)
    out = F.relu(batch_norm(state, p + '.bn1', out, training))
    out = _sn_conv(state, p + '.conv2', out, training)
    out = batch_norm(state, p + '.bn2', out, training)
    idt = x
    if has_down:                                       # AvgPool2d(2) + SN 1x1 + BN (resnet_enc.py:108-114)
        idt = F.avg_pool2d(x, 2, stride) if stride != 1 else x
        idt = _sn_conv(state, p + '.downsample.1', idt, training, padding=0)
        idt = batch_norm(state, p + '.downsample.2', idt, training)
    return F.relu(out + idt)


def _enc_layer(state, p, x, training, blocks, stride):
    for b in range(blocks):
        x = _enc_block(state, '%s.%d' % (p, b), x, training, stride if b == 0 else 1, b == 0 and stride != 1)
    return x


def _shortcut(state, p, x, training):
    """conv -> ReLU -> BN, twice: activation BEFORE norm (res_gca_enc.py:47-55)."""
    x = batch_norm(state, p + '.2', F.relu(_sn_conv(state, p + '.0', x, training)), training)
    x = batch_norm(state, p + '.5', F.relu(_sn_conv(state, p + '.3', x, training)), training)
    return x


def _guidance_head(state, p, img, training):
    """3x [ReflectionPad(1), SN conv3x3 s2 p0, ReLU, BN] (res_gca_enc.py:20-33)."""
    x = img
    for conv_i, bn_i in ((1, 3), (5, 7), (9, 11)):
        x = F.pad(x, (1, 1, 1, 1), mode='reflect')
        x = _sn_conv(state, '%s.%d' % (p, conv_i), x, training, stride=2, padding=0)
        x = batch_norm(state, '%s.%d' % (p, bn_i), F.relu(x), training)
    return x


def encoder_frame(state, x, training, prefix='encoder'):
    """ResGuidedCxtAtten.forward (res_gca_enc.py:57-90) on ONE frame.

    x [B,6,H,W] = normalised RGB + one-hot trimap {bg,unk,fg}.
    returns (embedding [B,512,H/32,W/32], mid dict)."""
    e = prefix
    out = F.relu(batch_norm(state, e + '.bn1', _sn_conv(state, e + '.conv1', x, training, stride=2), training))
    x1 = F.relu(batch_norm(state, e + '.bn2', _sn_conv(state, e + '.conv2', out, training), training))
    out = F.relu(batch_norm(state, e + '.bn3', _sn_conv(state, e + '.conv3', x1, training, stride=2), training))

    im_fea = _guidance_head(state, e + '.guidance_head', x[:, :3], training)
    unknown = x[:, 4:5, ::8, ::8]                      # nearest 1/8 of the "unknown" one-hot channel (:70-71)

    x2 = _enc_layer(state, e + '.layer1', out, training, 3, 1)
    x3 = _enc_layer(state, e + '.layer2', x2, training, 4, 2)
    x3, _ = guided_context_attention(state, e + '.gca', im_fea, x3, unknown, training)
    x4 = _enc_layer(state, e + '.layer3', x3, training, 4, 2)
    emb = _enc_layer(state, e + '.layer_bottleneck', x4, training, 2, 2)

    fea = tuple(_shortcut(state, '%s.shortcut.%d' % (e, i), t, training)
                for i, t in enumerate((x, x1, x2, x3, x4)))
    return emb, {'shortcut': fea, 'image_fea': im_fea, 'unknown': unknown}


# --------------------------------------------------------------------------- decoder
def _dec_block(state, p, x, training, up):
    """Decoder BasicBlock (resnet_dec.py:26-59): [ConvT4x4s2 | conv3x3] BN Leaky conv3x3 BN, +id, Leaky."""
    if up:
        out = _sn_convT(state, p + '.conv1', x, training)
    else:
        out = _sn_conv(state, p + '.conv1', x, training)
    out = F.leaky_relu(batch_norm(state, p + '.bn1', out, training), 0.2)
    out = _sn_conv(state, p + '.conv2', out, training)
    out = batch_norm(state, p + '.bn2', out, training)
    idt = x
    if up:                                             # NearestUp x2 + SN 1x1 + BN (resnet_dec.py:112-118)
        idt = F.interpolate(x, scale_factor=2, mode='nearest')
        idt = _sn_conv(state, p + '.upsample.1', idt, training, padding=0)
        idt = batch_norm(state, p + '.upsample.2', idt, training)
    return F.leaky_relu(out + idt, 0.2)


def _dec_layer(state, p, x, training, blocks):
    for b in range(blocks):
        x = _dec_block(state, '%s.%d' % (p, b), x, training, b == 0)
    return x


def decoder_front(state, emb, mid, training, prefix='decoder'):
    """extract_feature=True branch (VMN_GCA.py:27-34): feature handed to TAM, [B,128,H/8,W/8]."""
    d = prefix
    fea1, fea2, fea3, fea4, fea5 = mid['shortcut']
    x = _dec_layer(state, d + '.layer1', emb, training, 2) + fea5
    x = _dec_layer(state, d + '.layer2', x, training, 3) + fea4
    x, _ = guided_context_attention(state, d + '.gca', mid['image_fea'], x, mid['unknown'], training)
    return x


def decoder_tail(state, x, mid, training, prefix='decoder'):
    """Everything after the TAM (VMN_GCA.py:39-47): alpha [B,1,H,W]."""
    d = prefix
    fea1, fea2, fea3, fea4, fea5 = mid['shortcut']
    x = _dec_layer(state, d + '.layer3', x, training, 3) + fea3
    x = _dec_layer(state, d + '.layer4', x, training, 2) + fea2
    x = _sn_convT(state, d + '.conv1', x, training)
    x = F.leaky_relu(batch_norm(state, d + '.bn1', x, training), 0.2) + fea1
    x = F.conv2d(x, state[d + '.conv2.weight'], state[d + '.conv2.bias'], 1, 1)
    return (torch.tanh(x) + 1.0) / 2.0
