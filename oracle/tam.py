"""Oracle: Temporal Attention Module (reference `FeatureAggregationModule`,
/root/reference/models/VMN/VMN_model.py:9-68), dense masked formulation.

The reference loops over batch items, materialises F.unfold(k) [C, w*w, N],
gathers the unknown columns with torch.nonzero and scatters results back.  The
same numbers are obtained densely:

    logit[b, j, u] = <q[b,:,u], k[b,:,u+d_j]> / sqrt(C)   (zero-padded k, j = ky*w+kx)
    p              = softmax_j(logit)
    agg[b, :, u]   = sum_j p[j] * k[b,:,u+d_j]            (values ARE the keys, :53)
    both zeroed where mask[u] == 0; returned logits are the pre-softmax ones (:46-49)

TEST INFRASTRUCTURE — never imported by the product path.
"""
import math
import torch
import torch.nn.functional as F


def temporal_attention(q, k, mask, window):
    """q,k [B,C,H,W]; mask [B,1,H,W] bool/float.  -> (agg [B,C,H,W], logits [B,w*w,H*W])."""
    B, C, H, W = q.shape
    r = window // 2
    kp = F.pad(k, (r, r, r, r))                                    # zero pad: OOB keys take part with logit 0
    neigh = torch.stack([kp[:, :, dy:dy + H, dx:dx + W]
                         for dy in range(window) for dx in range(window)], dim=1)   # [B,w2,C,H,W]
    logits = (q.unsqueeze(1) * neigh).sum(dim=2) / math.sqrt(C)     # [B,w2,H,W]
    att = torch.softmax(logits, dim=1)
    agg = (att.unsqueeze(2) * neigh).sum(dim=1)                     # [B,C,H,W]
    m = mask.to(q.dtype)
    return agg * m, (logits * m).reshape(B, window * window, H * W)


def tam_forward(state, prefix, x, xb, xf, trimask, window):
    """FeatureAggregationModule.forward (VMN_model.py:18-68).

    x, xb, xf [B,C,h,w] features of the current / previous / next frame,
    trimask [B,1,8h,8w] full-res unknown mask.  key_conv is shared by both
    directions, q is computed once (:25,63-66).
    -> (v + agg_b + agg_f, attb [B,w2,hw], attf [B,w2,hw], small_mask bool [B,1,h,w])"""
    B, C, H, W = x.shape
    small = F.interpolate(trimask, size=(H, W), mode='nearest').bool()       # == trimask[..., ::8, ::8]
    conv = lambda name, t: F.conv2d(t, state['%s.%s.weight' % (prefix, name)],
                                    state['%s.%s.bias' % (prefix, name)], 1, 1)
    q = conv('query_conv', x)
    v = conv('value_conv', x)
    ab, lb = temporal_attention(q, conv('key_conv', xb), small, window)
    af, lf = temporal_attention(q, conv('key_conv', xf), small, window)
    return v + ab + af, lb, lf, small
