"""IndexNet base + TAM (`vmn_index`, models/VMN/__init__.py:22-24) on the HIP path.

Replaces  models/Index/net.py:25-83 (InvertedResidual), :85-249 (IndexMattingEncoder, output_stride 32: the strides of the
MobileNetV2 settings live in the index blocks, every conv runs at stride 1 / dilation 1), models/Index/hlindex.py:120-168
(DepthwiseM2OIndexBlock), hlaspp.py:87-142 (ASPP), hldecoder.py:115-135 (IndexedUpsamlping), net.py:16-22,252-280 (pred,
IndexMattingDecoder) and models/VMN/VMN_Index.py:7-28 (the decoder split around the TAM at os8, 32 channels).

The nn.Module tree mirrors the reference's so that state_dict keys and shapes are identical (pinned by
tests/golden/vmn_index_state_keys.npz); the modules only hold parameters, the work is done by
  * the conv engine for the 1x1 / 4x4 stride-2 / 5x5 / 3x3 convs (any C % 8 == 0) with BatchNorm + ReLU6 in the BN kernels,
  * `ops.dw_bn_act` (csrc/depthwise.hip) for the depthwise 3x3 convs,
  * `ops.head_conv` for pred[0][0] (32 -> 1),
  * `ops.index_pool` / `ops.index_up` (csrc/indexnet.hip) for the index normalisation + indexed pooling and the indexed up-sampling
    + concat, each with an analytic backward kernel,
  * small tensor expressions for what is left: the image-pooling branch of the ASPP ([N, 320] vectors), the zero ring of
    `fixed_padding`, the ASPP concat and dropout, and the 1-channel tail of `pred` (BatchNorm(1) + ReLU6 + 5x5 conv on a fp32 map).
`fixed_padding` (net.py:63-69) pads the BLOCK INPUT, so the 1x1 expand conv and its BatchNorm see the zero ring (it enters the
batch statistics, and after BN + ReLU6 the ring is relu6(shift), not zero): the padded tensor is materialised here as well.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ops import ACT_NONE, ACT_RELU6, ConvCfg, DwCfg
from .weights import ConvSpec, WeightBank

# expand_ratio, input channels, output channels, blocks (net.py:108-116)
SETTINGS = ((1, 32, 16, 1), (6, 16, 24, 2), (6, 24, 32, 3), (6, 32, 64, 4), (6, 64, 96, 3), (6, 96, 160, 3), (6, 160, 320, 1))
ASPP_DILATIONS = (1, 2, 4, 8)
H16 = ops.H16


def _conv_bn(inp, oup, k):
    """hlconv.py:36-41."""
    return nn.Sequential(nn.Conv2d(inp, oup, k, 1, padding=k // 2, bias=False), nn.BatchNorm2d(oup), nn.ReLU6(inplace=True))


class _Reg(object):
    """Registers conv / depthwise sites of a module tree with the bank under the reference's parameter names."""

    def __init__(self, bank, group):
        self.bank, self.group, self.cfgs = bank, group, {}

    def conv(self, name, conv, bn, act, needs_dgrad=True):
        spec = ConvSpec(name, conv.weight, None, None, None, False, conv.stride[0], conv.padding[0], self.group, needs_dgrad)
        self.bank.register(spec)
        self.cfgs[name] = ConvCfg(self.bank, spec, bn=bn, act=act)

    def dw(self, name, conv, bn, dilation, pad):
        self.cfgs[name] = DwCfg(self.bank, conv.weight, bn, dilation, pad, ACT_RELU6)


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, expand):
        super().__init__()
        hidden = round(inp * expand)
        self.inp, self.oup, self.expand = inp, oup, expand
        layers = []
        if expand != 1:
            layers += [nn.Conv2d(inp, hidden, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden), nn.ReLU6(inplace=True)]
        layers += [nn.Conv2d(hidden, hidden, 3, 1, 0, 1, groups=hidden, bias=False), nn.BatchNorm2d(hidden), nn.ReLU6(inplace=True),
                   nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def register(self, reg, name):
        c = self.conv
        o = 0
        if self.expand != 1:
            reg.conv(name + '.conv.0', c[0], c[1], ACT_RELU6)
            o = 3
        reg.dw(name + '.conv.%d' % o, c[o], c[o + 1], 1, 0)
        reg.conv(name + '.conv.%d' % (o + 3), c[o + 3], c[o + 4], ACT_NONE)
        self._names = (name + '.conv.0' if self.expand != 1 else None, name + '.conv.%d' % o, name + '.conv.%d' % (o + 3))

    def run(self, cf, x, token, training):
        pw, dw, lin = self._names
        t = F.pad(x, (0, 0, 1, 1, 1, 1))                              # fixed_padding on NHWC
        if pw is not None:
            t = ops.conv_bn_act(cf[pw], t, token, training)
        t = ops.dw_bn_act(cf[dw], t, token, training)
        return ops.conv_bn_act(cf[lin], t, token, training, res1=x if self.inp == self.oup else None)


class DepthwiseM2OIndexBlock(nn.Module):
    """hlindex.py:120-168 with use_nonlinear = use_context = True."""

    def __init__(self, inp):
        super().__init__()
        for k in range(1, 5):
            setattr(self, 'indexnet%d' % k, nn.Sequential(nn.Conv2d(inp, inp, 4, 2, 1, bias=False), nn.BatchNorm2d(inp),
                                                          nn.ReLU6(inplace=True), nn.Conv2d(inp, inp, 1, 1, 0, bias=False)))

    def register(self, reg, name):
        for k in range(1, 5):
            seq = getattr(self, 'indexnet%d' % k)
            reg.conv('%s.indexnet%d.0' % (name, k), seq[0], seq[1], ACT_RELU6)
            reg.conv('%s.indexnet%d.3' % (name, k), seq[3], None, ACT_NONE)
        self._name = name

    def run(self, cf, x, token, training):
        """x [N, H, W, C] -> (idx_en * x, 4 * avg_pool2(idx_en * x), idx_de)."""
        ys = []
        for k in range(1, 5):
            t = ops.conv_bn_act(cf['%s.indexnet%d.0' % (self._name, k)], x, token, training)
            ys.append(ops.conv_bn_act(cf['%s.indexnet%d.3' % (self._name, k)], t, token, training))
        # sigmoid, softmax over the four branches, pixel shuffle, idx_en * x and the 2x2 sum in one kernel (csrc/indexnet.hip)
        return ops.index_pool(ys[0], ys[1], ys[2], ys[3], x)


class _ASPPModule(nn.Module):
    def __init__(self, inp, planes, kernel_size, dilation):
        super().__init__()
        if kernel_size == 1:
            self.atrous_conv = nn.Sequential(nn.Conv2d(inp, planes, 1, 1, 0, bias=False), nn.BatchNorm2d(planes), nn.ReLU6(inplace=True))
        else:
            self.atrous_conv = nn.Sequential(nn.Conv2d(inp, inp, 3, 1, padding=dilation, dilation=dilation, groups=inp, bias=False),
                                             nn.BatchNorm2d(inp), nn.ReLU6(inplace=True),
                                             nn.Conv2d(inp, planes, 1, 1, 0, bias=False), nn.BatchNorm2d(planes), nn.ReLU6(inplace=True))


class _AllReduceSum(torch.autograd.Function):
    """Sum over the ranks of a process group, differentiable: every rank's loss depends on the total, so the gradient with
    respect to a rank's addend is the sum of the ranks' gradients with respect to the total (nn.SyncBatchNorm's backward
    all-reduce of sum_dy / sum_dy_xmu, written once for any statistic)."""

    @staticmethod
    def forward(ctx, t, group):
        ctx.group = group
        out = t.clone()
        torch.distributed.all_reduce(out, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        torch.distributed.all_reduce(g, group=ctx.group)
        return g, None


def _sync_batchnorm_expr(xf, bn, sync):
    """Train-mode SyncBatchNorm of xf [nf, count, C] (fp32; normalised over dim 1 of ALL ranks; the nf frames are separate
    calls) as tensor expressions -- the two tiny BatchNorms of IndexNet that do not run through the bank's kernels
    (train_ddp.py:271-273 converts every BatchNorm).  Returns (normalised xf, mean [nf, C], biased var [nf, C], global count)."""
    nf, cnt, Cc = xf.shape
    x64 = xf.double()
    sums = torch.stack([x64.sum(1), (x64 * x64).sum(1)], 1)            # [nf, 2, C]
    # (a one-rank loop-back mailbox -- bench.py --sync-bn on one GPU, tests -- has no process group: the sums are already global)
    multi = sync.world > 1 and torch.distributed.is_available() and torch.distributed.is_initialized()
    total = _AllReduceSum.apply(sums, sync.group) if multi else sums
    n = cnt * sync.world
    mean = total[:, 0] / n
    var = (total[:, 1] / n - mean * mean).clamp_min(0.0)
    out = (x64 - mean.unsqueeze(1)) * torch.rsqrt(var.unsqueeze(1) + bn.eps)
    return out.float(), mean.detach().float(), var.detach().float(), n



class ASPP(nn.Module):
    """hlaspp.py:87-142, output_stride 32."""

    def __init__(self, inp, oup):
        super().__init__()
        self.aspp1 = _ASPPModule(inp, 256, 1, ASPP_DILATIONS[0])
        self.aspp2 = _ASPPModule(inp, 256, 3, ASPP_DILATIONS[1])
        self.aspp3 = _ASPPModule(inp, 256, 3, ASPP_DILATIONS[2])
        self.aspp4 = _ASPPModule(inp, 256, 3, ASPP_DILATIONS[3])
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(inp, 256, 1, 1, 0, bias=False), nn.BatchNorm2d(256),
                                             nn.ReLU6(inplace=True))
        self.bottleneck_conv = nn.Sequential(nn.Conv2d(256 * 5, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup), nn.ReLU6(inplace=True))
        self.dropout = nn.Dropout(0.5)

    def register(self, reg, name):
        a = self.aspp1.atrous_conv
        reg.conv(name + '.aspp1.atrous_conv.0', a[0], a[1], ACT_RELU6)
        for k, d in zip((2, 3, 4), ASPP_DILATIONS[1:]):
            a = getattr(self, 'aspp%d' % k).atrous_conv
            reg.dw('%s.aspp%d.atrous_conv.0' % (name, k), a[0], a[1], d, d)
            reg.conv('%s.aspp%d.atrous_conv.3' % (name, k), a[3], a[4], ACT_RELU6)
        b = self.bottleneck_conv
        reg.conv(name + '.bottleneck_conv.0', b[0], b[1], ACT_RELU6)
        self._name = name
        object.__setattr__(self, '_bank', reg.bank)

    def _image_pool(self, x, nf, training):
        """AdaptiveAvgPool2d(1) + 1x1 conv + BatchNorm over the B samples of each frame + ReLU6 on [N, 320] vectors (a few KB:
        tensor expressions).  The nf frames of a frame-batched call are separate BatchNorm calls."""
        conv, bn = self.global_avg_pool[1], self.global_avg_pool[2]
        bank = self._bank
        N, h, w, Cc = x.shape
        g = x.float().mean((1, 2)) @ ops.param_in(conv.weight, bank).reshape(conv.weight.shape[0], Cc).t()          # [N, 256]
        if training:
            B = N // nf
            sync = ops._sync_group(bn)
            Bg = B * (sync.world if sync is not None else 1)           # SyncBatchNorm: the clips of all ranks
            if Bg < 2:
                raise ValueError('Expected more than 1 value per channel when training (the ASPP image-pooling BatchNorm sees '
                                 '[B, 256, 1, 1]: vmn_index needs at least 2 clips per step in train mode, as in the reference)')
            gf = g.reshape(nf, B, -1)
            if sync is not None:
                out, mean, var, _ = _sync_batchnorm_expr(gf, bn, sync)
                out, mean, var = out.reshape(N, -1), mean.unsqueeze(1), var.unsqueeze(1)
            else:
                mean = gf.mean(1, keepdim=True)
                var = gf.var(1, unbiased=False, keepdim=True)
                out = ((gf - mean) / torch.sqrt(var + bn.eps)).reshape(N, -1)
            with torch.no_grad():
                m = 0.1 if bn.momentum is None else bn.momentum
                for f in range(nf):                                   # running statistics: one EMA step per call, in call order
                    bn.running_mean.mul_(1 - m).add_(m * mean[f, 0])
                    bn.running_var.mul_(1 - m).add_(m * var[f, 0] * (Bg / (Bg - 1.0)))
                bn.num_batches_tracked += nf
        else:
            out = (g - bn.running_mean) / torch.sqrt(bn.running_var + bn.eps)
        out = F.relu6(out * ops.param_in(bn.weight, bank) + ops.param_in(bn.bias, bank))
        return out.to(H16).reshape(N, 1, 1, -1).expand(N, h, w, out.shape[1])

    def run(self, cf, x, token, training, nf):
        n = self._name
        outs = [ops.conv_bn_act(cf[n + '.aspp1.atrous_conv.0'], x, token, training)]
        for k in (2, 3, 4):
            t = ops.dw_bn_act(cf['%s.aspp%d.atrous_conv.0' % (n, k)], x, token, training)
            outs.append(ops.conv_bn_act(cf['%s.aspp%d.atrous_conv.3' % (n, k)], t, token, training))
        outs.append(self._image_pool(x, nf, training))
        t = ops.conv_bn_act(cf[n + '.bottleneck_conv.0'], torch.cat(outs, -1), token, training)
        return F.dropout(t, 0.5, training and self.dropout.training)


class IndexMattingEncoder(nn.Module):
    def __init__(self, bank=None):
        super().__init__()
        self.layer0 = _conv_bn(4, 32, 3)
        for i, (t, p, c, n) in enumerate(SETTINGS, 1):
            setattr(self, 'layer%d' % i, nn.Sequential(*[InvertedResidual(p if j == 0 else c, c, t) for j in range(n)]))
        for i, c in ((0, 32), (2, 24), (3, 32), (4, 64), (6, 160)):
            setattr(self, 'index%d' % i, DepthwiseM2OIndexBlock(c))
        self.dconv_pp = ASPP(320, 160)
        reg = _Reg(bank, 'frame')
        reg.conv('encoder.layer0.0', self.layer0[0], self.layer0[1], ACT_RELU6, needs_dgrad=False)
        for i in range(1, 8):
            for j, blk in enumerate(getattr(self, 'layer%d' % i)):
                blk.register(reg, 'encoder.layer%d.%d' % (i, j))
        for i in (0, 2, 3, 4, 6):
            getattr(self, 'index%d' % i).register(reg, 'encoder.index%d' % i)
        self.dconv_pp.register(reg, 'encoder.dconv_pp')
        object.__setattr__(self, '_cfgs', reg.cfgs)
        object.__setattr__(self, '_bank', bank)

    def _layer(self, i, x, token, training):
        for blk in getattr(self, 'layer%d' % i):
            x = blk.run(self._cfgs, x, token, training)
        return x

    def run(self, x8, unk_u8, token, training):
        """x8 [F*B, H, W, 8] bf16 (normalised RGB + 1-channel trimap, 4 zero channels) -> (l at os32 [.., 160],
        {'skip': (l6, l5, l4, l3, l2, l1, l0), 'idx': (idx6, idx4, idx3, idx2, idx0)})."""
        cf = self._cfgs
        l0 = ops.conv_bn_act(cf['encoder.layer0.0'], x8, token, training)
        l0, l0p, idx0 = self.index0.run(cf, l0, token, training)
        l1 = self._layer(1, l0p, token, training)
        l2 = self._layer(2, l1, token, training)
        l2, l2p, idx2 = self.index2.run(cf, l2, token, training)
        l3 = self._layer(3, l2p, token, training)
        l3, l3p, idx3 = self.index3.run(cf, l3, token, training)
        l4 = self._layer(4, l3p, token, training)
        l4, l4p, idx4 = self.index4.run(cf, l4, token, training)
        l5 = self._layer(5, l4p, token, training)
        l6 = self._layer(6, l5, token, training)
        l6, l6p, idx6 = self.index6.run(cf, l6, token, training)
        l7 = self._layer(7, l6p, token, training)
        l = self.dconv_pp.run(cf, l7, token, training, self._bank.frames_per_op)
        return l, {'skip': (l6, l5, l4, l3, l2, l1, l0), 'idx': (idx6, idx4, idx3, idx2, idx0)}


class IndexedUpsamlping(nn.Module):
    def __init__(self, inp, oup):
        super().__init__()
        self.dconv = _conv_bn(inp, oup, 5)


def _indexed_cat(l_encode, l_low, indices):
    """hldecoder.py:128-133 on NHWC: indices * nearest x2 of l_encode, concatenated with l_low (csrc/indexnet.hip)."""
    return ops.index_up(l_encode, indices, l_low)


class IndexMattingDecoder_VMN(nn.Module):
    """models/VMN/VMN_Index.py:7-28; with_fam=False: the single-image IndexMattingDecoder (models/Index/net.py:252-280)."""

    def __init__(self, reduction, window, freeze_backbone=False, bank=None, with_fam=True):
        super().__init__()
        from .vmn import FeatureAggregationModule
        self.freeze_backbone = freeze_backbone
        for i, (inp, oup) in zip((6, 5, 4, 3, 2, 1, 0), ((320, 96), (192, 64), (128, 32), (64, 24), (48, 16), (32, 32), (64, 32))):
            setattr(self, 'decoder_layer%d' % i, IndexedUpsamlping(inp, oup))
        self.pred = nn.Sequential(_conv_bn(32, 1, 5), nn.Conv2d(1, 1, 5, 1, padding=2, bias=False))
        cfgs = {}
        for i in (6, 5, 4, 3, 2, 1, 0):
            reg = _Reg(bank, 'frame' if i >= 4 else 'tail')
            seq = getattr(self, 'decoder_layer%d' % i).dconv
            reg.conv('decoder.decoder_layer%d.dconv.0' % i, seq[0], seq[1], ACT_RELU6)
            cfgs.update(reg.cfgs)
        object.__setattr__(self, '_cfgs', cfgs)
        object.__setattr__(self, '_bank', bank)
        if with_fam:
            self.fam = FeatureAggregationModule(32, reduction, window, bank=bank, prefix='decoder.fam')
        self.register_buffer('_zero_bias', torch.zeros(1), persistent=False)

    def _up(self, i, l, low, idx, token, training):
        return ops.conv_bn_act(self._cfgs['decoder.decoder_layer%d.dconv.0' % i], _indexed_cat(l, low, idx), token, training)

    def run_front(self, l, mid, token, training):
        l6, l5, l4 = mid['skip'][:3]
        idx6, idx4 = mid['idx'][:2]
        l = self._up(6, l, l6, idx6, token, training)
        l = self._up(5, l, l5, None, token, training)
        return self._up(4, l, l4, idx4, token, training)

    def run_tail(self, x, xb, xf, mask_u8, mid, token, training):
        l3, l2, l1, l0 = mid['skip'][3:]
        idx3, idx2, idx0 = mid['idx'][2:]
        x, attb, attf = self.fam.run(x, xb, xf, mask_u8.contiguous(), token, training)
        return self.run_tail_single(x, mid, token, training), attb, attf

    def run_tail_single(self, x, mid, token, training):
        l3, l2, l1, l0 = mid['skip'][3:]
        idx3, idx2, idx0 = mid['idx'][2:]
        l = self._up(3, x, l3, idx3, token, training)
        l = self._up(2, l, l2, idx2, token, training)
        l = self._up(1, l, l1, None, token, training)
        l = self._up(0, l, l0, idx0, token, training)
        # pred: 5x5 conv 32 -> 1 (HIP), then BatchNorm over ONE channel + ReLU6 + 5x5 conv 1 -> 1 on a [N, 1, H, W] fp32 map
        conv0, bn, conv1 = self.pred[0][0], self.pred[0][1], self.pred[1]
        p = ops.head_conv(l, ops.param_in(conv0.weight, self._bank), self._zero_bias, 5, 2)
        nf = self._bank.frames_per_op
        N = p.shape[0]
        # BatchNorm over ONE channel as tensor expressions (a library BatchNorm kernel runs a single-channel map on one
        # workgroup); the frames of a frame-batched call are separate BatchNorm calls
        if training:
            pf = p.reshape(nf, -1)
            sync = ops._sync_group(bn)
            cnt = pf.shape[1]
            if sync is not None:
                pn, mean, var, cnt = _sync_batchnorm_expr(pf.unsqueeze(2), bn, sync)
                p = pn.reshape(p.shape)
            else:
                var, mean = torch.var_mean(pf, dim=1, unbiased=False, keepdim=True)
                p = ((pf - mean) * torch.rsqrt(var + bn.eps)).reshape(p.shape)
            with torch.no_grad():
                m = 0.1 if bn.momentum is None else bn.momentum
                for f in range(nf):                                    # running statistics: one EMA step per call, in call order
                    bn.running_mean.mul_(1 - m).add_(m * mean[f])
                    bn.running_var.mul_(1 - m).add_(m * var[f] * (cnt / (cnt - 1.0)))
                bn.num_batches_tracked += nf
        else:
            p = (p - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps)
        p = p * ops.param_in(bn.weight, self._bank) + ops.param_in(bn.bias, self._bank)
        return ops.conv5x5_c1(F.relu6(p), ops.param_in(conv1.weight, self._bank))


def build_vmn_index(agg_window, agg_reduction=1, freeze_backbone=False):
    """models/VMN/__init__.py:22-27: VMN(IndexMattingEncoder(), IndexMattingDecoder_VMN(...))."""
    from .vmn import VMN
    bank = WeightBank()
    enc = IndexMattingEncoder(bank=bank)
    dec = IndexMattingDecoder_VMN(agg_reduction, agg_window, freeze_backbone, bank=bank)
    return VMN(enc, dec, bank, freeze_backbone=freeze_backbone)


class IndexMatting(nn.Module):
    """The single-image base (models/Index/net.py:284-293): IndexMattingEncoder + IndexMattingDecoder, no temporal module."""

    def __init__(self):
        super().__init__()
        bank = WeightBank()
        object.__setattr__(self, '_bank', bank)
        self.encoder = IndexMattingEncoder(bank=bank)
        self.decoder = IndexMattingDecoder_VMN(1, 1, bank=bank, with_fam=False)

    def run(self, x8):
        """x8 [B, H, W, 8] bf16 (normalised RGB + 1-channel trimap) -> raw alpha prediction [B, 1, H, W] fp32."""
        from .weights import bank_token
        training = self.training
        token = bank_token(self._bank, 1, training, self)
        l, mid = self.encoder.run(x8, None, token, training)
        pred = self.decoder.run_tail_single(self.decoder.run_front(l, mid, token, training), mid, token, training)
        self._bank.flush_bn_counters()
        return pred
