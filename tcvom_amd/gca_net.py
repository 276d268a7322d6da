"""GCA-Matting encoder / decoder for VMN on the HIP kernels.

Mirrors the reference module tree so that `NET.state_dict()` has the reference's 584 keys
(SURVEY.md §5): models/GCA/ops.py (SpectralNorm :12-80, GuidedCxtAtten :83-259),
models/GCA/encoders/resnet_enc.py + res_gca_enc.py, models/GCA/decoders/resnet_dec.py and
models/VMN/VMN_GCA.py.  The nn.Modules below only HOLD parameters/buffers under those names; the
math runs in libtcvom_hip.so through tcvom_amd.ops (NHWC bf16 activations, fp32 statistics).
"""
import math
import os as _os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_LEAKY, ConvCfg, H16
from .weights import ConvSpec, WeightBank, bank_token

TRIMAP_CHANNEL = 3
# The encoder stages of the FP16 ISLAND of the bf16 build (ops.F16_ISLAND): their forward runs on IEEE fp16 packed
# weights, IEEE fp16 activation twins and IEEE fp16 conv outputs -- tests/study_bf16_noise.py: >= 99 % of the bf16 storage noise of the alpha
# matte is injected in the stem, layer1 and layer2 (a third each from weights, conv outputs and stored activations).  Measured against the
# oracle, unknown-pixel alpha MSE at 256x320 / 512^2 / 544x960 / 1088x1920 (bound 1e-4): 1.21e-5 / 1.04e-5 / 9.98e-6 / 9.87e-6; the round-5
# scheme it replaces (doubled-tap stem weights, fp16 conv outputs in layer1 / layer2; removed, see DESIGN_HISTORY.md): 9.4e-5 / 7.7e-5 /
# 7.1e-5 / 6.75e-5 at the same step time (same-box A/B 22.76 / 22.90 ms against 22.69 / 22.73 ms).  TCVOM_NO_F16_ISLAND=1: plain bf16
# everywhere (1.6e-4 at 256x320: above the bound; kept as the A/B that shows what the island buys).  The fp16 build has no island.
ISLAND_LAYERS = ('conv1', 'conv2', 'conv3', 'layer1', 'layer2') if ops.F16_ISLAND else ()


# ----------------------------------------------------------------------------- parameter holders
class _ConvParams(nn.Module):
    """`.module` of a SpectralNorm wrapper: weight_u [h], weight_v [w], weight_bar (ops.py:57-72)."""

    def __init__(self, shape):
        super().__init__()
        h = shape[0]
        w = 1
        for s in shape[1:]:
            w *= s
        u = torch.randn(h)
        v = torch.randn(w)
        self.weight_u = nn.Parameter(u / (u.norm() + 1e-12), requires_grad=False)
        self.weight_v = nn.Parameter(v / (v.norm() + 1e-12), requires_grad=False)
        self.weight_bar = nn.Parameter(torch.empty(shape))
        nn.init.xavier_uniform_(self.weight_bar)


class SpectralNorm(nn.Module):
    """Holder with the reference's `<name>.module.weight_{u,v,bar}` layout.  `transposed` marks a
    ConvTranspose2d weight ([in, out, kh, kw]; the power-iteration matrix height is `in`, ops.py:30)."""

    def __init__(self, shape, stride=1, padding=1, transposed=False):
        super().__init__()
        self.module = _ConvParams(tuple(shape))
        self.stride, self.padding, self.transposed = stride, padding, transposed

    def spec(self, name, group, needs_dgrad=True):
        m = self.module
        return ConvSpec(name, m.weight_bar, m.weight_u, m.weight_v, None, self.transposed, self.stride,
                        self.padding, group, needs_dgrad)


def _plain_spec(name, conv, group, needs_dgrad=True):
    return ConvSpec(name, conv.weight, None, None, conv.bias, False, conv.stride[0], conv.padding[0], group, needs_dgrad)


def _conv3x3(cin, cout, stride=1):
    return SpectralNorm((cout, cin, 3, 3), stride=stride, padding=1)


def _conv1x1(cin, cout):
    return SpectralNorm((cout, cin, 1, 1), stride=1, padding=0)


# ----------------------------------------------------------------------------- guided contextual attention
class GuidedCxtAtten(nn.Module):
    """models/GCA/ops.py:83-259.  forward(f, alpha, unknown) -> (y, (offsets, softmax_scale)); the argmax
    `offsets` of the reference is a visualisation by-product that VMN discards (VMN_GCA.py:33) — None here."""

    def __init__(self, out_channels, guidance_channels, rate=2, bank=None, prefix='gca', group='frame'):
        super().__init__()
        assert rate == 2, 'only rate=2 (the value every reference model uses)'
        self.rate = rate
        self.guidance_conv = nn.Conv2d(guidance_channels, guidance_channels // 2, kernel_size=1)
        self.W = nn.Sequential(nn.Conv2d(out_channels, out_channels, kernel_size=1, bias=False),
                               nn.BatchNorm2d(out_channels))
        nn.init.xavier_uniform_(self.guidance_conv.weight)
        nn.init.constant_(self.guidance_conv.bias, 0)
        nn.init.xavier_uniform_(self.W[0].weight)
        nn.init.constant_(self.W[1].weight, 1e-3)
        nn.init.constant_(self.W[1].bias, 0)
        self._own_bank = bank is None
        bank = bank if bank is not None else WeightBank()
        object.__setattr__(self, '_bank', bank)
        sg = _plain_spec(prefix + '.guidance_conv', self.guidance_conv, group)
        sw = _plain_spec(prefix + '.W.0', self.W[0], group)
        bank.register(sg)
        bank.register(sw)
        self._cfg_g = ConvCfg(bank, sg)
        self._cfg_w = ConvCfg(bank, sw, bn=self.W[1], act=ACT_NONE)

    def run(self, im_fea, alpha, unk_u8, token, training):
        """NHWC bf16 fast path: im_fea [B,h8,w8,128], alpha [B,h8,w8,128], unk_u8 uint8 [B,h8,w8]."""
        g8 = ops.conv_bn_act(self._cfg_g, im_fea, token, training)
        y, scales = ops.gca_attention(g8, alpha, unk_u8)
        return ops.conv_bn_act(self._cfg_w, y, token, training, res1=alpha), scales

    def forward(self, f, alpha, unknown=None):
        assert self._own_bank, 'use .run() inside a network'
        training = self.training
        token = bank_token(self._bank, 1, training, self)
        to_nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(H16)
        if unknown is None:
            unknown = torch.ones_like(alpha[:, :1])
        y, scales = self.run(to_nhwc(f), to_nhwc(alpha), (unknown[:, 0] != 0).to(torch.uint8).contiguous(), token, training)
        self._bank.flush_bn_counters()
        return y.permute(0, 3, 1, 2).float(), (None, scales)


# ----------------------------------------------------------------------------- encoder
class EncBasicBlock(nn.Module):
    """resnet_enc.py:17-49."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResGuidedCxtAtten(nn.Module):
    """Encoder: ResNet_D-29 + shortcut branches + guidance head + GCA (res_gca_enc.py:8-90)."""

    def __init__(self, layers=(3, 4, 4, 2), bank=None):
        super().__init__()
        bank = bank if bank is not None else WeightBank()
        object.__setattr__(self, '_bank', bank)
        cin = 3 + TRIMAP_CHANNEL
        self.conv1 = _conv3x3(cin, 32, 2)
        self.conv2 = _conv3x3(32, 32, 1)
        self.conv3 = _conv3x3(32, 64, 2)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(32), nn.BatchNorm2d(32), nn.BatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer_bottleneck = self._make_layer(512, layers[3], 2)
        self.shortcut = nn.ModuleList()
        for cin_s, cout_s in ((cin, 32), (32, 32), (64, 64), (128, 128), (256, 256)):
            self.shortcut.append(nn.Sequential(_conv3x3(cin_s, cout_s), nn.ReLU(inplace=True), nn.BatchNorm2d(cout_s),
                                               _conv3x3(cout_s, cout_s), nn.ReLU(inplace=True), nn.BatchNorm2d(cout_s)))
        gh = []
        for ci, co in ((3, 16), (16, 32), (32, 128)):
            gh += [nn.ReflectionPad2d(1), SpectralNorm((co, ci, 3, 3), stride=2, padding=0), nn.ReLU(inplace=True),
                   nn.BatchNorm2d(co)]
        self.guidance_head = nn.Sequential(*gh)
        # zero-init of the last BN of each residual branch and of the trimap input channels
        # (resnet_enc.py:96-101)
        for m in self.modules():
            if isinstance(m, EncBasicBlock):
                nn.init.constant_(m.bn2.weight, 0)
        with torch.no_grad():
            self.conv1.module.weight_bar[:, 3:] = 0
        self.gca = GuidedCxtAtten(128, 128, bank=bank, prefix='encoder.gca')
        self._register(bank)

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1:
            down = nn.Sequential(nn.AvgPool2d(2, stride), _conv1x1(self.inplanes, planes), nn.BatchNorm2d(planes))
        layers = [EncBasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(EncBasicBlock(planes, planes, 1, None))
        return nn.Sequential(*layers)

    def _register(self, bank):
        def reg(name, sn, bn, act=ACT_NONE, pre_relu=False, needs_dgrad=True):
            spec = sn.spec('encoder.' + name, 'frame', needs_dgrad)
            spec.f16 = name.split('.')[0] in ISLAND_LAYERS          # forward in IEEE fp16 (the fp16 island of the bf16 build)
            bank.register(spec)
            return ConvCfg(bank, spec, bn=bn, act=act, pre_relu=pre_relu)
        self._stem = [reg('conv1', self.conv1, self.bn1, ACT_RELU, needs_dgrad=False),
                      reg('conv2', self.conv2, self.bn2, ACT_RELU),
                      reg('conv3', self.conv3, self.bn3, ACT_RELU)]
        self._layers = []
        for lname in ('layer1', 'layer2', 'layer3', 'layer_bottleneck'):
            blocks = []
            for i, blk in enumerate(getattr(self, lname)):
                p = '%s.%d' % (lname, i)
                c1 = reg(p + '.conv1', blk.conv1, blk.bn1, ACT_RELU)
                c2 = reg(p + '.conv2', blk.conv2, blk.bn2, ACT_RELU)
                dn = reg(p + '.downsample.1', blk.downsample[1], blk.downsample[2]) if blk.downsample is not None else None
                blocks.append((c1, c2, dn))
            self._layers.append(blocks)
        self._short = []
        for i, sc in enumerate(self.shortcut):
            p = 'shortcut.%d' % i
            self._short.append((reg(p + '.0', sc[0], sc[2], pre_relu=True, needs_dgrad=(i != 0)),
                                reg(p + '.3', sc[3], sc[5], pre_relu=True)))
        for a, b in self._short[:3]:            # fea1 .. fea3 (os1, os2, os4) feed the decoder tail only (VMN_GCA.py:39-47)
            a.tail_only = b.tail_only = True
            b.tail_last = True
        gh = self.guidance_head
        self._guid = [reg('guidance_head.1', gh[1], gh[3], pre_relu=True, needs_dgrad=False),
                      reg('guidance_head.5', gh[5], gh[7], pre_relu=True),
                      reg('guidance_head.9', gh[9], gh[11], pre_relu=True)]

    @staticmethod
    def _run_layer(blocks, x, token, training):
        for c1, c2, dn in blocks:
            idt = x
            if dn is not None:
                idt = ops.conv_bn_act(dn, ops.avgpool2(x), token, training)
            o = ops.conv_bn_act(c1, x, token, training)
            x = ops.conv_bn_act(c2, o, token, training, res1=idt)
        return x

    def run(self, x8, unk_u8, token, training):
        """x8: [B,H,W,8] bf16 (normalised RGB, one-hot trimap, 2 zero channels); unk_u8: uint8 [B,H/8,W/8].
        Returns (embedding, mid) like ResGuidedCxtAtten.forward (res_gca_enc.py:57-90)."""
        cba = ops.conv_bn_act
        o = cba(self._stem[0], x8, token, training)
        x1 = cba(self._stem[1], o, token, training)
        o = cba(self._stem[2], x1, token, training)
        im = x8
        for cfg in self._guid:
            im = cba(cfg, ops.reflect_pad1(im), token, training)
        x2 = self._run_layer(self._layers[0], o, token, training)
        x3 = self._run_layer(self._layers[1], x2, token, training)
        x3, _ = self.gca.run(im, x3, unk_u8, token, training)
        x4 = self._run_layer(self._layers[2], x3, token, training)
        emb = self._run_layer(self._layers[3], x4, token, training)
        feas = []
        for (a, b), t in zip(self._short, (x8, x1, x2, x3, x4)):
            feas.append(cba(b, cba(a, t, token, training), token, training))
        return emb, {'shortcut': tuple(feas), 'image_fea': im, 'unknown': unk_u8}


def resnet_gca_encoder_29(bank=None):
    return ResGuidedCxtAtten((3, 4, 4, 2), bank=bank)


# ----------------------------------------------------------------------------- decoder
class DecBasicBlock(nn.Module):
    """resnet_dec.py:23-59: first block of a layer up-samples with ConvTranspose2d(4, 2, 1)."""

    def __init__(self, inplanes, planes, up):
        super().__init__()
        if up:
            self.conv1 = SpectralNorm((inplanes, inplanes, 4, 4), stride=2, padding=1, transposed=True)
        else:
            self.conv1 = _conv3x3(inplanes, inplanes)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.conv2 = _conv3x3(inplanes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.upsample = None
        if up:
            self.upsample = nn.Sequential(nn.UpsamplingNearest2d(scale_factor=2), _conv1x1(inplanes, planes),
                                          nn.BatchNorm2d(planes))


class ResGuidedCxtAtten_FAM_Dec(nn.Module):
    """VMN decoder for GCA (models/VMN/VMN_GCA.py:8-48 over resnet_dec.py:62-144), split at os8:
    front = layer1, layer2, gca (per frame);  tail = TAM, layer3, layer4, conv1/bn1, conv2 (interior frames)."""

    def __init__(self, reduction, window, layers=(2, 3, 3, 2), freeze_backbone=False, bank=None, with_fam=True):
        super().__init__()
        assert reduction == 1, 'agg_reduction != 1 is inconsistent in the reference (SURVEY.md App. B 16)'
        from .vmn import FeatureAggregationModule
        bank = bank if bank is not None else WeightBank()
        object.__setattr__(self, '_bank', bank)
        self.freeze_backbone = freeze_backbone
        self.conv1 = SpectralNorm((32, 32, 4, 4), stride=2, padding=1, transposed=True)
        self.bn1 = nn.BatchNorm2d(32)
        self.conv2 = nn.Conv2d(32, 1, kernel_size=3, stride=1, padding=1)
        nn.init.xavier_uniform_(self.conv2.weight)
        self.inplanes = 512
        self.layer1 = self._make_layer(256, layers[0])
        self.layer2 = self._make_layer(128, layers[1])
        self.layer3 = self._make_layer(64, layers[2])
        self.layer4 = self._make_layer(32, layers[3])
        for m in self.modules():
            if isinstance(m, DecBasicBlock):
                nn.init.constant_(m.bn2.weight, 0)
        self.gca = GuidedCxtAtten(128, 128, bank=bank, prefix='decoder.gca')
        if with_fam:                            # without: the single-image decoder res_gca_decoder_22 (decoders/res_gca_dec.py)
            self.fam = FeatureAggregationModule(128, reduction, window, bank=bank, prefix='decoder.fam')
        self._register(bank)

    def _make_layer(self, planes, blocks):
        layers = [DecBasicBlock(self.inplanes, planes, True)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(DecBasicBlock(planes, planes, False))
        return nn.Sequential(*layers)

    def _register(self, bank):
        def reg(name, sn, bn, group, act=ACT_NONE, unbias_mult=1):
            spec = sn.spec('decoder.' + name, group)
            bank.register(spec)
            return ConvCfg(bank, spec, bn=bn, act=act, unbias_mult=unbias_mult)
        self._layers = []
        for lname, group in (('layer1', 'frame'), ('layer2', 'frame'), ('layer3', 'tail'), ('layer4', 'tail')):
            blocks = []
            for i, blk in enumerate(getattr(self, lname)):
                p = '%s.%d' % (lname, i)
                c1 = reg(p + '.conv1', blk.conv1, blk.bn1, group, ACT_LEAKY)
                c2 = reg(p + '.conv2', blk.conv2, blk.bn2, group, ACT_LEAKY)
                up = None
                if blk.upsample is not None:
                    # BN over the up-sampled map == BN over the low-res map (every value x4), except for the
                    # unbiased running_var correction -> unbias_mult=4
                    up = reg(p + '.upsample.1', blk.upsample[1], blk.upsample[2], group, unbias_mult=4)
                blocks.append((c1, c2, up))
            self._layers.append(blocks)
        self._out = reg('conv1', self.conv1, self.bn1, 'tail', ACT_LEAKY)

    @staticmethod
    def _run_layer(blocks, x, token, training, skip):
        n = len(blocks)
        for i, (c1, c2, up) in enumerate(blocks):
            idt = x
            if up is not None:
                idt = ops.upsample2(ops.conv_bn_act(up, x, token, training))
            o = ops.conv_bn_act(c1, x, token, training)
            x = ops.conv_bn_act(c2, o, token, training, res1=idt, res2=skip if i == n - 1 else None)
        return x

    def run_front(self, emb, mid, token, training):
        """extract_feature=True branch (VMN_GCA.py:27-34)."""
        fea1, fea2, fea3, fea4, fea5 = mid['shortcut']
        x = self._run_layer(self._layers[0], emb, token, training, fea5)
        x = self._run_layer(self._layers[1], x, token, training, fea4)
        x, _ = self.gca.run(mid['image_fea'], x, mid['unknown'], token, training)
        return x

    def run_tail(self, x, xb, xf, mask_u8, mid, token, training):
        """extract_feature=False branch (VMN_GCA.py:35-47): alpha fp32 [B,1,H,W], attb, attf."""
        x, attb, attf = self.fam.run(x, xb, xf, mask_u8, token, training)
        return self.run_tail_single(x, mid, token, training), attb, attf

    def run_tail_single(self, x, mid, token, training):
        """layer3, layer4, conv1/bn1, conv2 + (tanh + 1) / 2 (resnet_dec.py:105-120): alpha fp32 [B,1,H,W]."""
        fea1, fea2, fea3, fea4, fea5 = mid['shortcut']
        x = self._run_layer(self._layers[2], x, token, training, fea3)
        x = self._run_layer(self._layers[3], x, token, training, fea2)
        x = ops.conv_bn_act(self._out, x, token, training, res2=fea1)
        return ops.head_conv(x, ops.param_in(self.conv2.weight, self._bank), ops.param_in(self.conv2.bias, self._bank))

    def train(self, mode=True):
        super().train(mode)
        if self.freeze_backbone:
            print('Set GCA decoder feature extraction part in eval() mode.')
            self.layer1.eval()
            self.layer2.eval()
            self.gca.eval()
        return self


class Generator(nn.Module):
    """models/GCA/generators.py:8-41: the single-image GCA matting network (FullModel('gca')): resnet_gca_encoder_29 +
    res_gca_decoder_22, no temporal module."""

    def __init__(self, encoder='resnet_gca_encoder_29', decoder='res_gca_decoder_22', alpha_only=True):
        super().__init__()
        assert encoder == 'resnet_gca_encoder_29' and decoder == 'res_gca_decoder_22' and alpha_only
        self.alpha_only = alpha_only
        bank = WeightBank()
        object.__setattr__(self, '_bank', bank)
        self.encoder = resnet_gca_encoder_29(bank=bank)
        self.decoder = ResGuidedCxtAtten_FAM_Dec(1, 1, bank=bank, with_fam=False)

    def run(self, x8, unk_u8):
        """x8 [B,H,W,8] bf16 (normalised RGB + one-hot trimap), unk_u8 uint8 [B,H/8,W/8] -> alpha fp32 [B,1,H,W]."""
        training = self.training
        token = bank_token(self._bank, 1, training, self)
        emb, mid = self.encoder.run(x8, unk_u8, token, training)
        x = self.decoder.run_front(emb, mid, token, training)
        alpha = self.decoder.run_tail_single(x, mid, token, training)
        self._bank.flush_bn_counters()
        return alpha


def GCA(encoder='resnet_gca_encoder_29', decoder='res_gca_decoder_22', alpha_only=True):
    return Generator(encoder, decoder, alpha_only)
