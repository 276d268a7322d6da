"""FBA base with the Temporal Attention Module (BASELINE config 5, `FullModel_VMD('vmn_fba')`) on the HIP kernels.

Mirrors, with the reference's state_dict names (203 tensors, 36 463 271 parameters):
  * encoder = ResnetDilated(l_resnet50 GN+WS, dilate_scale=8) with the 11-channel stem
        models/FBA/resnet_GN_WS.py:50-137, models/FBA/models.py:43-66,183-236, models/FBA/layers_WS.py
  * decoder = vmn_fba_decoder (pyramid pooling, conv_up1..4, fusion) + FeatureAggregationModule(256)
        models/FBA/models.py:258-324, models/VMN/VMN_FBA.py:6-59
The module tree only HOLDS the parameters; the math runs in libtcvom_hip.so through tcvom_amd.ops on NHWC bf16:
  * weight standardisation is done once per step for all 60 convs by the weight bank (csrc/spectral.hip kind 8),
  * conv + GroupNorm(32) + ReLU / LeakyReLU(0.01) (+ residual) are the fused conv / norm ops of the GCA path, with the
    per-sample group statistics combined from the conv epilogue's channel sums (csrc/norm.hip gn_*),
  * every frame (and batch sample) of a window goes through a layer in ONE launch (frames_per_op),
  * the 7x7 stride-2 stem runs as a 4x4 conv over the 2x2 space-to-depth input that tcvom_fba_input writes,
  * the concat inputs of the decoder keep their 3072 / 320 channels (the conv engine takes any multiple of 8); only the 72-channel
    input of conv_up4 is zero-padded to 128 (9 taps x 128 and the 16 zero-padded taps 72 channels would need are the same work).
"""
import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LEAKY01, ACT_NONE, ACT_RELU, ConvCfg
from .vmn import FeatureAggregationModule
from .weights import ConvSpec, WeightBank, bank_token

PPM_SCALES = (1, 2, 3, 6)


def _gn(ch):
    return nn.GroupNorm(32, ch)


class Bottleneck(nn.Module):
    """models/FBA/resnet_GN_WS.py:50-92 (conv1 1x1, conv2 3x3 carrying the stride / dilation, conv3 1x1 x4)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = _gn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = _gn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _gn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResnetDilated(nn.Module):
    """models/FBA/models.py:183-236 over l_resnet50 ([3, 4, 6, 3] Bottlenecks): layer3 / layer4 lose their stride and get
    dilation 2 / 4 (their first blocks: dilation 1 / 2), so conv_out = [x, os2 64, os4 256, os8 512, os8 1024, os8 2048]."""

    def __init__(self, bank, in_channels=11):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = _gn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1, return_indices=True)
        cfgs = {}

        def reg(name, conv, norm, act, **kw):
            spec = ConvSpec(name, conv.weight, None, None, conv.bias, False, kw.pop('stride', 1), conv.padding[0], 'frame',
                            dilation=conv.dilation[0], ws=True, **kw)
            bank.register(spec)
            cfgs[name] = ConvCfg(bank, spec, bn=norm, act=act)

        reg('encoder.conv1', self.conv1, self.bn1, ACT_RELU, stride=2, stem=True, needs_dgrad=False)
        inplanes = 64
        for li, (planes, blocks, stride, dilate) in enumerate(((64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 1, 2), (512, 3, 1, 4)), 1):
            layer = []
            for b in range(blocks):
                first = b == 0
                down = None
                if first:
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), _gn(planes * 4))
                dil = 1 if dilate == 1 else (dilate // 2 if first else dilate)
                blk = Bottleneck(inplanes, planes, stride if first else 1, dil, down)
                p = 'encoder.layer%d.%d' % (li, b)
                reg(p + '.conv1', blk.conv1, blk.bn1, ACT_RELU)
                reg(p + '.conv2', blk.conv2, blk.bn2, ACT_RELU, stride=blk.conv2.stride[0])
                reg(p + '.conv3', blk.conv3, blk.bn3, ACT_RELU)           # ReLU after the residual add (res1)
                if first:
                    # a strided 1x1 conv reads every other pixel: sub-sample, then stride-1 conv
                    reg(p + '.downsample.0', down[0], down[1], ACT_NONE)
                layer.append(blk)
                inplanes = planes * 4
            setattr(self, 'layer%d' % li, nn.Sequential(*layer))
        object.__setattr__(self, '_cfgs', cfgs)

    def run(self, x2, token, training):
        """x2: [F, H/2, W/2, 64] bf16, the space-to-depth network input -> (os2 64, os4 256, os8 512, os8 1024, os8 2048)."""
        cf = self._cfgs
        c1 = ops.conv_bn_act(cf['encoder.conv1'], x2, token, training)
        x = ops.maxpool3s2(c1)
        outs = [c1]
        for li in range(1, 5):
            for b, blk in enumerate(getattr(self, 'layer%d' % li)):
                p = 'encoder.layer%d.%d' % (li, b)
                y = ops.conv_bn_act(cf[p + '.conv1'], x, token, training)
                y = ops.conv_bn_act(cf[p + '.conv2'], y, token, training)
                idt = x
                if blk.downsample is not None:
                    xs = x[:, ::2, ::2].contiguous() if blk.stride == 2 else x
                    idt = ops.conv_bn_act(cf[p + '.downsample.0'], xs, token, training)
                x = ops.conv_bn_act(cf[p + '.conv3'], y, token, training, res1=idt)
            outs.append(x)
        return outs


class vmn_fba_decoder(nn.Module):
    """models/VMN/VMN_FBA.py:6-59 over fba_decoder (models/FBA/models.py:258-324, batch_norm=False)."""

    def __init__(self, reduction, window, freeze_backbone=False, batch_norm=False, bank=None, with_fam=True):
        super().__init__()
        assert not batch_norm and bank is not None
        object.__setattr__(self, '_bank', bank)
        self.batch_norm = batch_norm
        self.freeze_backbone = freeze_backbone
        self.ppm = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(s), nn.Conv2d(2048, 256, 1, bias=True), _gn(256), nn.LeakyReLU())
                                  for s in PPM_SCALES])
        self.conv_up1 = nn.Sequential(nn.Conv2d(2048 + 4 * 256, 256, 3, padding=1, bias=True), _gn(256), nn.LeakyReLU(),
                                      nn.Conv2d(256, 256, 3, padding=1), _gn(256), nn.LeakyReLU())
        self.conv_up2 = nn.Sequential(nn.Conv2d(256 + 256, 256, 3, padding=1, bias=True), _gn(256), nn.LeakyReLU())
        self.conv_up3 = nn.Sequential(nn.Conv2d(256 + 64, 64, 3, padding=1, bias=True), _gn(64), nn.LeakyReLU())
        self.unpool = nn.MaxUnpool2d(2, stride=2)
        self.conv_up4 = nn.Sequential(nn.Conv2d(64 + 3 + 3 + 2, 32, 3, padding=1, bias=True), nn.LeakyReLU(),
                                      nn.Conv2d(32, 16, 3, padding=1, bias=True), nn.LeakyReLU(),
                                      nn.Conv2d(16, 7, 1, padding=0, bias=True))
        cfgs = {}

        def reg(name, conv, norm, group, ws=True, cpad=None, act=ACT_LEAKY01):
            spec = ConvSpec(name, conv.weight, None, None, conv.bias, False, 1, conv.padding[0], group, ws=ws, cpad=cpad)
            bank.register(spec)
            if norm is not None:
                cfgs[name] = ConvCfg(bank, spec, bn=norm, act=act)
            else:                                   # conv + bias + LeakyReLU(0.01) without a norm (conv_up4)
                cfgs[name] = ConvCfg(bank, spec, pre_relu=True, pre_slope=0.01)

        for i in range(4):
            reg('decoder.ppm.%d.1' % i, self.ppm[i][1], self.ppm[i][2], 'frame')
        reg('decoder.conv_up1.0', self.conv_up1[0], self.conv_up1[1], 'frame')
        reg('decoder.conv_up1.3', self.conv_up1[3], self.conv_up1[4], 'frame')
        reg('decoder.conv_up2.0', self.conv_up2[0], self.conv_up2[1], 'tail')
        reg('decoder.conv_up3.0', self.conv_up3[0], self.conv_up3[1], 'tail')
        reg('decoder.conv_up4.0', self.conv_up4[0], None, 'tail', ws=False, cpad=128)
        reg('decoder.conv_up4.2', self.conv_up4[2], None, 'tail', ws=False)
        object.__setattr__(self, '_cfgs', cfgs)
        if with_fam:                                # without: the single-image fba_decoder (models/FBA/models.py:258-353)
            self.fam = FeatureAggregationModule(256, reduction, window, bank=bank, prefix='decoder.fam')

    def train(self, mode=True):
        super().train(mode)
        if self.freeze_backbone:
            print('Set FBA decoder feature extraction part in eval() mode.')
            self.conv_up1.eval()
        return self

    def run_feature(self, conv5, token, training):
        """extract_feature=True (VMN_FBA.py:21-33): pyramid pooling + conv_up1 -> [F, h, w, 256] at os8."""
        cf = self._cfgs
        link = {}                                       # conv5 feeds the pooling AND the concat: one gradient kernel for both (ops._PyramidPool)
        pooled = ops.pyramid_pool(conv5, PPM_SCALES, link)
        maps = [ops.conv_bn_act(cf['decoder.ppm.%d.1' % i], pooled[i], token, training) for i in range(4)]
        x = ops.pyramid_concat(3072, conv5, maps, link)
        x = ops.conv_bn_act(cf['decoder.conv_up1.0'], x, token, training)
        return ops.conv_bn_act(cf['decoder.conv_up1.3'], x, token, training)

    def run_tail(self, x, xb, xf, mask_u8, os4, os2, extras, img, token, training):
        """extract_feature=False (VMN_FBA.py:35-59) for the interior frames: TAM, three up-sampling stages, head.
        os4 / os2: conv_out[-4] / conv_out[-5]; extras: [F, H, W, 8] bf16 (normalised RGB, RGB, bg, fg); img: fp32
        [F, 3, H, W] view of the scaled images -> (pred fp32 [F, 7, H, W], attb, attf)."""
        x, attb, attf = self.fam.run(x, xb, xf, mask_u8, token, training)
        return self.run_tail_single(x, os4, os2, extras, img, token, training), attb, attf

    def run_tail_single(self, x, os4, os2, extras, img, token, training):
        """The three up-sampling stages and the fused head (models/FBA/models.py:326-353) -> pred fp32 [F, 7, H, W]."""
        cf = self._cfgs
        x = ops.conv_bn_act(cf['decoder.conv_up2.0'], ops.up2_concat(512, x, os4), token, training)
        x = ops.conv_bn_act(cf['decoder.conv_up3.0'], ops.up2_concat(320, x, os2), token, training)
        x = ops.conv_bn_act(cf['decoder.conv_up4.0'], ops.up2_concat(128, x, extras), token, training)
        x = ops.conv_bn_act(cf['decoder.conv_up4.2'], x, token, training)
        return ops.fba_head(x, ops.param_in(self.conv_up4[4].weight, self._bank), ops.param_in(self.conv_up4[4].bias, self._bank), img)


class VMN_FBA(nn.Module):
    """VMN (models/VMN/VMN_model.py:70-113) for the FBA base.  `run` takes the tensors tcvom_fba_input wrote."""

    def __init__(self, encoder, decoder, bank, freeze_backbone=False):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.freeze_backbone = freeze_backbone
        object.__setattr__(self, '_bank', bank)

    def train(self, mode=True):
        super().train(mode)
        if self.freeze_backbone:
            print('Set VMN encoder to eval() mode.')
            self.encoder.eval()
        return self

    def run(self, x2, extras, imgs, unk_small):
        """x2 [B,S,H/2,W/2,64] bf16, extras [B,S,H,W,8] bf16, imgs fp32 [B,S,3,H,W], unk_small uint8 [B,S,H/8,W/8]
        -> (pred fp32 [B, S-2, 7, H, W] for the interior frames, attb, attf lists of S (None at the ends))."""
        B, S = x2.shape[:2]
        bank = self._bank
        training = self.training
        F = B * S
        token = bank_token(bank, F, training, self)
        fm = lambda t: t.transpose(0, 1).reshape((S * B,) + tuple(t.shape[2:]))          # frame-major [S*B, ...]
        X2, EX, U = fm(x2), fm(extras), fm(unk_small)
        lo, hi = B, (S - 1) * B
        try:
            bank.frames_per_op = F                     # one "frame" slot per sample: GroupNorm statistics are per sample
            if self.freeze_backbone:
                with torch.no_grad():
                    outs = self.encoder.run(X2, token, training)
                    feat = self.decoder.run_feature(outs[-1], token, training)
            else:
                outs = self.encoder.run(X2, token, training)
                feat = self.decoder.run_feature(outs[-1], token, training)
            bank.frames_per_op = hi - lo
            img = imgs.transpose(0, 1)[1:S - 1].reshape((hi - lo, 3) + tuple(imgs.shape[-2:]))
            pred, ab, af = self.decoder.run_tail(feat[lo:hi], feat[0:hi - B], feat[2 * B:hi + B], U[lo:hi].contiguous(),
                                                 ops.frame_slice(outs[1], lo, hi), ops.frame_slice(outs[0], lo, hi), EX[lo:hi], img, token,
                                                 training)       # (frame_slice: the skip gradients of the interior frames go into the
                                                                 #  producers' GroupNorm backward as row ranges, not as zero-padded tensors)
        finally:
            bank.frames_per_op = 1
        H, W = pred.shape[-2:]
        pred = pred.reshape(S - 2, B, 7, H, W).transpose(0, 1)
        attb, attf = [None] * S, [None] * S
        for i in range(1, S - 1):
            sl = slice((i - 1) * B, i * B)
            attb[i], attf[i] = ab[sl], af[sl]
        return pred, attb, attf


def build_vmn_fba(agg_window, agg_reduction=1, freeze_backbone=False):
    bank = WeightBank()
    enc = ResnetDilated(bank)
    dec = vmn_fba_decoder(agg_reduction, agg_window, freeze_backbone=freeze_backbone, bank=bank)
    return VMN_FBA(enc, dec, bank, freeze_backbone=freeze_backbone)


class MattingModule(nn.Module):
    """models/FBA/models.py:18-32: the single-image FBA network (FullModel('fba')), no temporal module."""

    def __init__(self):
        super().__init__()
        bank = WeightBank()
        object.__setattr__(self, '_bank', bank)
        self.encoder = ResnetDilated(bank)
        self.decoder = vmn_fba_decoder(1, 1, bank=bank, with_fam=False)

    def run(self, x2, extras, img):
        """x2 [B,H/2,W/2,64] bf16 space-to-depth input, extras [B,H,W,8] bf16, img fp32 [B,3,H,W] -> pred fp32 [B,7,H,W]."""
        bank, training = self._bank, self.training
        F = x2.shape[0]
        token = bank_token(bank, F, training, self)
        try:
            bank.frames_per_op = F
            outs = self.encoder.run(x2, token, training)
            feat = self.decoder.run_feature(outs[-1], token, training)
            return self.decoder.run_tail_single(feat, outs[1], outs[0], extras, img, token, training)
        finally:
            bank.frames_per_op = 1


def FBA():
    return MattingModule()
