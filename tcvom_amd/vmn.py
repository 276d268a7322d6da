"""Video Matting Network on the HIP kernels: per-frame encoder + decoder-front loop, Temporal
Attention Module, decoder tail.  Mirrors models/VMN/__init__.py:11-29 and VMN_model.py:9-113."""
import torch
import torch.nn as nn

from . import ops
from .ops import ConvCfg, H16
from .weights import ConvSpec, WeightBank, bank_token


def _to_nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(H16)


def _to_nchw(t):
    return t.permute(0, 3, 1, 2).float()


class FeatureAggregationModule(nn.Module):
    """TAM (models/VMN/VMN_model.py:9-68).  forward(x, b, f, mask) -> (feat, attb, attf, mask_small).

    q = query_conv(x), v = value_conv(x), k_b/k_f = key_conv(b / f) (shared key conv); attention over the
    window x window zero-padded neighbourhood of the adjacent frame at unknown os8 pixels only; the
    aggregated values are the KEYS; result = v + agg_b + agg_f."""

    def __init__(self, input_chn, reduction, window, bank=None, prefix='fam'):
        super().__init__()
        out_chn = input_chn // reduction
        self.key_conv = nn.Conv2d(input_chn, out_chn, kernel_size=3, padding=1)
        self.query_conv = nn.Conv2d(input_chn, out_chn, kernel_size=3, padding=1)
        self.value_conv = nn.Conv2d(input_chn, out_chn, kernel_size=3, padding=1)
        self.window = int(window)
        self._own_bank = bank is None
        bank = bank if bank is not None else WeightBank()
        object.__setattr__(self, '_bank', bank)
        self._cfg = {}
        for n in ('key_conv', 'query_conv', 'value_conv'):
            conv = getattr(self, n)
            spec = ConvSpec('%s.%s' % (prefix, n), conv.weight, None, None, conv.bias, False, 1, 1, 'tail')
            bank.register(spec)
            self._cfg[n] = ConvCfg(bank, spec)

    def run(self, x, xb, xf, mask_u8, token, training):
        """NHWC bf16 fast path; mask_u8: uint8 [B,h,w] (unknown at os8)."""
        q = ops.conv_bn_act(self._cfg['query_conv'], x, token, training)
        v = ops.conv_bn_act(self._cfg['value_conv'], x, token, training)
        kb = ops.conv_bn_act(self._cfg['key_conv'], xb, token, training)
        kf = ops.conv_bn_act(self._cfg['key_conv'], xf, token, training)
        return ops.tam_attention(q, kb, kf, v, mask_u8, self.window)

    def forward(self, x, b, f, mask):
        """Reference signature: NCHW fp32 features, mask [B,1,8H,8W] in {0,1}."""
        assert self._own_bank, 'use .run() inside a network'
        B, C, H, W = x.shape
        sh, sw = mask.shape[2] // H, mask.shape[3] // W
        small = mask[:, :, ::sh, ::sw] != 0                      # nearest down-sampling (VMN_model.py:22)
        token = bank_token(self._bank, 3, self.training, self)
        out, attb, attf = self.run(_to_nhwc(x), _to_nhwc(b), _to_nhwc(f), small[:, 0].to(torch.uint8).contiguous(),
                                   token, self.training)
        return _to_nchw(out), attb, attf, small


def _stack_frames(ts):
    """The S per-frame tensors [B, ...] as ONE frame-major batch [S*B, ...]: a view when they already are consecutive slices
    of one buffer (the facade's x8 [B, S, H, W, 8] with B = 1: 100 MB of torch.cat per 1080p window otherwise)."""
    t0 = ts[0]
    if len(ts) == 1:
        return t0
    tw = [ops.f16_twin(t) for t in ts]
    if all(t is not None for t in tw):          # the IEEE fp16 twins of the frames (bf16 build, fp16 island) travel with them
        return ops.set_f16_twin(_stack_frames_plain(ts), _stack_frames_plain(tw))
    return _stack_frames_plain(ts)


def _stack_frames_plain(ts):
    t0 = ts[0]
    n = t0.numel()
    if n > 0 and not t0.requires_grad and all(
            t.is_contiguous() and t.shape == t0.shape and t.dtype == t0.dtype and not t.requires_grad and
            t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() and t.storage_offset() == t0.storage_offset() + i * n
            for i, t in enumerate(ts)):
        return t0.new_empty(0).set_(t0.untyped_storage(), t0.storage_offset(), (len(ts) * t0.shape[0],) + tuple(t0.shape[1:]))
    return torch.cat(ts, 0)


def _mid_tensors(mid):
    """Every tensor an encoder handed to the decoder (skip features, pooling indices ...)."""
    out = []
    for v in mid.values():
        out += [t for t in v if torch.is_tensor(t)] if isinstance(v, (tuple, list)) else ([v] if torch.is_tensor(v) else [])
    return out


import os as _os
TAIL_SKIP = _os.environ.get('TCVOM_NO_TAIL_SKIP', '0') != '1'        # A/B switch (tools/ab_bench.sh TCVOM_NO_TAIL_SKIP)


class VMN(nn.Module):
    """models/VMN/VMN_model.py:70-113."""

    def __init__(self, encoder, decoder, bank, freeze_backbone=False):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.freeze_backbone = freeze_backbone
        self.frame_streams = True
        self.batched_frames = True          # push the S frames through every layer together (one launch per layer)
        object.__setattr__(self, '_bank', bank)
        object.__setattr__(self, '_streams', [])
        if freeze_backbone:
            # VMN_model.py:77-103 + VMN_GCA.py:18-24: encoder and decoder front (the per-frame feature extraction) run in eval
            # mode under no_grad inside a training window: BatchNorm uses (and keeps) its running statistics, SpectralNorm
            # takes sigma from the stored u, v without iterating, no gradient reaches them
            bank.frozen_groups = {'frame'}

    def train(self, mode=True):
        super().train(mode)
        if self.freeze_backbone:
            print('Set VMN encoder to eval() mode.')
            self.encoder.eval()
        return self

    def run(self, frames_x8, unk_u8):
        """frames_x8: list of S tensors [B,H,W,8] bf16; unk_u8: list of S uint8 [B,H/8,W/8].
        Returns (alphas list (None at the ends), attb, attf) for interior frames."""
        S = len(frames_x8)
        training = self.training
        token = bank_token(self._bank, S, training, self)
        front_training = training and not self.freeze_backbone
        if self.batched_frames:
            return self._run_batched(frames_x8, unk_u8, token, training, front_training)
        if self.freeze_backbone:
            with torch.no_grad():
                fronts = [self._front(frames_x8[i], unk_u8[i], token, front_training) for i in range(S)]
            mids, feats = [f[0] for f in fronts], [f[1] for f in fronts]
            preds, attb, attf = [None] * S, [None] * S, [None] * S
            for i in range(1, S - 1):
                preds[i], attb[i], attf[i] = self.decoder.run_tail(feats[i], feats[i - 1], feats[i + 1], unk_u8[i],
                                                                   mids[i], token, training)
            self._bank.flush_bn_counters()
            return preds, attb, attf
        mids, feats = [None] * S, [None] * S
        # The encoder + decoder-front of the S frames are independent (VMN_model.py:93-98 loops over them): each frame
        # runs on its own HIP stream so that the small-grid os16/os32 kernels of different frames overlap; autograd
        # replays the same streams in backward.  Order-dependent state (BN running statistics) is applied afterwards.
        main = torch.cuda.current_stream()
        # (SyncBatchNorm: the mailbox exchanges of a rank must run in ONE stream order -- tcvom_amd/mailbox.py)
        # (a stock nn.SyncBatchNorm carries no `.sync` until its first train-mode call -- ops._sync_group sets it lazily)
        if self.frame_streams and not any(getattr(m, 'sync', False) or isinstance(m, torch.nn.SyncBatchNorm)
                                          for m in self.modules()):
            ops.SIDE_STREAMS[0] = True                     # gradient deposits between ops carry stream events (ops._GradStash)
            if len(self._streams) < S:
                object.__setattr__(self, '_streams', [torch.cuda.Stream() for _ in range(S)])
            for i in range(S):
                st = self._streams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    emb, mids[i] = self.encoder.run(frames_x8[i], unk_u8[i], token, training)
                    feats[i] = self.decoder.run_front(emb, mids[i], token, training)
            for i in range(S):
                main.wait_stream(self._streams[i])
                for t in [feats[i]] + _mid_tensors(mids[i]):
                    t.record_stream(main)
        else:
            for i in range(S):
                emb, mids[i] = self.encoder.run(frames_x8[i], unk_u8[i], token, training)
                feats[i] = self.decoder.run_front(emb, mids[i], token, training)
        preds, attb, attf = [None] * S, [None] * S, [None] * S
        for i in range(1, S - 1):
            preds[i], attb[i], attf[i] = self.decoder.run_tail(feats[i], feats[i - 1], feats[i + 1], unk_u8[i],
                                                               mids[i], token, training)
        self._bank.flush_bn_counters()
        return preds, attb, attf

    def _front(self, x8, unk, token, training):
        emb, mid = self.encoder.run(x8, unk, token, training)
        return mid, self.decoder.run_front(emb, mid, token, training)

    def _run_batched(self, frames_x8, unk_u8, token, training, front_training=None):
        """The S frames as ONE frame-major batch [S*B, ...]: every encoder / decoder-front layer is a single launch for
        all frames (each frame keeps its own SpectralNorm call slot and BatchNorm statistics, ops._ConvBNAct), the
        decoder tail a single launch for the S-2 interior frames.  3x fewer, 3x larger launches than frame-by-frame."""
        S = len(frames_x8)
        B = frames_x8[0].shape[0]
        bank = self._bank
        X = _stack_frames(frames_x8)
        U = _stack_frames_plain(unk_u8) if S > 1 else unk_u8[0]
        front_training = training if front_training is None else front_training
        try:
            bank.frames_per_op = S
            # only the interior frames are decoded (VMN_model.py:107-110): ops whose output feeds the decoder tail alone skip
            # the backward of the end frames (ops._backward_active)
            bank.tail_frames = (1, S - 1) if TAIL_SKIP else None
            if self.freeze_backbone:
                with torch.no_grad():
                    mid, feat = self._front(X, U, token, front_training)
            else:
                mid, feat = self._front(X, U, token, front_training)
            bank.frames_per_op = S - 2
            lo, hi = B, (S - 1) * B                               # the interior frames 1 .. S-2
            fs = ops.frame_slice
            mid_c = {k: (tuple(fs(t, lo, hi) for t in v) if isinstance(v, (tuple, list)) else fs(v, lo, hi)) for k, v in mid.items()
                     if k != 'unknown'}
            mid_c['unknown'] = U[lo:hi]
            f_c, f_p, f_n = ops.neighbour_slices(feat, B, S) if feat.requires_grad else (feat[lo:hi], feat[0:hi - B], feat[2 * B:hi + B])
            pred, ab, af = self.decoder.run_tail(f_c, f_p, f_n, U[lo:hi], mid_c, token, training)
        finally:
            bank.frames_per_op = 1
            bank.tail_frames = None
        preds, attb, attf = [None] * S, [None] * S, [None] * S
        for i in range(1, S - 1):
            sl = slice((i - 1) * B, i * B)
            preds[i], attb[i], attf[i] = pred[sl], ab[sl], af[sl]
        bank.flush_bn_counters()
        return preds, attb, attf

    def forward(self, images, masks, extras=None):
        """Reference signature: images list[S] of [B,1,6,H,W] fp32, masks tuple[S] of [B,1,1,H,W]
        -> (preds[S], attb[S], attf[S], small_mask[S])."""
        S = len(images)
        frames, unks = [], []
        for i in range(S):
            img = images[i].squeeze(1)
            B, Cc, H, W = img.shape
            x8 = torch.zeros((B, H, W, 8), dtype=H16, device=img.device)
            x8[..., :Cc] = img.permute(0, 2, 3, 1).to(H16)
            if ops.F16_ISLAND and Cc == 6:      # (GCA input; bf16 build: the fp16 island's first conv reads the IEEE fp16 twin)
                x16 = torch.zeros((B, H, W, 8), dtype=torch.float16, device=img.device)
                x16[..., :Cc] = img.permute(0, 2, 3, 1).to(torch.float16)
                ops.set_f16_twin(x8, x16)
            frames.append(x8)
            m = masks[i].squeeze(1)
            unks.append((m[:, 0, ::8, ::8] != 0).to(torch.uint8).contiguous())
        preds, attb, attf = self.run(frames, unks)
        small = [None] * S
        for i in range(1, S - 1):
            small[i] = unks[i].bool().unsqueeze(1)
        preds[0] = torch.zeros_like(preds[1])
        preds[-1] = torch.zeros_like(preds[-2])
        return preds, attb, attf, small


def build_vmn_gca(agg_window, agg_reduction=1, freeze_backbone=False):
    from .gca_net import resnet_gca_encoder_29, ResGuidedCxtAtten_FAM_Dec
    bank = WeightBank()
    enc = resnet_gca_encoder_29(bank=bank)
    dec = ResGuidedCxtAtten_FAM_Dec(agg_reduction, agg_window, freeze_backbone=freeze_backbone, bank=bank)
    return VMN(enc, dec, bank, freeze_backbone=freeze_backbone)


def get_VMN_models(arch, agg_window, agg_reduction=1, freeze_backbone=False, **kwargs):
    """models/VMN/__init__.py:11-29.  `vmn_gca` (configs 2-4), `vmn_fba` (config 5), `vmn_dim` and `vmn_index` run on the HIP path.

    agg_reduction != 1: the reference CONSTRUCTS such a model but its first forward raises for every architecture --
    FeatureAggregationModule._attention reshapes the C / reduction-channel keys with the input's channel count
    (VMN_model.py:33-37: "shape '[128, -1, 64]' is invalid"; checked against the imported reference, all four archs, reductions 2
    and 4) and the GCA decoder's layer_multi is inconsistent with it (VMN_GCA.py:12-15).  No behaviour to reproduce: refused here,
    at construction, with the reason."""
    if agg_reduction != 1:
        raise ValueError('agg_reduction=%r: only 1 is usable -- the reference\'s FeatureAggregationModule fails in its first forward '
                         'for any other value (models/VMN/VMN_model.py:33-37 reshapes C/reduction-channel keys as C channels)' % (agg_reduction,))
    if arch == 'vmn_gca':
        return build_vmn_gca(agg_window, agg_reduction, freeze_backbone)
    if arch == 'vmn_fba':
        from .fba_net import build_vmn_fba
        return build_vmn_fba(agg_window, agg_reduction, freeze_backbone)
    if arch == 'vmn_dim':
        from .dim_net import build_vmn_dim
        return build_vmn_dim(agg_window, agg_reduction, freeze_backbone)
    if arch == 'vmn_index':
        from .index_net import build_vmn_index
        return build_vmn_index(agg_window, agg_reduction, freeze_backbone)
    raise ValueError
