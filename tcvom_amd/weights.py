"""WeightBank: every conv weight of the network, normalised + packed for the MFMA kernels.

The reference wraps 70 convs of `vmn_gca` in SpectralNorm (models/GCA/ops.py:12-80), which runs one
power iteration per forward CALL — ~1.2 k tiny kernels per window.  The power iteration depends only
on (W_bar, u, v), never on activations, so the bank performs ALL iterations of a window up front
(S chained iterations for layers used once per frame, S-2 for decoder-tail layers), writes the
bf16 packed copies the igemm kernels consume, and after the backward pass turns the accumulated
packed-weight gradients into gradients of `weight_bar` — a handful of launches in total.

State (u, v, weight_bar) stays in ordinary nn.Parameters with the reference's state_dict names.
"""
import ctypes as C
import os
import struct

import torch

from . import _lib as L
from ._lib import ACT_DTYPE as H16

TILED_PACK = os.environ.get('TCVOM_SN_PACK_TILED', '1') != '0'      # 0: one thread per packed element (A/B, tests)
WS_FRAG = os.environ.get('TCVOM_NO_WSCONV') is None                 # the same switch that disables the kernel in the library
SN_WORDS = 24
(SN_W, SN_U, SN_V, SN_H, SN_WD, SN_KIND, SN_K, SN_C, SN_T, SN_CPAD, SN_FWD_OFF, SN_BWD_OFF,
 SN_T_OFF, SN_S_OFF, SN_DW_OFF, SN_GRAD_OFF, SN_NUMEL, SN_WS_STATS) = range(18)


def _r8(n):
    return (n + 7) // 8 * 8


def _nch(norm):
    """Channels of a BatchNorm2d / GroupNorm."""
    return norm.num_features if hasattr(norm, 'num_features') else norm.num_channels


class ConvSpec(object):
    """Static description of one conv weight registered in the bank."""

    def __init__(self, name, weight, u, v, bias, transposed, stride, pad, group, needs_dgrad=True,
                 dilation=1, ws=False, stem=False, cpad=None):
        """ws: weight-standardised conv of the FBA base (models/FBA/layers_WS.py:13-23).  stem: the 7x7 stride-2 pad-3
        input conv of its ResNet, run as a 4x4 stride-1 conv over the 2x2 space-to-depth input (16 taps x 64 channels)."""
        self.name = name
        self.dilation, self.ws, self.stem = dilation, ws, stem
        # fp16 island (bf16 build; gca_net.py F16_ISLAND): the FORWARD pack is IEEE fp16 and the forward conv runs on IEEE fp16
        # activations (tcvom_conv_desc.in_f16); the data-gradient pack, the weight gradient and everything downstream stay bf16
        self.f16 = False
        self.weight, self.u, self.v, self.bias = weight, u, v, bias
        self.transposed = transposed
        shp = tuple(weight.shape)
        if transposed:                       # ConvTranspose2d weight [Cin][Kout][R][S]
            self.C, self.K = shp[0], shp[1]
        else:
            self.K, self.C = shp[0], shp[1]
        self.R, self.S = shp[2], shp[3]
        self.T = self.R * self.S
        self.stride, self.pad = stride, pad
        self.spectral = u is not None
        self.group = group                   # 'frame' (S calls per window) | 'tail' (S-2 calls)
        self.needs_dgrad = needs_dgrad
        # cpad: channels of the (zero-padded) input buffer: a multiple of 8 (16-byte channel chunks)
        self.cpad = max(8, _r8(self.C)) if cpad is None else int(cpad)
        assert self.cpad >= self.C and self.cpad % 8 == 0
        if stem:
            assert shp[2:] == (7, 7) and stride == 2 and pad == 3 and self.C <= 16 and not transposed and not needs_dgrad
            self.T, self.cpad = 16, 64          # packed geometry; C / R / S keep the parameter's shape
        # fragment-major packed weights for the weight-stationary conv kernel (csrc/wsconv.hip): stride-1 3x3 layers with
        # C == K in {64, 128}; its persistent workgroups then load their 288 KB of A fragments as contiguous 1 KiB blocks
        self.frag = (WS_FRAG and not transposed and not stem and shp[2:] == (3, 3) and stride == 1 and pad == 1 and
                     dilation == 1 and self.C == self.K and self.C in (64, 128) and self.cpad == self.C)
        self.numel = weight.numel()
        self.h = shp[0]
        self.wd = self.numel // shp[0]
        self.layer_id = -1


_WS_CAP = []


def _wgrad_ws_cap():
    if not _WS_CAP:
        _WS_CAP.append(int(L.call('tcvom_wgrad_ws_max_problems')))
    return _WS_CAP[0]


_WS_GEO_CAP = []


def _wgrad_ws_geo_cap():
    if not _WS_GEO_CAP:
        _WS_GEO_CAP.append(int(L.call('tcvom_wgrad_ws_max_geometries')))
    return _WS_GEO_CAP[0]


WGRAD_HETERO = os.environ.get('TCVOM_NO_WGRAD_HETERO') is None          # A/B switch: one launch per geometry


WGRAD_GROUP_LAYERS = os.environ.get('TCVOM_NO_WGRAD_GROUP_LAYERS') is None        # A/B switch: one launch per layer
# A/B switch: the weight gradients that stay on the implicit-GEMM TT kernel (1x1, stride-2, transposed, small-channel convs) as one launch
# per layer geometry instead of one per TILE SHAPE (csrc/igemm.hip: igemm_tt_hetero_kernel)
WGRAD_TT_HETERO = os.environ.get('TCVOM_NO_WGRAD_TT_HETERO') is None


def _wgrad_tt_tile(geo):
    """(tm, tn) of the implicit-GEMM TT kernel when the weight gradient of this geometry runs there, else None (cached on the object)."""
    t = getattr(geo, '_tt_tile', False)
    if t is False:
        t = None
        v = L._FNS['tcvom_wgrad_igemm_variant'](C.byref(geo.wgrad[0])).decode()
        if v.startswith('igemm_tt<'):
            a = v[len('igemm_tt<'):].split(',')
            t = (int(a[0]), int(a[1]))
        geo._tt_tile = t
    return t



def _wgrad_geo_key(geo, K):
    """Signature of a layer's weight-gradient launch (all phases): layers with equal signatures share batched launches."""
    key = getattr(geo, '_wg_key', None)
    if key is None:
        key = (K,) + tuple((d.N, d.H, d.W, d.C, d.OH, d.OW, d.K, d.PH, d.PW, d.in_step, d.out_step, d.out_off_h, d.out_off_w, d.wt, d.ldo,
                            tuple((d.tap_dh[t], d.tap_dw[t], d.tap_w[t]) for t in range(d.ntaps))) for d in geo.wgrad)
        geo._wg_key = key
    return key


def _wgrad_ws_key(geo):
    """Geometry signature of a layer whose weight gradient runs on the accumulator-stationary kernel, else None (cached
    on the geometry object)."""
    key = getattr(geo, '_ws_key', False)
    if key is False:
        key = None
        if len(geo.wgrad) == 1:
            d = geo.wgrad[0]
            if L._FNS['tcvom_wgrad_igemm_variant'](C.byref(d)).startswith(b'wgrad_ws'):
                key = (d.N, d.H, d.W, d.C, d.K, d.wt, tuple((d.tap_dh[t], d.tap_dw[t], d.tap_w[t]) for t in range(d.ntaps)))
        geo._ws_key = key
    return key


PACK_ALL_CALLS = os.environ.get('TCVOM_NO_PACK_ALL', '0') != '1'       # A/B switch (tools/ab_bench.sh TCVOM_NO_PACK_ALL)


class WeightBank(object):
    def __init__(self):
        self.specs = []
        self._ptr_sig = None
        self._plans = {}
        self.max_calls = 0
        self.bns, self.bn_ch, self._bn_sig, self._bn_cap = [], 0, None, 0
        self.bn_mask, self.window_id = [], 0
        self._bn_sites = None             # (weakref(root), [(parent module, attribute, bank index)], count) of adopt_norm_modules
        self.frozen_groups = set()          # spec groups run in eval mode inside a training window (VMN freeze_backbone: {'frame'})
        self.frames_per_op = 1              # >1 while VMN.run pushes the S frames of a window through the layers together
        self._deferred = []
        self.grad_span_hook = None        # callable(flat_grad, lo, hi): a finished span of the flat gradient (ddp.py)
        self.loss_scale = 1.0             # the facade sets ops.LOSS_SCALE on the banks of an fp16 network (ops.py)

    # ------------------------------------------------------------------ BatchNorm bookkeeping
    # Every BatchNorm call of a window gets a fixed slot in one arena for its (scale, shift) and (mean, invstd)
    # vectors, and every BatchNorm one slot in a gradient buffer that the S calls add into atomically: no
    # per-call allocations, no per-call gradient accumulation kernels, ONE launch for all deferred EMAs.
    def register_bn(self, bn):
        idx = getattr(bn, '_tcvom_bank_idx', None)
        if idx is not None and idx < len(self.bns) and self.bns[idx] is bn:
            return idx
        idx = len(self.bns)
        bn._tcvom_bank_idx, bn._tcvom_ch_off = idx, self.bn_ch
        self.bns.append(bn)
        self.bn_ch += _nch(bn)
        self._bn_sig = None
        return idx

    def adopt_norm_modules(self, root):
        """The ops find their BatchNorm through `bank.bns`; `torch.nn.SyncBatchNorm.convert_sync_batchnorm` (train_ddp.py:271-273)
        REPLACES every BatchNorm of the module tree by a new nn.SyncBatchNorm built around the same Parameters -- and a later
        `.to(device)` moves the buffers of the modules IN the tree only.  Called at the start of every window with the network
        that owns the bank: each registered site (parent module, attribute name) is looked up again and a replaced module takes
        the slot of the one it replaced, so that `.train()` / `.eval()`, the running statistics and the process group the
        kernels see are those of the module the caller's tree holds.  (72 dict lookups per window.)"""
        sites = self._bn_sites
        if sites is None or sites[0]() is not root:
            import weakref
            # (a module converted BEFORE the first window is recognised by the Parameter it shares with the one it replaced)
            by_weight = {id(bn.weight): i for i, bn in enumerate(self.bns) if getattr(bn, 'weight', None) is not None}
            found = []
            for parent in root.modules():
                for name, child in parent._modules.items():
                    idx = by_weight.get(id(getattr(child, 'weight', None))) if child is not None else None
                    if idx is not None and isinstance(child, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm)):
                        found.append((parent, name, idx))
            sites = self._bn_sites = (weakref.ref(root), found, len(self.bns))
        if sites[2] != len(self.bns):                      # a BatchNorm registered since (lazy construction): walk again
            self._bn_sites = None
            return self.adopt_norm_modules(root)
        for parent, name, idx in sites[1]:
            cur, old = parent._modules[name], self.bns[idx]
            if cur is old:
                continue
            if not isinstance(cur, torch.nn.modules.batchnorm._BatchNorm) or _nch(cur) != _nch(old) or cur.weight is None:
                raise TypeError('tcvom_amd: %s.%s was replaced by %s: the HIP network runs BatchNorm / SyncBatchNorm modules with '
                                'affine parameters of the original width there' % (type(parent).__name__, name, type(cur).__name__))
            cur._tcvom_bank_idx, cur._tcvom_ch_off = idx, old._tcvom_ch_off
            self.bns[idx] = cur
            self._bn_sig = None

    def _ensure_bn(self, frames, dev):
        if not self.bns:
            return
        cap = max(int(frames), 1)
        sig = (cap, dev) + tuple((bn.weight.data_ptr(), bn.running_mean.data_ptr() if hasattr(bn, 'running_mean') else 0) for bn in self.bns)
        if sig != self._bn_sig:
            self._bn_sig, self._bn_cap = sig, cap
            self.bn_arena = torch.empty(cap * 4 * self.bn_ch, dtype=torch.float32, device=dev)
            self.bn_grad = torch.zeros(2 * self.bn_ch, dtype=torch.float32, device=dev)
            self._bn_arena_ptr, self._bn_grad_ptr = self.bn_arena.data_ptr(), self.bn_grad.data_ptr()
            rows = []
            for bn in self.bns:
                Cn = _nch(bn)
                mom = 0.1 if getattr(bn, 'momentum', None) is None else float(bn.momentum)
                packed = int.from_bytes(struct.pack('<ff', mom, float(bn.eps)), 'little', signed=True)
                if not hasattr(bn, 'running_mean'):           # GroupNorm: never part of an EMA launch (mask stays 0)
                    rows.append([0, 0, 0, Cn, 4 * self.bn_ch, packed])
                    continue
                rows.append([bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                             self._bn_arena_ptr + 4 * (4 * bn._tcvom_ch_off + 2 * Cn), Cn, 4 * self.bn_ch, packed])
            self.bn_table = torch.tensor(rows, dtype=torch.int64).to(dev)
        n = len(self.bns)
        self.bn_calls, self.bn_mask, self.bn_unbias, self.bn_touched = [0] * n, [0] * n, [1.0] * n, [False] * n

    def bn_slots(self, bn, nf, training, unbias_count):
        """Device addresses of the (scale, shift) and (mean, invstd) vectors of the next `nf` calls (frames) of `bn`
        and the float stride between consecutive calls' vectors."""
        idx = bn._tcvom_bank_idx
        c = self.bn_calls[idx]
        if c + nf > self._bn_cap:
            raise RuntimeError('BatchNorm called %d times in one window (bank prepared for %d frames)' % (c + nf, self._bn_cap))
        self.bn_calls[idx] = c + nf
        if training:
            self.bn_mask[idx] |= ((1 << nf) - 1) << c
            self.bn_unbias[idx] = unbias_count / (unbias_count - 1.0) if unbias_count > 1 else 1.0
        base = self._bn_arena_ptr + 4 * (c * 4 * self.bn_ch + 4 * bn._tcvom_ch_off)
        return base, base + 8 * _nch(bn), 4 * self.bn_ch

    def bn_slot(self, bn, training, unbias_count):
        return self.bn_slots(bn, 1, training, unbias_count)[:2]

    def bn_grad_ptrs(self, bn):
        self.bn_touched[bn._tcvom_bank_idx] = True
        base = self._bn_grad_ptr + 8 * bn._tcvom_ch_off
        return base, base + 4 * _nch(bn)

    def bn_params(self):
        out = []
        for bn in self.bns:
            out += [bn.weight, bn.bias]
        return out

    def bn_backward(self):
        """Gradients of (weight, bias) of every BatchNorm that ran a backward in this window (None for the others)."""
        if not self.bns:
            return []
        inv = 1.0 / self.loss_scale
        g = self.bn_grad.clone() if inv == 1.0 else self.bn_grad * inv      # (fp16 build: the backward ran under the loss scale)
        out = []
        for bn, hit in zip(self.bns, self.bn_touched):
            o, Cn = 2 * bn._tcvom_ch_off, _nch(bn)
            out += [g[o:o + Cn], g[o + Cn:o + 2 * Cn]] if hit else [None, None]
        return out

    def flush_bn_counters(self):
        """Apply the deferred BatchNorm state updates of the train-mode calls since the last flush, in call order:
        running_mean/var EMA (one launch for all of them) and num_batches_tracked (one foreach launch)."""
        if not self.bns or not any(self.bn_mask):
            return
        n = len(self.bns)
        masks = (C.c_uint32 * n)(*self.bn_mask)
        unb = (C.c_float * n)(*self.bn_unbias)
        L.call('tcvom_bn_ema_multi', L.ptr(self.bn_table), n, C.cast(masks, C.c_void_p), C.cast(unb, C.c_void_p), L.stream_ptr())
        by_n = {}
        for bn, m in zip(self.bns, self.bn_mask):
            if m:
                by_n.setdefault(bin(m).count('1'), []).append(bn.num_batches_tracked)
        with torch.no_grad():
            for k, bufs in by_n.items():
                torch._foreach_add_(bufs, k)
        self.bn_mask = [0] * n

    # ------------------------------------------------------------------ registration
    def register(self, spec):
        spec.layer_id = len(self.specs)
        spec.sn_dot = None
        self.specs.append(spec)
        return spec.layer_id

    def set_sn_dot(self, spec, ok):
        """A conv site declares whether the BatchNorm backward behind this conv delivers SpectralNorm's <dW~, weight_bar> (ops._sn_dot,
        tcvom_sn_dot); it does when EVERY site of the layer says so -- then the layer has no rows in the sn_bwd_inner work list."""
        new = bool(ok) and spec.spectral if spec.sn_dot is None else (spec.sn_dot and bool(ok))
        if new != spec.sn_dot:
            spec.sn_dot = new
            self._plans = {}

    def weight_params(self):
        return [s.weight for s in self.specs]

    # ------------------------------------------------------------------ device tables
    def _signature(self):
        return tuple((s.weight.data_ptr(), 0 if s.u is None else s.u.data_ptr()) for s in self.specs)

    def _build(self, device):
        specs = self.specs
        nl = len(specs)
        tab = torch.zeros(nl, SN_WORDS, dtype=torch.int64)
        fwd_off = bwd_off = t_off = s_off = dw_off = g_off = 0
        for i, s in enumerate(specs):
            assert s.weight.device == device and s.weight.dtype == torch.float32 and s.weight.is_contiguous()
            tab[i, SN_W] = s.weight.data_ptr()
            tab[i, SN_U] = s.u.data_ptr() if s.spectral else 0
            tab[i, SN_V] = s.v.data_ptr() if s.spectral else 0
            tab[i, SN_H], tab[i, SN_WD] = s.h, s.wd
            tab[i, SN_KIND] = ((1 if s.transposed else 0) | (0 if s.spectral else 2) |
                               (8 if s.ws else 0) | (16 if s.stem else 0) | (32 if s.frag else 0) | (64 if s.f16 else 0))
            assert not (s.f16 and (s.ws or s.stem or s.T > 9)), 'fp16 forward pack: plain 3x3 / 1x1 layers'

            assert not (s.ws and (s.spectral or s.transposed)) and (s.ws or not s.stem)
            tab[i, SN_K], tab[i, SN_C], tab[i, SN_T], tab[i, SN_CPAD] = s.K, s.C, s.T, s.cpad
            tab[i, SN_FWD_OFF] = fwd_off
            s.fwd_off = fwd_off
            fwd_off += s.K * s.T * s.cpad
            if s.needs_dgrad:
                tab[i, SN_BWD_OFF] = bwd_off
                s.bwd_off = bwd_off
                bwd_off += s.C * s.T * s.K
            else:
                tab[i, SN_BWD_OFF] = -1
                s.bwd_off = -1
            tab[i, SN_T_OFF], tab[i, SN_S_OFF] = t_off, s_off
            if s.spectral:
                t_off += s.wd
                s_off += s.h
            tab[i, SN_DW_OFF] = dw_off
            s.dw_off = dw_off
            dw_off += s.K * s.T * s.cpad
            tab[i, SN_GRAD_OFF] = g_off
            s.grad_off = g_off
            g_off += s.numel
            tab[i, SN_NUMEL] = s.numel
        ws = [s for s in specs if s.ws]
        self.ws_stats = torch.zeros(max(1, 4 * sum(s.h for s in ws)), dtype=torch.float32, device=device)
        off = 0
        for s in ws:
            tab[s.layer_id, SN_WS_STATS] = self.ws_stats.data_ptr() + 4 * off
            off += 4 * s.h
        rows = [(s.layer_id, r) for s in ws for r in range(s.h)]
        self.work_ws = torch.tensor(rows, dtype=torch.int32).reshape(-1).to(device) if rows else None
        self._ws_rows = rows
        self.n_ws = len(rows)
        self.device = device
        self.table = tab.to(device)
        self.fwd_stride, self.bwd_stride, self.dw_stride = _r8(fwd_off), _r8(bwd_off), _r8(dw_off)
        self.sum_wd, self.sum_h, self.grad_numel = max(t_off, 1), max(s_off, 1), g_off
        i32 = lambda rows: torch.tensor(rows, dtype=torch.int32).reshape(-1).to(device)
        sn = [s for s in specs if s.spectral]
        self.sn_ids = i32([s.layer_id for s in sn])
        self.n_sn = len(sn)
        wtu = [(s.layer_id, r0) for s in sn for r0 in range(0, s.h, 16)]
        wv = [(s.layer_id, r0) for s in sn for r0 in range(0, s.h, 4)]
        self.work_wtu, self.n_wtu = i32(wtu), len(wtu)
        self.work_wv, self.n_wv = i32(wv), len(wv)

        rows_all, rows_sn = self._pack_rows(specs), self._pack_rows(sn)
        self.work_pack_all, self.n_pack_all = i32(rows_all), len(rows_all)
        self.work_pack_sn, self.n_pack_sn = i32(rows_sn), len(rows_sn)
        for s in specs:                                  # work rows of sn_bwd_apply_kernel (256 (k, c) pairs or 256 elements each)
            s.apply_blocks = L.call('tcvom_sn_apply_blocks', int(tab[s.layer_id, SN_KIND]), s.K, s.C, s.T, s.numel)
        app = [(s.layer_id, b) for s in specs for b in range(s.apply_blocks)]
        self.work_apply, self.n_apply = i32(app), len(app)
        self.tvec = torch.zeros(self.sum_wd, device=device)
        self.svec = torch.zeros(self.sum_h, device=device)
        self._plans = {}
        self.max_calls = 0
        self._ptr_sig = self._signature()

    def _ensure_calls(self, ncalls):
        if ncalls <= self.max_calls:
            return
        dev = self.device
        nl = len(self.specs)
        self.max_calls = ncalls
        self.fwd_arena = torch.zeros(ncalls * self.fwd_stride, dtype=H16, device=dev)
        self.bwd_arena = torch.zeros(max(1, ncalls * self.bwd_stride), dtype=H16, device=dev)
        # (+ the <dy, y> slots [call][layer] of tcvom_sn_dot at the end: zeroed with the arena at the start of every window)
        self.dw_arena = torch.zeros(ncalls * self.dw_stride + ncalls * nl, dtype=torch.float32, device=dev)
        self.sn_dots = self.dw_arena[ncalls * self.dw_stride:]
        self.sigma = torch.ones(ncalls * nl, device=dev)
        self.uhist = torch.zeros(ncalls * self.sum_h, device=dev)
        self.vhist = torch.zeros(ncalls * self.sum_wd, device=dev)
        self.inner = torch.zeros(ncalls * nl, device=dev)
        self.scratch = L.SnScratch(self.tvec.data_ptr(), self.svec.data_ptr(), self.sigma.data_ptr(),
                                   self.uhist.data_ptr(), self.vhist.data_ptr(), self.sum_h, self.sum_wd, nl)
        self._plans = {}

    def _plan(self, frames, training):
        key = (frames, training, frozenset(self.frozen_groups))
        if key in self._plans:
            return self._plans[key]
        calls = {}
        for s in self.specs:
            if not s.spectral or not training or s.group in self.frozen_groups:
                calls[s.layer_id] = 1           # eval mode (also a frozen backbone in a training window): no iteration, one copy
            else:
                calls[s.layer_id] = frames if s.group == 'frame' else max(frames - 2, 1)
        inner = [(s.layer_id, c, b) for s in self.specs if s.spectral and not s.sn_dot
                 for c in range(calls[s.layer_id]) for b in range((s.numel + 8191) // 8192)]      # SN_INNER_BLOCK of csrc/spectral.hip
        dev = self.device
        dots = [1 if (s.spectral and s.sn_dot) else 0 for s in self.specs]
        plan = {
            'ncalls': calls,
            'ncalls_dev': torch.tensor([calls[i] for i in range(len(self.specs))], dtype=torch.int32, device=dev),
            'work_inner': torch.tensor(inner, dtype=torch.int32).reshape(-1).to(dev),
            'n_inner': len(inner),
            'dot_layers': torch.tensor(dots, dtype=torch.int32, device=dev) if any(dots) else None,
            'iters': max(calls.values()),
        }
        self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------ per-window work
    def prepare(self, frames, training):
        """Run every power iteration of this window and write the packed weights.  Returns the plan."""
        dev = self.specs[0].weight.device
        if self.loss_scale != 1.0 and dev.type == 'cuda':
            from . import ops                               # fp16 build: the saturation counters exist (and the library knows them)
            if ops.SCALER.enabled:                          # before the FIRST backward of this bank, not from the first optimizer step
                ops.SCALER.counter(dev)
        self.flush_bn_counters()
        if self._ptr_sig is None or self._ptr_sig != self._signature():
            self._build(dev)
        plan = self._plan(frames, training)
        self._ensure_calls(plan['iters'])
        plan = self._plan(frames, training)
        st = L.stream_ptr()
        sc = C.byref(self.scratch)
        self.dw_arena.zero_()
        self._deferred = []
        self._ensure_bn(frames, dev)
        if self.bns and training:
            self.bn_grad.zero_()
        self.window_id += 1
        if self.n_ws:
            L.call('tcvom_ws_stats', L.ptr(self.table), L.ptr(self.work_ws), self.n_ws, st)
        for call in range(plan['iters']):
            if training and call > 0:
                # layers with fewer calls keep iterating harmlessly only if still needed; tail layers
                # (S-2 calls) must NOT be advanced further than the reference does -> restrict tables
                runs = [(self._restricted(plan, call), 1)]
            elif training and self.frozen_groups:
                # frozen backbone (VMN_model.py:77-103: encoder.eval() under no_grad): its SpectralNorm layers take sigma
                # from the stored u, v without iterating, the others iterate
                frozen = lambda s: s.group in self.frozen_groups
                runs = [(self._subset(plan, 'trainable', lambda s: not frozen(s)), 1), (self._subset(plan, 'frozen', frozen), 0)]
            else:
                runs = [((self.sn_ids, self.n_sn, self.work_wtu, self.n_wtu, self.work_wv, self.n_wv), 1 if training else 0)]
            for (ids, n_sn, wtu, n_wtu, wv, n_wv), flag in runs:
                if n_sn > 0:
                    L.call('tcvom_sn_power_iteration', L.ptr(self.table), sc, L.ptr(wtu), n_wtu, L.ptr(wv), n_wv,
                           L.ptr(ids), n_sn, call, flag, st)
            if plan['iters'] > 1 and PACK_ALL_CALLS:
                continue                                   # packed below, all calls in one launch
            if call == 0:
                wp, npk = self.work_pack_all, self.n_pack_all
            else:
                wp, npk = self._restricted_pack(plan, call)
            if npk > 0:
                L.call('tcvom_sn_pack', L.ptr(self.table), sc, L.ptr(wp), npk, call, L.ptr(self.fwd_arena),
                       L.ptr(self.bwd_arena), self.fwd_stride, self.bwd_stride, st)
        if plan['iters'] > 1 and PACK_ALL_CALLS:
            # sigma / u / v of every call are kept per call (sigma[call][layer]), the packs do not feed the iterations: ONE pack
            # launch for all calls of the window (3 launches reading the 102 MB of fp32 weights each: 0.36 ms per 1080p step)
            wp, npk = self._pack_all_calls(plan)
            L.call('tcvom_sn_pack', L.ptr(self.table), sc, L.ptr(wp), npk, -1, L.ptr(self.fwd_arena),
                   L.ptr(self.bwd_arena), self.fwd_stride, self.bwd_stride, st)
        self.current_plan = plan
        self.call_counter = [0] * len(self.specs)
        return plan

    def _subset(self, plan, key, pred):
        """Power-iteration work tables of the SpectralNorm layers selected by `pred` (cached in the plan)."""
        if key not in plan:
            sel = [s for s in self.specs if s.spectral and pred(s)]
            i32 = lambda rows: torch.tensor(rows, dtype=torch.int32).reshape(-1).to(self.device)
            wtu = [(s.layer_id, r0) for s in sel for r0 in range(0, s.h, 16)]
            wv = [(s.layer_id, r0) for s in sel for r0 in range(0, s.h, 4)]
            plan[key] = (i32([s.layer_id for s in sel]), len(sel), i32(wtu), len(wtu), i32(wv), len(wv))
        return plan[key]

    def _restricted(self, plan, call):
        key = ('restrict', call)
        if key not in plan:
            sel = [s for s in self.specs if s.spectral and plan['ncalls'][s.layer_id] > call]
            i32 = lambda rows: torch.tensor(rows, dtype=torch.int32).reshape(-1).to(self.device)
            wtu = [(s.layer_id, r0) for s in sel for r0 in range(0, s.h, 16)]
            wv = [(s.layer_id, r0) for s in sel for r0 in range(0, s.h, 4)]
            plan[key] = (i32([s.layer_id for s in sel]), len(sel), i32(wtu), len(wtu), i32(wv), len(wv))
        return plan[key]

    def _pack_all_calls(self, plan):
        """Work list of ONE tcvom_sn_pack launch covering every call of the window: the per-call lists (call 0: all layers, later
        calls: the layers that have that many calls) with the call in bits 8.. of `which`, ordered tile-major so that the calls of
        one tile run close together."""
        key = 'pack_all_calls'
        if key not in plan:
            rows = []
            for layer, which, blk in self._pack_rows(self.specs):
                nc = plan['ncalls'][layer] if self.specs[layer].spectral else 1
                rows += [(layer, which | (c << 8), blk) for c in range(max(nc, 1))]
            plan[key] = (torch.tensor(rows, dtype=torch.int32).reshape(-1).to(self.device), len(rows))
        return plan[key]

    def _restricted_pack(self, plan, call):
        key = ('restrict_pack', call)
        if key not in plan:
            rows = self._pack_rows([s for s in self.specs if s.spectral and plan['ncalls'][s.layer_id] > call])
            plan[key] = (torch.tensor(rows, dtype=torch.int32).reshape(-1).to(self.device), len(rows))
        return plan[key]

    @staticmethod
    def _pack_rows(sel):
        """Work list of tcvom_sn_pack: (layer, which, block) triples."""
        rows = []
        for s in sel:
            slots = s.T
            lds_slots = s.T * (2 if getattr(s, 'f16', False) else 1)     # (sn_pack_tile keeps both formats of an f16 layer)
            if TILED_PACK and not getattr(s, 'stem', False) and lds_slots <= 18:
                # csrc/spectral.hip sn_pack_tile: 32 (k) x 64 (c) tiles, which = 3 also writes the data-gradient pack
                ntile = ((s.K + 31) // 32) * ((s.cpad + 63) // 64)
                rows += [(s.layer_id, 3 if s.needs_dgrad else 2, b) for b in range(ntile)]
                continue
            rows += [(s.layer_id, 0, b) for b in range((s.K * slots * s.cpad + 255) // 256)]
            if s.needs_dgrad:
                rows += [(s.layer_id, 1, b) for b in range((s.C * s.T * s.K + 255) // 256)]
        return rows

    def next_call(self, spec):
        """Call slot of this use of `spec` within the current window."""
        return self.next_calls(spec, 1)[0]

    def next_calls(self, spec, nf):
        """`nf` consecutive call slots (the frames of a frame-batched op): (first slot, element stride between the
        frames' forward weight copies, ... data-gradient copies); strides are 0 when all frames share one copy
        (eval mode, layers without SpectralNorm)."""
        n = self.current_plan['ncalls'][spec.layer_id]
        c = self.call_counter[spec.layer_id]
        self.call_counter[spec.layer_id] = c + nf
        if n == 1:
            return 0, 0, 0
        assert c + nf <= n, 'layer %s: call %d of %d in this window' % (spec.name, c + nf, n)
        return c, (self.fwd_stride if nf > 1 else 0), (self.bwd_stride if nf > 1 else 0)

    def fwd_ptr(self, spec, call):
        return C.c_void_p(self.fwd_arena.data_ptr() + 2 * (call * self.fwd_stride + spec.fwd_off))

    def bwd_ptr(self, spec, call):
        return C.c_void_p(self.bwd_arena.data_ptr() + 2 * (call * self.bwd_stride + spec.bwd_off))

    def dw_ptr(self, spec, call):
        return C.c_void_p(self.dw_arena.data_ptr() + 4 * (call * self.dw_stride + spec.dw_off))

    # ------------------------------------------------------------------ deferred weight gradients
    def defer_wgrad(self, spec, call, dy, x, geo, nf=1):
        """Queue dW~[call + f] += dy_f^T * im2col(x_f) for the nf frames of a conv call.  All calls of a layer in a
        window have the same shape, so `run_deferred_wgrads` issues them as one batched launch: 3x the workgroups per
        launch means the pixel reduction is split 3x less (3x fewer atomic partial sums) and 3x fewer launches."""
        n = self.current_plan['ncalls'][spec.layer_id]
        st = torch.cuda.current_stream()
        dyb, xb = dy.numel() // nf * dy.element_size(), x.numel() // nf * x.element_size()
        for f in range(nf):
            self._deferred.append((spec, (call + f) if n > 1 else 0, dy, x, geo, st, f * dyb, f * xb, dyb + xb))

    def _tt_hetero_cap(self):
        c = getattr(self, '_tt_het_cap', None)
        if c is None:
            c = self._tt_het_cap = int(L.call('tcvom_wgrad_igemm_hetero_max_problems'))
        return c

    def _launch_tt_hetero(self, part, tm, tn, st):
        """One igemm_tt launch for problems of different descriptors (tcvom_wgrad_igemm_hetero).  The descriptor table and the work list
        depend on the geometries only: planned on the host once per set of problems and kept on the device (the operand pointers are
        per step: they travel in the kernel arguments)."""
        cache = self.__dict__.setdefault('_tt_het_tables', {})
        key = (tm, tn, self.device) + tuple(id(e[4]) for e in part)
        ent = cache.get(key)
        n = len(part)
        if ent is None:
            descs = [d for e in part for d in e[4].wgrad]
            arr = (L.ConvDesc * len(descs))(*descs)
            nph = (C.c_int32 * n)(*[len(e[4].wgrad) for e in part])
            ldys = (C.c_int32 * n)(*[e[0].K for e in part])
            nwork = L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldys, C.c_void_p), n, tm, tn, None, 0)
            if nwork <= 0:
                raise RuntimeError('tcvom_wgrad_igemm_hetero_plan: %s' % L.last_error())
            work = (C.c_int32 * (8 * nwork))()
            got = L.call('tcvom_wgrad_igemm_hetero_plan', arr, C.cast(nph, C.c_void_p), C.cast(ldys, C.c_void_p), n, tm, tn,
                         C.cast(work, C.c_void_p), nwork)
            assert got == nwork
            dtab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            wtab = torch.frombuffer(bytearray(bytes(work)), dtype=torch.int32).to(self.device)
            ent = cache[key] = (dtab, wtab, nwork, [e[4] for e in part])        # (the geometries stay alive: their ids are the key)
        dtab, wtab, nwork, _ = ent
        dys = (C.c_void_p * n)(*[e[2].data_ptr() + e[6] for e in part])
        xs = (C.c_void_p * n)(*[e[3].data_ptr() + e[7] for e in part])
        dws = (C.c_void_p * n)(*[self.dw_ptr(e[0], e[1]).value for e in part])
        info = None
        if L.PROFILE is not None:                        # bench.py's event-instrumented step: the launch's algorithmic work
            def taps(g):
                return sum(len({(d.tap_dh[t], d.tap_dw[t]) for t in range(d.ntaps) if d.tap_w[t] >= 0}) for d in g.wgrad)
            info = {'variant': 'igemm_tt<%d,%d>+%dprob' % (tm, tn, n),
                    'gflop': sum(2.0 * e[4].wgrad[0].N * e[4].wgrad[0].PH * e[4].wgrad[0].PW * e[0].K * taps(e[4]) * e[4].wgrad[0].C for e in part) / 1e9,
                    'algo_bytes': sum(e[8] + 4 * e[0].K * e[0].T * e[0].cpad for e in part)}
        L.call_with_info('tcvom_wgrad_igemm_hetero', info or {}, C.cast(dys, C.c_void_p), C.cast(xs, C.c_void_p), C.cast(dws, C.c_void_p), n,
                         L.ptr(dtab), L.ptr(wtab), nwork, tm, tn, st)

    def run_deferred_wgrads(self, layers=None):
        """Issue the queued weight-gradient launches; `layers` = (lo, hi): only those of layer ids lo <= id < hi (the
        chunked bank backward), the others stay queued."""
        from .ops import _phase_array
        if layers is None:
            pend, self._deferred = self._deferred, []
        else:
            pend = [e for e in self._deferred if layers[0] <= e[0].layer_id < layers[1]]
            self._deferred = [e for e in self._deferred if not (layers[0] <= e[0].layer_id < layers[1])]
        if not pend:
            return
        cur = torch.cuda.current_stream()
        for s in {e[5] for e in pend}:
            if s != cur:
                cur.wait_stream(s)
        for e in pend:                                      # operands produced on another stream: no early reuse of their memory
            if e[5] != cur:
                e[2].record_stream(cur)
                e[3].record_stream(cur)
        # Layers of one geometry share launches of the accumulator-stationary kernel (csrc/wgradws.hip): with every call of every
        # such layer in one launch a block of dw is owned by one or two workgroups instead of ~40.  Other shapes: the calls of
        # a layer as one batched launch.
        groups, multi, small_keys = {}, {}, set()
        for e in pend:
            geo = e[4]
            key = _wgrad_ws_key(geo)
            if key is not None:
                multi.setdefault(key, []).append(e)
            else:
                # the other shapes: the calls of every layer of one geometry, 8 problems per launch (the more problems a launch has,
                # the less its pixel reduction is split and the fewer atomic partial sums it adds)
                # -- except for the large 1x1 problems (>= 32 MB of operands: the 40 - 80 MB pointwise convs of the FBA bottlenecks stream
                # at 2.3 - 3.8 TB/s as launches of 3 and lose as launches of 8: 47.88 -> 48.12 ms); the 1x1 / stride-2 / transposed /
                # 32-channel convs of the GCA trunk, 20 - 100 us per layer, gain: GCA+TAM 24.25 -> 24.01 ms
                # (e[8]: operand bytes of this problem, one frame's dy + x)
                small = WGRAD_GROUP_LAYERS and (e[8] < (32 << 20) or geo.wgrad[0].ntaps > 1)
                gkey = _wgrad_geo_key(geo, e[0].K) if small else (e[0].layer_id, id(geo))
                groups.setdefault(gkey, []).append(e)
                if small:
                    small_keys.add(gkey)
        st = L.stream_ptr()
        cap = _wgrad_ws_cap()
        if WGRAD_TT_HETERO and WGRAD_GROUP_LAYERS:
            # the small problems of the implicit-GEMM TT kernel, whatever their descriptors: one launch per tile shape
            tt = {}
            for key in [k for k in groups if k in small_keys]:
                tile = _wgrad_tt_tile(groups[key][0][4])
                if tile is not None:
                    tt.setdefault(tile, []).extend(groups.pop(key))
            hcap = self._tt_hetero_cap()
            for (tm, tn), items in tt.items():
                for i in range(0, len(items), hcap):
                    self._launch_tt_hetero(items[i:i + hcap], tm, tn, st)
        if WGRAD_HETERO:
            # ... and the geometries that share the kernel's channel window (C a multiple of 128, or not) share launches too: the
            # atomic flush costs a launch ~45 us however few problems it has (csrc/wgradws.hip: tcvom_wgrad_ws_hetero)
            gcap = _wgrad_ws_geo_cap()
            # (one launch = one channel window AND one tap -> weight-slot map, k[5:] = (wt, taps): tcvom_wgrad_ws_hetero refuses mixed maps)
            for cw in sorted({((k[3] % 128 == 0),) + tuple(k[5:]) for k in multi}, key=repr):
                keys = [k for k in multi if ((k[3] % 128 == 0),) + tuple(k[5:]) == cw]
                batch, descs = [], []                   # problems (entry, geometry index) and descriptors of the launch being filled

                def launch():
                    n = len(batch)
                    dys = (C.c_void_p * n)(*[e[2].data_ptr() + e[6] for e, _ in batch])
                    xs = (C.c_void_p * n)(*[e[3].data_ptr() + e[7] for e, _ in batch])
                    dws = (C.c_void_p * n)(*[self.dw_ptr(e[0], e[1]).value for e, _ in batch])
                    gidx = (C.c_int32 * n)(*[g for _, g in batch])
                    arr = (type(descs[0]) * len(descs))(*descs)
                    L.call('tcvom_wgrad_ws_hetero', C.cast(dys, C.c_void_p), C.cast(xs, C.c_void_p), C.cast(dws, C.c_void_p), n,
                           arr, len(descs), C.cast(gidx, C.c_void_p), st)
                for k in keys:
                    items = multi[k]
                    d = items[0][4].wgrad[0]
                    for i in range(0, len(items), cap):
                        part = items[i:i + cap]
                        if batch and (len(batch) + len(part) > cap or len(descs) >= gcap):
                            launch()
                            batch, descs = [], []
                        descs.append(d)
                        batch += [(e, len(descs) - 1) for e in part]
                if batch:
                    launch()
            multi = {}
        for items in multi.values():
            arr = _phase_array(items[0][4].wgrad)
            for i in range(0, len(items), cap):
                part = items[i:i + cap]
                n = len(part)
                dys = (C.c_void_p * n)(*[e[2].data_ptr() + e[6] for e in part])
                xs = (C.c_void_p * n)(*[e[3].data_ptr() + e[7] for e in part])
                dws = (C.c_void_p * n)(*[self.dw_ptr(e[0], e[1]).value for e in part])
                L.call('tcvom_wgrad_ws_multi', C.cast(dys, C.c_void_p), C.cast(xs, C.c_void_p), C.cast(dws, C.c_void_p), n,
                       arr, items[0][0].K, st)
        for items in groups.values():
            spec, geo = items[0][0], items[0][4]
            arr = _phase_array(geo.wgrad)
            for i in range(0, len(items), 8):
                part = items[i:i + 8]
                n = len(part)
                dys = (C.c_void_p * n)(*[e[2].data_ptr() + e[6] for e in part])
                xs = (C.c_void_p * n)(*[e[3].data_ptr() + e[7] for e in part])
                dws = (C.c_void_p * n)(*[self.dw_ptr(e[0], e[1]).value for e in part])
                L.call('tcvom_wgrad_igemm_batched', C.cast(dys, C.c_void_p), C.cast(xs, C.c_void_p), C.cast(dws, C.c_void_p), n,
                       arr, len(geo.wgrad), spec.K, st)
        for e in pend:
            if e[5] != cur:                      # allocated on a frame stream, consumed here: keep the blocks alive
                e[2].record_stream(cur)
                e[3].record_stream(cur)

    GRAD_CHUNKS = 4          # spans of the flat gradient handed to `grad_span_hook` as they complete

    def _chunks(self, plan):
        """Layer ranges of ~equal gradient size, each with its slice of the sn_backward work lists (both lists are in
        layer order): [(layer lo, layer hi, grad lo, grad hi, inner row lo, n, apply row lo, n)]."""
        ck = plan.get('chunks')
        if ck is not None:
            return ck
        specs = self.specs
        n = max(1, min(self.GRAD_CHUNKS, len(specs)))
        target = self.grad_numel / float(n)
        bounds, acc = [0], 0
        for i, sp in enumerate(specs):
            acc += sp.numel
            if len(bounds) < n and acc >= target * len(bounds) and i + 1 < len(specs):
                bounds.append(i + 1)
        bounds.append(len(specs))
        calls = plan['ncalls']
        inner_rows = lambda sp: calls[sp.layer_id] * ((sp.numel + 8191) // 8192) if (sp.spectral and not sp.sn_dot) else 0
        apply_rows = lambda sp: sp.apply_blocks
        ck, i0, a0 = [], 0, 0
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            ni = sum(inner_rows(sp) for sp in specs[lo:hi])
            na = sum(apply_rows(sp) for sp in specs[lo:hi])
            ck.append((lo, hi, specs[lo].grad_off, specs[hi - 1].grad_off + specs[hi - 1].numel, i0, ni, a0, na))
            i0 += ni
            a0 += na
        assert i0 == plan['n_inner'] and a0 == self.n_apply
        plan['chunks'] = ck
        return ck

    def backward(self, plan):
        """dW~ arena -> list of weight_bar gradients (views of one flat fp32 buffer).

        Runs in GRAD_CHUNKS layer ranges: the deferred weight-gradient launches of a range, its SpectralNorm backward,
        then `grad_span_hook(flat, lo, hi)` -- tcvom_amd.ddp.GradientAverager starts the all-reduce of that span of the
        flat buffer there, so the collective of one range overlaps the weight-gradient kernels of the next
        (DDP's bucketed overlap, train_ddp.py:275-280)."""
        grad = torch.empty(self.grad_numel, dtype=torch.float32, device=self.device)
        st = L.stream_ptr()
        hook = self.grad_span_hook
        inv = 1.0 / self.loss_scale                    # (fp16 build: the backward ran under the loss scale, ops.py; folded into sn_backward's writes)
        chunks = self._chunks(plan) if hook is not None else [(0, len(self.specs), 0, self.grad_numel, 0, plan['n_inner'], 0, self.n_apply)]
        for lo, hi, g0, g1, i0, ni, a0, na in chunks:
            self.run_deferred_wgrads(None if len(chunks) == 1 else (lo, hi))
            wi = C.c_void_p(plan['work_inner'].data_ptr() + 12 * i0) if ni else None
            wa = C.c_void_p(self.work_apply.data_ptr() + 8 * a0)
            L.call('tcvom_sn_backward', L.ptr(self.table), C.byref(self.scratch), wi, ni, wa, na, L.ptr(plan['ncalls_dev']),
                   L.ptr(self.dw_arena), self.dw_stride, L.ptr(self.inner), self.max_calls, L.ptr(grad), inv,
                   L.ptr(self.sn_dots) if plan['dot_layers'] is not None else None, L.ptr(plan['dot_layers']), st)
            if self.n_ws:
                rows = [(i, r) for i, r in self._ws_rows if lo <= i < hi]
                if rows:
                    w0 = self._ws_rows.index(rows[0])
                    L.call('tcvom_ws_backward', L.ptr(self.table), C.c_void_p(self.work_ws.data_ptr() + 8 * w0), len(rows),
                           L.ptr(grad), st)
            if hook is not None:
                hook(grad, g0, g1)
        self.run_deferred_wgrads()                     # (nothing left unless a layer id fell outside the ranges)
        # a frozen backbone (VMN freeze_backbone: the 'frame' group runs under no_grad) gets NO gradient, as in the reference --
        # not zeros: Adam folds weight_decay * p into the gradient and would move the "frozen" weights (holes in the flat
        # buffer; GradientAverager reduces across them)
        frozen = self.frozen_groups
        return [None if s.group in frozen else grad[s.grad_off:s.grad_off + s.numel].view(s.weight.shape) for s in self.specs]


class _BankToken(torch.autograd.Function):
    """Ties the bank into autograd: forward = prepare (power iterations + packing), output = a scalar
    token every conv op takes as an input; backward (which therefore runs after every conv's backward
    has deposited its packed weight gradient) = SpectralNorm backward for all layers at once."""

    @staticmethod
    def forward(ctx, bank, frames, training, *weights):
        ctx.bank = bank
        ctx.plan = bank.prepare(frames, training)
        return torch.zeros((), device=weights[0].device)

    @staticmethod
    def backward(ctx, gtoken):
        grads = ctx.bank.backward(ctx.plan)
        return (None, None, None) + tuple(grads) + tuple(ctx.bank.bn_backward())


def bank_token(bank, frames, training, root=None):
    """root: the network that owns the bank -- BatchNorm modules replaced in its tree since the last window are adopted first."""
    if root is not None and bank.bns:
        bank.adopt_norm_modules(root)
    return _BankToken.apply(bank, frames, training, *(bank.weight_params() + bank.bn_params()))
