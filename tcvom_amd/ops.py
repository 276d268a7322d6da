"""Autograd wrappers around the HIP kernels (one torch.autograd.Function per fused op).

PyTorch is used here for device memory (caching allocator), streams and the autograd tape only:
every forward/backward below is one or more launches from libtcvom_hip.so on the current stream.
Tensors between ops are NHWC bf16 (`[N,H,W,C]`, contiguous).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L
from .conv_plan import ConvGeometry, dense_desc, dense_tt_desc

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_LEAKY01 = 0, 1, 2, 3        # LeakyReLU(0.2) (GCA decoder) / nn.LeakyReLU() default 0.01 (FBA)
ACT_RELU6 = 4                                                  # IndexNet / MobileNetV2 blocks
H16 = L.ACT_DTYPE            # torch.bfloat16 or torch.float16: the 16-bit storage type of the loaded library


# ---- loss scale of the fp16 build
# fp16 has 5 exponent bits: activation GRADIENTS (1e-5 .. 1e-9 at 1080p: the losses are means over ~1e5 unknown pixels) would fall
# into its denormals.  A network built by the facade in the fp16 build therefore runs its whole backward under a constant scale,
# invisibly to the caller: the gradients entering the network (d loss / d prediction, d loss / d attention logits: fp32 tensors
# produced by the loss kernels) are multiplied by `bank.loss_scale` (`enter_backward`), and every parameter gradient leaving it
# is multiplied by its inverse -- the bank's flat weight gradient and BatchNorm arena (weights.py), the bias gradients of the
# conv ops below (they know their bank), and the parameters that enter through plain tensor expressions or bank-less ops
# (`param_in` at the call site) -- so .grad holds the same values as in the bf16 build and in the reference.  With the formula
# weights the per-layer gradient maxima sit at 1e-5 .. 4e-3: 2^16 puts them at 1 .. 300, far from both ends of fp16 (65504 /
# 6e-5); the 16-bit conversions saturate instead of producing inf (csrc/common.h).  A WeightBank starts with loss_scale 1:
# ops called directly (tests, tools) are not scaled.  bf16 (8 exponent bits) needs none.
import os as _os
LOSS_SCALE = float(_os.environ.get('TCVOM_LOSS_SCALE', '65536' if L.DTYPE_NAME == 'fp16' else '1'))
class LossScaler(object):
    """Overflow guard + dynamic loss scale of the fp16 build (torch.cuda.amp.GradScaler's rule, without a host synchronisation).

    The 16-bit conversions saturate at +-65504 instead of producing inf (csrc/common.h), so an overflowing activation gradient
    would be clipped silently.  The BatchNorm-backward reductions -- which read every activation gradient of the network --
    count workgroups that saw an element AT the saturation value into a device counter (`tcvom_overflow_sink`; forward: conv
    outputs read by the BatchNorm apply pass, counter [1]).  Per optimizer step (FusedAdam.step):
      * the Adam kernel reads the counter ON THE DEVICE and updates nothing when it is non-zero (`tcvom_adam_mt_guarded`);
      * the counter is copied to pinned memory asynchronously and zeroed; the host looks at it one step later (the copy of step
        t - 1 has long finished when step t's optimizer call runs): after an overflow the scale of every registered bank halves
        (floor 1) and the Adam step counters of the skipped step are taken back; after `growth_interval` clean steps it doubles,
        up to its initial value (the calibrated 2^16: gradient maxima of 1e-5 .. 4e-3 sit at 1 .. 300).
    With several ranks GradientAverager all-reduces (MAX) the counter, so every rank skips the same steps."""

    def __init__(self, init_scale, growth_interval=1000):
        import weakref
        self.init_scale, self.scale, self.growth_interval = float(init_scale), float(init_scale), int(growth_interval)
        self.banks = weakref.WeakSet()
        self.counters = {}           # device index -> (device int32[2], pinned int32[2, 2], [event, event])
        self.clean_steps, self.skipped_steps, self.forward_saturations, self.t = 0, 0, 0, 0
        self.reduced_over_ranks = False   # set by GradientAverager.average() (MAX over the ranks), consumed by FusedAdam.step()
        self._ignore_next = False         # the read-back that follows a detected skip belongs to a backward run at the OLD scale
        self.enabled = L.DTYPE_NAME == 'fp16' and self.init_scale != 1.0 and _os.environ.get('TCVOM_NO_OVERFLOW_GUARD') is None

    def register(self, bank, device=None):
        """`device`: the bank's device when known -- the counter pair is then created and announced to the library BEFORE the
        first backward (a lazily created one missed the first step)."""
        bank.loss_scale = self.scale
        self.banks.add(bank)
        if self.enabled and device is not None and torch.device(device).type == 'cuda':
            self.counter(device)

    def counter(self, device):
        """The device counter pair of `device` (created on first use and announced to the library)."""
        idx = torch.device(device).index or 0
        ent = self.counters.get(idx)
        if ent is None:
            # the library holds ONE process-wide sink pointer (csrc/norm.hip: g_overflow_sink; one process per GPU is the
            # deployment): a second device in the same process would have its kernels count into the first device's memory
            if self.counters:
                raise RuntimeError('tcvom_amd: the fp16 overflow guard supports one GPU per process (counter registered on '
                                   'cuda:%d, asked for cuda:%d); run one process per GPU or set TCVOM_NO_OVERFLOW_GUARD=1'
                                   % (next(iter(self.counters)), idx))
            dev = torch.zeros(2, dtype=torch.int32, device=torch.device('cuda', idx))
            ent = self.counters[idx] = (dev, torch.zeros((2, 2), dtype=torch.int32).pin_memory(), [None, None])
            L.call('tcvom_overflow_sink', L.ptr(dev))
        return ent[0]

    def _set_scale(self, s):
        self.scale = s
        for bank in self.banks:
            bank.loss_scale = s

    def before_step(self, device):
        """Host side, at the start of an optimizer step: evaluate the read-back of the PREVIOUS step.  Returns True when that
        step was skipped on the device (the caller takes its step counters back)."""
        ent = self.counters.get(torch.device(device).index or 0)
        if ent is None:
            return False
        ev = ent[2][(self.t - 1) & 1]
        if ev is None:
            return False
        ev.synchronize()                                   # recorded a whole step ago: returns at once
        ent[2][(self.t - 1) & 1] = None
        bwd, fwd = (int(v) for v in ent[1][(self.t - 1) & 1])
        self.forward_saturations += fwd
        if self._ignore_next:
            # step t-1's backward had already run at the old scale when step t-2's overflow became known: whatever it counted
            # (the device dropped or applied it by its own counter) must not halve the scale a second time
            self._ignore_next = False
            if bwd:
                self.skipped_steps += 1
                return True
            return False
        if bwd:
            self._ignore_next = True
            self.skipped_steps += 1
            self.clean_steps = 0
            self._set_scale(max(self.scale * 0.5, 1.0))
            return True
        self.clean_steps += 1
        if self.clean_steps >= self.growth_interval and self.scale < self.init_scale:
            self.clean_steps = 0
            self._set_scale(min(self.scale * 2.0, self.init_scale))
        return False

    def after_step(self, device):
        """Stream side, after the guarded Adam launch: read the counters back asynchronously and zero them for the next step."""
        ent = self.counters[torch.device(device).index or 0]
        slot = self.t & 1
        ent[1][slot].copy_(ent[0], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ent[2][slot] = ev
        ent[0].zero_()
        self.t += 1
        self.reduced_over_ranks = False


SCALER = LossScaler(LOSS_SCALE)
RES_MASK = _os.environ.get('TCVOM_NO_RES_MASK') is None          # A/B switch: activation bitmask of the residual sites (tcvom_bn_apply_mask)
SN_DOT = _os.environ.get('TCVOM_NO_SN_DOT') is None            # A/B switch: SpectralNorm's <dW~, weight_bar> from the BatchNorm backward


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, s):
        ctx.s = s
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def scale_grad(t, s):
    """Identity whose backward multiplies the gradient by s."""
    if t is None or s == 1.0 or not torch.is_tensor(t) or not t.requires_grad:
        return t
    return _ScaleGrad.apply(t, s)


def enter_backward(outs, bank):
    """Network outputs on their way to the loss kernels: the gradients coming back are scaled by bank.loss_scale."""
    s = bank.loss_scale
    if s == 1.0:
        return outs
    if isinstance(outs, (list, tuple)):
        return type(outs)(enter_backward(o, bank) for o in outs)
    return scale_grad(outs, s)


def param_in(p, bank):
    """A parameter that enters the network through a tensor expression or a bank-less op (head convs): its gradient is
    unscaled on the way out."""
    return scale_grad(p, 1.0 / bank.loss_scale)


def _unscale_(bank, *grads):
    """Parameter gradients computed by an op inside the scaled backward, on their way out (in place)."""
    if bank.loss_scale != 1.0:
        for g in grads:
            if g is not None:
                g.mul_(1.0 / bank.loss_scale)


class _GradStash(list):
    """The deposit list of a conv + BatchNorm output (`z._tcvom_grad_stash`).  Deposits bypass autograd's edges, hence also its
    stream synchronisation: when the depositing op runs on another HIP stream than the op that consumes the list (the shortcut
    branches of the GCA encoder on their side stream, gca_net.py), the deposit records an event and the consumer waits for it."""
    __slots__ = ('events',)

    def deposit(self, item):
        self.append(item)
        if SIDE_STREAMS[0]:
            ev = torch.cuda.Event()
            ev.record()
            if getattr(self, 'events', None) is None:
                self.events = []
            self.events.append(ev)

    def collect(self):
        """Called by the consumer before it reads the deposits: wait for their streams, keep their memory from being reused early."""
        evs = getattr(self, 'events', None)
        if evs:
            cur = torch.cuda.current_stream()
            for ev in evs:
                cur.wait_event(ev)
            for t in self:
                g = t[1] if isinstance(t, tuple) else t
                g.record_stream(cur)
            self.events = None


SIDE_STREAMS = [False]          # set once a network runs part of its window on other streams (deposits then carry events)
# (Round 4, measured and dropped: the tail-only shortcut branches of the GCA encoder -- os1 / os2 / os4 halo convs + BatchNorm passes,
#  ~2.5 ms of HBM-bound work per 1080p step -- on a SIDE stream beside the MFMA-bound trunk, autograd replaying the stream in backward:
#  parity-green, but 40.0 against 41.1 windows/s on the same box at any stream priority: the streaming kernels take CU slots and L2
#  from the attention GEMMs, which lose more than the overlap gains.)


# ---------------------------------------------------------------------------------------------
# The fp16 island of the bf16 build.  tests/study_bf16_noise.py: >= 99 % of the storage noise of the alpha matte
# is injected in the encoder stem, layer1 and layer2 -- a third each by the rounding of weights, conv outputs and stored activations.
# Those layers therefore run their FORWARD in IEEE fp16 (same bytes, same MFMA rate, 11 instead of 8 significant bits): fp16 packed
# weights (weights.ConvSpec.f16), fp16 conv outputs (tcvom_conv_desc.out_fp32 = 2) and an IEEE fp16 "twin" of every activation of the
# island, written by the apply pass next to the bf16 tensor (tcvom_bn_apply_f16).  The twin is what the next forward conv of the island
# and the residual input of its block read; the bf16 tensor is what autograd sees: the backward (data gradient, weight gradient,
# BatchNorm backward) and every consumer outside the island are unchanged.  TCVOM_NO_F16_ISLAND=1: the round-5 scheme (doubled taps in
# the stem, fp16 conv outputs in layer1 / layer2) for A/B runs.  The fp16 build has no island (it is one).
F16_ISLAND = H16 == torch.bfloat16 and _os.environ.get('TCVOM_NO_F16_ISLAND') is None


def f16_twin(t):
    """The IEEE fp16 twin of an activation of the fp16 island (same shape), or None."""
    return getattr(t, '_tcvom_f16', None) if t is not None else None


def set_f16_twin(t, t16):
    assert t16.dtype == torch.float16 and t16.shape == t.shape
    t._tcvom_f16 = t16
    return t


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError('tcvom_amd ops run on the GPU through libtcvom_hip.so only (no CPU fallback); got a %s tensor'
                           % t.device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# =============================================================================================
# conv (+ bias) (+ ReLU) (+ BatchNorm) (+ residual) (+ activation) (+ residual)
# =============================================================================================
class ConvCfg(object):
    """Static configuration of one conv(+BN) site of the network."""

    def __init__(self, bank, spec, bn=None, act=ACT_NONE, pre_relu=False, unbias_mult=1, pre_slope=0.0):
        """bn: nn.BatchNorm2d, nn.GroupNorm (FBA base: statistics per sample, no running state) or None.
        pre_relu: activation fused into the conv epilogue BEFORE the norm (or the layer's only activation when there is
        no norm): ReLU, or LeakyReLU(pre_slope) for pre_slope == 0.01."""
        self.bank, self.spec = bank, spec
        # the norm module is read through the bank (property `bn`): WeightBank.adopt_norm_modules follows replacements in the
        # module tree (torch's SyncBatchNorm conversion)
        self._bn_idx = bank.register_bn(bn) if bn is not None else None
        assert pre_slope in (0.0, 0.01)
        self.act, self.pre_relu, self.unbias_mult, self.pre_slope = act, pre_relu, unbias_mult, pre_slope
        self.group_norm = isinstance(bn, torch.nn.GroupNorm)
        # SpectralNorm's <dW~, weight_bar> comes out of the BatchNorm backward of this site (tcvom_sn_dot) instead of a pass over
        # the weight gradient: bias-free conv straight into a BatchNorm
        bank.set_sn_dot(spec, SN_DOT and bn is not None and not self.group_norm and spec.bias is None)
        # the output of this op is consumed by the decoder TAIL only (shortcut branches fea1..fea3 of the GCA encoder): in a
        # frame-batched window only the interior frames (bank.tail_frames) can receive a gradient, see _ConvBNAct.backward
        self.tail_only = False
        self.tail_last = False              # ... and it is the last op of such a branch (its output is the tail's input)
        self._geo = {}

    bn = property(lambda self: None if self._bn_idx is None else self.bank.bns[self._bn_idx])

    def geometry(self, N, H, W):
        key = (N, H, W)
        g = self._geo.get(key)
        if g is None:
            g = self._geo[key] = ConvGeometry(self.spec, N, H, W)
        return g


def _sn_dot(cfg, call, training, sync):
    """The tcvom_sn_dot of a BatchNorm-backward finalize (ctypes byref), or None: the slot of weight call `call` (frame f of a
    batched op adds to call + f; a layer with one call per window -- eval mode, frozen backbone -- has one slot)."""
    spec, bank = cfg.spec, cfg.bank
    if not getattr(spec, 'sn_dot', False):
        return None
    n = bank.current_plan['ncalls'][spec.layer_id]
    nl = len(bank.specs)
    slot = call if n > 1 else 0
    world = sync.world if sync is not None else 1
    return C.byref(L.SnDot(bank.sn_dots.data_ptr() + 4 * (slot * nl + spec.layer_id), nl if n > 1 else 0, float(cfg.bn.eps),
                           1 if training else 0, 1.0 / world))


SYNC_ALLREDUCES = [0]          # diagnostic: SyncBatchNorm all-reduce calls issued by the fallback transport


class _Sync(tuple):
    """(process group, world size, PeerMailbox or None) of a SyncBatchNorm call."""
    group = property(lambda self: self[0])
    world = property(lambda self: self[1])
    mailbox = property(lambda self: self[2])


def _sync_group(bn):
    """_Sync when `bn` takes its train-mode statistics over the ranks and more than one rank is running (or the module carries a
    one-rank loop-back mailbox: bench.py --sync-bn on one GPU, tests), else None.  Two ways in, same result:
    `tcvom_amd.ddp.convert_sync_batchnorm` (marks the modules, keeps their class) and torch's own
    `nn.SyncBatchNorm.convert_sync_batchnorm` -- the line the reference runs, train_ddp.py:271-273 -- whose nn.SyncBatchNorm
    modules are adopted on their first train-mode call: process group = the module's `process_group`, transport = the peer
    mailbox of that group when it is available (a collective decision, taken at the same call on every rank), else one
    all-reduce per BatchNorm call.  With a mailbox the statistics are exchanged inside the finalize kernels
    (tcvom_amd/mailbox.py); without one through an all-reduce of the process group."""
    stock = isinstance(bn, torch.nn.SyncBatchNorm)
    if not getattr(bn, 'sync', False) and not stock:
        return None
    mb = getattr(bn, 'sync_mailbox', None)
    if mb is not None and mb.group is None and mb.world == 1:
        return _Sync((None, 1, mb))
    if not dist.is_available() or not dist.is_initialized():
        return None
    group = bn.sync_group if hasattr(bn, 'sync_group') else bn.process_group
    world = dist.get_world_size(group)
    if world <= 1:
        return None
    if not hasattr(bn, 'sync_mailbox'):          # a stock nn.SyncBatchNorm seen for the first time
        from .mailbox import mailbox_for
        mb = mailbox_for(group) if bn.weight.is_cuda else None
        bn.sync, bn.sync_group, bn.sync_mailbox = True, group, mb
    return _Sync((group, world, mb))


def _phase_array(descs):
    """ctypes array of the phase descriptors (cached on the list object's first element)."""
    arr = getattr(descs[0], '_phase_array', None)
    if arr is None:
        arr = (L.ConvDesc * len(descs))(*descs)
        descs[0]._phase_array = arr
        descs[0]._phase_groups = {}
    return arr


def _set_frames(arr, n, nf, w_stride):
    """Frame-batched launch: the S frames of a window go through a layer in ONE launch, as `batch` independent problems
    of N = B samples each with their own weight copy (w_stride elements apart; 0 = shared)."""
    for i in range(n):
        d = arr[i]
        d.batch = nf
        if nf > 1:
            d.in_bstride = d.N * d.H * d.W * d.C
            d.out_bstride = d.N * d.OH * d.OW * d.ldo
            d.w_bstride = w_stride
            d.vec_bstride = 0


def _y_mode(t):
    """The conv-output type code of the C ABI (tcvom_conv_desc.out_fp32 / y_fp32 of the norm kernels): 0 = the build's 16-bit storage
    type, 1 = fp32 (dense GEMM outputs only), 2 = IEEE fp16 in a build that stores bf16 (the fp16 island)."""
    if t.dtype == torch.float32:
        return 1
    return 2 if (t.dtype == torch.float16 and H16 != torch.float16) else 0


def _stats_groups(descs, nf=1):
    """Statistics groups ONE frame of a launch of these phases writes (depends on the tile the library picks, which
    depends on the total workgroup count, hence on nf)."""
    arr = _phase_array(descs)
    cache = descs[0]._phase_groups
    if nf not in cache:
        _set_frames(arr, len(descs), nf, 0)
        cache[nf] = L.call('tcvom_conv_stats_groups', C.byref(arr[0]), len(descs)) * len(descs)
    return cache[nf]


def _launch_conv(descs, x, wptr, out, bias, stats, act, st, nf=1, w_stride=0):
    """All phases of a conv in ONE launch (stride-2 data gradients / ConvTranspose forwards have 4), for nf frames."""
    arr = _phase_array(descs)
    n = len(descs)
    gf = _stats_groups(descs, nf)                  # groups per frame (all phases)
    _set_frames(arr, n, nf, w_stride)
    f32 = _y_mode(out)
    for i in range(n):
        arr[i].act = act
        arr[i].out_fp32 = f32
        arr[i].stats_group_offset = i * (gf // n)
        arr[i].stats_bstride = gf
    L.call('tcvom_conv_igemm_phases', L.ptr(x), wptr, L.ptr(out), L.ptr(bias), L.ptr(stats), arr, n, st)


class _ConvBNAct(torch.autograd.Function):
    """conv (+bias) (+ReLU) (+BatchNorm) (+residual) (+activation) (+residual) of one layer, for `bank.frames_per_op`
    frames at once: x is [nf*B, H, W, C] frame-major; every frame has its own SpectralNorm call slot (weight copy) and
    its own BatchNorm statistics, exactly as in the reference's per-frame encoder calls (VMN_model.py:93-98)."""

    @staticmethod
    def forward(ctx, x, token, gamma, beta, bias, res1, res2, cfg, training, stash=None):
        _need_cuda(x)
        spec, bank, bn = cfg.spec, cfg.bank, cfg.bn
        # skip-branch gradients (see conv_bn_act): `stash` collects what residual consumers of THIS op's output hand back;
        # res1's own producer may offer such a list too
        ctx.stash = stash
        ctx.res1_stash = getattr(res1, '_tcvom_grad_stash', None) if res1 is not None else None
        ctx.x_stash = getattr(x, '_tcvom_grad_stash', None)
        ctx.x_tail_rows = getattr(x, '_tcvom_tail_rows', None)      # x comes from a tail-only op: it takes row-range gradients
        ctx.x_pad_unread = getattr(x, '_tcvom_pad_unread', False)   # x is a concat buffer whose backward never reads the padding channels
        ctx.set_materialize_grads(False)             # every consumer may have deposited: then autograd hands over None
        x16 = f16_twin(x)
        x = _c(x)
        NT, H, W, Cx = x.shape
        assert Cx == spec.cpad and x.dtype == H16, 'conv %s: input %s %s, expected %d channels' % (
            spec.name, tuple(x.shape), x.dtype, spec.cpad)
        nf = bank.frames_per_op
        assert NT % nf == 0
        N = NT // nf
        geo = cfg.geometry(N, H, W)
        call, wsf, wsb = bank.next_calls(spec, nf)
        st = L.stream_ptr()
        K = spec.K
        has_bn = bn is not None
        # fp16 island (bf16 build): IEEE fp16 input twin x IEEE fp16 packed weight -> IEEE fp16 conv output (the geometry's forward
        # descriptors carry in_f16 = 1 / out_fp32 = 2, conv_plan.py); an input without a twin (a caller that built x itself) is converted
        island = getattr(spec, 'f16', False)
        if island:
            assert has_bn and H16 == torch.bfloat16, 'fp16 island: conv + BatchNorm sites of the bf16 build'
            x16 = _c(x16) if x16 is not None else x.to(torch.float16)
        ydt = torch.float16 if island else H16
        y = torch.empty((NT, geo.OH, geo.OW, K), dtype=ydt, device=x.device)
        stats = None
        gn = cfg.group_norm
        if gn:
            assert N == 1, 'GroupNorm: one sample per frame slot (VMN.run flattens the batch into frames)'
        if has_bn and (training or gn):
            stats = torch.empty(nf * _stats_groups(geo.fwd, nf) * 2 * K, dtype=torch.float32, device=x.device)
        pre_act = (ACT_LEAKY01 if cfg.pre_slope else ACT_RELU) if cfg.pre_relu else ACT_NONE
        _launch_conv(geo.fwd, x16 if island else x, bank.fwd_ptr(spec, call), y, bias, stats, pre_act, st, nf, wsf)
        ctx.cfg, ctx.training, ctx.call, ctx.geo, ctx.nf, ctx.wsb = cfg, training, call, geo, nf, wsb
        tf = getattr(bank, 'tail_frames', None)
        ctx.active = tf if (cfg.tail_only and tf is not None and nf > 1 and 0 <= tf[0] < tf[1] <= nf and tf[1] - tf[0] < nf) else None
        ctx.has_res1, ctx.has_res2, ctx.has_bias = res1 is not None, res2 is not None, bias is not None
        ctx.res_mask = False
        if not has_bn:
            assert res1 is None and res2 is None and cfg.act == ACT_NONE
            if cfg.pre_relu:                 # conv + bias + ReLU without BatchNorm (DIM decoder): the mask needs y
                ctx.save_for_backward(x, y)
            else:
                ctx.save_for_backward(x)
            return y
        P = geo.out_pixels                   # pixels of ONE frame
        sync = _sync_group(bn) if training else None
        if sync is not None:
            P = P * sync[1]
        ss_i, saved_i, slot_stride = bank.bn_slots(bn, nf, training and not gn, P * cfg.unbias_mult)
        ss, saved = C.c_void_p(ss_i), C.c_void_p(saved_i)
        if gn:
            # GroupNorm: per-sample statistics in train AND eval mode, no running state
            groups = stats.numel() // (2 * K * nf)
            scratch = torch.empty(nf * 128 * K, dtype=torch.float64, device=x.device) if groups > 256 else None
            L.call('tcvom_gn_finalize', L.ptr(stats), groups, K, P, bn.num_groups, L.ptr(gamma), L.ptr(beta), float(bn.eps),
                   ss, saved, L.ptr(scratch), nf, slot_stride, st)
        elif training:
            groups = stats.numel() // (2 * K * nf)
            scratch = torch.empty(nf * 128 * K, dtype=torch.float64, device=x.device) if groups > 256 else None
            if sync is None:
                # running statistics / num_batches_tracked are updated after the window, in call order: see
                # WeightBank.flush_bn_counters
                L.call('tcvom_bn_finalize', L.ptr(stats), groups, K, P, P * cfg.unbias_mult,
                       L.ptr(gamma), L.ptr(beta), None, None,
                       float(bn.momentum), float(bn.eps), ss, saved, L.ptr(scratch), nf, slot_stride, st)
            elif sync.mailbox is not None and sync.mailbox.fits(nf, K):
                # SyncBatchNorm: statistics over the clips of ALL ranks (every rank holds the same crop size), exchanged
                # INSIDE the finalize kernel through the peer mailbox: same launches as the local path, no collective call
                L.call('tcvom_bn_finalize_sync', L.ptr(stats), groups, K, P, P * cfg.unbias_mult, L.ptr(gamma), L.ptr(beta),
                       float(bn.eps), ss, saved, L.ptr(scratch), nf, slot_stride, sync.mailbox.next(), st)
            else:
                # fallback (ranks on several nodes, TCVOM_SYNCBN=rccl): the nf frames of a frame-batched call keep separate
                # statistics and share ONE all-reduce
                sums = torch.empty(nf * 2 * K, dtype=torch.float64, device=x.device)
                L.call('tcvom_bn_reduce_sums', L.ptr(stats), groups, K, L.ptr(sums), L.ptr(scratch), nf, st)
                SYNC_ALLREDUCES[0] += 1
                dist.all_reduce(sums, group=sync[0])
                L.call('tcvom_bn_finalize_sums', L.ptr(sums), K, P, P * cfg.unbias_mult, L.ptr(gamma), L.ptr(beta),
                       float(bn.eps), ss, saved, nf, slot_stride, st)
        else:
            slot_stride = 0                  # eval: one (scale, shift) for all frames
            L.call('tcvom_bn_eval_coeffs', K, L.ptr(gamma), L.ptr(beta), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                   float(bn.eps), ss, saved, st)
        ctx.sync, ctx.ss, ctx.saved, ctx.window_id, ctx.slot_stride = sync, ss, saved, bank.window_id, slot_stride
        z = torch.empty((NT, geo.OH, geo.OW, K), dtype=H16, device=x.device)
        r1 = _c(res1) if res1 is not None else None
        r2 = _c(res2) if res2 is not None else None
        if island:
            # z twice: bf16 (autograd's tensor: saved for the weight gradient, read by everything outside the island) and IEEE fp16 (the
            # twin: read by the island's next forward conv and as the residual input of its block, here through res1's own twin)
            assert ctx.active is None and not gn
            z16 = torch.empty((NT, geo.OH, geo.OW, K), dtype=torch.float16, device=x.device)
            r1_16 = f16_twin(res1)
            r1_16 = _c(r1_16) if r1_16 is not None else None
            want_mask = r1 is not None and RES_MASK and cfg.act != ACT_RELU6 and any(ctx.needs_input_grad[:6])
            amask = torch.empty(NT * geo.out_pixels * (K // 8), dtype=torch.uint8, device=x.device) if want_mask else None
            L.call('tcvom_bn_apply_f16', L.ptr(y), ss, L.ptr(r1_16 if r1_16 is not None else r1), 1 if r1_16 is not None else 0, L.ptr(r2),
                   L.ptr(z), L.ptr(z16), L.ptr(amask), geo.out_pixels, K, cfg.act, _y_mode(y), nf, slot_stride, st)
            cfg._z16 = z16                       # conv_bn_act attaches it to the tensor autograd returns
            if want_mask:
                ctx.res_mask = True
                ctx.save_for_backward(x, y, gamma, amask)
                return z
        elif ctx.active is not None and cfg.tail_last and r1 is None and r2 is None:
            # last op of a tail-only branch: its output is read for the interior frames only -- the end frames still went
            # through the conv (their batch statistics feed the BatchNorm's running statistics, as in the reference) but are
            # not normalised / stored (z is uninitialised there; tcvom_amd.vmn slices the interior frames)
            f0, f1 = ctx.active
            L.call('tcvom_bn_apply', L.ptr(y[f0 * N:f1 * N]), C.c_void_p(ss.value + 4 * f0 * slot_stride), None, None,
                   L.ptr(z[f0 * N:f1 * N]), geo.out_pixels, K, cfg.act, _y_mode(y), f1 - f0, slot_stride, st)
        elif r1 is not None and RES_MASK and cfg.act != ACT_RELU6 and any(ctx.needs_input_grad[:6]):
            # residual site: one byte per 8 channels with the signs of norm(y) + res1 -- the backward takes the activation slope from it
            # instead of reading res1 again in both of its passes (saved in place of res1)
            amask = torch.empty(NT * geo.out_pixels * (K // 8), dtype=torch.uint8, device=x.device)
            L.call('tcvom_bn_apply_mask', L.ptr(y), ss, L.ptr(r1), L.ptr(r2), L.ptr(z), L.ptr(amask), geo.out_pixels, K, cfg.act,
                   _y_mode(y), nf, slot_stride, st)
            ctx.res_mask = True
            ctx.save_for_backward(x, y, gamma, amask)
            return z
        else:
            L.call('tcvom_bn_apply', L.ptr(y), ss, L.ptr(r1), L.ptr(r2), L.ptr(z), geo.out_pixels, K, cfg.act, _y_mode(y),
                   nf, slot_stride, st)
        ctx.save_for_backward(x, y, gamma, r1)
        return z

    @staticmethod
    def backward(ctx, dz):
        cfg, geo, nf = ctx.cfg, ctx.geo, ctx.nf
        spec, bank = cfg.spec, cfg.bank
        st = L.stream_ptr()
        K = spec.K
        extra, ranged = [], []
        if ctx.stash:                                      # gradients consumers deposited instead of returning them to autograd
            ctx.stash.collect()
            for t in ctx.stash:
                if isinstance(t, tuple):                   # ('rows', g, lo, hi): frame_slice's gradient of rows lo .. hi only
                    ranged.append(t)
                else:
                    extra.append(_c(t))
            del ctx.stash[:]
        dz2_rng = None
        # three addends inside the BatchNorm-backward kernels (dz for all frames + two more, each a whole tensor or the rows of a frame
        # range): the stage outputs of the encoder feed the next stage's conv1, its down-sampling branch and the shortcut branch --
        # without this the third gradient costs an element-wise add (and a zero-padded copy when it covers the interior frames only)
        add3 = None
        if (THREE_ADDENDS and ctx.active is None and cfg.bn is not None and not ctx.has_res2 and len(extra) + len(ranged) + (dz is not None) == 3
                and len(extra) + (dz is not None) >= 1 and (dz if dz is not None else extra[0]).dtype == H16
                and all(lo % geo.N == 0 and hi % geo.N == 0 and g.dtype == H16 for _, g, lo, hi in ranged)):
            full = ([dz] if dz is not None else []) + extra
            others = [(_c(t), 0, nf) for t in full[1:]] + [(_c(g), lo // geo.N, hi // geo.N) for _, g, lo, hi in ranged]
            if all(t.shape[1:] == full[0].shape[1:] and t.dtype == H16 for t, _, _ in others) and full[0].shape[0] == geo.N * nf:
                dz, extra, ranged, add3 = full[0], [], [], others
        if ranged and (ctx.active is None or any((lo, hi) != (ctx.active[0] * geo.N, ctx.active[1] * geo.N) for _, _, lo, hi in ranged)):
            # this op ran (and is differentiated) for all frames; a consumer deposited the gradient of some frames only
            _, g0, lo0, hi0 = ranged[0]
            if (RANGED_DZ2 and ctx.active is None and len(ranged) == 1 and cfg.bn is not None and not ctx.has_res2 and len(extra) + (dz is not None) == 1
                    and lo0 % geo.N == 0 and hi0 % geo.N == 0 and g0.dtype == H16):
                # the BatchNorm-backward kernels add it for those frames (dz2 with a frame range): no zero-padded copy
                dz2_rng = (_c(g0), lo0 // geo.N, hi0 // geo.N)
            else:
                for _, g, lo, hi in ranged:                # the plain zero-padded gradient
                    full = g.new_zeros((geo.N * nf,) + tuple(g.shape[1:]))
                    full[lo:hi] = g
                    extra.append(full)
            ranged = []
        if dz is None:
            if not extra and not ranged:
                return (None,) * 10                        # nothing arrived: the output did not reach the loss
            if extra:
                dz = extra.pop(0)
        dz = _c(dz) if dz is not None else None
        dgamma = dbeta = dbias = dres1 = dz2 = dsum = None
        if extra:
            dz2 = extra[0] if len(extra) == 1 else sum(extra[1:], extra[0])
            assert dz is not None and dz2.shape == dz.shape and dz2.dtype == dz.dtype
        P = geo.out_pixels
        if ctx.active is not None and cfg.bn is not None and not ctx.has_res1 and not ctx.has_res2 and not ctx.has_bias:
            return _ConvBNAct._backward_active(ctx, dz, dz2, [_c(g) for _, g, _, _ in ranged])
        assert not ranged and dz is not None
        if cfg.bn is None:
            if cfg.pre_relu:
                x, y = ctx.saved_tensors
                dy = torch.empty_like(dz)
                L.call('tcvom_relu_bwd', L.ptr(dz), L.ptr(y), L.ptr(dy), dz.numel(), float(cfg.pre_slope), st)
            else:
                (x,) = ctx.saved_tensors
                dy = dz
            if ctx.has_bias:
                dbias = torch.empty(K, dtype=torch.float32, device=dz.device)
                L.call('tcvom_colsum', L.ptr(dy), L.ptr(dbias), P * nf, K, K, st)
        else:
            x, y, gamma, r1 = ctx.saved_tensors
            if ctx.window_id != bank.window_id:
                raise RuntimeError('conv %s: backward of a window after a newer forward of the same network is not '
                                   'supported (the per-window BatchNorm / weight arenas were reused)' % spec.name)
            ss, saved, stride = ctx.ss, ctx.saved, ctx.slot_stride
            yf = _y_mode(y)
            zf0, zf1 = 0, nf
            if dz2_rng is not None:
                assert dz2 is None
                dz2, zf0, zf1 = dz2_rng
            dy = torch.empty(y.shape, dtype=H16, device=dz.device)
            if ctx.has_res1 and ctx.needs_input_grad[5]:
                dres1 = torch.empty(y.shape, dtype=H16, device=dz.device)
        if cfg.bn is not None:
            groups = L.call('tcvom_bn_bwd_groups_n', P, K, nf)
            partial = torch.empty(nf * groups * 2 * K, dtype=torch.float32, device=dz.device)
            if add3 is not None:
                (t2, a0, a1), (t3, b0, b1) = add3
                L.call('tcvom_bn_bwd_reduce3', L.ptr(dz), L.ptr(t2), a0, a1, L.ptr(t3), b0, b1, L.ptr(y), None if ctx.res_mask else L.ptr(r1),
                       L.ptr(r1) if ctx.res_mask else None, ss, saved, L.ptr(partial), P, K, cfg.act, yf, nf, stride, st)
            else:
                L.call('tcvom_bn_bwd_reduce_mask' if ctx.res_mask else 'tcvom_bn_bwd_reduce_ranged', L.ptr(dz), L.ptr(dz2), L.ptr(y), L.ptr(r1), ss,
                       saved, L.ptr(partial), P, K, cfg.act, yf, nf, stride, zf0, zf1, st)      # (r1 = the activation mask when res_mask)
            # gamma / beta gradients of the S calls of one BatchNorm add up in the bank (delivered by the bank token)
            dgp, dbp = (C.c_void_p(a) for a in bank.bn_grad_ptrs(cfg.bn))
            coef = torch.empty(nf * 3 * K, dtype=torch.float32, device=dz.device)
            scratch = torch.empty(nf * 128 * K, dtype=torch.float64, device=dz.device) if groups > 256 else None
            sync = ctx.sync if ctx.training else None
            if cfg.group_norm:
                L.call('tcvom_gn_bwd_finalize', L.ptr(partial), groups, K, P, cfg.bn.num_groups, L.ptr(gamma), saved, dgp, dbp,
                       L.ptr(coef), L.ptr(scratch), nf, stride, st)
            elif sync is None:
                L.call('tcvom_bn_bwd_finalize', L.ptr(partial), groups, K, P, L.ptr(gamma), saved, dgp, dbp,
                       L.ptr(coef), L.ptr(scratch), 1, nf, stride, _sn_dot(cfg, ctx.call, ctx.training, None), st)
            elif sync.mailbox is not None and sync.mailbox.fits(nf, K):
                L.call('tcvom_bn_bwd_finalize_sync', L.ptr(partial), groups, K, P * sync.world, L.ptr(gamma), saved, dgp, dbp,
                       L.ptr(coef), L.ptr(scratch), 1, nf, stride, sync.mailbox.next(), _sn_dot(cfg, ctx.call, ctx.training, sync), st)
            else:
                group, world = sync.group, sync.world
                local = torch.empty(nf * 2 * K, dtype=torch.float64, device=dz.device)
                L.call('tcvom_bn_reduce_sums', L.ptr(partial), groups, K, L.ptr(local), L.ptr(scratch), nf, st)
                total = local.clone()
                SYNC_ALLREDUCES[0] += 1
                dist.all_reduce(total, group=group)
                L.call('tcvom_bn_bwd_finalize_sums', L.ptr(total), L.ptr(local), K, P * world, L.ptr(gamma), saved,
                       dgp, dbp, L.ptr(coef), 1, nf, stride, _sn_dot(cfg, ctx.call, ctx.training, sync), st)
            if add3 is not None:
                (t2, a0, a1), (t3, b0, b1) = add3
                L.call('tcvom_bn_bwd_apply3', L.ptr(dz), L.ptr(t2), a0, a1, L.ptr(t3), b0, b1, L.ptr(y), None if ctx.res_mask else L.ptr(r1),
                       L.ptr(r1) if ctx.res_mask else None, ss, saved, L.ptr(coef), L.ptr(dy), L.ptr(dres1), None, P, K, cfg.act,
                       1 if (ctx.training or cfg.group_norm) else 0, 1 if cfg.pre_relu else 0, yf, nf, stride, st)
            elif THREE_ADDENDS and ctx.has_res2 and dz2 is not None and dz2_rng is None and ctx.needs_input_grad[6]:
                # res2 is added after the activation: its gradient is the op's whole incoming gradient dz + dz2 -- written by the same pass
                dsum = torch.empty_like(dz)
                L.call('tcvom_bn_bwd_apply3', L.ptr(dz), L.ptr(dz2), 0, nf, None, 0, 0, L.ptr(y), None if ctx.res_mask else L.ptr(r1),
                       L.ptr(r1) if ctx.res_mask else None, ss, saved, L.ptr(coef), L.ptr(dy), L.ptr(dres1), L.ptr(dsum), P, K, cfg.act,
                       1 if (ctx.training or cfg.group_norm) else 0, 1 if cfg.pre_relu else 0, yf, nf, stride, st)
            else:
                L.call('tcvom_bn_bwd_apply_mask' if ctx.res_mask else 'tcvom_bn_bwd_apply_ranged', L.ptr(dz), L.ptr(dz2), L.ptr(y), L.ptr(r1), ss, saved, L.ptr(coef), L.ptr(dy),
                       L.ptr(dres1), P, K, cfg.act, 1 if (ctx.training or cfg.group_norm) else 0, 1 if cfg.pre_relu else 0, yf, nf, stride,
                       zf0, zf1, st)
        if cfg.bn is not None and ctx.has_bias:
            # a conv bias in front of a BatchNorm (DIM encoder): its gradient is the column sum of dy -- ~0 with
            # batch statistics, but not None (weight decay still acts on it in the reference's Adam)
            dbias = torch.empty(K, dtype=torch.float32, device=dz.device)
            L.call('tcvom_colsum', L.ptr(dy), L.ptr(dbias), P * nf, K, K, st)
        dx = None
        if spec.needs_dgrad and ctx.needs_input_grad[0]:
            # zero-padded concat inputs (spec.cpad > spec.C): the gradient keeps the padded layout, zeros in the padding
            cx = spec.cpad if spec.cpad > 8 else spec.C
            alloc = torch.zeros if (cx != spec.C and not ctx.x_pad_unread) else torch.empty
            dx = alloc((geo.N * nf, geo.H, geo.W, cx), dtype=H16, device=dz.device)
            _launch_conv(geo.dgrad, dy, bank.bwd_ptr(spec, ctx.call), dx, None, None, ACT_NONE, st, nf, ctx.wsb)
        # the weight gradients are deferred: the bank runs the S calls of a layer as ONE launch at the end of backward
        bank.defer_wgrad(spec, ctx.call, dy, x, geo, nf)
        # res2 is added after the activation: its gradient is the op's WHOLE incoming gradient, deposited part included
        dres2 = (dsum if dsum is not None else dz if dz2 is None else dz + dz2) if ctx.has_res2 else None
        if dres1 is not None and ctx.res1_stash is not None:
            # the producer of res1 (a conv + BatchNorm op further up) adds this to its incoming gradient inside its
            # BatchNorm-backward kernels (dz2 of tcvom_bn_bwd_reduce / tcvom_bn_bwd_apply): no element-wise add pass
            ctx.res1_stash.deposit(dres1)
            dres1 = None
        if dx is not None and ctx.x_stash is not None and dx.shape == x.shape:
            ctx.x_stash.deposit(dx)                        # likewise the data gradient, when the input came from such an op
            dx = None
        _unscale_(bank, dbias)
        return dx, None, dgamma, dbeta, dbias, dres1, dres2, None, None, None


def _backward_active(ctx, dz, dz2, ranged=()):
    """Backward of a frame-batched conv + BatchNorm op whose output reaches the loss through the interior frames only
    (ctx.active = (f0, f1) of nf frames; the shortcut branches fea1 .. fea3 of the GCA encoder feed the decoder TAIL, which runs
    for the interior frames -- VMN_model.py:107-110 -- so the reference's autograd never visits them for the end frames).  The
    incoming gradient of the other frames is identically zero: their BatchNorm backward, data gradient and weight gradient are
    skipped instead of computed on zeros (at 1080p the os1 / os2 / os4 branches are HBM-bound passes over 400 MB tensors).  The
    returned input gradient is a full tensor, zero in the skipped frames."""
    cfg, geo, nf = ctx.cfg, ctx.geo, ctx.nf
    spec, bank = cfg.spec, cfg.bank
    f0, f1 = ctx.active
    nfa, N = f1 - f0, geo.N
    st = L.stream_ptr()
    K = spec.K
    P = geo.out_pixels
    x, y, gamma, _r1 = ctx.saved_tensors
    if ctx.window_id != bank.window_id:
        raise RuntimeError('conv %s: backward of a window after a newer forward of the same network is not supported' % spec.name)
    fr = lambda t: t[f0 * N:f1 * N] if t is not None else None          # frame-major: a contiguous slice
    # the incoming gradient of the active frames: slices of full-size gradients and / or the row-range gradients frame_slice
    # deposited (those never existed at full size: no zero fill, no copy); the kernels take two addends
    parts = [t for t in (fr(dz), fr(dz2)) if t is not None] + list(ranged)
    assert parts, 'no gradient for the active frames'
    while len(parts) > 2:
        parts = [parts[0] + parts[1]] + parts[2:]
    dza, dz2a = parts[0], (parts[1] if len(parts) > 1 else None)
    ya, xa = fr(y), fr(x)
    stride = ctx.slot_stride
    ss = C.c_void_p(ctx.ss.value + 4 * f0 * stride)
    saved = C.c_void_p(ctx.saved.value + 4 * f0 * stride)
    groups = L.call('tcvom_bn_bwd_groups_n', P, K, nfa)
    dev = dza.device
    partial = torch.empty(nfa * groups * 2 * K, dtype=torch.float32, device=dev)
    yf = _y_mode(y)
    L.call('tcvom_bn_bwd_reduce', L.ptr(dza), L.ptr(dz2a), L.ptr(ya), None, ss, saved, L.ptr(partial), P, K, cfg.act, yf, nfa, stride, st)
    dgp, dbp = (C.c_void_p(a) for a in bank.bn_grad_ptrs(cfg.bn))
    coef = torch.empty(nfa * 3 * K, dtype=torch.float32, device=dev)
    scratch = torch.empty(nfa * 128 * K, dtype=torch.float64, device=dev) if groups > 256 else None
    sync = ctx.sync if ctx.training else None
    if cfg.group_norm:
        L.call('tcvom_gn_bwd_finalize', L.ptr(partial), groups, K, P, cfg.bn.num_groups, L.ptr(gamma), saved, dgp, dbp,
               L.ptr(coef), L.ptr(scratch), nfa, stride, st)
    elif sync is None:
        L.call('tcvom_bn_bwd_finalize', L.ptr(partial), groups, K, P, L.ptr(gamma), saved, dgp, dbp,
               L.ptr(coef), L.ptr(scratch), 1, nfa, stride, _sn_dot(cfg, ctx.call + f0, ctx.training, None), st)
    elif sync.mailbox is not None and sync.mailbox.fits(nfa, K):
        # (every rank skips the same frames: the exchange carries the sums of the active frames only)
        L.call('tcvom_bn_bwd_finalize_sync', L.ptr(partial), groups, K, P * sync.world, L.ptr(gamma), saved, dgp, dbp,
               L.ptr(coef), L.ptr(scratch), 1, nfa, stride, sync.mailbox.next(), _sn_dot(cfg, ctx.call + f0, ctx.training, sync), st)
    else:
        local = torch.empty(nfa * 2 * K, dtype=torch.float64, device=dev)
        L.call('tcvom_bn_reduce_sums', L.ptr(partial), groups, K, L.ptr(local), L.ptr(scratch), nfa, st)
        total = local.clone()
        SYNC_ALLREDUCES[0] += 1
        dist.all_reduce(total, group=sync.group)
        L.call('tcvom_bn_bwd_finalize_sums', L.ptr(total), L.ptr(local), K, P * sync.world, L.ptr(gamma), saved,
               dgp, dbp, L.ptr(coef), 1, nfa, stride, _sn_dot(cfg, ctx.call + f0, ctx.training, sync), st)
    dya = torch.empty(ya.shape, dtype=H16, device=dev)
    L.call('tcvom_bn_bwd_apply', L.ptr(dza), L.ptr(dz2a), L.ptr(ya), None, ss, saved, L.ptr(coef), L.ptr(dya),
           None, P, K, cfg.act, 1 if (ctx.training or cfg.group_norm) else 0, 1 if cfg.pre_relu else 0, yf, nfa, stride, st)
    dx = None
    if spec.needs_dgrad and ctx.needs_input_grad[0]:
        cx = spec.cpad if spec.cpad > 8 else spec.C
        if ctx.x_stash is not None and cx == x.shape[3] and (RANGED_DZ2 or ctx.x_tail_rows == (f0 * N, f1 * N)):
            # hand the producer of x the gradient of the active rows only (no full-size tensor): it either skips the same frames
            # or adds the rows inside its BatchNorm-backward kernels (dz2 with a frame range)
            dxa = torch.empty((N * nfa, geo.H, geo.W, cx), dtype=H16, device=dev)
            _launch_conv(geo.dgrad, dya, bank.bwd_ptr(spec, ctx.call + f0), dxa, None, None, ACT_NONE, st, nfa, ctx.wsb)
            ctx.x_stash.deposit(('rows', dxa, f0 * N, f1 * N))
        else:
            dx = torch.zeros((geo.N * nf, geo.H, geo.W, cx), dtype=H16, device=dev)
            _launch_conv(geo.dgrad, dya, bank.bwd_ptr(spec, ctx.call + f0), fr(dx), None, None, ACT_NONE, st, nfa, ctx.wsb)
    bank.defer_wgrad(spec, ctx.call + f0, dya, xa, geo, nfa)
    if dx is not None and ctx.x_stash is not None and dx.shape == x.shape:
        ctx.x_stash.deposit(dx)
        dx = None
    return dx, None, None, None, None, None, None, None, None, None


_ConvBNAct._backward_active = staticmethod(_backward_active)


# =============================================================================================
# Depthwise 3x3 + BatchNorm + ReLU6 (IndexNet base: models/Index/net.py:38-61, hlaspp.py:38-46)
# =============================================================================================
class DwCfg(object):
    """One depthwise 3x3 conv + BatchNorm (+ activation) site: weight [C, 1, 3, 3] fp32 (read by the kernel as it is -- 9 C
    values, not worth a packed copy), stride 1."""

    def __init__(self, bank, weight, bn, dilation=1, pad=0, act=ACT_RELU6):
        self.bank, self.weight, self.dilation, self.pad, self.act = bank, weight, int(dilation), int(pad), act
        self._bn_idx = bank.register_bn(bn)

    bn = property(lambda self: self.bank.bns[self._bn_idx])


class _DwBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, weight, gamma, beta, cfg, training):
        _need_cuda(x)
        bank, bn = cfg.bank, cfg.bn
        x = _c(x)
        NT, H, W, Cc = x.shape
        assert x.dtype == H16 and tuple(weight.shape) == (Cc, 1, 3, 3)
        sync = _sync_group(bn) if training else None
        nf = bank.frames_per_op
        assert NT % nf == 0
        N = NT // nf
        d, p = cfg.dilation, cfg.pad
        OH, OW = H + 2 * p - 2 * d, W + 2 * p - 2 * d
        st = L.stream_ptr()
        wt = weight.detach().reshape(Cc, 9).t().contiguous().float()          # [9][C] tap-major
        y = torch.empty((NT, OH, OW, Cc), dtype=H16, device=x.device)
        P = N * OH * OW
        stats, groups = None, 0
        if training:
            groups = L.call('tcvom_dw3x3_stats_groups', P, Cc)
            stats = torch.empty(nf * groups * 2 * Cc, dtype=torch.float32, device=x.device)
        L.call('tcvom_dw3x3', L.ptr(x), L.ptr(wt), L.ptr(y), L.ptr(stats), N, H, W, Cc, d, p, 0, nf, st)
        Pg = P * (sync[1] if sync is not None else 1)                  # SyncBatchNorm: statistics over the clips of all ranks
        ss_i, saved_i, slot_stride = bank.bn_slots(bn, nf, training, Pg)
        ss, saved = C.c_void_p(ss_i), C.c_void_p(saved_i)
        if training:
            scratch = torch.empty(nf * 128 * Cc, dtype=torch.float64, device=x.device) if groups > 256 else None
            if sync is None:
                L.call('tcvom_bn_finalize', L.ptr(stats), groups, Cc, P, P, L.ptr(gamma), L.ptr(beta), None, None,
                       float(bn.momentum), float(bn.eps), ss, saved, L.ptr(scratch), nf, slot_stride, st)
            elif sync.mailbox is not None and sync.mailbox.fits(nf, Cc):
                L.call('tcvom_bn_finalize_sync', L.ptr(stats), groups, Cc, Pg, Pg, L.ptr(gamma), L.ptr(beta), float(bn.eps), ss, saved,
                       L.ptr(scratch), nf, slot_stride, sync.mailbox.next(), st)
            else:
                sums = torch.empty(nf * 2 * Cc, dtype=torch.float64, device=x.device)
                L.call('tcvom_bn_reduce_sums', L.ptr(stats), groups, Cc, L.ptr(sums), L.ptr(scratch), nf, st)
                SYNC_ALLREDUCES[0] += 1
                dist.all_reduce(sums, group=sync[0])
                L.call('tcvom_bn_finalize_sums', L.ptr(sums), Cc, Pg, Pg, L.ptr(gamma), L.ptr(beta), float(bn.eps), ss, saved, nf,
                       slot_stride, st)
        else:
            slot_stride = 0
            L.call('tcvom_bn_eval_coeffs', Cc, L.ptr(gamma), L.ptr(beta), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                   float(bn.eps), ss, saved, st)
        z = torch.empty_like(y)
        L.call('tcvom_bn_apply', L.ptr(y), ss, None, None, L.ptr(z), P, Cc, cfg.act, 0, nf, slot_stride, st)
        ctx.cfg, ctx.training, ctx.nf, ctx.dims = cfg, training, nf, (N, H, W, OH, OW, Cc)
        ctx.ss, ctx.saved, ctx.slot_stride, ctx.window_id, ctx.sync = ss, saved, slot_stride, bank.window_id, sync
        ctx.save_for_backward(x, y, gamma, wt)
        return z

    @staticmethod
    def backward(ctx, dz):
        cfg, nf = ctx.cfg, ctx.nf
        bank = cfg.bank
        x, y, gamma, wt = ctx.saved_tensors
        N, H, W, OH, OW, Cc = ctx.dims
        if ctx.window_id != bank.window_id:
            raise RuntimeError('depthwise conv: backward of a window after a newer forward of the same network is not supported')
        st = L.stream_ptr()
        dz = _c(dz)
        P = N * OH * OW
        ss, saved, stride = ctx.ss, ctx.saved, ctx.slot_stride
        groups = L.call('tcvom_bn_bwd_groups_n', P, Cc, nf)
        partial = torch.empty(nf * groups * 2 * Cc, dtype=torch.float32, device=dz.device)
        L.call('tcvom_bn_bwd_reduce', L.ptr(dz), None, L.ptr(y), None, ss, saved, L.ptr(partial), P, Cc, cfg.act, 0, nf, stride, st)
        dgp, dbp = (C.c_void_p(a) for a in bank.bn_grad_ptrs(cfg.bn))
        coef = torch.empty(nf * 3 * Cc, dtype=torch.float32, device=dz.device)
        scratch = torch.empty(nf * 128 * Cc, dtype=torch.float64, device=dz.device) if groups > 256 else None
        if ctx.sync is None or not ctx.training:
            L.call('tcvom_bn_bwd_finalize', L.ptr(partial), groups, Cc, P, L.ptr(gamma), saved, dgp, dbp, L.ptr(coef), L.ptr(scratch),
                   1, nf, stride, None, st)
        elif ctx.sync.mailbox is not None and ctx.sync.mailbox.fits(nf, Cc):
            L.call('tcvom_bn_bwd_finalize_sync', L.ptr(partial), groups, Cc, P * ctx.sync.world, L.ptr(gamma), saved, dgp, dbp,
                   L.ptr(coef), L.ptr(scratch), 1, nf, stride, ctx.sync.mailbox.next(), None, st)
        else:
            group, world = ctx.sync.group, ctx.sync.world
            local = torch.empty(nf * 2 * Cc, dtype=torch.float64, device=dz.device)
            L.call('tcvom_bn_reduce_sums', L.ptr(partial), groups, Cc, L.ptr(local), L.ptr(scratch), nf, st)
            total = local.clone()
            SYNC_ALLREDUCES[0] += 1
            dist.all_reduce(total, group=group)
            L.call('tcvom_bn_bwd_finalize_sums', L.ptr(total), L.ptr(local), Cc, P * world, L.ptr(gamma), saved, dgp, dbp, L.ptr(coef),
                   1, nf, stride, None, st)
        dy = torch.empty(y.shape, dtype=H16, device=dz.device)
        L.call('tcvom_bn_bwd_apply', L.ptr(dz), None, L.ptr(y), None, ss, saved, L.ptr(coef), L.ptr(dy), None, P, Cc, cfg.act,
               1 if ctx.training else 0, 0, 0, nf, stride, st)
        d, p = cfg.dilation, cfg.pad
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=H16, device=dz.device)
            L.call('tcvom_dw3x3', L.ptr(dy), L.ptr(wt), L.ptr(dx), None, N, OH, OW, Cc, d, 2 * d - p, 1, nf, st)
        dw = torch.empty((9, Cc), dtype=torch.float32, device=dz.device)
        L.call('tcvom_dw3x3_wgrad', L.ptr(dy), L.ptr(x), L.ptr(dw), N, H, W, Cc, d, p, nf, st)
        _unscale_(bank, dw)
        return dx, None, dw.t().reshape(Cc, 1, 3, 3), None, None, None, None


def dw_bn_act(cfg, x, token, training):
    """Depthwise 3x3 (cfg.dilation, cfg.pad) + BatchNorm + cfg.act on NHWC bf16; the BatchNorm parameter gradients travel
    through the bank like those of conv_bn_act."""
    return _DwBNAct.apply(x, token, cfg.weight, cfg.bn.weight, cfg.bn.bias, cfg, training)


class _IndexPool(torch.autograd.Function):
    """x1..x4 [N, h2, w2, C], l [N, 2 h2, 2 w2, C] -> (idx_en * l, 4 * avg_pool2d(idx_en * l), idx_de) (csrc/indexnet.hip)."""

    @staticmethod
    def forward(ctx, x1, x2, x3, x4, l):
        x1, x2, x3, x4, l = _c(x1), _c(x2), _c(x3), _c(x4), _c(l)
        N, h2, w2, Cc = x1.shape
        assert tuple(l.shape) == (N, 2 * h2, 2 * w2, Cc) and l.dtype == H16
        xe, de = torch.empty_like(l), torch.empty_like(l)
        pooled = torch.empty_like(x1)
        L.call('tcvom_index_pool_fwd', L.ptr(x1), L.ptr(x2), L.ptr(x3), L.ptr(x4), L.ptr(l), L.ptr(xe), L.ptr(pooled), L.ptr(de),
               N, h2, w2, Cc, L.stream_ptr())
        ctx.save_for_backward(x1, x2, x3, x4, l)
        return xe, pooled, de

    @staticmethod
    def backward(ctx, dxe, dpooled, dde):
        x1, x2, x3, x4, l = ctx.saved_tensors
        N, h2, w2, Cc = x1.shape
        g = [None if t is None else _c(t) for t in (dxe, dpooled, dde)]
        dxs = [torch.empty_like(x1) for _ in range(4)]
        dl = torch.empty_like(l)
        L.call('tcvom_index_pool_bwd', L.ptr(x1), L.ptr(x2), L.ptr(x3), L.ptr(x4), L.ptr(l), L.ptr(g[0]), L.ptr(g[1]), L.ptr(g[2]),
               L.ptr(dxs[0]), L.ptr(dxs[1]), L.ptr(dxs[2]), L.ptr(dxs[3]), L.ptr(dl), N, h2, w2, Cc, L.stream_ptr())
        return dxs[0], dxs[1], dxs[2], dxs[3], dl


class _IndexUp(torch.autograd.Function):
    """concat(idx * nearest_x2(enc), low) along the channels (idx None: concat(enc, low)) (csrc/indexnet.hip)."""

    @staticmethod
    def forward(ctx, enc, idx, low):
        enc, low = _c(enc), _c(low)
        idx = _c(idx) if idx is not None else None
        N, H, W, C2 = low.shape
        C1 = enc.shape[3]
        out = torch.empty((N, H, W, C1 + C2), dtype=H16, device=low.device)
        L.call('tcvom_index_up_fwd', L.ptr(enc), L.ptr(idx), L.ptr(low), L.ptr(out), N, H, W, C1, C2, L.stream_ptr())
        ctx.save_for_backward(enc, idx)
        ctx.dims = (N, H, W, C1, C2)
        return out

    @staticmethod
    def backward(ctx, dout):
        enc, idx = ctx.saved_tensors
        N, H, W, C1, C2 = ctx.dims
        dout = _c(dout)
        denc = torch.empty_like(enc)
        didx = torch.empty_like(idx) if idx is not None else None
        dlow = torch.empty((N, H, W, C2), dtype=H16, device=dout.device)
        L.call('tcvom_index_up_bwd', L.ptr(dout), L.ptr(enc), L.ptr(idx), L.ptr(denc), L.ptr(didx), L.ptr(dlow), N, H, W, C1, C2,
               L.stream_ptr())
        return denc, didx, dlow


class _Conv5x5C1(torch.autograd.Function):
    """nn.Conv2d(1, 1, 5, padding=2, bias=False) on a fp32 [N, 1, H, W] map (the last layer of the IndexNet decoder)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = _c(x.float())
        N, _, H, W = x.shape
        w = _c(weight.detach().float().reshape(25))
        y = torch.empty_like(x)
        L.call('tcvom_conv5x5_c1', L.ptr(x), L.ptr(w), L.ptr(y), N, H, W, 0, L.stream_ptr())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, _, H, W = x.shape
        dy = _c(dy.float())
        dx = torch.empty_like(x)
        L.call('tcvom_conv5x5_c1', L.ptr(dy), L.ptr(w), L.ptr(dx), N, H, W, 1, L.stream_ptr())
        dw = torch.empty(25, dtype=torch.float32, device=x.device)
        L.call('tcvom_conv5x5_c1_wgrad', L.ptr(dy), L.ptr(x), L.ptr(dw), N, H, W, L.stream_ptr())
        return dx, dw.reshape(1, 1, 5, 5)


conv5x5_c1 = _Conv5x5C1.apply
index_pool = _IndexPool.apply
index_up = _IndexUp.apply


class _FrameSlice(torch.autograd.Function):
    """Rows [lo, hi) of a frame-major batch (the interior frames on their way to the decoder tail).  When the producer is a
    conv + BatchNorm op (`_tcvom_grad_stash` on the tensor) the gradient of the slice is DEPOSITED with it as a row-range
    gradient instead of being padded to full size: no zero fill of the end frames, no copy (at 1080p the os1 branch alone:
    400 MB of fill + 134 MB of copy per step).  A tail-only producer skips the other frames in its backward altogether; any
    other one adds the rows inside its BatchNorm-backward kernels."""

    @staticmethod
    def forward(ctx, t, lo, hi, stash):
        ctx.stash, ctx.rng, ctx.rows = stash, (lo, hi), t.shape[0]
        ctx.set_materialize_grads(False)
        return t[lo:hi]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        lo, hi = ctx.rng
        if ctx.stash is not None:
            ctx.stash.deposit(('rows', g, lo, hi))
            return None, None, None, None
        full = g.new_zeros((ctx.rows,) + tuple(g.shape[1:]))
        full[lo:hi] = g
        return full, None, None, None


THREE_ADDENDS = _os.environ.get('TCVOM_NO_ADD3') is None       # A/B switch: a third gradient addend inside the BatchNorm backward (round 6)
RANGED_DZ2 = _os.environ.get('TCVOM_NO_RANGED') is None          # A/B switch: row-range gradients added inside the BatchNorm backward


class _NeighbourSlices(torch.autograd.Function):
    """(centre, previous, next) frames of a frame-major batch for the decoder tail (VMN_model.py:107-110: frame i is decoded
    against frames i - 1 and i + 1): rows [B, (S-1)B), [0, (S-2)B), [2B, SB).  As three plain slices autograd zero-pads each
    gradient to full size and adds them (3 fills + 3 copies + 2 adds of the os8 feature map per step); here the backward writes
    the full gradient once -- for S = 3 the three gradients are simply its three frames."""

    @staticmethod
    def forward(ctx, t, B, S):
        ctx.B, ctx.S = B, S
        ctx.set_materialize_grads(False)
        return t[B:(S - 1) * B], t[0:(S - 2) * B], t[2 * B:S * B]

    @staticmethod
    def backward(ctx, gc, gp, gn):
        B, S = ctx.B, ctx.S
        ref = next((g for g in (gc, gp, gn) if g is not None), None)
        if ref is None:
            return None, None, None
        if S == 3 and gc is not None and gp is not None and gn is not None:
            return torch.cat([gp, gc, gn], 0), None, None
        full = ref.new_zeros((S * B,) + tuple(ref.shape[1:]))
        for g, lo in ((gc, B), (gp, 0), (gn, 2 * B)):
            if g is not None:
                full[lo:lo + (S - 2) * B] += g
        return full, None, None


def neighbour_slices(t, B, S):
    if not RANGED_DZ2:
        return t[B:(S - 1) * B], t[0:(S - 2) * B], t[2 * B:S * B]
    return _NeighbourSlices.apply(t, B, S)


def frame_slice(t, lo, hi):
    """t[lo:hi] along the frame-major batch dimension, see _FrameSlice."""
    if not torch.is_tensor(t):
        return t
    rows = getattr(t, '_tcvom_tail_rows', None)
    stash = getattr(t, '_tcvom_grad_stash', None)
    # a tail-only producer takes exactly its own rows; any other conv + BatchNorm op adds a row-range gradient inside its
    # BatchNorm-backward kernels (dz2 with a frame range)
    if not t.requires_grad or (rows != (lo, hi) if rows is not None else (stash is None or not RANGED_DZ2)):
        return t[lo:hi]
    return _FrameSlice.apply(t, lo, hi, stash)


def conv_bn_act(cfg, x, token, training, res1=None, res2=None):
    """conv (+BatchNorm +activation +residuals).  Skip-branch gradients: the output z of an op WITH a BatchNorm carries a list
    (`z._tcvom_grad_stash`); a later op that takes z as its residual input `res1` (the `out += identity` of a BasicBlock, whose
    conv1 reads the same z) deposits d(res1) there in its backward instead of returning it to autograd, and the op that produced z
    reads it as a second addend of its incoming gradient.  Autograd still orders the two backward calls (res1 is an input of the
    consumer), it just has one tensor less to add: 33 element-wise add launches per 1080p step."""
    bn = cfg.bn
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    stash = _GradStash() if (bn is not None and torch.is_grad_enabled()) else None
    z = _ConvBNAct.apply(x, token, gamma, beta, cfg.spec.bias, res1, res2, cfg, training, stash)
    z16 = cfg.__dict__.pop('_z16', None)
    if z16 is not None:
        set_f16_twin(z, z16)
    if stash is not None and z.requires_grad:
        z._tcvom_grad_stash = stash
        tf, nf = getattr(cfg.bank, 'tail_frames', None), cfg.bank.frames_per_op
        if (cfg.tail_only and tf is not None and nf > 1 and 0 <= tf[0] < tf[1] <= nf and tf[1] - tf[0] < nf and res1 is None
                and res2 is None and cfg.spec.bias is None):
            # (the condition under which _ConvBNAct.backward takes the frame-skipping path: frame_slice may deposit with it)
            n = z.shape[0] // nf
            z._tcvom_tail_rows = (tf[0] * n, tf[1] * n)
    return z


# =============================================================================================
# resampling / padding
# =============================================================================================
class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=H16, device=x.device)
        x16 = _AvgPool2.twin_in
        if x16 is not None:                      # fp16 island: pool the IEEE fp16 twin, write both formats
            y16 = torch.empty((N, H // 2, W // 2, Cc), dtype=torch.float16, device=x.device)
            L.call('tcvom_avgpool2_f16', L.ptr(_c(x16)), L.ptr(y), L.ptr(y16), N, H, W, Cc, L.stream_ptr())
            _AvgPool2.twin_out = y16
        else:
            L.call('tcvom_avgpool2', L.ptr(x), L.ptr(y), N, H, W, Cc, L.stream_ptr())
        ctx.shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, dtype=H16, device=dy.device)
        L.call('tcvom_upsample2', L.ptr(_c(dy)), L.ptr(dx), N, H, W, Cc, 0.25, L.stream_ptr())
        return dx


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        y = _c(y)
        N, h, w, Cc = y.shape
        x = torch.empty((N, 2 * h, 2 * w, Cc), dtype=H16, device=y.device)
        L.call('tcvom_upsample2', L.ptr(y), L.ptr(x), N, 2 * h, 2 * w, Cc, 1.0, L.stream_ptr())
        ctx.shape = (N, h, w, Cc)
        return x

    @staticmethod
    def backward(ctx, dx):
        N, h, w, Cc = ctx.shape
        dy = torch.empty(ctx.shape, dtype=H16, device=dx.device)
        L.call('tcvom_sumpool2', L.ptr(_c(dx)), L.ptr(dy), N, 2 * h, 2 * w, Cc, 1.0, L.stream_ptr())
        return dy


class _ReflectPad1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, H + 2, W + 2, Cc), dtype=H16, device=x.device)
        L.call('tcvom_reflect_pad1', L.ptr(x), L.ptr(y), N, H, W, Cc, L.stream_ptr())
        ctx.shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, dtype=H16, device=dy.device)
        L.call('tcvom_reflect_pad1_bwd', L.ptr(_c(dy)), L.ptr(dx), N, H, W, Cc, L.stream_ptr())
        return dx


_AvgPool2.twin_in = _AvgPool2.twin_out = None


def avgpool2(x):
    """AvgPool2d(2, 2); an input of the fp16 island (ops.f16_twin) is pooled from its IEEE fp16 twin and the result carries one."""
    _AvgPool2.twin_in, _AvgPool2.twin_out = f16_twin(x), None
    try:
        y = _AvgPool2.apply(x)
    finally:
        _AvgPool2.twin_in = None
    if _AvgPool2.twin_out is not None:
        set_f16_twin(y, _AvgPool2.twin_out)
        _AvgPool2.twin_out = None
    return y
upsample2 = _Upsample2.apply
reflect_pad1 = _ReflectPad1.apply


# =============================================================================================
# DIM base: MaxPool2d(2, return_indices) / MaxUnpool2d(2), and the 7x7 conv6 as unfold + dense GEMM
# =============================================================================================
class _MaxPool2Idx(torch.autograd.Function):
    """F.max_pool2d(x, 2, 2, return_indices=True) on NHWC bf16; idx = uint8 position inside the 2x2 window."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=H16, device=x.device)
        idx = torch.empty((N, H // 2, W // 2, Cc), dtype=torch.uint8, device=x.device)
        L.call('tcvom_maxpool2_idx', L.ptr(x), L.ptr(y), L.ptr(idx), N, H, W, Cc, L.stream_ptr())
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, Cc)
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        (idx,) = ctx.saved_tensors
        N, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, dtype=H16, device=dy.device)
        L.call('tcvom_unpool2', L.ptr(_c(dy)), L.ptr(idx), L.ptr(dx), N, H, W, Cc, L.stream_ptr())
        return dx


class _Unpool2(torch.autograd.Function):
    """F.max_unpool2d(y, idx, 2, 2): every value goes to the position its max-pool partner came from."""

    @staticmethod
    def forward(ctx, y, idx):
        y = _c(y)
        N, h, w, Cc = y.shape
        x = torch.empty((N, 2 * h, 2 * w, Cc), dtype=H16, device=y.device)
        L.call('tcvom_unpool2', L.ptr(y), L.ptr(idx), L.ptr(x), N, 2 * h, 2 * w, Cc, L.stream_ptr())
        ctx.save_for_backward(idx)
        return x

    @staticmethod
    def backward(ctx, dx):
        (idx,) = ctx.saved_tensors
        N, H, W, Cc = dx.shape
        dy = torch.empty((N, H // 2, W // 2, Cc), dtype=H16, device=dx.device)
        L.call('tcvom_pick2', L.ptr(_c(dx)), L.ptr(idx), L.ptr(dy), N, H, W, Cc, L.stream_ptr())
        return dy, None


class _ConvUnfoldDense(torch.autograd.Function):
    """Large-kernel conv + bias + ReLU as im2col + dense GEMM (DIM conv6: 7x7, 512 -> 4096 at os32, vggnet.py:56,97)."""

    @staticmethod
    def forward(ctx, x, token, bias, cfg):
        _need_cuda(x)
        spec, bank = cfg.spec, cfg.bank
        x = _c(x)
        N, H, W, Cc = x.shape
        assert Cc == spec.cpad == spec.C and spec.R == spec.S and spec.stride == 1 and spec.pad == spec.R // 2
        call = bank.next_call(spec)
        st = L.stream_ptr()
        P, K, kred = N * H * W, spec.K, spec.T * Cc
        u = torch.empty((P, kred), dtype=H16, device=x.device)
        L.call('tcvom_unfold', L.ptr(x), L.ptr(u), N, H, W, Cc, spec.R, st)
        y = torch.empty((N, H, W, K), dtype=H16, device=x.device)
        d = dense_desc(P, K, kred, K)
        d.act = ACT_RELU
        L.call('tcvom_conv_igemm', L.ptr(u), bank.fwd_ptr(spec, call), L.ptr(y), L.ptr(bias), None, None, None, C.byref(d), st)
        ctx.cfg, ctx.call, ctx.shape = cfg, call, (N, H, W, Cc)
        ctx.save_for_backward(u, y)
        return y

    @staticmethod
    def backward(ctx, dz):
        u, y = ctx.saved_tensors
        spec, bank = ctx.cfg.spec, ctx.cfg.bank
        N, H, W, Cc = ctx.shape
        st = L.stream_ptr()
        P, K, kred = N * H * W, spec.K, spec.T * Cc
        dz = _c(dz)
        dy = torch.empty_like(dz)
        L.call('tcvom_relu_bwd', L.ptr(dz), L.ptr(y), L.ptr(dy), dz.numel(), 0.0, st)
        dbias = torch.empty(K, dtype=torch.float32, device=dz.device)
        L.call('tcvom_colsum', L.ptr(dy), L.ptr(dbias), P, K, K, st)
        # du[p][c*T + t] = sum_k dy[p][k] w[k][t][c]: the data-gradient weights are packed [C][T][K]
        du = torch.empty((P, kred), dtype=H16, device=dz.device)
        L.call('tcvom_conv_igemm', L.ptr(dy), bank.bwd_ptr(spec, ctx.call), L.ptr(du), None, None, None, None,
               C.byref(dense_desc(P, kred, K, kred)), st)
        dx = torch.empty((N, H, W, Cc), dtype=H16, device=dz.device)
        L.call('tcvom_fold', L.ptr(du), L.ptr(dx), N, H, W, Cc, spec.R, st)
        L.call('tcvom_wgrad_igemm', L.ptr(dy), L.ptr(u), bank.dw_ptr(spec, ctx.call), C.byref(dense_tt_desc(P, K, kred)), K, st)
        _unscale_(bank, dbias)
        return dx, None, dbias, None


def conv_unfold_dense(cfg, x, token):
    return _ConvUnfoldDense.apply(x, token, cfg.spec.bias, cfg)


maxpool2_idx = _MaxPool2Idx.apply
unpool2 = _Unpool2.apply


# =============================================================================================
# decoder head: conv 3x3 (C -> 1, bias) + (tanh + 1)/2      -> alpha fp32 [N,1,H,W]
# =============================================================================================
class _HeadConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, ksize, mode):
        x = _c(x)
        N, H, W, Cc = x.shape
        T = ksize * ksize
        wt = weight.detach().reshape(Cc, T).t().contiguous()          # [T][C] tap-major
        alpha = torch.empty((N, 1, H, W), dtype=torch.float32, device=x.device)
        L.call('tcvom_head_conv_fwd', L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(alpha), N, H, W, Cc, ksize, mode, L.stream_ptr())
        ctx.save_for_backward(x, wt, alpha)
        ctx.ksize, ctx.mode = ksize, mode
        return alpha

    @staticmethod
    def backward(ctx, dalpha):
        x, wt, alpha = ctx.saved_tensors
        N, H, W, Cc = x.shape
        T = ctx.ksize * ctx.ksize
        dalpha = _c(dalpha.float())
        dx = torch.empty_like(x)
        dpre = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
        reps = 16                                               # replicas of dw: 16x less atomic contention
        dw = torch.empty((reps, T, Cc), dtype=torch.float32, device=x.device)
        db = torch.empty(reps, dtype=torch.float32, device=x.device)
        L.call('tcvom_head_conv_bwd', L.ptr(dalpha), L.ptr(alpha), L.ptr(x), L.ptr(wt), L.ptr(dx), L.ptr(dpre), L.ptr(dw),
               L.ptr(db), N, H, W, Cc, ctx.ksize, ctx.mode, reps, L.stream_ptr())
        dw = dw.sum(0) if reps > 1 else dw[0]
        db = db.sum(0, keepdim=True)
        return dx, dw.t().reshape(1, Cc, ctx.ksize, ctx.ksize), db, None, None


def head_conv(x, weight, bias, ksize=3, mode=0):
    """Final conv C -> 1 with its output map: ksize 3 / mode 0 = (tanh + 1) / 2 (GCA), ksize 5 / mode 1 = clamp(0, 1) (DIM)."""
    return _HeadConv.apply(x, weight, bias, ksize, mode)




# =============================================================================================
# FBA base: MaxPool2d(3, 2, 1), pyramid pooling, bilinear up-sampling into concat buffers, fused head
# =============================================================================================
class _MaxPool3S2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) (models/FBA/resnet_GN_WS.py:101) on NHWC bf16."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, Cc = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, OH, OW, Cc), dtype=H16, device=x.device)
        idx = torch.empty((N, OH, OW, Cc), dtype=torch.uint8, device=x.device)
        L.call('tcvom_maxpool3s2', L.ptr(x), L.ptr(y), L.ptr(idx), N, H, W, Cc, L.stream_ptr())
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, dtype=H16, device=dy.device)
        L.call('tcvom_maxpool3s2_bwd', L.ptr(_c(dy)), L.ptr(idx), L.ptr(dx), N, H, W, Cc, L.stream_ptr())
        return dx


class _PyramidPool(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(s) for every s of `scales` in one op (models/VMN/VMN_FBA.py:26-30): x [N,h,w,C] bf16 ->
    tuple of [N,s,s,C] bf16.  The backward adds the gradients of all scales in one pass over x."""

    @staticmethod
    def forward(ctx, x, scales, link=None):
        # link (a dict shared with the _PyramidConcat that takes the same x): the concat's backward leaves its gradient slice there
        # and this backward -- which autograd runs later: the pooled maps feed the concat -- adds it inside the pooling-gradient kernel
        ctx.link = link
        x = _c(x)
        N, h, w, Cc = x.shape
        st = L.stream_ptr()
        n = len(scales)
        if 1 <= n <= 4 and 256 % (Cc // 8) == 0:
            # one pass over x for all scales (tcvom_adaptive_avgpool_multi)
            o32 = [torch.empty((N, s, s, Cc), dtype=torch.float32, device=x.device) for s in scales]
            ptrs = (C.c_void_p * n)(*[o.data_ptr() for o in o32])
            sc = (C.c_int32 * n)(*[int(s) for s in scales])
            nfl = L.call('tcvom_adaptive_avgpool_scratch_floats', C.cast(sc, C.c_void_p), n, N, h, w, Cc)
            if nfl > 0:          # per-cell partial sums + one combine launch: no atomics (8.8 M of them per 1080p window otherwise)
                scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
                L.call('tcvom_adaptive_avgpool_multi_ws', L.ptr(x), C.cast(ptrs, C.c_void_p), C.cast(sc, C.c_void_p), n, L.ptr(scratch),
                       N, h, w, Cc, st)
            else:
                L.call('tcvom_adaptive_avgpool_multi', L.ptr(x), C.cast(ptrs, C.c_void_p), C.cast(sc, C.c_void_p), n, N, h, w, Cc, st)
            outs = [o.to(H16) for o in o32]
        else:
            outs = []
            for s in scales:
                o = torch.empty((N, s, s, Cc), dtype=torch.float32, device=x.device)
                L.call('tcvom_adaptive_avgpool', L.ptr(x), L.ptr(o), N, h, w, Cc, s, st)
                outs.append(o.to(H16))
        ctx.shape, ctx.scales = (N, h, w, Cc), tuple(scales)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        N, h, w, Cc = ctx.shape
        gs = [_c(d).float() for d in douts]
        n = len(gs)
        ptrs = (C.c_void_p * n)(*[g.data_ptr() for g in gs])
        sc = (C.c_int32 * n)(*ctx.scales)
        dx = torch.empty(ctx.shape, dtype=H16, device=gs[0].device)
        pend = ctx.link.pop('dbuf', None) if ctx.link is not None else None
        if pend is not None:
            dbuf, cpad = pend
            L.call('tcvom_adaptive_avgpool_bwd_add', C.cast(ptrs, C.c_void_p), C.cast(sc, C.c_void_p), n, L.ptr(dx), L.ptr(dbuf), cpad,
                   N, h, w, Cc, L.stream_ptr())
        else:
            L.call('tcvom_adaptive_avgpool_bwd', C.cast(ptrs, C.c_void_p), C.cast(sc, C.c_void_p), n, L.ptr(dx), N, h, w, Cc, L.stream_ptr())
        return dx, None, None


class _PyramidConcat(torch.autograd.Function):
    """cat(conv5, bilinear(pooled_s -> h x w) for every scale) zero-padded to `cpad` channels (VMN_FBA.py:25-31)."""

    @staticmethod
    def forward(ctx, cpad, link, x, *maps):
        x = _c(x)
        N, h, w, Cx = x.shape
        st = L.stream_ptr()
        ctx.link = link
        full = Cx + sum(m.shape[3] for m in maps) == cpad      # every channel gets written: no zero fill (3072 = 2048 + 4 x 256)
        buf = (torch.empty if full else torch.zeros)((N, h, w, cpad), dtype=H16, device=x.device)
        buf[..., :Cx].copy_(x)
        off = Cx
        shapes = []
        for m in maps:
            m = _c(m)
            _, hs, ws, Cm = m.shape
            L.call('tcvom_bilinear', L.ptr(m), L.ptr(buf), N, hs, ws, h, w, Cm, Cm, 0, cpad, off, st)
            shapes.append((hs, ws, Cm, off))
            off += Cm
        assert off <= cpad
        ctx.geo = (N, h, w, Cx, cpad, shapes)
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        N, h, w, Cx, cpad, shapes = ctx.geo
        dbuf = _c(dbuf)
        st = L.stream_ptr()
        if ctx.link is not None and cpad % 8 == 0 and PPM_LINK:
            ctx.link['dbuf'] = (dbuf, cpad)                     # added by the pooling backward of the same x (no copy, no autograd add)
            dx = None
        else:
            dx = dbuf[..., :Cx].contiguous()
        dmaps = []
        for hs, ws, Cm, off in shapes:
            d = torch.empty((N, hs, ws, Cm), dtype=torch.float32, device=dbuf.device)
            L.call('tcvom_bilinear_small_bwd', L.ptr(dbuf), L.ptr(d), N, hs, ws, h, w, Cm, cpad, off, st)
            dmaps.append(d.to(H16))
        return (None, None, dx) + tuple(dmaps)


class _Up2Concat(torch.autograd.Function):
    """cat(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), skip) zero-padded to `cpad` channels
    (models/VMN/VMN_FBA.py:37-48): the up-sampling writes straight into its slice of the concat buffer."""

    @staticmethod
    def forward(ctx, cpad, x, skip):
        x, skip = _c(x), _c(skip)
        N, h, w, Cx = x.shape
        Cs = skip.shape[3]
        assert skip.shape[:3] == (N, 2 * h, 2 * w) and Cx + Cs <= cpad and Cx % 8 == 0
        buf = torch.empty((N, 2 * h, 2 * w, cpad), dtype=H16, device=x.device)
        if Cs % 8 == 0 and cpad % 8 == 0:
            # up-sampling, skip copy and zero padding in one pass of whole rows
            L.call('tcvom_up2_concat', L.ptr(x), L.ptr(skip), L.ptr(buf), N, h, w, Cx, Cs, cpad, L.stream_ptr())
        else:
            L.call('tcvom_bilinear', L.ptr(x), L.ptr(buf), N, h, w, 2 * h, 2 * w, Cx, Cx, 0, cpad, 0, L.stream_ptr())
            buf[..., Cx:Cx + Cs].copy_(skip)
            if Cx + Cs < cpad:
                buf[..., Cx + Cs:].zero_()
        ctx.geo = (N, h, w, Cx, Cs, cpad)
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        N, h, w, Cx, Cs, cpad = ctx.geo
        dbuf = _c(dbuf)
        dx = torch.empty((N, h, w, Cx), dtype=H16, device=dbuf.device)
        L.call('tcvom_bilinear_up2_bwd', L.ptr(dbuf), L.ptr(dx), N, h, w, Cx, cpad, 0, L.stream_ptr())
        dskip = dbuf[..., Cx:Cx + Cs].contiguous() if ctx.needs_input_grad[2] else None
        return None, dx, dskip


class _FbaHead(torch.autograd.Function):
    """conv_up4[4] (1x1, 16 -> 7) + clamp / sigmoid + fba_fusion (models/VMN/VMN_FBA.py:50-57): x [N,H,W,16] bf16,
    img fp32 [N,3,H,W] (a view with image stride img.stride(0)) -> pred fp32 [N,7,H,W] = (alpha, F, B)."""
    REPLICAS = 16

    @staticmethod
    def forward(ctx, x, weight, bias, img):
        x = _c(x)
        N, H, W, Cc = x.shape
        assert Cc == 16 and weight.numel() == 7 * 16 and img.shape == (N, 3, H, W) and img.stride()[1:] == (H * W, W, 1)
        w2 = weight.detach().reshape(7, 16).contiguous().float()
        pred = torch.empty((N, 7, H, W), dtype=torch.float32, device=x.device)
        L.call('tcvom_fba_head_fwd', L.ptr(x), L.ptr(w2), L.ptr(bias), L.ptr(img), L.ptr(pred), N, H * W, img.stride(0), 7 * H * W,
               L.stream_ptr())
        ctx.save_for_backward(x, w2, bias, img)
        ctx.wshape = weight.shape
        return pred

    @staticmethod
    def backward(ctx, dpred):
        x, w2, bias, img = ctx.saved_tensors
        N, H, W, _ = x.shape
        dpred = _c(dpred)
        R = _FbaHead.REPLICAS
        dx = torch.empty_like(x)
        dw = torch.zeros((R, 7, 16), dtype=torch.float32, device=x.device)
        db = torch.zeros((R, 7), dtype=torch.float32, device=x.device)
        L.call('tcvom_fba_head_bwd', L.ptr(x), L.ptr(w2), L.ptr(bias), L.ptr(img), L.ptr(dpred), L.ptr(dx), L.ptr(dw), L.ptr(db), R, N,
               H * W, img.stride(0), 7 * H * W, L.stream_ptr())
        return dx, dw.sum(0).reshape(ctx.wshape), db.sum(0), None


maxpool3s2 = _MaxPool3S2.apply


PPM_LINK = _os.environ.get('TCVOM_NO_PPM_LINK') is None              # A/B switch


def pyramid_pool(x, scales, link=None):
    return _PyramidPool.apply(x, tuple(scales), link)


def pyramid_concat(cpad, x, maps, link=None):
    buf = _PyramidConcat.apply(cpad, link, x, *maps)
    buf._tcvom_pad_unread = True               # (its backward reads the channel slices of x and the maps only)
    return buf


def up2_concat(cpad, x, skip):
    buf = _Up2Concat.apply(cpad, x, skip)
    buf._tcvom_pad_unread = True
    return buf


fba_head = _FbaHead.apply


# =============================================================================================
# Temporal attention core
# =============================================================================================
class _TamAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kb, kf, v, mask_u8, window):
        ctx.set_materialize_grads(False)         # the attention maps may not reach the loss: no zero tensors for them
        q, kb, kf, v = _c(q), _c(kb), _c(kf), _c(v)
        B, H, W, Cc = q.shape
        out = torch.empty_like(q)
        w2 = window * window
        attb = torch.empty((B, w2, H * W), dtype=torch.float32, device=q.device)
        attf = torch.empty((B, w2, H * W), dtype=torch.float32, device=q.device)
        work = torch.empty(B * H * W + 1, dtype=torch.int32, device=q.device)      # compacted list of the unknown pixels
        L.call('tcvom_tam_fwd', L.ptr(q), L.ptr(kb), L.ptr(kf), L.ptr(v), L.ptr(mask_u8), L.ptr(out), L.ptr(attb),
               L.ptr(attf), L.ptr(work), B, H, W, Cc, window, L.stream_ptr())
        ctx.save_for_backward(q, kb, kf, mask_u8, work)
        ctx.window = window
        return out, attb, attf

    @staticmethod
    def backward(ctx, dout, dattb, dattf):
        q, kb, kf, mask, work = ctx.saved_tensors
        B, H, W, Cc = q.shape
        w2 = ctx.window * ctx.window
        if dout is None and dattb is None and dattf is None:
            return (None,) * 6
        dout = _c(dout) if dout is not None else torch.zeros_like(q)
        dq, dkb, dkf = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        pbuf = torch.empty((B, 2, w2, H * W), dtype=torch.float32, device=q.device)
        dsbuf = torch.empty_like(pbuf)
        db = _c(dattb) if dattb is not None else None
        df = _c(dattf) if dattf is not None else None
        L.call('tcvom_tam_bwd', L.ptr(q), L.ptr(kb), L.ptr(kf), L.ptr(mask), L.ptr(dout), L.ptr(db), L.ptr(df), L.ptr(dq),
               L.ptr(dkb), L.ptr(dkf), L.ptr(pbuf), L.ptr(dsbuf), L.ptr(work), B, H, W, Cc, ctx.window, L.stream_ptr())
        return dq, dkb, dkf, dout, None, None


tam_attention = _TamAttention.apply


# =============================================================================================
# Guided contextual attention core:  (g8, alpha, unknown) -> fold(P V)/4
# =============================================================================================
def _r64(n):
    return (n + 63) // 64 * 64


GCA_KMAJOR_A = _os.environ.get('TCVOM_NO_GCA_KMAJOR_A') is None              # A/B switch: forward O = P V reads V k-major (no V^T)
GCA_KMAJOR = _os.environ.get('TCVOM_NO_GCA_KMAJOR') is None                  # A/B switch: dV / M' read P / T k-major (no P^T / T^T)
GCA_FUSED_SOFTMAX = _os.environ.get('TCVOM_NO_FUSED_SOFTMAX', '0') != '1'      # A/B switch (tools/ab_bench.sh TCVOM_NO_FUSED_SOFTMAX)



GCA_KEEP = None


class _GcaAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g8, alpha, unk_u8):
        g8, alpha = _c(g8), _c(alpha)
        B, h8, w8, CG = g8.shape
        Ca = alpha.shape[3]
        dev = g8.device
        st = L.stream_ptr()
        N = (h8 // 2) * (w8 // 2)
        ld = _r64(N)
        D, DV = 9 * CG, 16 * Ca
        G = torch.empty((B, N, D), dtype=H16, device=dev)
        scales = torch.empty((B, 2), dtype=torch.float32, device=dev)
        cvec = torch.empty((B, N), dtype=torch.float32, device=dev)
        dvec = torch.empty((B, N), dtype=torch.float32, device=dev)
        nrm = torch.empty((B, N), dtype=torch.float32, device=dev)
        L.call('tcvom_gca_prepare', L.ptr(g8), L.ptr(unk_u8), L.ptr(G), L.ptr(scales), L.ptr(cvec), L.ptr(dvec), L.ptr(nrm),
               B, h8, w8, CG, st)
        V = torch.empty((B, N, DV), dtype=H16, device=dev)
        L.call('tcvom_gca_value_patches', L.ptr(alpha), L.ptr(V), B, h8, w8, Ca, st)
        # the value aggregation reads V as it lies in memory (k-major A operand of the 256-tile GEMM) when the shape fills the chip;
        # else an NT GEMM on the transposed copy V^T
        kmajor = GCA_KMAJOR and GCA_KMAJOR_A and ld % 256 == 0 and N >= 256 and DV >= 256 and ((N + 255) // 256) * (DV // 256) * B >= 192
        P = torch.empty((B, N, ld), dtype=H16, device=dev)
        # O[i][v] = sum_j P[i][j] V[j][v]             (rows m = v, columns n = queries i, reduce j)
        # (fp32: the backward forms sum_j P dP as <dO_i, O_i>; with a peaked softmax dP[i][i] - <dO_i, O_i> cancels to ~0 and
        # a bf16-rounded O would leave its rounding error as the gradient)
        O = torch.empty((B, N, DV), dtype=torch.float32, device=dev)
        # (measured and dropped, round 3: scores -> softmax -> P V frame by frame, so that a frame's P is still in the Infinity Cache
        #  when its GEMM reads it: 28.16 vs 28.13 ms per step -- the kernel is not bound by where P comes from)
        # S'[i][j] = c_j <G_i, G_j> - d_j [i==j]     (rows m = keys j, columns n = queries i)
        tiles = ((N + 255) // 256) * (ld // 256) * B if ld % 256 == 0 else 0
        if GCA_FUSED_SOFTMAX and tiles >= 192 and L.call('tcvom_gca_scores_softmax_ok', N, D, ld, B):
            # scores + softmax without the fp32 N x N matrix: the score GEMM's epilogue writes exp(S' - tile row max) and the
            # per-tile (max, sum) of every row, a second pass rescales the rows in place (csrc/gemm256.hip EPI 3)
            stats = torch.empty((B, N, ld // 256, 2), dtype=torch.float32, device=dev)
            L.call('tcvom_gca_scores_exp', L.ptr(G), L.ptr(cvec), L.ptr(dvec), L.ptr(P), L.ptr(stats), N, D, ld, B, st)
            L.call('tcvom_gca_softmax_rescale', L.ptr(P), L.ptr(stats), N, ld, B, st)
        else:
            S = torch.empty((B, N, ld), dtype=torch.float32, device=dev)
            d = dense_desc(N, N, D, ld, batch=B, in_bstride=N * D, w_bstride=N * D, out_bstride=N * ld, vec_bstride=N, out_fp32=True)
            L.call('tcvom_conv_igemm', L.ptr(G), L.ptr(G), L.ptr(S), None, L.ptr(cvec), L.ptr(dvec), None, C.byref(d), st)
            L.call('tcvom_row_softmax', L.ptr(S), L.ptr(P), B * N, N, ld, ld, st)
            del S
        if kmajor:
            L.call('tcvom_gca_pv', L.ptr(P), L.ptr(V), L.ptr(O), N, DV, ld, B, st)
        else:
            Vt = torch.empty((B, DV, ld), dtype=H16, device=dev)
            L.call('tcvom_transpose_bf16', L.ptr(V), L.ptr(Vt), N, DV, DV, ld, B, N * DV, DV * ld, st)
            d2 = dense_desc(N, DV, ld, DV, batch=B, in_bstride=N * ld, w_bstride=DV * ld, out_bstride=N * DV, out_fp32=True)
            L.call('tcvom_conv_igemm', L.ptr(P), L.ptr(Vt), L.ptr(O), None, None, None, None, C.byref(d2), st)
            del Vt
        y = torch.empty((B, h8, w8, Ca), dtype=H16, device=dev)
        L.call('tcvom_gca_fold_f32', L.ptr(O), L.ptr(y), B, h8, w8, Ca, st)
        ctx.save_for_backward(G, P, V, cvec, nrm, O)
        if GCA_KEEP is not None:
            GCA_KEEP.append(P)                       # (tools/gca_sparsity.py: statistics of the attention matrix)
        ctx.dims = (B, h8, w8, CG, Ca, N, ld)
        ctx.mark_non_differentiable(scales)
        return y, scales

    @staticmethod
    def backward(ctx, dy, _dscales):
        G, P, V, cvec, nrm, O = ctx.saved_tensors
        B, h8, w8, CG, Ca, N, ld = ctx.dims
        D, DV = 9 * CG, 16 * Ca
        dev = G.device
        st = L.stream_ptr()
        dy = _c(dy)
        dO = torch.empty((B, N, DV), dtype=H16, device=dev)
        L.call('tcvom_gca_unfold', L.ptr(dy), L.ptr(dO), B, h8, w8, Ca, st)
        # T = softmax_bwd(P, dP) * c_j with dP[i][j] = sum_v dO[i][v] V[j][v]: ONE GEMM whose epilogue applies the softmax
        # backward (the row sums sum_j P dP are <dO_i, O_i>): no fp32 N x N dP matrix, no separate softmax-backward pass
        delta = torch.empty((B, N), dtype=torch.float32, device=dev)
        L.call('tcvom_rowdot_bf16', L.ptr(dO), L.ptr(O), 1, L.ptr(delta), B * N, DV, st)
        T = torch.empty((B, N, ld), dtype=H16, device=dev)
        dV = torch.empty((B, N, DV), dtype=torch.float32, device=dev)
        dWq = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        Mp = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        Gt = torch.empty((B, D, ld), dtype=H16, device=dev)
        L.call('tcvom_transpose_bf16', L.ptr(G), L.ptr(Gt), N, D, D, ld, B, N * D, D * ld, st)
        # dV[j][v]  = sum_i P[i][j] dO[i][v]
        # dWq[i][d] = sum_j T[i][j] G[j][d]            (rows m = d, columns n = queries i, reduce j)
        # M'[j][d]  = sum_i T[i][j] G[i][d]
        if GCA_KMAJOR and ld % 256 == 0 and N >= 256 and DV >= 256 and ((N + 255) // 256) * B >= 24:
            # dV and M' contract the ROW index of P / T: the 256-tile GEMM reads them as they lie in memory (k-major operand through
            # the transposing LDS read), so the softmax-backward epilogue writes T alone -- no P^T / T^T (2 x 400 MB written and
            # read back per 3-frame launch at 1080p)
            L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T), None, None,
                   N, DV, ld, B, st)
            L.call('tcvom_gca_dv', L.ptr(P), L.ptr(dO), L.ptr(dV), N, DV, ld, B, st)
            L.call('tcvom_gca_dq_dk', L.ptr(T), L.ptr(Gt), L.ptr(dWq), L.ptr(Mp), N, D, ld, B, st)
        else:
            # NT GEMMs on transposed copies (Pt = P^T, Tt = T^T), written by the same epilogue when the padded row length is a whole
            # number of 256-wide tiles, else by two transpose passes
            fused_t = ld % 256 == 0
            Pt = torch.empty((B, ld, ld), dtype=H16, device=dev)
            Tt = torch.empty((B, ld, ld), dtype=H16, device=dev)
            L.call('tcvom_gca_dp_softmax_bwd', L.ptr(dO), L.ptr(V), L.ptr(P), L.ptr(delta), L.ptr(cvec), L.ptr(T),
                   L.ptr(Tt) if fused_t else None, L.ptr(Pt) if fused_t else None, N, DV, ld, B, st)
            if not fused_t:
                L.call('tcvom_transpose_bf16', L.ptr(P), L.ptr(Pt), N, ld, ld, ld, B, N * ld, ld * ld, st)
                L.call('tcvom_transpose_bf16', L.ptr(T), L.ptr(Tt), N, ld, ld, ld, B, N * ld, ld * ld, st)
            dOt = torch.empty((B, DV, ld), dtype=H16, device=dev)
            L.call('tcvom_transpose_bf16', L.ptr(dO), L.ptr(dOt), N, DV, DV, ld, B, N * DV, DV * ld, st)
            d4 = dense_desc(N, DV, ld, DV, batch=B, in_bstride=ld * ld, w_bstride=DV * ld, out_bstride=N * DV, out_fp32=True)
            L.call('tcvom_conv_igemm', L.ptr(Pt), L.ptr(dOt), L.ptr(dV), None, None, None, None, C.byref(d4), st)
            del dOt
            # both products share the weight operand Gt and go out as ONE launch (tcvom_gemm_pair)
            d3 = dense_desc(N, D, ld, D, batch=B, in_bstride=N * ld, w_bstride=D * ld, out_bstride=N * D, out_fp32=True)
            L.call('tcvom_gemm_pair', L.ptr(T), L.ptr(Tt), L.ptr(Gt), L.ptr(dWq), L.ptr(Mp), C.byref(d3), ld * ld, st)
            del Pt, Tt
        dalpha = torch.empty((B, h8, w8, Ca), dtype=H16, device=dev)
        L.call('tcvom_gca_value_patches_bwd', L.ptr(dV), L.ptr(dalpha), B, h8, w8, Ca, st)
        dg8 = torch.empty((B, h8, w8, CG), dtype=H16, device=dev)
        L.call('tcvom_gca_patches_bwd', L.ptr(dWq), L.ptr(Mp), L.ptr(G), L.ptr(nrm), L.ptr(dg8), B, h8, w8, CG, st)
        return dg8, dalpha, None


gca_attention = _GcaAttention.apply
