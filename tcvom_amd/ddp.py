"""Data-parallel gradient synchronisation for one-process-per-GPU training (RCCL over xGMI).

Mirrors what `torch.nn.parallel.DistributedDataParallel` does for the reference (train_ddp.py:275-280):
parameters/buffers are broadcast from rank 0 once, and after every backward the gradients are averaged
over ranks.  Clips (windows) are independent, so this all-reduce is the only data-path collective.
Gradients are reduced as a few large flat buckets (default 64 MB) instead of DDP's 25 MB default: xGMI
is point-to-point (7 links x ~153 GB/s per GPU), large messages amortise the per-collective latency.
`backend="nccl"` is RCCL on ROCm; the same code runs on `gloo` for the CPU tests.
"""
import torch
import torch.distributed as dist


def broadcast_module_state(module, src=0):
    """DDP constructor semantics: every rank starts from rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for group in by_type.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def convert_sync_batchnorm(module, process_group=None, mailbox='auto'):
    """`nn.SyncBatchNorm.convert_sync_batchnorm` (train_ddp.py:271-273) for the HIP network: every BatchNorm of `module` takes
    its train-mode statistics over the clips of all ranks of `process_group`.  The modules keep their class and state_dict
    keys; tcvom_amd.ops.conv_bn_act reads the attributes set here.

    mailbox: 'auto' -- the ranks of one node exchange the [frames][2][C] fp64 sums INSIDE the BatchNorm finalize kernels
    through hipIpc-mapped peer mailboxes (tcvom_amd/mailbox.py; a collective call: every rank must convert); None -- one
    all-reduce of the process group per BatchNorm call, forward and backward (what remains when the ranks span nodes or
    TCVOM_SYNCBN=rccl); or a PeerMailbox (a one-rank loop-back mailbox measures the exchange on one GPU)."""
    mb = None
    if mailbox == 'auto':
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1 and torch.cuda.is_available():
            from .mailbox import mailbox_for
            mb = mailbox_for(process_group)
    else:
        mb = mailbox
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.sync = True
            m.sync_group = process_group
            m.sync_mailbox = mb
    return module


def sync_batchnorm_info(module):
    """(transport, exchanges issued so far) of a converted module: 'mailbox' / 'allreduce' / None; 'pending' when the module
    was converted by torch's own function and no BatchNorm has made a train-mode call yet (ops._sync_group adopts each
    nn.SyncBatchNorm on its first one)."""
    pending = False
    for m in module.modules():
        if getattr(m, 'sync', False):
            mb = getattr(m, 'sync_mailbox', None)
            if mb is not None:
                return ('mailbox', mb.exchanges)
            from .ops import SYNC_ALLREDUCES
            return ('allreduce', SYNC_ALLREDUCES[0])
        pending = pending or isinstance(m, torch.nn.SyncBatchNorm)
    return ('pending', 0) if pending else (None, None)


class GradientAverager(object):
    """All-reduce(mean) of the gradients of `params` (DDP's reduction, train_ddp.py:275-280).  Parameters whose
    .grad is None on this rank (unused in this step: DDP's find_unused_parameters=True case) contribute zeros when another
    rank has a gradient for them and stay None when no rank has (`_fill_locally_unused`).

    The weight bank hands autograd views of ONE flat fp32 buffer (WeightBank.backward) and the BatchNorm arenas do
    the same, so most of the 25.6 M gradient elements already sit in a few contiguous spans: those are all-reduced
    IN PLACE (no pack / unpack copies, no per-parameter kernels).  Whatever is left (biases, stray tensors) is
    packed into flat buckets of `bucket_bytes`.  With RCCL the mean is taken by the collective (ReduceOp.AVG)."""

    MIN_SPAN = 1 << 14           # elements; shorter runs are cheaper packed together than as their own collective

    def __init__(self, params, bucket_bytes=64 << 20, banks=(), force=False):
        """banks: the WeightBanks of the model (`tcvom_amd.weights.banks_of(model)`).  Their backward finishes the flat
        gradient in WeightBank.GRAD_CHUNKS layer ranges and calls back after each: the all-reduce of a finished span is
        started right there, on RCCL's stream, while the weight-gradient kernels of the next range still run -- DDP's
        overlap of bucket reductions with the tail of backward.  `average()` then only waits for those and reduces the
        small remainder (BatchNorm arena, biases).  force: also in a one-rank group (tests)."""
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self.last_plan = None    # (elements reduced in place, number of spans, elements packed, number of buckets)
        self.force = force
        self._early = []         # [(flat view, work, storage ptr, first element, count)] started from the bank hooks
        self._early_ok = True
        # the parameters whose gradients ARE the bank's flat buffer (assigned by autograd after the bank's backward returns;
        # other leaves -- BatchNorm scales, biases -- legitimately have their .grad by then)
        self._bank_params = {id(bank): bank.weight_params() for bank in banks}
        self.early_spans = 0     # diagnostic: spans started before backward returned, last step
        self.globally_unused = 0  # diagnostic: parameters no rank had a gradient for, last step (left None)
        self.host_presence_ms = 0.0          # diagnostic: wall time of the host-side presence all-reduce (_fill_locally_unused), last step
        self.host_presence_ms_total, self.host_presence_calls = 0.0, 0
        self._cpu_group = None
        if self._active():       # (a one-process run keeps the bank's single-launch backward)
            import functools
            for bank in banks:
                bank.grad_span_hook = functools.partial(self._on_span, id(bank))

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force)

    def _on_span(self, bank_id, flat, lo, hi):
        """WeightBank.backward finished flat[lo:hi] (fp32, final values of this step).  Only valid when no parameter of that bank
        still holds an older .grad that autograd would ADD these values to after the all-reduce of the new ones has started
        (zero_grad(set_to_none=True) each step, as bench.py / train_ddp.py do).  Checked at the FIRST span of every bank backward
        (lo == 0): a second backward before average() (gradient accumulation) finds the .grad of the first and then (a) starts
        nothing early and (b) waits for the spans the first backward started and forgets them -- autograd is about to accumulate
        into those buffers, and average() reduces the sums again (an all-reduce(mean) of already averaged values is the identity,
        so the first backward's share stays correct)."""
        if not self._active() or hi <= lo:
            return
        if lo == 0:
            ok = all(p.grad is None for p in self._bank_params.get(bank_id, ()))
            if not ok and self._early:
                for e in self._early:
                    e[1].wait()
                    if dist.get_backend() != 'nccl':           # (gloo sums; RCCL's ReduceOp.AVG already divided)
                        e[0].div_(dist.get_world_size())
                self._early = []
            self._early_ok = ok
        if not self._early_ok:
            return
        avg = dist.get_backend() == 'nccl'
        view = flat[lo:hi]
        work = dist.all_reduce(view, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
        self._early.append((view, work, flat.untyped_storage().data_ptr(), flat.storage_offset() + lo, hi - lo))

    def _host_group(self):
        """A process group whose collectives run on HOST tensors (the default group when it is gloo, else a gloo group created
        collectively on first use): the per-step presence vector travels there without touching the GPU stream."""
        if self._cpu_group is None:
            if dist.get_backend() == 'gloo':
                self._cpu_group = dist.group.WORLD
            else:
                try:
                    self._cpu_group = dist.new_group(backend='gloo')
                except Exception as e:                   # noqa: BLE001 -- (every rank fails or none: same build, same store)
                    import warnings
                    warnings.warn('GradientAverager: no host-side process group (%s); locally unused parameters contribute zeros '
                                  'even when no rank used them' % e)
                    self._cpu_group = False
        return self._cpu_group

    def _fill_locally_unused(self):
        """DDP's find_unused_parameters=True rule (train_ddp.py:275-280): a parameter without a gradient on THIS rank contributes
        zeros when ANY rank has one -- and keeps `.grad = None` when NO rank has one (DDP leaves the gradient of a globally unused
        parameter untouched), so that Adam skips it: its weight decay must not move a frozen backbone (VMN freeze_backbone),
        exactly as in a one-process run.  The presence vector (one int32 per parameter) is summed over the ranks on the HOST (a
        2.4 KB gloo all-reduce: no device synchronisation); its last element carries the SyncBatchNorm mailbox status, so a
        timed-out exchange on one rank raises MailboxTimeout on EVERY rank before the optimizer step."""
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        group = self._host_group()
        self._partial = set()
        if group is False:
            return [self.params[i] for i in missing]
        vec = torch.ones(len(self.params) + 1, dtype=torch.int32)
        vec[missing] = 0
        vec[-1] = 0
        for mb in self._mailboxes():
            vec[-1] += int(mb.status[0] != 0)
        dist.all_reduce(vec, group=group)
        if int(vec[-1]) != 0:
            from .mailbox import MailboxTimeout
            for mb in self._mailboxes():
                mb.check()                                  # the rank that saw the timeout reports the exchange number
            raise MailboxTimeout('SyncBatchNorm mailbox: an exchange timed out on another rank; the statistics of this step are '
                                 'invalid (no optimizer step taken)')
        self.globally_unused = sum(1 for i in missing if int(vec[i]) == 0)
        # parameters only SOME ranks have a gradient for: the ranks that lack one substitute a fresh zero tensor, which is not part of
        # any flat buffer -- such parameters must not shape the in-place spans (every rank has to issue the same collectives)
        ws = dist.get_world_size(group)
        self._partial = {i for i in range(len(self.params)) if 0 < int(vec[i]) < ws}
        return [self.params[i] for i in missing if int(vec[i]) != 0]

    def _mailboxes(self):
        from .mailbox import _MAILBOXES
        return [mb for mb in _MAILBOXES.values() if mb is not None]

    @staticmethod
    def plan(grads, min_span):
        """Split `grads` into (spans, rest): a span is (tensor list, storage, first element, element count) of
        contiguous gradients lying back to back in one storage."""
        keyed = {}
        rest = []
        for g in grads:
            if g.is_contiguous() and g.numel() > 0:
                keyed.setdefault((g.untyped_storage().data_ptr(), g.dtype, g.device), []).append(g)
            else:
                rest.append(g)
        spans = []
        for gs in keyed.values():
            gs.sort(key=lambda t: t.storage_offset())
            run = [gs[0]]
            for g in gs[1:] + [None]:
                if g is not None and g.storage_offset() == run[-1].storage_offset() + run[-1].numel():
                    run.append(g)
                    continue
                n = run[-1].storage_offset() + run[-1].numel() - run[0].storage_offset()
                if n >= min_span:
                    spans.append((run, run[0].untyped_storage(), run[0].storage_offset(), n))
                else:
                    rest.extend(run)
                run = [g]
        return spans, rest

    def average(self, force=False):
        """force: run the collectives even in a one-rank group (tests/test_gpu_ddp.py drives the RCCL code path on one GPU)."""
        early, self._early = self._early, []
        self.early_spans = len(early)
        if not (dist.is_available() and dist.is_initialized()):
            return
        ws = dist.get_world_size()
        if ws < 2 and not (force or self.force):
            return
        avg = dist.get_backend() == 'nccl'
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        # (the host-side presence all-reduce is timed: it is the one blocking host collective of a step -- a 2.4 KB gloo all-reduce whose
        #  cost with 8 ranks is reported as `host_presence_allreduce_ms` by bench.py and the world-8 pre-flight test)
        import time as _time
        t0 = _time.perf_counter()
        fill = self._fill_locally_unused()
        self.host_presence_ms = 1e3 * (_time.perf_counter() - t0)
        self.host_presence_ms_total += self.host_presence_ms
        self.host_presence_calls += 1
        for p in fill:
            p.grad = torch.zeros_like(p)
        # fp16 build: an overflowed backward on ANY rank drops the step on every rank (ops.LossScaler: the guarded Adam reads it)
        from .ops import SCALER
        ovf = None
        if SCALER.enabled and self.params and self.params[0].is_cuda:
            ovf = dist.all_reduce(SCALER.counter(self.params[0].device), op=dist.ReduceOp.MAX, async_op=True)
            SCALER.reduced_over_ranks = True

        def covered(g):          # already being reduced by a span started from the bank hook
            if not early or not g.is_contiguous():
                return False
            sp, off = g.untyped_storage().data_ptr(), g.storage_offset()
            return any(sp == e[2] and e[3] <= off and off + g.numel() <= e[3] + e[4] for e in early)
        partial = getattr(self, '_partial', set())
        grads = [p.grad for i, p in enumerate(self.params) if p.grad is not None and i not in partial]
        if early:
            # (parameters of the bank that do not require a gradient -- a frozen backbone -- leave holes in the spans: reduced
            #  along with the rest, harmless)
            grads = [g for g in grads if not covered(g)]
        spans, rest = self.plan(grads, self.MIN_SPAN)
        rest = rest + [p.grad for i, p in enumerate(self.params) if p.grad is not None and i in partial and not covered(p.grad)]
        works = [(e[0], None, e[1]) for e in early]
        for run, storage, first, n in spans:
            flat = run[0].new_empty(0).set_(storage, first, (n,))
            works.append((flat, None, dist.all_reduce(flat, op=op, async_op=True)))
        buckets, cur, size = [], [], 0
        for g in rest:
            nbytes = g.numel() * g.element_size()
            if cur and (size + nbytes > self.bucket_bytes or g.dtype != cur[0].dtype):
                buckets.append(cur)
                cur, size = [], 0
            cur.append(g)
            size += nbytes
        if cur:
            buckets.append(cur)
        for bucket in buckets:
            flat = torch.cat([g.reshape(-1) for g in bucket])
            works.append((flat, bucket, dist.all_reduce(flat, op=op, async_op=True)))
        if ovf is not None:
            ovf.wait()
        for flat, bucket, work in works:
            work.wait()
            if not avg:
                flat.div_(ws)
            if bucket is not None:
                torch._foreach_copy_(bucket, [t.view_as(g) for t, g in zip(flat.split([g.numel() for g in bucket]), bucket)])
        self.last_plan = (sum(s[3] for s in spans) + sum(e[4] for e in early), len(spans) + len(early),
                          sum(g.numel() for g in rest), len(buckets))


def banks_of(module):
    """The WeightBanks behind a HIP network (each owns one flat gradient buffer): for GradientAverager(banks=...)."""
    from .weights import WeightBank
    seen, out = set(), []
    for m in module.modules():
        for v in vars(m).values():
            if isinstance(v, WeightBank) and id(v) not in seen:
                seen.add(id(v))
                out.append(v)
    return out


def reduce_tensor(inp):
    """utils/utils.py:45-59 without the extra barrier: mean of a scalar over ranks (logging only)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            out = inp.clone()
            dist.all_reduce(out)
            return out / dist.get_world_size()
    return inp
