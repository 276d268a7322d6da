"""Multi-tensor Adam on one HIP launch (tcvom_adam_mt) — same update rule as
`torch.optim.Adam(params, lr, weight_decay=wd)` used by the reference (train_ddp.py:296-297):
L2 weight decay folded into the gradient, eps 1e-8, betas (0.9, 0.999), bias correction.

Keeps the torch.optim.Optimizer surface (`param_groups[0]['lr']` is what `poly_lr` mutates,
utils/utils.py:185-188; `state_dict()` round-trips exp_avg / exp_avg_sq / step)."""
import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self.table_rebuilds = 0        # diagnostic: how often the pointer table had to be rebuilt
        self._last_step_of = {}        # id(state) -> step number handed to the last launch (taken back when that step was dropped)

    def _table(self, key, plist):
        # the kernel writes through the cached pointers: exp_avg / exp_avg_sq are part of the signature, so that a
        # load_state_dict() (which replaces them) rebuilds the table
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr())
                    for p in plist)
        ent = self._tables.get(key)
        if ent is not None and ent[0] == sig:
            return ent
        # The gradient buffers may move between steps (they are views of a buffer the bank allocates per backward), so a
        # rebuild must not stall the stream: the block list depends on the sizes only and is cached on the device; the
        # pointer rows go up from pinned memory without a host synchronisation.
        rows = [[p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr(),
                 p.numel()] for p in plist]
        dev = plist[0].device
        wkey = ('work', dev, tuple(p.numel() for p in plist))
        went = self._tables.get(wkey)
        if went is None:
            counts = torch.tensor([(p.numel() + 1023) // 1024 for p in plist], dtype=torch.int64)
            idx = torch.repeat_interleave(torch.arange(len(plist), dtype=torch.int64), counts)
            first = torch.cumsum(counts, 0) - counts
            blk = torch.arange(int(counts.sum()), dtype=torch.int64) - first[idx]
            work = torch.stack([idx, blk], 1).to(torch.int32).reshape(-1)
            went = (work.to(dev), int(counts.sum()))
            self._tables[wkey] = went
        self.table_rebuilds += 1
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        ent = (sig, host.to(dev, non_blocking=True), went[0], went[1])
        self._tables[key] = ent
        return ent

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}
        for st in self.state.values():                      # torch stores `step` as a tensor in newer checkpoints
            if torch.is_tensor(st.get('step')):
                st['step'] = int(st['step'].item())

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        # fp16 build: overflow guard + dynamic loss scale (ops.LossScaler): the step is dropped ON THE DEVICE when the backward
        # saturated; the host learns of it one step later and takes the step counters of the dropped step back
        from .ops import SCALER
        guard, dev0 = None, None
        if SCALER.enabled:
            # (the device of the PARAMETERS, with or without a gradient on this rank in this step: whether the ranks meet in the
            #  all-reduce below must not depend on which of them received gradients -- ADVICE round 5: a rank without any would skip
            #  the collective the others block in)
            dev0 = next((p.device for g in self.param_groups for p in g['params'] if p.is_cuda), None)
            if dev0 is not None:
                if SCALER.before_step(dev0):
                    for st in self.state.values():
                        if st and int(st.get('step', 0)) == self._last_step_of.get(id(st), -1):
                            st['step'] = int(st['step']) - 1
                guard = L.ptr(SCALER.counter(dev0))
                # several ranks WITHOUT GradientAverager (stock DistributedDataParallel averages the gradients itself,
                # train_ddp.py:275-280): the counter is rank-local, so a rank whose backward saturated would drop the step alone and
                # the replicas would drift apart -- take the MAX over the ranks here, on the stream, before the guarded launch
                import torch.distributed as dist
                if (not SCALER.reduced_over_ranks and dist.is_available() and dist.is_initialized()
                        and dist.get_world_size() > 1):
                    dist.all_reduce(SCALER.counter(dev0), op=dist.ReduceOp.MAX)
                    SCALER.reduced_over_ranks = True
        self._last_step_of = {}
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            by_step = {}
            for p in plist:
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise RuntimeError('FusedAdam: parameters and gradients must be contiguous fp32 CUDA tensors')
                # the bias correction uses each parameter's OWN step count (torch.optim.Adam): a parameter that receives its
                # first gradient later than the others (find_unused_parameters) runs in its own launch
                st['step'] = int(st['step']) + 1
                self._last_step_of[id(st)] = st['step']
                by_step.setdefault(st['step'], []).append(p)
            b1, b2 = group['betas']
            if len(self._tables) > 64:                      # step groups come and go: no unbounded cache of pointer tables
                self._tables = {k: v for k, v in self._tables.items() if k[0] == 'work'}
            for step, ps in by_step.items():
                # several step groups: keyed by the parameter set (stable from step to step), not by the step count
                key = (gi, False) if len(by_step) == 1 else (gi, tuple(id(p) for p in ps))
                _, table, work, nblk = self._table(key, ps)
                L.call('tcvom_adam_mt_guarded', L.ptr(table), L.ptr(work), nblk, float(group['lr']), float(b1), float(b2),
                       float(group['eps']), float(group['weight_decay']), int(step), float(grad_scale), guard, L.stream_ptr())
        if guard is not None:
            SCALER.after_step(dev0)
        return loss
