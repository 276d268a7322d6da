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


def convert_sync_batchnorm(module, process_group=None):
    """`nn.SyncBatchNorm.convert_sync_batchnorm` (train_ddp.py:213) for the HIP network: every BatchNorm of
    `module` takes its train-mode statistics over the clips of all ranks of `process_group`.  The modules keep
    their class and state_dict keys; tcvom_amd.ops.conv_bn_act reads the two attributes set here and inserts one
    [2][C] fp64 all-reduce per BatchNorm call in forward and one in backward."""
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.sync = True
            m.sync_group = process_group
    return module


class GradientAverager(object):
    """All-reduce(mean) of the gradients of `params` in flat buckets.  Parameters whose .grad is None on
    this rank (unused in this step: DDP's find_unused_parameters=True case) contribute zeros."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)

    def average(self):
        if not (dist.is_available() and dist.is_initialized()):
            return
        ws = dist.get_world_size()
        if ws < 2:
            return
        works = []
        for bucket in self.buckets:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            works.append((bucket, flat, dist.all_reduce(flat, async_op=True)))
        for bucket, flat, work in works:
            work.wait()
            flat.div_(ws)
            off = 0
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n


def reduce_tensor(inp):
    """utils/utils.py:45-59 without the extra barrier: mean of a scalar over ranks (logging only)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            out = inp.clone()
            dist.all_reduce(out)
            return out / dist.get_world_size()
    return inp
