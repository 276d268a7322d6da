"""The N>1 path on CPU: two gloo ranks broadcast module state from rank 0 and average gradients in flat
buckets exactly like DDP's all-reduce(mean) (train_ddp.py:275-280), including parameters unused on one rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _plain(o):
    """Tensors -> numpy arrays, recursively: a worker's results cross the queue BY VALUE (a tensor would travel as a file descriptor
    that the parent may try to fetch after the worker has exited)."""
    if torch.is_tensor(o):
        return o.detach().cpu().numpy()
    if isinstance(o, (tuple, list)):
        return type(o)(_plain(v) for v in o)
    return o


def _tensors(o):
    import numpy as np
    if isinstance(o, np.ndarray):
        return torch.from_numpy(o)
    if isinstance(o, (tuple, list)):
        return type(o)(_tensors(v) for v in o)
    return o


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tcvom_amd.ddp import GradientAverager, broadcast_module_state, reduce_tensor
    torch.manual_seed(100 + rank)                           # different init per rank on purpose
    net = nn.Sequential(nn.Linear(5, 7), nn.BatchNorm1d(7), nn.Linear(7, 3))
    broadcast_module_state(net)
    params = [p for p in net.parameters()]
    flat_state = torch.cat([p.detach().reshape(-1) for p in params] + [b.detach().float().reshape(-1) for b in net.buffers()])
    for i, p in enumerate(params):
        p.grad = None if (rank == 1 and i == 0) else torch.full_like(p, float(rank + 1) * (i + 1))
    GradientAverager(params, bucket_bytes=64).average()     # tiny buckets -> several collectives
    grads = torch.cat([p.grad.reshape(-1) for p in params])
    loss = reduce_tensor(torch.tensor(float(rank)))
    # gradients that are views of one flat buffer (what WeightBank.backward hands to autograd): reduced in place
    ps = [nn.Parameter(torch.zeros(n)) for n in (6, 10, 4, 3)]
    arena = torch.arange(40, dtype=torch.float32) * (rank + 1)
    ps[0].grad, ps[1].grad, ps[2].grad = arena[2:8], arena[8:18], arena[18:22]      # one run of 20 elements
    ps[3].grad = arena[30:33]                                                         # short run -> packed bucket
    av = GradientAverager(ps)
    av.MIN_SPAN = 8
    av.average()
    q.put(_plain((rank, flat_state, grads, float(loss), arena.clone(), av.last_plan, [p.grad.data_ptr() - arena.data_ptr() for p in ps])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_broadcast():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=120)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, g0, l0, a0, plan0, offs0), (_, s1, g1, l1, a1, plan1, offs1) = res
    base = torch.arange(40, dtype=torch.float32)
    for arena, scale in ((a0, 1.0), (a1, 2.0)):
        want = base * scale                               # outside the gradient views: untouched, rank-specific
        want[2:22] = base[2:22] * 1.5
        want[30:33] = base[30:33] * 1.5
        assert torch.equal(arena, want)
    assert plan0 == plan1 == (20, 1, 3, 1)
    assert offs0 == offs1 == [8, 32, 72, 120]            # .grad still aliases the arena: no copies were made
    assert torch.equal(s0, s1), 'state must equal rank 0 after the broadcast'
    assert torch.equal(g0, g1), 'averaged gradients must agree on all ranks'
    net = nn.Sequential(nn.Linear(5, 7), nn.BatchNorm1d(7), nn.Linear(7, 3))
    expect = []
    for i, p in enumerate(net.parameters()):
        r0, r1 = 1.0 * (i + 1), (0.0 if i == 0 else 2.0 * (i + 1))
        expect.append(torch.full((p.numel(),), (r0 + r1) / 2))
    assert torch.allclose(g0, torch.cat(expect))
    assert l0 == l1 == 0.5


def _overlap_worker(rank, world, port, q):
    """The overlapped path: the bank calls back per finished span of its flat gradient (here a stand-in with the same
    contract as WeightBank.backward), the all-reduce of a span starts inside the callback, average() waits and reduces
    the remainder."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tcvom_amd.ddp import GradientAverager

    class Bank(object):
        grad_span_hook = None

        def weight_params(self):
            return ps[:5]

    bank = Bank()
    sizes = [50000, 30000, 20000, 40000, 10000]                   # five "layers", handed over in 3 spans
    ps = [nn.Parameter(torch.zeros(n)) for n in sizes] + [nn.Parameter(torch.zeros(7))]
    av = GradientAverager(ps, banks=[bank])
    assert bank.grad_span_hook is not None
    results = []
    for step in range(2):
        for p in ps:
            p.grad = None                                          # zero_grad(set_to_none=True)
        flat = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1) + step
        started = []
        for lo, hi in ((0, 80000), (80000, 100000), (100000, 150000)):
            bank.grad_span_hook(flat, lo, hi)                      # "backward" finished this span
            started.append(len(av._early))
        off = 0
        for p, n in zip(ps, sizes):
            p.grad = flat[off:off + n]
            off += n
        ps[5].grad = torch.full((7,), float(rank + 1))
        av.average()
        results.append((started, av.early_spans, av.last_plan, flat.clone(), ps[5].grad.clone(),
                        [p.grad.data_ptr() - flat.data_ptr() for p in ps[:5]]))
    # a parameter that kept its old .grad: autograd would ADD the new values to it after the early all-reduce had started, so no
    # span may start early in such a backward; average() then reduces everything itself
    for p in ps:
        p.grad = None
    ps[1].grad = torch.full((sizes[1],), 7.0)                      # left over from an earlier step (no zero_grad)
    flat = torch.ones(sum(sizes)) * (rank + 1)
    bank.grad_span_hook(flat, 0, 150000)
    stale = len(av._early)
    off = 0
    for p, n in zip(ps, sizes):
        p.grad = (p.grad + flat[off:off + n]) if p.grad is not None else flat[off:off + n]      # what AccumulateGrad does
        off += n
    ps[5].grad = torch.full((7,), float(rank + 1))
    av.average()
    stale = (stale, av.early_spans, float(ps[1].grad[0]), float(ps[0].grad[0]))
    # gradient accumulation: a SECOND backward before average().  The first backward's spans started early; the second finds the
    # first's .grad at its first span, waits for those all-reduces and forgets them, starts nothing early; autograd then
    # accumulates into the (already averaged) buffers and average() reduces the sums: mean over ranks of (g1 + g2)
    for p in ps:
        p.grad = None
    f1 = torch.full((sum(sizes),), 10.0 * (rank + 1))
    for lo, hi in ((0, 80000), (80000, 150000)):
        bank.grad_span_hook(f1, lo, hi)
    early_first = len(av._early)
    off = 0
    for p, n in zip(ps, sizes):
        p.grad = f1[off:off + n]
        off += n
    f2 = torch.full((sum(sizes),), 1.0 * (rank + 1))
    for lo, hi in ((0, 80000), (80000, 150000)):
        bank.grad_span_hook(f2, lo, hi)
    early_second = len(av._early)
    off = 0
    for p, n in zip(ps, sizes):
        p.grad += f2[off:off + n]                                  # AccumulateGrad, in place into the first backward's buffer
        off += n
    ps[5].grad = torch.full((7,), float(rank + 1))
    av.average()
    accum = (early_first, early_second, float(ps[0].grad[0]), float(ps[4].grad[-1]))
    q.put(_plain((rank, results, stale, accum)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_gradient_spans():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=120)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    base = torch.arange(150000, dtype=torch.float32)
    for rank, results, stale, accum in res:
        assert stale == (0, 0, 8.5, 1.5)                           # nothing started early; mean of (7 + 1, 7 + 2) and of (1, 2)
        assert accum == (2, 0, 16.5, 16.5)                         # mean of (10 + 1, 20 + 2): both backwards averaged exactly once
        for step, (started, early, plan, flat, small, offs) in enumerate(results):
            assert started == [1, 2, 3] and early == 3             # one collective per span, started before average()
            assert plan == (150000, 3, 7, 1)                       # 3 early spans in place + one packed bucket
            assert torch.equal(flat, base * 1.5 + step)            # mean of (x + step, 2x + step)
            assert torch.equal(small, torch.full((7,), 1.5))
            assert offs == [0, 200000, 320000, 400000, 560000]     # .grad still aliases the flat buffer


def _sync_expr_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tcvom_amd import ops
    from tcvom_amd.index_net import _sync_batchnorm_expr
    torch.manual_seed(7)
    x_all = torch.randn(3, 4, 5) * 2 + 0.5                       # [frames, clips of both ranks, channels]
    w_all = torch.randn(3, 4, 5)                                  # d(loss)/d(out) of the one-process run
    bn = nn.BatchNorm1d(5)
    x = x_all[:, 2 * rank:2 * rank + 2].clone().requires_grad_(True)
    out, mean, var, n = _sync_batchnorm_expr(x, bn, ops._Sync((None, world, None)))
    (out * w_all[:, 2 * rank:2 * rank + 2]).sum().backward()
    q.put((rank, out.detach().tolist(), x.grad.tolist(), mean.tolist(), var.tolist(), n))      # by value: the worker exits
    dist.barrier()
    dist.destroy_process_group()


def test_tensor_expression_sync_batchnorm_matches_one_process_batch():
    """The two BatchNorms of IndexNet that are tensor expressions (ASPP image pooling over [B, 256] vectors, the 1-channel
    decoder tail) under SyncBatchNorm: statistics AND input gradients of two ranks with 2 clips each equal one process with 4
    clips (train_ddp.py:271-273 converts every BatchNorm; the backward all-reduces the statistic gradients)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_expr_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=120)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(7)
    x_all = (torch.randn(3, 4, 5) * 2 + 0.5).requires_grad_(True)
    w_all = torch.randn(3, 4, 5)
    mean = x_all.mean(1, keepdim=True)
    var = x_all.var(1, unbiased=False, keepdim=True)
    ref = (x_all - mean) / torch.sqrt(var + 1e-5)
    (ref * w_all).sum().backward()
    for rank, out, gx, m, v, n in res:
        sl = slice(2 * rank, 2 * rank + 2)
        out, gx, m, v = (torch.tensor(t) for t in (out, gx, m, v))
        assert n == 4
        assert torch.allclose(out, ref.detach()[:, sl], atol=1e-5)
        assert torch.allclose(gx, x_all.grad[:, sl], atol=1e-5)
        assert torch.allclose(m, mean.detach()[:, 0], atol=1e-6) and torch.allclose(v, var.detach()[:, 0], atol=1e-5)
