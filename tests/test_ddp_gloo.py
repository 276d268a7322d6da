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
    # a parameter NO rank has a gradient for (frozen backbone: VMN freeze_backbone) keeps .grad = None, as DDP leaves globally
    # unused parameters untouched -- Adam's weight decay must not move it; one that only THIS rank lacks contributes zeros
    fz = [nn.Parameter(torch.full((4,), 3.0)), nn.Parameter(torch.full((4,), 5.0)), nn.Parameter(torch.full((2,), 1.0))]
    fz[1].grad = torch.full((4,), float(rank + 1))
    if rank == 0:
        fz[2].grad = torch.full((2,), 4.0)
    av2 = GradientAverager(fz)
    av2.average()
    opt = torch.optim.Adam(fz, lr=0.1, weight_decay=0.5)
    opt.step()
    frozen = (fz[0].grad is None, av2.globally_unused, fz[0].detach().clone(), fz[1].grad.clone(), fz[2].grad.clone())
    q.put(_plain((rank, flat_state, grads, float(loss), arena.clone(), av.last_plan, [p.grad.data_ptr() - arena.data_ptr() for p in ps], frozen)))
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
    (_, s0, g0, l0, a0, plan0, offs0, fz0), (_, s1, g1, l1, a1, plan1, offs1, fz1) = res
    for none, n_unused, w, g1_, g2_ in (fz0, fz1):
        assert none and n_unused == 1                          # left without a gradient on both ranks ...
        assert torch.equal(w, torch.full((4,), 3.0))           # ... so the optimizer (weight decay 0.5) did not move it
        assert torch.equal(g1_, torch.full((4,), 1.5)) and torch.equal(g2_, torch.full((2,), 2.0))
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


def _stock_worker(rank, world, port, q):
    """The reference's own lines (train_ddp.py:271-280) on the product model, host side: torch's SyncBatchNorm conversion
    REPLACES the BatchNorm modules of the tree; the HIP ops must follow (WeightBank.adopt_norm_modules) and treat an
    nn.SyncBatchNorm as converted (ops._sync_group); DistributedDataParallel(find_unused_parameters=True) must construct."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from models.model import FullModel_VMD
    from tcvom_amd import ops
    from tcvom_amd.ddp import banks_of, sync_batchnorm_info
    torch.manual_seed(rank)
    model = FullModel_VMD('vmn_gca', agg_window=7)
    bank = banks_of(model)[0]
    before = list(bank.bns)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)                  # train_ddp.py:273
    # (DDP refuses nn.SyncBatchNorm inside a CPU module -- the converted model under DDP runs in tests/test_gpu_syncbn.py; here DDP
    #  wraps the un-converted product model, as train_ddp.py:271-280 does for FBA)
    torch.manual_seed(rank)
    plain = torch.nn.parallel.DistributedDataParallel(FullModel_VMD('vmn_gca', agg_window=7), find_unused_parameters=True)
    net = model.NET
    info0 = sync_batchnorm_info(model)
    bank.adopt_norm_modules(net)                                                   # what every window does first
    tree = {id(m) for m in net.modules()}
    adopted = all(isinstance(b, nn.SyncBatchNorm) and id(b) in tree for b in bank.bns)
    replaced = all(a is not b and a.weight is b.weight and a.running_mean is b.running_mean for a, b in zip(before, bank.bns))
    cfg = net.encoder._cfgs['conv1'] if hasattr(net.encoder, '_cfgs') else None
    sync = ops._sync_group(bank.bns[0])
    model.eval()
    eval_follows = not bank.bns[0].training and not bank.bns[-1].training
    model.train()
    w0 = torch.cat([p.detach().reshape(-1)[:4] for p in list(plain.module.NET.parameters())[:8]])      # DDP broadcast rank 0's weights
    q.put(_plain((rank, info0[0], adopted, replaced, len(bank.bns), sync is not None and sync.world == 2 and sync.mailbox is None,
                  eval_follows, w0, sync_batchnorm_info(model)[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_reference_ddp_lines_on_product_model_host_side():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stock_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=300)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, info0, adopted, replaced, nbn, sync_ok, eval_follows, w0, info1 in res:
        assert info0 == 'pending' and info1 == 'allreduce'       # adopted on the first _sync_group call (CPU weights: no mailbox)
        assert adopted and replaced and nbn == 72
        assert sync_ok and eval_follows
    assert torch.equal(res[0][7], res[1][7])


# ------------------------------------------------------------------------------------------------ world 8 (BASELINE config 4's rank count)
def _world8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tcvom_amd.ddp import GradientAverager
    # gradients as views of one flat buffer (the bank's layout) + stray tensors; parameter 1 has a gradient on the even ranks only,
    # parameter 4 on rank 5 only, parameter 5 on no rank
    ps = [nn.Parameter(torch.zeros(n)) for n in (6, 10, 4, 3, 2, 5)]
    arena = torch.arange(40, dtype=torch.float32) * (rank + 1)
    ps[0].grad = arena[2:8]
    if rank % 2 == 0:
        ps[1].grad = arena[8:18]
    ps[2].grad, ps[3].grad = arena[18:22], arena[30:33]
    if rank == 5:
        ps[4].grad = torch.full((2,), 16.0)
    av = GradientAverager(ps)
    av.MIN_SPAN = 4
    for _ in range(3):                                       # (the presence vector is exchanged every step)
        av.average()
        first = [None if p.grad is None else p.grad.clone() for p in ps] if _ == 0 else first
        for p, g in zip(ps, (arena[2:8], arena[8:18] if rank % 2 == 0 else None, arena[18:22], arena[30:33],
                             torch.full((2,), 16.0) if rank == 5 else None, None)):
            p.grad = g
        arena.copy_(torch.arange(40, dtype=torch.float32) * (rank + 1))
    q.put(_plain((rank, first, av.globally_unused, av.host_presence_calls, av.host_presence_ms_total)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gradient_average_and_presence_vector():
    """GradientAverager with 8 ranks (every other multi-process test has 2): the mean over 8, parameters that only SOME ranks have a
    gradient for (zeros from the others: DDP's find_unused_parameters rule), one that no rank has (stays None), and the host-side
    presence all-reduce with 8 participants, three steps in a row."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=300)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    base = torch.arange(40, dtype=torch.float32)
    mean_scale = sum(r + 1 for r in range(world)) / world                                 # 4.5
    even_scale = sum(r + 1 for r in range(0, world, 2)) / world                           # ranks 0, 2, 4, 6 only: 16 / 8
    for rank, grads, unused, calls, ms in res:
        assert unused == 1 and calls == 3 and ms >= 0.0
        assert grads[5] is None
        assert torch.allclose(grads[0], base[2:8] * mean_scale)
        assert torch.allclose(grads[1], base[8:18] * even_scale)
        assert torch.allclose(grads[2], base[18:22] * mean_scale) and torch.allclose(grads[3], base[30:33] * mean_scale)
        assert torch.allclose(grads[4], torch.full((2,), 2.0))
    print('host presence all-reduce, 8 ranks (gloo, this host): %.2f ms per step' % (sum(r[4] for r in res) / sum(r[3] for r in res)))
