"""SyncBatchNorm statistic exchange (train_ddp.py:271-273): two ranks holding one clip each must produce what ONE
process holding both clips produces.  Both ranks share cuda:0 (RCCL refuses two ranks on one device; the process group
is gloo and only carries the hipIpc handles).  Two transports, both the code the 8-GPU job runs: 'mailbox' -- the sums
travel through hipIpc-mapped peer mailboxes INSIDE the BatchNorm finalize kernels (tcvom_amd/mailbox.py, the default) --
and 'rccl' -- one [frames][2][C] fp64 all-reduce per BatchNorm call (the fallback across nodes)."""
import os
import socket

import pytest
import torch

from tcvom_amd._lib import ACT_DTYPE as H16      # the 16-bit storage type of the loaded build (bf16 / fp16)
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _block(dev, sync, tag):
    import torch.nn as nn
    from tcvom_amd import ops
    from tcvom_amd.ddp import convert_sync_batchnorm
    from tcvom_amd.synthetic import formula_tensor
    from tcvom_amd.weights import ConvSpec, WeightBank
    Ci, Co = 64, 128
    w = nn.Parameter((formula_tensor('sync.w', (Co, Ci, 3, 3)) * 0.1).to(dev))
    bn = nn.BatchNorm2d(Co).to(dev)
    with torch.no_grad():
        bn.weight.copy_(formula_tensor('sync.gamma', (Co,)) * 0.5 + 1.0)
        bn.bias.copy_(formula_tensor('sync.beta', (Co,)) * 0.2)
    if sync:
        convert_sync_batchnorm(bn)
    bank = WeightBank()
    spec = ConvSpec(tag, w, None, None, None, False, 1, 1, 'frame')
    bank.register(spec)
    return w, bn, bank, ops.ConvCfg(bank, spec, bn=bn, act=ops.ACT_RELU)


def _run(cfg, bank, bn, w, x, dz):
    from tcvom_amd import ops
    from tcvom_amd.weights import bank_token
    token = bank_token(bank, 1, True)
    x = x.clone().requires_grad_(True)
    z = ops.conv_bn_act(cfg, x, token, True)
    bank.flush_bn_counters()
    z.backward(dz)
    torch.cuda.synchronize()
    return (z.detach().float(), x.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(),
            bn.running_mean.clone(), bn.running_var.clone())


def _worker(rank, world, port, out, transport):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TCVOM_SYNCBN=transport, TCVOM_MBOX_TIMEOUT_S='20')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        from tcvom_amd.synthetic import formula_tensor
        xs = (formula_tensor('sync.x', (2, 24, 40, 64)) * 2).to(dev)
        xs[1] = xs[1] * 1.7 + 0.3                      # the two clips have different statistics
        xs = xs.to(H16)
        dzs = formula_tensor('sync.dz', (2, 24, 40, 128)).to(dev).to(H16)
        w, bn, bank, cfg = _block(dev, True, 'sync')
        assert (bn.sync_mailbox is not None) == (transport == 'mailbox')
        got = _run(cfg, bank, bn, w, xs[rank:rank + 1].contiguous(), dzs[rank:rank + 1].contiguous())
        if bn.sync_mailbox is not None:
            assert bn.sync_mailbox.exchanges == 2 and bn.sync_mailbox.world == 2      # one exchange forward, one backward
            bn.sync_mailbox.check()
        dg = got[2].clone(); db = got[3].clone()
        dist.all_reduce(dg); dist.all_reduce(db)
        if rank == 0:
            w2, bn2, bank2, cfg2 = _block(dev, False, 'nosync')
            ref = _run(cfg2, bank2, bn2, w2, xs, dzs)
            res = dict(z=(got[0] - ref[0][0:1]).abs().max().item(), zmax=ref[0].abs().max().item(),
                       dx=(got[1] - ref[1][0:1]).abs().max().item(), dxmax=ref[1].abs().max().item(),
                       dg=(dg - ref[2]).abs().max().item(), dgmax=ref[2].abs().max().item(),
                       db=(db - ref[3]).abs().max().item(), dbmax=ref[3].abs().max().item(),
                       rm=(got[4] - ref[4]).abs().max().item(), rv=(got[5] - ref[5]).abs().max().item())
            torch.save(res, out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('transport', ['mailbox', 'rccl'])
def test_sync_batchnorm_two_ranks_match_one_process_batch(tmp_path, transport):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_worker, args=(2, _free_port(), out, transport), nprocs=2, join=True)
    r = torch.load(out)
    print(transport, r)
    # the fp64 sums of the two ranks add up to the one-process sums to the last fp64 bits: the fp32 (mean, invstd) and with them
    # the bf16 outputs and input gradients of the two-rank run EQUAL the batch-of-2 run
    assert r['z'] == 0.0 and r['dx'] <= 1e-2 * r['dxmax'], r
    assert r['dg'] <= 2e-3 * r['dgmax'] and r['db'] <= 2e-3 * r['dbmax'], r
    assert r['rm'] <= 1e-5 and r['rv'] <= 1e-5, r


def _window_worker(rank, world, port, out, transport):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TCVOM_SYNCBN=transport, TCVOM_MBOX_TIMEOUT_S='20')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        from tcvom_amd.ddp import convert_sync_batchnorm
        from tcvom_amd.facade import FullModel_VMD
        from tcvom_amd.synthetic import formula_tensor, synthetic_window

        def make(sync):
            m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
            m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
            m = m.to(dev).train()
            return convert_sync_batchnorm(m) if sync else m
        clips = [[t.to(dev) for t in synthetic_window(1, 3, 128, 160, seed=20 + i)] for i in range(2)]
        m = make(True)
        assert m.NET.batched_frames                       # SyncBatchNorm keeps the frame-batched launches: the statistics of
        a, fg, bg = clips[rank]                            # the 3 frames of a layer travel in ONE [3][2][C] all-reduce
        outs = m(a, fg, bg)
        alpha = outs[7].detach().float().clone()
        from tcvom_amd.facade import train_step_loss
        train_step_loss(outs).backward()                   # the backward collectives pair up on the two ranks as well
        gnorm = torch.stack([p.grad.float().norm() for p in m.parameters() if p.grad is not None])
        assert bool(torch.isfinite(gnorm).all()) and float(gnorm.max()) > 0
        from tcvom_amd.ddp import sync_batchnorm_info
        kind, n = sync_batchnorm_info(m)
        assert kind == ('mailbox' if transport == 'mailbox' else 'allreduce') and n > 100, (kind, n)
        if kind == 'mailbox':
            next(b for b in m.modules() if getattr(b, 'sync', False)).sync_mailbox.check()
        if rank == 0:
            m2 = make(False)
            a2, fg2, bg2 = [torch.cat([clips[0][k], clips[1][k]], 0) for k in range(3)]
            outs2 = m2(a2, fg2, bg2)
            ref = outs2[7].detach().float()[0:1]
            torch.save(dict(mse=((alpha - ref) ** 2).mean().item()), out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('transport', ['mailbox', 'rccl'])
def test_sync_batchnorm_window_forward(tmp_path, transport):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_window_worker, args=(2, _free_port(), out, transport), nprocs=2, join=True)
    r = torch.load(out)
    print(r)
    assert r['mse'] <= 1e-4, r


def test_sync_batchnorm_loopback_mailbox_equals_local_statistics():
    """A one-rank mailbox (what `bench.py --gpus 1 --sync-bn` measures): the sums make the round trip through the uncached
    mailbox memory inside the finalize kernels and come back unchanged -- outputs and gradients equal the plain BatchNorm's bit
    for bit, over several calls (ring slots are reused from the 5th exchange on)."""
    from tcvom_amd.ddp import convert_sync_batchnorm
    from tcvom_amd.mailbox import PeerMailbox
    from tcvom_amd.synthetic import formula_tensor
    dev = torch.device('cuda:0')
    xs = (formula_tensor('sync.x', (2, 24, 40, 64)) * 2).to(dev).to(H16)
    dzs = formula_tensor('sync.dz', (2, 24, 40, 128)).to(dev).to(H16)
    w, bn, bank, cfg = _block(dev, False, 'loop.sync')
    mb = PeerMailbox(loopback=True, timeout_s=5)
    convert_sync_batchnorm(bn, mailbox=mb)
    w2, bn2, bank2, cfg2 = _block(dev, False, 'loop.local')
    for it in range(6):
        got = _run(cfg, bank, bn, w, xs, dzs)
        ref = _run(cfg2, bank2, bn2, w2, xs, dzs)
        for g, r in zip(got, ref):
            assert torch.equal(g, r), it
        bn.weight.grad = bn.bias.grad = bn2.weight.grad = bn2.bias.grad = None
    assert mb.exchanges == 12
    mb.check()
    mb.close()


def test_sync_batchnorm_mailbox_timeout_is_reported_not_hung():
    """A peer that never arrives: the polling kernel gives up after the timeout, stores the exchange number into the pinned
    status word and terminates; the host raises MailboxTimeout at the next check (no device hang)."""
    import time
    from tcvom_amd.ddp import convert_sync_batchnorm
    from tcvom_amd.mailbox import MailboxTimeout, PeerMailbox
    from tcvom_amd.synthetic import formula_tensor
    dev = torch.device('cuda:0')
    xs = (formula_tensor('sync.x', (1, 24, 40, 64)) * 2).to(dev).to(H16)
    dzs = formula_tensor('sync.dz', (1, 24, 40, 128)).to(dev).to(H16)
    w, bn, bank, cfg = _block(dev, False, 'timeout')
    # pretend to be rank 0 of 2 whose peer is silent: both table entries point at the own mailbox (sized for one sender: the
    # second sender region is ring slot memory nobody writes with this tag)
    mb2 = PeerMailbox(loopback=True, timeout_s=0.25, capacity=4096)
    mb2._sync.world = 2
    mb2.world = 2
    mb2._sync.capacity = 1024                       # [ring 4][2 senders][1024] fits the 4 x 1 x 4096 allocation
    mb2.capacity = 1024
    mb2.table = torch.tensor([mb2._ptr.value, mb2._ptr.value], dtype=torch.int64, device=dev)
    mb2._sync.peers = mb2.table.data_ptr()
    convert_sync_batchnorm(bn, mailbox=mb2)
    bn.sync_group = None
    import tcvom_amd.ops as ops
    real = ops._sync_group
    ops._sync_group = lambda b: ops._Sync((None, 2, b.sync_mailbox)) if getattr(b, 'sync', False) else None
    try:
        t0 = time.time()
        got = _run(cfg, bank, bn, w, xs, dzs)
        assert time.time() - t0 < 10.0
        # the statistics of the timed-out exchange are poisoned, not computed from partial sums: NaN running statistics and
        # BatchNorm parameter gradients (the 16-bit activations saturate instead: common.h h16_clamp)
        assert bool(torch.isnan(got[4]).any()) and bool(torch.isnan(got[2]).any())
        with pytest.raises(MailboxTimeout):
            mb2.check()
    finally:
        ops._sync_group = real
        mb2.close()



def _stock_worker(rank, world, port, out, transport):
    """The reference's own lines (train_ddp.py:271-280: torch's SyncBatchNorm conversion, .to(device), DistributedDataParallel with
    find_unused_parameters=True) on the product model against the tcvom_amd.ddp path (convert_sync_batchnorm +
    broadcast_module_state + GradientAverager): same clips, same weights -> same alphas, gradients and running statistics."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TCVOM_SYNCBN=transport, TCVOM_MBOX_TIMEOUT_S='20')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        from tcvom_amd.ddp import GradientAverager, banks_of, broadcast_module_state, convert_sync_batchnorm, sync_batchnorm_info
        from tcvom_amd.facade import FullModel_VMD, train_step_loss
        from tcvom_amd.synthetic import formula_tensor, synthetic_window

        damp = os.environ.get('TCVOM_TEST_DAMP') == '1'

        def fresh():
            m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
            sd = {k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()}
            if damp:                                    # residual gains x 0.15: a well-conditioned backward map (test_gpu_window.py:
                for k in sd:                            # test_gradient_noise_on_a_conditioned_network) -> tight bounds below
                    if k.endswith('bn2.weight'):
                        sd[k] = sd[k] * 0.15
            m.NET.load_state_dict(sd)
            return m
        a, fg, bg = [t.to(dev) for t in synthetic_window(1, 3, 128, 160, seed=20 + rank)]

        def step(model, net, average):
            outs = model(a, fg, bg)
            train_step_loss(outs).backward()
            average()
            torch.cuda.synchronize()
            grads = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
            stats = {k: b.detach().float().clone() for k, b in net.named_buffers() if 'running' in k}
            return outs[7].detach().float().clone(), grads, stats

        # (A) the reference's lines, verbatim
        model = fresh()
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = model.to(dev)
        model = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True, device_ids=[0], output_device=0)
        model.train()
        assert sync_batchnorm_info(model)[0] == 'pending'
        alpha_a, grads_a, stats_a = step(model, model.module.NET, lambda: None)
        kind, n = sync_batchnorm_info(model)
        assert kind == ('mailbox' if transport == 'mailbox' else 'allreduce') and n > 100, (kind, n)
        bank = banks_of(model)[0]
        assert all(isinstance(b, torch.nn.SyncBatchNorm) for b in bank.bns)
        # (B) this repo's counterparts of the same lines, twice: the forward is not bit-reproducible from run to run (fp32 atomics
        # in the statistics / K-split sums, amplified by the formula-initialised network), so (A) is held to the distance between
        # two runs of (B)
        def ours():
            m2 = fresh().to(dev).train()
            convert_sync_batchnorm(m2)
            broadcast_module_state(m2)
            av = GradientAverager([p for p in m2.parameters() if p.requires_grad], banks=banks_of(m2))
            return step(m2, m2.NET, av.average)
        alpha_b, grads_b, stats_b = ours()
        alpha_c, grads_c, stats_c = ours()
        assert set(grads_a) == set(grads_b) and len(grads_a) > 200
        flat = lambda g: torch.cat([g[k].reshape(-1) for k in sorted(g)])
        fa, fb, fc = flat(grads_a), flat(grads_b), flat(grads_c)
        cos = lambda x, y: float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-30))
        # the gradients DDP left on the two ranks are the same (it averaged them)
        other = fa.clone()
        dist.broadcast(other, 0)
        torch.save(dict(mse_ab=float(((alpha_a - alpha_b) ** 2).mean()), mse_bc=float(((alpha_b - alpha_c) ** 2).mean()),
                        cos_ab=cos(fa, fb), cos_bc=cos(fb, fc), norm_ab=float(fa.norm() / fb.norm()), norm_bc=float(fc.norm() / fb.norm()),
                        serr_ab=max(float((stats_a[k] - stats_b[k]).abs().max()) for k in stats_a),
                        serr_bc=max(float((stats_c[k] - stats_b[k]).abs().max()) for k in stats_a),
                        ranks_agree=float((fa - other).abs().max())), out + str(rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('transport', ['mailbox', 'rccl'])
def test_reference_ddp_lines_equal_tcvom_ddp_path(tmp_path, transport):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_stock_worker, args=(2, _free_port(), out, transport), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(out + str(rank))
        print(transport, rank, r)
        # floors at the observed run-to-run scale of the SAME path (atomics order -> ReLU mask flips) on this noise-amplifying network
        # (the tight, absolute bounds live in the conditioned-network test below).  fp16 build: alpha mse 3-6e-7, gradient-norm ratio
        # 1 +- 0.2 .. 0.7 %, cosine 0.976 .. 0.985.  bf16 build, 16 runs of each transport on one box (round 5): alpha mse 1.1-2.3e-5 for
        # BOTH distances, norm ratio 1 +- 2.2 % (a-b) / 1.6 % (b-c), cosine 0.856 .. 0.902, statistics error 1.8-3.6e-3 -- the fp16 floors
        # sat at ~1 sigma of that spread (one failure in ~20 runs of the suite).  The reference distance (b vs c) is itself a sample: a
        # lucky near-zero one must not turn the bound into a tighter one than the path can meet against itself
        from helpers import tol
        assert r['mse_ab'] <= max(3 * r['mse_bc'], tol(5e-5, 2e-6)) and r['mse_ab'] <= 1e-4, r        # alphas: within the run-to-run distance
        assert r['cos_ab'] >= min(0.97, 1 - 3 * (1 - r['cos_bc'])) - 1e-4, r               # whole-network gradient direction
        assert abs(r['norm_ab'] - 1) <= max(3 * abs(r['norm_bc'] - 1), tol(5e-2, 1.5e-2)), r
        assert r['serr_ab'] <= max(3 * r['serr_bc'], tol(8e-3, 1e-5)), r                   # BatchNorm running statistics
        assert r['ranks_agree'] == 0.0, r


def test_reference_ddp_lines_equal_tcvom_ddp_path_tight_on_a_conditioned_network(tmp_path, monkeypatch):
    """The same comparison on a network whose backward map does not amplify last-bit noise (BasicBlock bn2 gains x 0.15): there the
    bounds can be absolute and tight -- a 3 % direction error or a 2x error in one tensor's gradient fails."""
    monkeypatch.setenv('TCVOM_TEST_DAMP', '1')
    out = str(tmp_path / 'res.pt')
    mp.spawn(_stock_worker, args=(2, _free_port(), out, 'mailbox'), nprocs=2, join=True)
    from helpers import tol
    for rank in range(2):
        r = torch.load(out + str(rank))
        print('damped', rank, r)
        assert r['mse_ab'] <= 1e-5, r
        assert r['cos_ab'] >= tol(0.995, 0.999) and r['cos_bc'] >= tol(0.995, 0.999), r         # measured 0.9985 (bf16) / 0.9998 (fp16)
        assert abs(r['norm_ab'] - 1) <= tol(2e-2, 5e-3), r
        assert r['ranks_agree'] == 0.0, r


def _frozen_worker(rank, world, port, out):
    """with world_size > 1 the averager used to fill the gradients of a frozen backbone with zeros and Adam's weight
    decay moved the "frozen" encoder in multi-GPU runs only."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TCVOM_MBOX_TIMEOUT_S='20')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        from tcvom_amd.ddp import GradientAverager, banks_of, broadcast_module_state, convert_sync_batchnorm
        from tcvom_amd.facade import FullModel_VMD, train_step_loss
        from tcvom_amd.optim import FusedAdam
        from tcvom_amd.synthetic import formula_tensor, synthetic_window
        m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12, freeze_backbone=True)
        m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
        m = m.to(dev).train()
        convert_sync_batchnorm(m)
        broadcast_module_state(m)
        params = [p for p in m.parameters() if p.requires_grad]
        av = GradientAverager(params, banks=banks_of(m))
        opt = FusedAdam(params, lr=1e-3, weight_decay=1e-2)
        before = {k: p.detach().clone() for k, p in m.NET.named_parameters()}
        a, fg, bg = [t.to(dev) for t in synthetic_window(1, 3, 128, 160, seed=30 + rank)]
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            train_step_loss(m(a, fg, bg)).backward()
            av.average()
            opt.step()
        torch.cuda.synchronize()
        moved = [k for k, p in m.NET.named_parameters() if not torch.equal(p.detach(), before[k])]
        frozen_moved = [k for k in moved if k.startswith('encoder.')]
        torch.save(dict(moved=len(moved), frozen_moved=frozen_moved, unused=av.globally_unused), out + str(rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frozen_backbone_stays_frozen_under_two_ranks_with_weight_decay(tmp_path):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_frozen_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(out + str(rank))
        print(rank, r['moved'], r['unused'], r['frozen_moved'][:3])
        assert r['frozen_moved'] == [] and r['moved'] > 0 and r['unused'] > 100, r


# ------------------------------------------------------------------------------------------------ world 8 (BASELINE config 4's rank count)
def _world8_worker(rank, world, port, out):
    """Config 4's process layout on ONE GPU: `world` ranks share cuda:0 (gloo carries the hipIpc handles and the host collectives; RCCL
    refuses several ranks per device), one 3 x 128 x 160 clip each, SyncBatchNorm through the peer mailboxes, forward + backward +
    GradientAverager + FusedAdam for 3 steps (train_ddp.py:271-280,292-297).  Damped residual gains (bn2 x 0.15) so that the comparison
    with the one-process batch of `world` clips can be tight."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TCVOM_SYNCBN='mailbox', TCVOM_MBOX_TIMEOUT_S='60')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        from tcvom_amd.ddp import GradientAverager, banks_of, broadcast_module_state, convert_sync_batchnorm, sync_batchnorm_info
        from tcvom_amd.facade import FullModel_VMD, train_step_loss
        from tcvom_amd.optim import FusedAdam
        from tcvom_amd.synthetic import formula_tensor, synthetic_window

        def fresh():
            m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
            sd = {k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()}
            for k in sd:
                if k.endswith('bn2.weight'):
                    sd[k] = sd[k] * 0.15
            m.NET.load_state_dict(sd)
            return m.to(dev).train()
        clips = [synthetic_window(1, 3, 128, 160, seed=40 + i) for i in range(world)]
        a, fg, bg = [t.to(dev) for t in clips[rank]]
        m = convert_sync_batchnorm(fresh())
        broadcast_module_state(m)
        params = [p for p in m.parameters() if p.requires_grad]
        av = GradientAverager(params, banks=banks_of(m))
        opt = FusedAdam(params, lr=1e-4, weight_decay=1e-4)
        mb = next(b for b in m.modules() if getattr(b, 'sync', False)).sync_mailbox
        assert mb is not None and mb.world == world
        per_step, presence, first = [], [], None
        for it in range(3):
            n0 = mb.exchanges
            outs = m(a, fg, bg)
            m.zero_grad(set_to_none=True)
            train_step_loss(outs).backward()
            av.average()
            presence.append(av.host_presence_ms)
            if it == 0:
                torch.cuda.synchronize()
                first = (outs[7].detach().float().cpu().clone(),
                         {k: p.grad.detach().float().cpu().clone() for k, p in m.NET.named_parameters() if p.grad is not None},
                         {k: b.detach().float().cpu().clone() for k, b in m.NET.named_buffers() if 'running' in k})
            opt.step()
            torch.cuda.synchronize()
            mb.check()
            per_step.append(mb.exchanges - n0)
        kind, n = sync_batchnorm_info(m)
        assert kind == 'mailbox' and n == sum(per_step)
        # every ring slot was reused many times (the ring holds mb.ring exchanges; a step issues > 100)
        assert per_step[0] > 100 and per_step[0] == per_step[1] == per_step[2] and sum(per_step) > 3 * mb.ring
        # all ranks end the three steps with the same weights (same averaged gradients, same Adam)
        wsum = torch.stack([p.detach().double().sum() for p in params]).cpu()
        ws = [None] * world
        dist.all_gather_object(ws, wsum)
        # the host-side presence all-reduce by itself (GradientAverager._fill_locally_unused: one int32 per parameter over the averager's
        # gloo group), all ranks arriving together: inside the steps above its wall time is the ARRIVAL SKEW of eight ranks time-slicing
        # one GPU (their launch queues fill up while the in-kernel mailbox waits spin), not the cost of the collective
        import time
        hg = av._host_group()
        vec = torch.ones(len(params) + 1, dtype=torch.int32)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(vec, group=hg if hg else None)
        pure_ms = 1e3 * (time.perf_counter() - t0) / 20
        res = dict(rank=rank, exchanges_per_step=per_step[0], ring=mb.ring, presence_ms=presence, presence_pure_ms=pure_ms,
                   weights_agree=float(max((w - ws[0]).abs().max() for w in ws)))
        if rank == 0:
            # ONE process, the `world` clips as one batch, plain BatchNorm: what SyncBatchNorm + gradient averaging must reproduce
            m2 = fresh()
            a2, fg2, bg2 = [torch.cat([c[k] for c in clips], 0).to(dev) for k in range(3)]
            o2 = m2(a2, fg2, bg2)
            train_step_loss(o2).backward()
            torch.cuda.synchronize()
            g2 = {k: p.grad.detach().float().cpu() for k, p in m2.NET.named_parameters() if p.grad is not None}
            s2 = {k: b.detach().float().cpu() for k, b in m2.NET.named_buffers() if 'running' in k}
            keys = sorted(k for k in first[1] if k in g2)
            fa = torch.cat([first[1][k].reshape(-1) for k in keys]).double()
            fb = torch.cat([g2[k].reshape(-1) for k in keys]).double()
            res.update(alpha_mse=float(((first[0] - o2[7].detach().float().cpu()[0:1]) ** 2).mean()),
                       stats_err=max(float((first[2][k] - s2[k]).abs().max()) for k in s2),
                       grad_cos=float((fa * fb).sum() / (fa.norm() * fb.norm())), grad_norm_ratio=float(fa.norm() / fb.norm()),
                       n_grads=len(keys))
        torch.save(res, out + str(rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world8_preflight_on_one_gpu(tmp_path):
    """BASELINE config 4 has 8 ranks; every other distributed test here has 2.  Eight ranks on one GPU: the mailbox lanes sl < world, the ring
    reuse with 8 senders (norm.hip: slot_off = seq % ring * world * cap2) and the host-side presence all-reduce with 8 participants all run;
    SyncBatchNorm statistics, alphas and the AVERAGED gradient are compared with the one-process batch of 8 clips."""
    from helpers import tol
    world = 8
    out = str(tmp_path / 'res.pt')
    mp.spawn(_world8_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rs = [torch.load(out + str(r)) for r in range(world)]
    r0 = rs[0]
    print('world 8 on one GPU:', {k: v for k, v in r0.items() if k != 'presence_ms'})
    print('host presence all-reduce inside the steps (= arrival skew of 8 ranks sharing one GPU), ms per step, per rank:',
          [[round(x, 1) for x in r['presence_ms']] for r in rs])
    print('host presence all-reduce by itself (8 ranks, gloo, %d int32, aligned arrival): %.3f ms (max over ranks)'
          % (229, max(r['presence_pure_ms'] for r in rs)))
    assert all(r['weights_agree'] == 0.0 for r in rs), [r['weights_agree'] for r in rs]
    assert all(r['exchanges_per_step'] == r0['exchanges_per_step'] for r in rs)
    # SyncBatchNorm over 8 ranks == BatchNorm over the batch of 8: running statistics to fp32 rounding, alphas to the storage type
    assert r0['stats_err'] <= tol(2e-3, 1e-4), r0
    assert r0['alpha_mse'] <= tol(1e-5, 1e-6), r0
    assert r0['n_grads'] > 200 and r0['grad_cos'] >= tol(0.99, 0.998) and abs(r0['grad_norm_ratio'] - 1) <= tol(3e-2, 1e-2), r0
    # the one blocking host collective of a step, by itself: fail only if it is pathological
    assert max(r['presence_pure_ms'] for r in rs) < 50.0
