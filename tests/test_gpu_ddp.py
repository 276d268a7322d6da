"""The N > 1 gradient exchange on the REAL backend: a one-rank RCCL group on the GPU box runs exactly the collectives the
driver's 2 / 4 / 8-GPU bench runs issue (in-place ReduceOp.AVG all-reduce over spans of the flat gradient arena + bucketed
stragglers, SyncBatchNorm statistic all-reduce, parameter broadcast).  The two-rank semantics are covered on CPU over gloo
(tests/test_ddp_gloo.py); this one checks that RCCL accepts the calls."""
import os
import socket

import pytest
import torch
import torch.distributed as dist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_rccl_gradient_average_and_sync_bn_on_one_rank():
    from tcvom_amd.ddp import GradientAverager, broadcast_module_state, reduce_tensor
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(_free_port())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend='nccl', init_method='env://', world_size=1, rank=0, device_id=dev)      # as bench.py / train_ddp.py
    try:
        # gradients laid out like WeightBank.backward's arena: views of one flat buffer (-> in-place span all-reduce) + strays
        arena = torch.arange(3 * 70000, dtype=torch.float32, device=dev) / 1000.0
        params = [torch.nn.Parameter(torch.zeros(70000, device=dev)) for _ in range(3)] + \
                 [torch.nn.Parameter(torch.zeros(17, device=dev)), torch.nn.Parameter(torch.zeros(5, 3, device=dev))]
        for i in range(3):
            params[i].grad = arena[i * 70000:(i + 1) * 70000]
        params[3].grad = torch.full((17,), 2.5, device=dev)
        params[4].grad = None                                      # a parameter without a gradient this step
        want = [None if p.grad is None else p.grad.clone() for p in params]
        av = GradientAverager(params)
        av.average(force=True)
        torch.cuda.synchronize()
        numel_spans, nspans, numel_rest, nbuckets = av.last_plan
        assert nspans >= 1 and numel_spans == 3 * 70000 and nbuckets >= 1
        for p, w in zip(params, want):
            # AVG over one rank = identity; a parameter NO rank has a gradient for keeps .grad = None (DDP leaves globally unused
            # parameters untouched: Adam must not decay a frozen backbone)
            assert (p.grad is None) if w is None else torch.equal(p.grad, w)
        assert av.globally_unused == 1
        assert params[0].grad.data_ptr() == arena.data_ptr()                           # reduced in place
        assert float(reduce_tensor(torch.tensor(3.0, device=dev))) == 3.0
        # the overlapped path on a real network: the bank hands over its flat gradient in >= 3 spans, each all-reduced
        # (RCCL, in place) from inside backward; the result equals the plain backward
        from tcvom_amd.ddp import banks_of
        from tcvom_amd.facade import FullModel_VMD, train_step_loss
        from tcvom_amd.synthetic import formula_tensor, synthetic_window
        a, fg, bg = [t.to(dev) for t in synthetic_window(1, 3, 128, 160, seed=3)]
        model = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
        model.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in model.NET.state_dict().items()})
        model = model.to(dev).train()
        ps = [p for p in model.parameters() if p.requires_grad]
        banks = banks_of(model)
        assert len(banks) == 1
        bank = banks[0]
        av2 = GradientAverager(ps, banks=banks, force=True)
        train_step_loss(model(a, fg, bg)).backward()
        spans = [(e[3], e[4]) for e in av2._early]
        av2.average()
        torch.cuda.synchronize()
        assert av2.early_spans >= 3 and av2.last_plan[1] >= 3, (av2.early_spans, av2.last_plan)
        assert spans[0][0] == 0 and sum(n for _, n in spans) == bank.grad_numel                    # the spans tile the flat gradient
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        assert not bank._deferred                                                                   # every weight gradient was issued
        assert all(bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0 for p in ps if p.grad is not None)
        # the layer-range form of the SpectralNorm backward equals the single-launch form (same dW~ arena; the inner products
        # <dW~, u v^T> are accumulated with fp32 atomics, hence not bit for bit; the
        # whole-network gradients of two passes are not comparable at this size: their atomics reorder and the 4x5-pixel
        # BatchNorms amplify that)
        plan = bank.current_plan
        chunked = torch.cat([g.reshape(-1) for g in bank.backward(plan)])
        bank.grad_span_hook = None
        whole = torch.cat([g.reshape(-1) for g in bank.backward(plan)])
        torch.cuda.synchronize()
        assert float((chunked - whole).abs().max()) <= 1e-5 * float(whole.abs().max())
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4)).to(dev)
        broadcast_module_state(m)
        t = torch.ones(8, device=dev)
        dist.all_reduce(t)
        dist.barrier()
        assert float(t.sum()) == 8.0
    finally:
        dist.destroy_process_group()


def _overflow_worker(rank, world, port, out):
    """FusedAdam under stock DDP semantics (no GradientAverager): a saturated backward on ONE rank must drop the step on BOTH."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)      # both ranks share cuda:0 (RCCL refuses that); gloo carries the MAX
    try:
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        from tcvom_amd import ops
        from tcvom_amd.optim import FusedAdam
        assert ops.SCALER.enabled
        p = torch.nn.Parameter(torch.ones(4096, device=dev))
        opt = FusedAdam([p], lr=0.1)
        res = {}
        for step, sat_rank in enumerate((None, 1, None)):
            p.grad = torch.full_like(p, 0.5)               # identical ("already averaged") gradients on both ranks
            if sat_rank == rank:
                ops.SCALER.counter(dev)[0] = 3             # what bn_bwd_reduce counts when it reads a gradient at +-65504
            before = p.detach().clone()
            opt.step()
            torch.cuda.synchronize()
            res['moved%d' % step] = bool((p.detach() != before).any())
        res['scale'] = ops.SCALER.scale
        res['skipped'] = ops.SCALER.skipped_steps
        res['step'] = int(opt.state[p]['step'])
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        if rank == 0:
            torch.save(gathered, out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_overflow_on_one_rank_drops_the_adam_step_on_every_rank_without_an_averager(tmp_path):
    """with stock DistributedDataParallel nothing but FusedAdam sees the fp16 saturation counter;
    it must take the MAX over the ranks itself, or the replicas drift apart."""
    import torch.multiprocessing as mp
    from tcvom_amd import ops
    if not ops.SCALER.enabled:
        pytest.skip('the overflow guard belongs to the fp16 build (TCVOM_DTYPE=fp16)')
    out = str(tmp_path / 'ovf.pt')
    mp.spawn(_overflow_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out)
    print(r0, r1)
    for r in (r0, r1):
        assert r['moved0'] and not r['moved1'] and r['moved2'], r      # step 1 dropped on BOTH ranks, steps 0 and 2 applied
    assert r0 == r1                                                    # same scale, same skip count, same Adam step counter
