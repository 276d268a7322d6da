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
