"""debug: which of the reference's two lines (SyncBatchNorm conversion, DDP wrapper) changes the result vs the tcvom_amd.ddp path"""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TCVOM_MBOX_TIMEOUT_S='20')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
    from tcvom_amd.ddp import GradientAverager, banks_of, broadcast_module_state, convert_sync_batchnorm
    from tcvom_amd.facade import FullModel_VMD, train_step_loss
    from tcvom_amd.synthetic import formula_tensor, synthetic_window

    def fresh():
        m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
        m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
        return m
    a, fg, bg = [t.to(dev) for t in synthetic_window(1, 3, 128, 160, seed=20 + rank)]

    def fwd(model):
        with torch.no_grad():
            outs = model(a, fg, bg)
        torch.cuda.synchronize()
        return outs[7].detach().float().clone()
    res = {}
    m = fresh().to(dev).train(); convert_sync_batchnorm(m); res['ours'] = fwd(m)
    m = fresh().to(dev).train(); convert_sync_batchnorm(m); res['ours2'] = fwd(m)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(fresh()).to(dev).train(); res['stock_conv_then_to'] = fwd(m)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(fresh().to(dev)).train(); res['to_then_stock_conv'] = fwd(m)
    m = fresh().to(dev).train(); convert_sync_batchnorm(m)
    m = torch.nn.parallel.DistributedDataParallel(m, find_unused_parameters=True, device_ids=[0], output_device=0); res['ours_ddp'] = fwd(m)
    m = fresh().to(dev).train(); res['nosync'] = fwd(m)
    if rank == 0:
        for k, v in res.items():
            print(k, float((v - res['ours']).abs().max()))
    dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
