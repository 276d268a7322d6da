#!/usr/bin/env python
"""Training entry point with the reference's call surface (train_ddp.py:40-100,178-349): same config keys,
same model construction, loss composition L_alpha + L_comp + L_grad + 0.5 L_dt + 0.25 L_att, Adam with weight decay,
poly learning-rate schedule, checkpoint of `model.NET.state_dict()` per epoch — on the MI355X HIP path, one process
per GPU, gradients averaged over RCCL.

    python train_ddp.py --cfg cfgs/vmd_vmn_gca_synthetic.yaml TRAIN.TOTAL_STEPS 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_ddp.py --cfg ...

With `DATASET.PATH` set the clips come from a VideoMatting108-style tree through `dataset.VMD.VideoMattingDataset`
(tcvom_amd/data.py: PNG decode in loader workers, crop / resize on the device; train_ddp.py:224-240 of the reference);
with `DATASET.PATH ''` they are the synthetic windows of tcvom_amd/synthetic.py with the same output contract
(fg, bg, a, idx: float32 0..255, BGR, [B,S,C,H,W]; dataset/VMD.py:293-301).
"""
import argparse
import logging
import os
import random
import shutil
import time

import torch
import torch.distributed as dist

from models.model import FullModel_VMD
from tcvom_amd.config import get_cfg_defaults
from tcvom_amd.ddp import GradientAverager, banks_of, broadcast_module_state, convert_sync_batchnorm, reduce_tensor
from tcvom_amd.optim import FusedAdam
from tcvom_amd.synthetic import synthetic_window


def poly_lr(optimizer, base_lr, max_iters, cur_iters, power=0.9):
    """utils/utils.py:185-188."""
    lr = base_lr * ((1 - float(cur_iters) / max_iters) ** power)
    optimizer.param_groups[0]['lr'] = lr
    return lr


def const_lr(optimizer, base_lr, max_iters, cur_iters):
    return base_lr


STR_DICT = {'poly': poly_lr, 'const': const_lr}


class SyntheticClips(object):
    """Stands in for DataLoader(VideoMattingDataset(...)): yields (fg, bg, a, idx) batches."""

    def __init__(self, batch, frames, size, steps, seed):
        self.batch, self.frames, self.size, self.steps, self.seed = batch, frames, size, steps, seed

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            a, fg, bg = synthetic_window(self.batch, self.frames, self.size[0], self.size[1], seed=self.seed + 1000 * i)
            yield fg, bg, a, torch.arange(self.batch)


class DiskClips(object):
    """DataLoader(VideoMattingDataset, sampler=DistributedSampler) of train_ddp.py:232-240: every rank takes its stride of a
    per-epoch permutation (drop_last), worker processes decode the PNGs, the training process finishes the clips on its GPU."""

    def __init__(self, dataset, batch, rank, world, workers, seed):
        self.ds, self.batch, self.rank, self.world, self.workers, self.seed = dataset, batch, rank, world, workers, seed
        self.epoch = 0

    def __len__(self):
        return (len(self.ds) // self.world) // self.batch

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        self.epoch += 1
        perm = torch.randperm(len(self.ds), generator=g).tolist()
        mine = perm[self.rank::self.world][:len(self) * self.batch]
        raws = torch.utils.data.DataLoader(torch.utils.data.Subset(self.ds.raw_view(), mine), batch_size=None, shuffle=False,
                                           num_workers=self.workers)
        buf = []
        for raw in raws:
            buf.append(self.ds.transform(raw))
            if len(buf) == self.batch:
                yield tuple(torch.stack([b[k] for b in buf]) for k in range(4))
                buf = []


def train(epoch, loader, base_lr, total_epochs, optimizer, averager, model, adjust_lr, print_freq, rank, device):
    model.train()
    steps_per_epoch = len(loader)
    tic = time.time()
    for i_iter, (fg, bg, a, _) in enumerate(loader):
        out = model(a.to(device), fg.to(device), bg.to(device))
        L_alpha, L_comp, L_grad, L_dt, L_att = (out[k].mean() for k in range(5))
        loss = L_alpha + L_comp + L_grad + 0.5 * L_dt + 0.25 * L_att          # train_ddp.py:56-61
        model.zero_grad(set_to_none=True)
        loss.backward()
        averager.average()
        optimizer.step()
        reduced = reduce_tensor(loss.detach())
        cur = epoch * steps_per_epoch + i_iter
        lr = adjust_lr(optimizer, base_lr, total_epochs * steps_per_epoch, cur)
        if i_iter % print_freq == 0 and rank == 0:
            logging.info('Iter:[%d/%d], Time: %.2f, lr: %.3e, Loss: %.6f, L_alpha: %.4f L_comp: %.4f L_grad: %.4f '
                         'L_dt: %.4f L_att: %.4f', cur, total_epochs * steps_per_epoch, time.time() - tic, lr,
                         float(reduced), float(L_alpha), float(L_comp), float(L_grad), float(L_dt), float(L_att))
            tic = time.time()


def validate(dataset, model, rank, world, workers):
    """train_ddp.py:102-166: mean of L_alpha + L_comp + L_grad over the 3-frame validation samples plus, on rank 0, the
    temporal indicator mean_unknown |(a_t - a_t+1) - (g_t - g_t+1)| over consecutive centre-frame predictions.  The reference
    round-trips (pred, unknown mask, gt) through 8-bit PNGs in /dev/shm; here the same 8-bit triplets stay in memory."""
    model.eval()
    c = dataset.sample_length // 2
    mine = list(range(len(dataset)))[rank::world]
    raws = torch.utils.data.DataLoader(torch.utils.data.Subset(dataset.raw_view(), mine), batch_size=None, shuffle=False,
                                       num_workers=workers)
    total, count, store = 0.0, 0, {}
    with torch.no_grad():
        for raw in raws:
            fg, bg, a, idx = dataset.transform(raw)
            out = model(a[None], fg[None], bg[None])
            loss = out[0].mean() + out[1].mean() + out[2].mean()
            total += float(loss)
            count += 1
            tri = out[6][0, c, 0].float() * 255
            store[dataset.samples[int(idx)][c]] = (torch.floor(out[7][0, c, 0].float() * 255).to(torch.uint8).cpu(),     # np.uint8(alpha * 255)
                                                   ((tri > 0) & (tri < 255)).cpu(), a[c, 0].to(torch.uint8).cpu())
    model.train()
    if world > 1:                                   # ranks may hold one sample more or less: one reduction of (sum, count)
        acc = torch.tensor([total, float(count)], dtype=torch.float64, device=next(model.parameters()).device)
        dist.all_reduce(acc)
        total, count = float(acc[0]), int(acc[1])
    val_loss = total / max(count, 1)
    if world > 1:
        parts = [None] * world if rank == 0 else None       # only rank 0 evaluates the temporal indicator
        dist.gather_object(store, parts, dst=0)
        store = {k: v for part in parts for k, v in part.items()} if rank == 0 else {}
    if rank == 0:
        logging.info('Validation loss: %.6f', val_loss)
        res = 0.0
        for sample in dataset.samples:
            pa, m, g = store[sample[c]]
            ha, _, hg = store[sample[c + 1]]
            if not bool(m.any()):
                continue
            d = (pa.float() - ha.float()) / 255.0 - (g.float() - hg.float()) / 255.0
            res += float(d[m].abs().mean())
        res /= float(len(dataset.samples))
        logging.info('Average L_dt: %.6f', res)
        val_loss += res
    if world > 1:
        dist.barrier()
    return val_loss


def main(cfg_name, cfg, steps_per_epoch, frames):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group(backend='nccl', init_method='env://', device_id=device)
    if cfg.SYSTEM.RANDOM_SEED > 0:                                 # train_ddp.py:192-197 (the crop search of the loader uses `random`)
        random.seed(cfg.SYSTEM.RANDOM_SEED + rank)
        torch.manual_seed(cfg.SYSTEM.RANDOM_SEED + rank)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format='%(asctime)-15s %(message)s')
    out_dir = os.path.join(cfg.SYSTEM.OUTDIR, cfg_name + cfg.SYSTEM.EXP_SUFFIX)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)

    model = FullModel_VMD(model=cfg.MODEL, agg_window=cfg.AGG_WINDOW)
    if cfg.TRAIN.LOAD_CKPT:
        dct = torch.load(cfg.TRAIN.LOAD_CKPT, map_location='cpu')
        missing, unexpected = model.NET.load_state_dict(dct, strict=False)
        logging.info('Missing keys: %s', sorted(missing))
        logging.info('Unexpected keys: %s', sorted(unexpected))
    model = model.to(device)
    if not cfg.MODEL.endswith('fba'):
        convert_sync_batchnorm(model)                              # train_ddp.py:271-273: SyncBN unless FBA
    broadcast_module_state(model)                                  # DDP constructor semantics (train_ddp.py:275-280)
    params = [p for p in model.parameters() if p.requires_grad]
    logging.info('=> Total Parameters: %d', sum(p.numel() for p in params))
    assert cfg.TRAIN.OPTIMIZER == 'adam', 'only Adam (the optimizer of every reference config) is on the HIP path'
    optimizer = FusedAdam(params, lr=cfg.TRAIN.BASE_LR, weight_decay=cfg.TRAIN.WEIGHT_DECAY)
    start_step = 0
    if cfg.TRAIN.LOAD_OPT:
        # resume (train_ddp.py:300-304): the epoch to continue from is parsed from 'optimizer_<N>.pth.tar', so that the poly
        # learning rate, the epoch shuffle, the checkpoint numbering and the validation delay pick up where the run stopped
        optimizer.load_state_dict(torch.load(cfg.TRAIN.LOAD_OPT, map_location='cpu'))
        start_step = int(os.path.basename(cfg.TRAIN.LOAD_OPT).split('_')[-1][:-8])
    averager = GradientAverager(params, banks=banks_of(model))      # gradient spans all-reduced as the bank finishes them
    adjust_lr = STR_DICT[cfg.TRAIN.LR_STRATEGY]
    test_dataset, best_loss = None, 1e+8
    if cfg.DATASET.PATH:
        from dataset.VMD import VideoMattingDataset
        train_dataset = VideoMattingDataset(data_root=cfg.DATASET.PATH, image_shape=cfg.TRAIN.TRAIN_INPUT_SIZE, mode='train',
                                            use_subset=cfg.DATASET.SUBSET, plus1=cfg.MODEL.startswith('vmn_res'), no_flow=True,
                                            sample_length=frames, device=device)
        loader = DiskClips(train_dataset, cfg.TRAIN.BATCH_SIZE_PER_GPU, rank, world, cfg.SYSTEM.NUM_WORKERS,
                           seed=max(cfg.SYSTEM.RANDOM_SEED, 0))
        if os.path.exists(os.path.join(cfg.DATASET.PATH, 'val_videos_subset.txt' if cfg.DATASET.SUBSET else 'val_videos.txt')):
            test_dataset = VideoMattingDataset(data_root=cfg.DATASET.PATH, image_shape=cfg.TRAIN.VAL_INPUT_SIZE, mode='val',
                                               use_subset=cfg.DATASET.SUBSET, plus1=cfg.MODEL.startswith('vmn_res'), no_flow=True,
                                               sample_length=3, device=device)          # train_ddp.py:242-250
    else:
        loader = SyntheticClips(cfg.TRAIN.BATCH_SIZE_PER_GPU, frames, tuple(cfg.TRAIN.TRAIN_INPUT_SIZE), steps_per_epoch,
                                seed=max(cfg.SYSTEM.RANDOM_SEED, 0) + rank)
    if hasattr(loader, 'epoch'):
        loader.epoch = start_step                                  # DistributedSampler.set_epoch(epoch) (train_ddp.py:317-318)
    for epoch in range(start_step, cfg.TRAIN.TOTAL_STEPS):
        train(epoch, loader, cfg.TRAIN.BASE_LR, cfg.TRAIN.TOTAL_STEPS, optimizer, averager, model, adjust_lr,
              cfg.TRAIN.PRINT_FREQ, rank, device)
        if world > 1:
            dist.barrier()
        val_loss = best_loss
        if test_dataset is not None and epoch >= cfg.TRAIN.VAL_START_EPOCH:
            val_loss = validate(test_dataset, model, rank, world, cfg.SYSTEM.NUM_WORKERS)
        if rank == 0:
            weight_fn = os.path.join(out_dir, 'checkpoint_%d.pth.tar' % (epoch + 1))
            torch.save(model.NET.state_dict(), weight_fn)           # the reference's checkpoint format (train_ddp.py:338)
            torch.save(optimizer.state_dict(), os.path.join(out_dir, 'optimizer_%d.pth.tar' % (epoch + 1)))
            logging.info('=> saved %s', weight_fn)
            if val_loss < best_loss:                                # train_ddp.py:340-343
                best_loss = val_loss
                shutil.copyfile(weight_fn, os.path.join(out_dir, 'best.pth'))
                logging.info('=> new minimum loss. copy to best.pth')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description='Train network')
    ap.add_argument('--cfg', required=True, type=str)
    ap.add_argument('--local_rank', type=int, default=-1)
    ap.add_argument('--steps-per-epoch', type=int, default=4, help='synthetic clips per epoch')
    ap.add_argument('--frames', type=int, default=5, help='sample_length (dataset/VMD.py:28)')
    ap.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cfg = get_cfg_defaults()
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    main(os.path.splitext(os.path.basename(args.cfg))[0], cfg, args.steps_per_epoch, args.frames)
