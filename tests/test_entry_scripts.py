"""The reference's entry-script call surface: config keys (CPU) and train_ddp.py / pred_vmn.py on the GPU."""
import os

import numpy as np
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_keys_and_overrides():
    from tcvom_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    assert cfg.MODEL == 'vmn50' and cfg.AGG_WINDOW == 9 and cfg.TRAIN.OPTIMIZER == 'adam'      # config.py:3-44 defaults
    cfg.merge_from_file(os.path.join(REPO, 'cfgs', 'vmd_vmn_gca_synthetic.yaml'))
    assert cfg.MODEL == 'vmn_gca' and cfg.AGG_WINDOW == 7 and cfg.TRAIN.LR_STRATEGY == 'poly'
    assert cfg.TRAIN.TRAIN_INPUT_SIZE == (512, 512) and cfg.TRAIN.BASE_LR == 1e-4
    cfg.merge_from_list(['TRAIN.BASE_LR', '2e-4', 'TRAIN.TRAIN_INPUT_SIZE', '(256, 320)', 'SYSTEM.EXP_SUFFIX', '_x'])
    assert cfg.TRAIN.BASE_LR == 2e-4 and cfg.TRAIN.TRAIN_INPUT_SIZE == (256, 320) and cfg.SYSTEM.EXP_SUFFIX == '_x'
    with pytest.raises(KeyError):
        cfg.merge({'NOT_A_KEY': 1})


def test_poly_lr_matches_reference_formula():
    sys.path.insert(0, REPO)
    import train_ddp
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    lr = train_ddp.poly_lr(opt, 1e-4, 100, 25)
    assert abs(lr - 1e-4 * (0.75 ** 0.9)) < 1e-12 and opt.param_groups[0]['lr'] == lr


@pytest.mark.gpu
@pytest.mark.parametrize('arch,nkeys', [('vmn_gca', 584), ('vmn_fba', 203), ('vmn_dim', 113), ('vmn_index', 555)])
def test_train_ddp_two_steps_and_checkpoint(tmp_path, arch, nkeys):
    """train_ddp.py counterpart for every VMN architecture on the HIP path: two optimizer steps on 5-frame synthetic
    clips (L_dt active), checkpoint with the reference's state_dict layout that loads back strictly."""
    sys.path.insert(0, REPO)
    import train_ddp
    from tcvom_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(REPO, 'cfgs', 'vmd_vmn_gca_synthetic.yaml'))
    cfg.merge_from_list(['TRAIN.TRAIN_INPUT_SIZE', '(128, 160)', 'SYSTEM.OUTDIR', str(tmp_path), 'TRAIN.TOTAL_STEPS', '1',
                         'MODEL', arch] + (['TRAIN.BATCH_SIZE_PER_GPU', '2'] if arch == 'vmn_index' else []))   # (IndexNet: B >= 2 in train mode)
    train_ddp.main('vmd_%s_synthetic' % arch, cfg, steps_per_epoch=2, frames=5)
    ck = os.path.join(str(tmp_path), 'vmd_%s_synthetic_agg7_synthetic' % arch, 'checkpoint_1.pth.tar')
    sd = torch.load(ck, map_location='cpu')
    assert len(sd) == nkeys and all(torch.isfinite(v.float()).all() for v in sd.values())
    # the checkpoint loads back through the reference's loading code path
    from models.model import FullModel_VMD
    m = FullModel_VMD(arch, agg_window=7, dilate_kernel=12)
    missing, unexpected = m.NET.load_state_dict(sd, strict=False)
    assert not missing and not unexpected


def test_disk_clip_sharding_matches_a_distributed_sampler():
    """train_ddp.DiskClips (the DataLoader + DistributedSampler pair of train_ddp.py:232-240): per epoch every rank takes its
    stride of ONE shared permutation, ranks are disjoint, drop_last per rank, and the permutation changes with the epoch."""
    sys.path.insert(0, REPO)
    import train_ddp

    class FakeDataset(object):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def raw_view(self):
            ds = self

            class View(torch.utils.data.Dataset):
                def __len__(self):
                    return ds.n

                def __getitem__(self, i):
                    return {'idx': i}
            return View()

        def transform(self, raw):
            i = int(raw['idx'])
            return torch.full((2, 1), float(i)), torch.zeros(2, 1), torch.zeros(2, 1), torch.tensor(i)

    ds, world, batch = FakeDataset(23), 3, 2
    loaders = [train_ddp.DiskClips(ds, batch, r, world, 0, seed=5) for r in range(world)]
    assert all(len(ld) == (23 // world) // batch == 3 for ld in loaders)
    epochs = []
    for _ in range(2):
        seen = []
        for ld in loaders:
            idxs = []
            for fg, bg, a, idx in ld:
                assert tuple(fg.shape) == (batch, 2, 1) and tuple(idx.shape) == (batch,)
                assert torch.equal(fg[:, 0, 0], idx.float())
                idxs += idx.tolist()
            assert len(idxs) == 6
            seen.append(idxs)
        flat = [i for s in seen for i in s]
        assert len(set(flat)) == len(flat) == 18 and all(0 <= i < 23 for i in flat)
        epochs.append(seen)
    assert epochs[0] != epochs[1]
    g = torch.Generator()
    g.manual_seed(5)
    perm = torch.randperm(23, generator=g).tolist()
    assert epochs[0][1] == perm[1::3][:6]


@pytest.mark.gpu
def test_train_ddp_resume_continues_at_the_saved_epoch(tmp_path):
    """TRAIN.LOAD_OPT = .../optimizer_<N>.pth.tar resumes at epoch N (train_ddp.py:300-304,316): the run does not start over at
    epoch 0 -- it writes checkpoint_<N+1> only, continues the poly learning rate and keeps the Adam moments."""
    sys.path.insert(0, REPO)
    import train_ddp
    from tcvom_amd.config import get_cfg_defaults

    def cfg_for(total, extra=()):
        cfg = get_cfg_defaults()
        cfg.merge_from_file(os.path.join(REPO, 'cfgs', 'vmd_vmn_gca_synthetic.yaml'))
        cfg.merge_from_list(['TRAIN.TRAIN_INPUT_SIZE', '(64, 64)', 'SYSTEM.OUTDIR', str(tmp_path), 'TRAIN.TOTAL_STEPS', str(total)] + list(extra))
        return cfg
    train_ddp.main('resume_a', cfg_for(3), steps_per_epoch=1, frames=3)
    out_a = os.path.join(str(tmp_path), 'resume_a_agg7_synthetic')
    assert all(os.path.exists(os.path.join(out_a, 'checkpoint_%d.pth.tar' % e)) for e in (1, 2, 3))
    opt2 = os.path.join(out_a, 'optimizer_2.pth.tar')
    lr2 = torch.load(opt2, map_location='cpu')['param_groups'][0]['lr']
    train_ddp.main('resume_b', cfg_for(3, ['TRAIN.LOAD_CKPT', os.path.join(out_a, 'checkpoint_2.pth.tar'), 'TRAIN.LOAD_OPT', opt2]),
                   steps_per_epoch=1, frames=3)
    out_b = os.path.join(str(tmp_path), 'resume_b_agg7_synthetic')
    assert sorted(f for f in os.listdir(out_b) if f.startswith('checkpoint_')) == ['checkpoint_3.pth.tar']
    lr3a = torch.load(os.path.join(out_a, 'optimizer_3.pth.tar'), map_location='cpu')['param_groups'][0]['lr']
    lr3b = torch.load(os.path.join(out_b, 'optimizer_3.pth.tar'), map_location='cpu')['param_groups'][0]['lr']
    assert lr3b == lr3a and lr3b < lr2, 'the poly schedule continues from epoch 2'


@pytest.mark.gpu
def test_train_ddp_from_a_clip_directory(tmp_path):
    """DATASET.PATH set: train_ddp.py reads a VideoMatting108-style tree (1080p RGBA foregrounds, backgrounds, frame_corr.json,
    train_videos.txt) through dataset.VMD.VideoMattingDataset with PNG-decoding worker processes, and trains on it."""
    import json
    import numpy as np
    from PIL import Image
    sys.path.insert(0, REPO)
    import train_ddp
    from tcvom_amd.config import get_cfg_defaults
    root = os.path.join(str(tmp_path), 'vm108')
    H, W = 1080, 1920
    yy, xx = np.mgrid[0:H, 0:W]
    corr = {}
    os.makedirs(os.path.join(root, 'FG_done', 'clip0'))
    os.makedirs(os.path.join(root, 'BG_done', 'bg0'))
    for k in range(3):
        rgb = np.dstack([(xx + 40 * k) % 256, (yy * 2) % 256, (xx + yy) % 256]).astype(np.uint8)
        d = np.sqrt((xx - (900 + 30 * k)) ** 2 + (yy - 540) ** 2)
        alpha = np.clip((420 - d) * 2 + 128, 0, 255).astype(np.uint8)
        Image.fromarray(np.dstack([rgb, alpha]), 'RGBA').save(os.path.join(root, 'FG_done', 'clip0', '%04d.png' % k), compress_level=1)
        Image.fromarray(np.dstack([yy % 256, xx % 256, (xx * 3) % 256]).astype(np.uint8), 'RGB').save(
            os.path.join(root, 'BG_done', 'bg0', '%04d.png' % k), compress_level=1)
        corr['clip0/%04d.png' % k] = 'bg0/%04d.png' % k
    with open(os.path.join(root, 'frame_corr.json'), 'w') as f:
        json.dump(corr, f)
    for name in ('train_videos.txt', 'val_videos.txt'):
        with open(os.path.join(root, name), 'w') as f:
            f.write('clip0\n')
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(REPO, 'cfgs', 'vmd_vmn_gca_synthetic.yaml'))
    cfg.merge_from_list(['TRAIN.TRAIN_INPUT_SIZE', '(128, 128)', 'SYSTEM.OUTDIR', str(tmp_path), 'TRAIN.TOTAL_STEPS', '1',
                         'DATASET.PATH', root, 'SYSTEM.NUM_WORKERS', '2', 'TRAIN.VAL_INPUT_SIZE', '(128, 224)',
                         'TRAIN.VAL_START_EPOCH', '0'])
    train_ddp.main('vmd_vmn_gca_disk', cfg, steps_per_epoch=0, frames=5)
    out_dir = os.path.join(str(tmp_path), 'vmd_vmn_gca_disk_agg7_synthetic')
    sd = torch.load(os.path.join(out_dir, 'checkpoint_1.pth.tar'), map_location='cpu')
    assert len(sd) == 584 and all(torch.isfinite(v.float()).all() for v in sd.values())
    # the validation pass (3-frame samples, losses + the 8-bit temporal indicator) ran and kept the epoch as best.pth
    best = torch.load(os.path.join(out_dir, 'best.pth'), map_location='cpu')
    assert best.keys() == sd.keys() and all(torch.equal(best[k], sd[k]) for k in sd)


@pytest.mark.gpu
def test_adam_step_matches_torch_adam():
    """FusedAdam (one HIP launch) vs torch.optim.Adam on the same gradients, two steps, weight decay on."""
    from tcvom_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (128,), (1,), (17, 5)]
    p1 = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in p1]
    o1 = FusedAdam(p1, lr=1e-3, weight_decay=1e-4)
    o2 = torch.optim.Adam(p2, lr=1e-3, weight_decay=1e-4)
    for step in range(2):
        for a, b in zip(p1, p2):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # a parameter that gets its first gradient later keeps its OWN step count (bias correction), as in torch.optim.Adam
    p1.append(torch.nn.Parameter(torch.randn(33, device='cuda')))
    p2.append(torch.nn.Parameter(p1[-1].detach().clone()))
    o1.add_param_group({'params': [p1[-1]]})
    o2.add_param_group({'params': [p2[-1]]})
    o1.param_groups[0]['params'].append(o1.param_groups.pop()['params'][0])      # same group: two step counts in one group
    for step in range(2):
        for a, b in zip(p1, p2):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert o1.state[p1[0]]['step'] == 4 and o1.state[p1[-1]]['step'] == 2
    # load_state_dict replaces exp_avg / exp_avg_sq: the cached pointer table must follow
    sd = o1.state_dict()
    o3 = FusedAdam(p1, lr=1e-3, weight_decay=1e-4)
    o3.load_state_dict(sd)
    for a, b in zip(p1, p2):
        g = torch.randn_like(a)
        a.grad, b.grad = g.clone(), g.clone()
    o3.step()
    o2.step()
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('base', ['gca', 'fba', 'dim'])
def test_pred_vmn_runs_at_1080p(capsys, base):
    sys.path.insert(0, REPO)
    import pred_vmn
    import argparse
    pred_vmn.main(argparse.Namespace(model=base, load=None, trimap='medium', agg_window=7, clips=1, save=None))
    out = capsys.readouterr().out
    assert 'L_alpha' in out and 'L_total' in out


def _write_val_tree(root, nframes=2):
    """1080p RGBA foregrounds / RGB backgrounds of one clip + frame_corr.json + val_videos.txt (VideoMatting108 layout)."""
    import json
    import numpy as np
    from PIL import Image
    H, W = 1080, 1920
    yy, xx = np.mgrid[0:H, 0:W]
    os.makedirs(os.path.join(root, 'FG_done', 'clipA'))
    os.makedirs(os.path.join(root, 'BG_done', 'bgA'))
    corr = {}
    for k in range(nframes):
        rgb = np.dstack([(xx + 25 * k) % 256, yy % 256, (xx // 2 + yy) % 256]).astype(np.uint8)
        d = np.sqrt((xx - (960 + 20 * k)) ** 2 + (yy - 540) ** 2)
        alpha = np.clip((400 - d) * 3 + 128, 0, 255).astype(np.uint8)
        Image.fromarray(np.dstack([rgb, alpha]), 'RGBA').save(os.path.join(root, 'FG_done', 'clipA', '%04d.png' % k), compress_level=1)
        Image.fromarray(np.dstack([yy % 256, (xx * 2) % 256, xx % 256]).astype(np.uint8), 'RGB').save(
            os.path.join(root, 'BG_done', 'bgA', '%04d.png' % k), compress_level=1)
        corr['clipA/%04d.png' % k] = 'bgA/%04d.png' % k
    with open(os.path.join(root, 'frame_corr.json'), 'w') as f:
        json.dump(corr, f)
    with open(os.path.join(root, 'val_videos.txt'), 'w') as f:
        f.write('clipA\n')
    return H, W


@pytest.mark.gpu
def test_pred_vmn_from_a_precomputed_validation_tree(tmp_path, capsys):
    """pred_vmn.py --data: 1080p RGBA / RGB PNG clips -> dataset.VMD (val, precomputed, padded to 1088) -> one
    `_pred.png` and `_tri.png` per frame, cropped back to 1080x1920 (pred_vmn.py:120-134)."""
    import argparse
    import json
    import numpy as np
    from PIL import Image
    sys.path.insert(0, REPO)
    import pred_vmn
    root = os.path.join(str(tmp_path), 'val')
    H, W = _write_val_tree(root, 2)
    out = os.path.join(str(tmp_path), 'out')
    pred_vmn.main(argparse.Namespace(model='gca', load=None, trimap='medium', agg_window=7, clips=1, save=out, data=root, subset=False,
                                     n_threads=2))
    assert 'L_alpha' in capsys.readouterr().out
    for k in range(2):
        for suffix in ('_pred.png', '_tri.png'):
            im = np.asarray(Image.open(os.path.join(out, 'clipA', '%04d%s' % (k, suffix))))
            assert im.shape == (H, W) and im.dtype == np.uint8
        tri = np.asarray(Image.open(os.path.join(out, 'clipA', '%04d_tri.png' % k)))
        assert set(np.unique(tri).tolist()) <= {0, 127, 128, 255} and (tri == 255).any() and (tri == 0).any()


@pytest.mark.gpu
def test_pred_test_folder_inference(tmp_path):
    """pred_test.py counterpart: a synthetic 4-frame `*_rgb.png` / `*_trimap.png` folder (70x100, so the reflect padding
    to 96x128 is exercised) -> one alpha PNG per frame, equal to a direct EvalModel call on the padded sample."""
    from PIL import Image
    import pred_test
    from models.model import EvalModel
    from tcvom_amd.synthetic import formula_tensor, synthetic_window
    H, W, T = 70, 100, 4
    a, fg, bg = synthetic_window(1, T, H, W, seed=5)
    al = a / 255.0
    imgs = torch.round(fg * al + bg * (1 - al))[0]                 # [T,3,H,W] BGR
    tris = torch.where(a <= 0, torch.zeros_like(a), torch.where(a >= 255, torch.full_like(a, 255.0), torch.full_like(a, 128.0)))[0]
    vdir = tmp_path / 'data' / 'clip0'
    vdir.mkdir(parents=True)
    for t in range(T):
        Image.fromarray(imgs[t].permute(1, 2, 0).numpy().astype(np.uint8)[..., ::-1].copy()).save(str(vdir / ('%04d_rgb.png' % t)))
        Image.fromarray(tris[t, 0].numpy().astype(np.uint8)).save(str(vdir / ('%04d_trimap.png' % t)))
    from models.model import FullModel_VMD
    fm = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=12)
    fm.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in fm.NET.state_dict().items()})
    fm = fm.to('cuda').train()
    with torch.no_grad():                                          # calibrate the BatchNorm running statistics
        for _ in range(2):
            fm(*(t.cuda() for t in synthetic_window(1, 3, 96, 128, seed=0)))
    em = EvalModel('vmn_gca', agg_window=7, dilate_kernel=2)
    em.NET.load_state_dict(fm.NET.state_dict())
    ck = str(tmp_path / 'net.pth')
    torch.save(em.NET.state_dict(), ck)
    args = pred_test.parse(['--data', str(tmp_path / 'data'), '--load', ck, '--save', str(tmp_path / 'out'), '--dilation', '2'])
    outs = pred_test.main(args)
    assert len(outs) == T and all(os.path.exists(o) for o in outs)
    ds = pred_test.TestFolder(str(tmp_path / 'data'), [])
    assert [os.path.basename(s[1][0]) for s in ds.samples] == ['%04d_rgb.png' % t for t in range(T)]
    assert os.path.basename(ds.samples[0][0][0]) == '0001_rgb.png' and os.path.basename(ds.samples[-1][2][0]) == '0002_rgb.png'
    x, tr, (h, w) = ds[1]
    assert tuple(x.shape) == (3, 3, 96, 128) and (h, w) == (H, W)
    assert torch.equal(x[1, :, :H, :W], imgs[1]) and torch.equal(x[1, :, H:, :W], torch.flip(imgs[1][:, H - 1 - (96 - H):H - 1, :], [1]))
    em = em.to('cuda').eval()
    direct = em(x.cuda().unsqueeze(0), tr.cuda().unsqueeze(0)).squeeze()[1][:H, :W].cpu().numpy()
    got = np.asarray(Image.open(outs[1])).astype(np.float32)
    assert np.isfinite(direct).all() and 0.0 < direct.mean() < 1.0
    # the clip path (features computed once per frame, 4 frames per launch) vs the per-sample window: the same per-frame math --
    # every frame's tiles are computed by the same instructions whatever the frame batch, so the PNGs are identical
    assert got.shape == (H, W) and np.abs(got - np.uint8(direct * 255).astype(np.float32)).max() <= 1
    args2 = pred_test.parse(['--data', str(tmp_path / 'data'), '--load', ck, '--save', str(tmp_path / 'out2'), '--dilation', '2', '--per_sample'])
    outs2 = pred_test.main(args2)
    for a_fn, b_fn in zip(outs, outs2):
        pa, pb = np.asarray(Image.open(a_fn)).astype(np.float32), np.asarray(Image.open(b_fn)).astype(np.float32)
        assert np.array_equal(pa, pb), (np.abs(pa - pb).mean(), np.abs(pa - pb).max())
    got2 = np.asarray(Image.open(outs2[1])).astype(np.float32)
    assert np.abs(got2 - np.uint8(direct * 255).astype(np.float32)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize('base', ['dim', 'gca', 'fba', 'index'])
def test_pred_single_dim_config1(base):
    """BASELINE.json config 1: pred_single.py, DIM base, one 512 x 512 synthetic frame + trimap (eval mode); the GCA and FBA
    single-image bases run through the same script."""
    import pred_single
    out = pred_single.main(pred_single.parse(['--model', base, '--trimap', 'medium', '--frames', '1']))
    assert set(out) == {'L_alpha', 'L_comp', 'L_grad', 'L_total', 'mSAD', 'MSE'}
    assert all(np.isfinite(v) for v in out.values()) and abs(out['L_total'] - out['L_alpha'] - out['L_comp'] - out['L_grad']) < 1e-5


@pytest.mark.gpu
def test_pred_single_from_a_precomputed_validation_tree(tmp_path):
    """pred_single.py --data (the `vmd` branch of the reference script): single-image GCA over the 3-frame validation samples,
    centre frame scored over the unknown band on rows :1080, PNGs written."""
    from PIL import Image
    import pred_single
    root = os.path.join(str(tmp_path), 'val')
    H, W = _write_val_tree(root, 2)
    out_dir = os.path.join(str(tmp_path), 'out')
    out = pred_single.main(pred_single.parse(['--model', 'gca', '--trimap', 'narrow', '--data', root, '--save', out_dir, '--n_threads', '2']))
    assert set(out) == {'L_alpha', 'L_comp', 'L_grad', 'L_total', 'mSAD', 'MSE'} and all(np.isfinite(v) for v in out.values())
    assert out['mSAD'] > 0
    for k in range(2):
        assert np.asarray(Image.open(os.path.join(out_dir, 'clipA', '%04d_pred.png' % k))).shape == (H, W)


@pytest.mark.gpu
def test_calc_metric_folder(tmp_path):
    """calc_metric.py counterpart on a synthetic prediction folder (2 videos x 3 frames of _pred / _tri PNGs, FG_done RGBA
    ground truth, frame_corr.json): the per-frame and averaged SAD / MSE / SSDA / dtSSD equal the oracle's numpy values."""
    import json
    from PIL import Image
    import calc_metric
    from oracle import metrics as om
    from helpers import hu
    H, W = 40, 56
    pred, data = tmp_path / 'pred', tmp_path / 'data'
    frames, arrays = {}, {}
    for v in ('vidA', 'vidB'):
        (pred / v).mkdir(parents=True)
        (data / 'FG_done' / v).mkdir(parents=True)
        for t in range(3):
            fn = '%s/%04d' % (v, t)
            a = np.uint8((hu('cm.a.' + fn, (H, W)).numpy() * 0.5 + 0.5) * 255)
            g = np.uint8(np.clip(a.astype(np.float32) + hu('cm.g.' + fn, (H, W)).numpy() * 30, 0, 255))
            u = hu('cm.t.' + fn, (H, W)).numpy()
            tri = np.where(u < -0.3, 0, np.where(u > 0.4, 255, 128)).astype(np.uint8)
            Image.fromarray(a).save(str(pred / (fn + '_pred.png')))
            Image.fromarray(tri).save(str(pred / (fn + '_tri.png')))
            Image.fromarray(np.dstack([g, g, g, g])).save(str(data / 'FG_done' / (fn + '.png')))
            frames[fn + '.png'] = []
            arrays[fn] = (np.float32(a / 255.0), np.float32(g / 255.0), tri)
    with open(str(data / 'frame_corr.json'), 'w') as f:
        json.dump(frames, f)
    out = str(tmp_path / 'metric.json')
    calc_metric.main(argparse_ns(pred=str(pred), data=str(data), output=out, vis=False, n_threads=None))
    res = json.load(open(out))
    assert sorted(res['all'].keys()) == ['vidA', 'vidB']
    for v in ('vidA', 'vidB'):
        sad = []
        for t in range(3):
            fn = '%s/%04d' % (v, t)
            a, g, tri = arrays[fn]
            nxt = arrays.get('%s/%04d' % (v, t + 1))
            want = om.frame_metrics(a, g, tri, *(nxt[:2] if nxt else (None, None)))
            got = res['all'][v]['all'][fn]
            assert got['pixel_count'] == want['pixels']
            for k, wk in (('mSAD', 'SAD'), ('MSE', 'MSE'), ('SSDA', 'SSDA')):
                assert abs(got[k] - want[wk]) <= 1e-5 * want[wk]
            assert abs(got['dtSSD'] - want.get('dtSSD', 0.0)) <= 1e-5 * max(want.get('dtSSD', 0.0), 1e-9)
            sad.append(want['SAD'])
        assert abs(res['all'][v]['avg']['mSAD'] - np.mean(sad)) <= 1e-5
    assert abs(res['avg']['mSAD'] - np.mean([res['all'][v]['avg']['mSAD'] for v in ('vidA', 'vidB')])) <= 1e-9


def argparse_ns(**kw):
    import argparse
    return argparse.Namespace(**kw)


@pytest.mark.gpu
def test_bench_two_ranks_dry_run():
    """bench.py's N > 1 path end to end (rendezvous, state broadcast, in-place gradient all-reduce, max-over-ranks timing,
    one JSON line from rank 0) with two ranks sharing this GPU over gloo (RCCL wants one GPU per rank)."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TCVOM_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--height', '128', '--width', '160', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch_clips'] == 2 and res['value'] > 0 and res['scaling'] == 'weak'
    assert res['dist']['world_size'] == 2 and res['dist']['grad_spans_overlapped_with_backward'] >= 3, res['dist']
    # N > 1 runs train_ddp.py's semantics: SyncBatchNorm, exchanged through the peer mailboxes (72 BatchNorms forward + backward)
    assert res['config']['sync_bn'] is True and res['dist']['sync_bn_transport'] == 'mailbox', res['dist']
    assert res['dist']['sync_bn_exchanges_per_step'] == 144, res['dist']
    assert np.isfinite(res['final_loss'])


@pytest.mark.gpu
def test_bench_two_ranks_full_size_syncbn_preflight():
    """BASELINE config 4 has never run on 8 GPUs (the driver's to take); this is what one GPU can check of it: the SAME command at
    the FULL 3 x 1088 x 1920 geometry with two ranks sharing this device (gloo for the hipIpc handles), every one of the 72
    BatchNorms exchanging its [3 frames][2][C <= 512] sums through the peer mailboxes in forward and backward (train_ddp.py:271-280):
    144 exchanges per step over a ring of 4 (36 reuses per slot and step), mailbox capacity for the widest layer, per-rank
    diagnostics present, and the per-rank-BatchNorm A/B (`dist.no_sync_bn_ms_per_step`) in the same JSON line."""
    import json
    import socket
    import subprocess
    from tcvom_amd.mailbox import CAPACITY, RING
    assert 3 * 2 * 512 <= CAPACITY and RING >= 4          # PeerMailbox.fits(3, 512): the os32 bottleneck, three frames per exchange
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TCVOM_DIST_BACKEND='gloo', TCVOM_MBOX_TIMEOUT_S='60')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    d = res['dist']
    print({k: v for k, v in d.items() if k != 'per_rank'}, d['per_rank'])
    assert res['n_gpus'] == 2 and res['config']['height'] == 1088 and res['config']['width'] == 1920 and res['config']['sync_bn'] is True
    assert d['sync_bn_transport'] == 'mailbox' and d['sync_bn_exchanges_per_step'] == 144, d
    assert len(d['per_rank']) == 2
    for r in d['per_rank']:
        assert r['sync_bn_transport'] == 'mailbox' and r['mailbox_exchanges_per_step'] == 144 and r['mailbox_world'] == 2, r
        assert r['mailbox_self_test'] == 'passed' and r['mailbox_max_wait_ms'] >= 0.0 and r['ipc_peer_devices'], r
        assert r['allreduce_exposed_ms_per_step'] >= 0.0
    assert d['grad_spans_overlapped_with_backward'] >= 3
    assert d['no_sync_bn_ms_per_step'] > 0 and d['no_sync_bn_steps'] >= 2, d
    assert np.isfinite(res['final_loss']) and res['value'] > 0


@pytest.mark.gpu
def test_bench_fba_config_line():
    """`bench.py --config fba` (BASELINE.json config 5) prints the same one-line JSON contract as the headline run."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--config', 'fba', '--steps', '2', '--warmup', '1', '--height', '128',
           '--width', '160', '--no-cpu-baseline']
    out = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert 'FBA+TAM' in res['metric'] and res['unit'] == 'windows/s' and res['value'] > 0 and res['n_gpus'] == 1
    assert res['roofline']['bound'] == 'mfma' and res['roofline']['achieved'] > 0 and 'all_igemm' in res['roofline']
