#!/usr/bin/env python
"""Headline benchmark: synthetic 1080p (1088x1920) 3-frame windows/s, GCA+TAM, forward + backward
(L_alpha + 0.5 L_dt + 0.25 L_att, train_ddp.py:56-61) + gradient all-reduce + Adam; 16-bit activations / packed weights (bf16, the
north star's type, for the GCA+TAM window; fp16 for --config fba as BASELINE config 5 names it; TCVOM_DTYPE overrides: same MFMA rate),
fp32 accumulation, statistics, softmax, losses and master weights.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; each rank trains on its own clip (weak scaling, clips are independent — the only
data-path collective is the gradient all-reduce over RCCL/xGMI).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of one window (FlopCounterMode on the reference, 2 flop/MAC)
GFLOP_FWD_CONV_1080P = 1737.3      # non-GCA convolutions, forward
GFLOP_FWD_GCA_1080P = 2106.3       # 6 guided-contextual-attention calls, forward (quadratic in pixels)
GFLOP_WINDOW_1080P = 11179.81      # forward + backward
GFLOP_WINDOW_FBA_1080P = 21200.0   # BASELINE config 5 (FBA+TAM): 2663 GFLOP fwd+bwd at 384x672 x 7.96 pixels
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16 (MI355X_MICROARCH.md)
FULL_H, FULL_W = 1088, 1920
PROFILE_DIR = os.path.join(ROOT, 'profiles')


TRACED_STEPS = 3                   # the PMC passes trace `bench.py --steps 2 --warmup 1` (the instrumented extra step switched off)


def profile_traffic(variant, calls_per_step):
    """HBM bytes per C-ABI LAUNCH of the kernel family behind `variant`, read from the newest committed PMC summary
    (profiles/r*_hbm_traffic_pmc.md: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes of this script, fetch
    side doubled per MI355X_MICROARCH.md's gfx950 correction): the family's counter bytes per STEP divided by `calls_per_step`, the
    number of C-ABI calls the event-instrumented step counted -- the same denominator as `algorithmic_mib_per_launch` (a C-ABI call may
    be two kernels: tcvom_gca_dq_dk; dividing by the table's kernel-launch count made the two figures incomparable).
    bench.py cannot run rocprofv3 on itself; returns (bytes or None, file name or None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(PROFILE_DIR, 'r*_hbm_traffic_pmc*.md')))
    if not files:
        return None, None
    m = re.match(r'(\w+?)(<.*>)?$', variant)
    base, targs = m.group(1), (m.group(2) or '')
    if base == 'gemm_nt256':
        key = 'gemm_nt256_kernel<'                     # the family: every instantiation of the 256 x 256 GEMM, launch-weighted
    else:
        key = base + '_kernel' + (targs[:-1].replace(',', ', ') if targs else '')    # the table truncates long names
    tot, launches = 0.0, 0
    for line in open(files[-1]):
        if line.startswith('| `') and key in line.replace('void ', ''):
            cols = [c.strip() for c in line.split('|')]
            try:
                n = int(cols[2])
                tot += n * (float(cols[3]) + float(cols[4])) * 2 ** 20
                launches += n
            except (ValueError, IndexError):
                pass
    return (tot / TRACED_STEPS / calls_per_step if (launches and calls_per_step) else None), os.path.basename(files[-1])


ALGO_GB_WINDOW_1080P = 17.8        # ideal conv+BN+act fusion, every layer reads its input and writes its output once, x3 for fwd+bwd


def profile_step_traffic():
    """(GiB of counter traffic per step, file) from the newest committed profiles/r*_hbm_traffic_pmc*.md: its last line is the total
    over the traced process, which runs `bench.py --steps 2 --warmup 1` (3 steps; the instrumented extra step is switched off)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(PROFILE_DIR, 'r*_hbm_traffic_pmc*.md')))
    if not files:
        return None, None
    m = re.search(r'total HBM traffic over the traced process: ([0-9.]+) GiB', open(files[-1]).read())
    return (round(float(m.group(1)) / TRACED_STEPS, 2) if m else None), os.path.basename(files[-1])


def tam_all_unknown(device, h, w, C=128, window=7, reps=20):
    """The Temporal Attention Module launches on a window whose EVERY os8 pixel is unknown (the synthetic bench window has ~3 %):
    algorithmic bytes / HIP-event time of `tcvom_tam_fwd` and `tcvom_tam_bwd` called through the C ABI back to back (the launchers'
    own memsets / copies included, no autograd or ATen work around them -- tools/tam_kernels.py is the same measurement with other
    masks), forward and forward + backward, against the 8 TB/s HBM peak."""
    import tcvom_amd._lib as L
    g = torch.Generator(device='cpu').manual_seed(0)
    mk = lambda: torch.randn(1, h, w, C, generator=g).to(device).to(L.ACT_DTYPE)
    q, kb, kf, v, dout = mk(), mk(), mk(), mk(), mk()
    out, dq, dkb, dkf = (torch.empty_like(q) for _ in range(4))
    mask = torch.ones(1, h, w, dtype=torch.uint8, device=device)
    w2 = window * window
    attb = torch.empty(1, w2, h * w, device=device)
    attf = torch.empty_like(attb)
    datt = torch.ones_like(attb)
    pbuf = torch.empty(1, 2, w2, h * w, device=device)
    dsbuf = torch.empty_like(pbuf)
    work = torch.empty(h * w + 1, dtype=torch.int32, device=device)
    st = L.stream_ptr()
    fwd_bytes = 5 * h * w * C * 2 + h * w + 2 * w2 * h * w * 4
    bwd_bytes = 7 * h * w * C * 2 + h * w + 2 * w2 * h * w * 4 * 3

    def fwd():
        L.call('tcvom_tam_fwd', L.ptr(q), L.ptr(kb), L.ptr(kf), L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(attb), L.ptr(attf), L.ptr(work),
               1, h, w, C, window, st)

    def bwd():
        L.call('tcvom_tam_bwd', L.ptr(q), L.ptr(kb), L.ptr(kf), L.ptr(mask), L.ptr(dout), L.ptr(datt), L.ptr(datt), L.ptr(dq), L.ptr(dkb),
               L.ptr(dkf), L.ptr(pbuf), L.ptr(dsbuf), L.ptr(work), 1, h, w, C, window, st)
    res = {}
    for name, nbytes, fns in (('fwd', fwd_bytes, (fwd,)), ('fwd+bwd', fwd_bytes + bwd_bytes, (fwd, bwd))):
        for _ in range(3):
            for f in fns:
                f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for f in fns:
                f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[name] = {'ms': round(ms, 4), 'GBps': round(nbytes / ms / 1e6, 1), 'frac_of_8TBps': round(nbytes / ms / 1e6 / 8000.0, 4)}
    return res


def window_gflop(H, W, config='gca'):
    """fwd+bwd GFLOP of one 3-frame window at HxW, scaled from the measured 1088x1920 figures."""
    r = (H * W) / float(FULL_H * FULL_W)
    if config == 'fba':
        return GFLOP_WINDOW_FBA_1080P * r          # no attention term quadratic in the pixels
    fwd = GFLOP_FWD_CONV_1080P * r + GFLOP_FWD_GCA_1080P * r * r
    return fwd * (GFLOP_WINDOW_1080P / (GFLOP_FWD_CONV_1080P + GFLOP_FWD_GCA_1080P))


def build(device, H, W, seed, config='gca'):
    from models.model import FullModel_VMD
    from tcvom_amd.synthetic import formula_tensor, synthetic_window
    model = FullModel_VMD({'fba': 'vmn_fba', 'index': 'vmn_index'}.get(config, 'vmn_gca'), agg_window=7, dilate_kernel=12)
    model.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in model.NET.state_dict().items()})
    model = model.to(device).train()
    if os.environ.get('TCVOM_FRAME_STREAMS', '1') == '0' and hasattr(model.NET, 'frame_streams'):      # profiling aid
        model.NET.frame_streams = False
    if os.environ.get('TCVOM_BATCHED_FRAMES', '1') == '0' and hasattr(model.NET, 'batched_frames'):    # A/B aid
        model.NET.batched_frames = False
    a, fg, bg = synthetic_window(2 if config == 'index' else 1, 3, H, W, seed=seed)
    return model, a.to(device), fg.to(device), bg.to(device)


def igemm_profile(step_fn):
    """One extra, event-instrumented step: every igemm launch is bracketed by HIP events on the launch
    stream.  Returns {variant: (launches, total_ms, total_gflop)}."""
    import tcvom_amd._lib as L
    rec = []
    L.PROFILE = rec
    step_fn()
    torch.cuda.synchronize()
    L.PROFILE = None
    agg = {}
    tam = {}
    algo = {}
    for name, desc, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        if 'bytes' in desc:                          # Temporal Attention Module launches: an HBM-bound kernel family
            n, t, b = tam.get(desc['variant'], (0, 0.0, 0))
            tam[desc['variant']] = (n + 1, t + ms, b + desc['bytes'])
            continue
        real_taps = sum(1 for t in range(desc['ntaps']) if desc['tap_w'][t] >= 0)
        gflop = desc['gflop'] if 'gflop' in desc else 2.0 * desc['P'] * desc['K'] * real_taps * desc['C'] * max(desc['batch'], 1) / 1e9
        var = desc['variant']                       # the instantiation the library selected for this shape
        n, t, g = agg.get(var, (0, 0.0, 0.0))
        agg[var] = (n + 1, t + ms, g + gflop)
        nb, bb = algo.get(var, (0, 0))
        algo[var] = (nb + 1, bb + int(desc.get('algo_bytes', 0)))
    igemm_profile.tam = tam
    igemm_profile.algo = algo                       # {variant: (launches, algorithmic bytes)}: every operand once
    return agg


def cpu_baseline_fba(sample_hw=(544, 960), threads=32):
    """Config 5 (FBA+TAM): the oracle's fba_window_forward + backward on a BOUNDED sample (one 3x544x960 window after a
    small warm-up; the ResNet-50 GN+WS trunk at os8 makes a full 1080p pass minutes long on the host), scaled to
    3x1088x1920 by the pixel ratio (the FBA path has no term quadratic in the pixels)."""
    import oracle.fba_net as fba
    from tcvom_amd.synthetic import formula_tensor, synthetic_window
    import numpy as np
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, threads)
    torch.set_num_threads(cores)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'fba_state_keys.npz'))
    state = {}
    for k, shp in zip(g['keys'], g['shapes']):
        shape = tuple(int(d) for d in str(shp).split(',')) if str(shp) else ()
        state[str(k)] = formula_tensor(str(k), shape).requires_grad_(True)

    def one(H, W):
        a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
        t0 = time.time()
        out, _ = fba.fba_window_forward(state, a, fg, bg, window=7, dilate_kernel=12)
        (out[0].mean() + out[1].mean() + out[2].mean() + 0.5 * out[3].mean() + 0.25 * out[4].mean()).backward()
        for v in state.values():
            v.grad = None
        return time.time() - t0

    one(128, 160)
    H, W = sample_hw
    dt = one(H, W)
    ratio = window_gflop(H, W, 'fba') / window_gflop(FULL_H, FULL_W, 'fba')
    return {'value': round(ratio / dt, 6), 'unit': 'windows/s', 'cores': cores, 'host_logical_cores': host_cores, 'kind': 'port',
            'sample': 'one fwd+bwd pass over a 3x%dx%d window in %.1f s on %d threads after a 3x128x160 warm-up, scaled to '
                      '3x%dx%d by the pixel ratio %.4f' % (H, W, dt, cores, FULL_H, FULL_W, ratio)}


def cpu_baseline(sample_hw=(FULL_H, FULL_W), threads=32, reps=3):
    """The oracle (pure PyTorch fp32 restatement of the reference, pinned by tests/golden) timed on the host cores on the
    SAME 3x1088x1920 window as the GPU line: one small warm-up window (thread pool, allocator), then `reps` timed
    forward+backward passes, median reported.  `cores` is the thread count actually used: PyTorch's CPU kernels get
    SLOWER on these shapes beyond ~32 threads (measured on the 256-logical-core MI355X host: 19-21 s per window at 32
    threads, 30 s at 64, 55 s at 128 -- tests/cpu_oracle_time.py), so 32 is the best the port does on this box."""
    import statistics
    import oracle
    from oracle.state_spec import vmn_gca_state_spec
    from tcvom_amd.synthetic import formula_tensor, synthetic_window
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, threads)
    torch.set_num_threads(cores)
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
             for k, s in vmn_gca_state_spec().items()}
    for k, v in state.items():
        if v.is_floating_point() and not any(t in k for t in ('weight_u', 'weight_v', 'running_')):
            v.requires_grad_(True)

    keep = {}

    def one(H, W):
        a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
        t0 = time.time()
        out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True)
        oracle.train_step_loss(out).backward()
        for v in state.values():
            v.grad = None
        dt = time.time() - t0
        if (H, W) == tuple(sample_hw) and 'alpha' not in keep:      # the checker's matte of this window, for the `parity` object (after the clock)
            keep['alpha'], keep['tris_vis'] = out[7].detach().clone(), out[6].detach().clone()
        return dt

    one(256, 448)                                            # warm-up, not timed
    H, W = sample_hw
    times = sorted(one(H, W) for _ in range(reps))
    cpu_baseline.oracle_outputs = keep
    dt = statistics.median(times)
    ratio = window_gflop(H, W) / window_gflop(FULL_H, FULL_W)
    return {'value': round(ratio / dt, 6), 'unit': 'windows/s', 'cores': cores, 'host_logical_cores': host_cores, 'kind': 'port',
            'sample': 'median of %d fwd+bwd passes over one 3x%dx%d window after a 3x256x448 warm-up: %s s on %d threads '
                      '(more threads measured slower, tests/cpu_oracle_time.py)%s'
                      % (reps, H, W, '/'.join('%.1f' % t for t in times), cores,
                         '' if (H, W) == (FULL_H, FULL_W) else ', scaled to 3x%dx%d by the algorithmic FLOP ratio %.4f' % (FULL_H, FULL_W, ratio))}


def other_configs(steps, warmup):
    """The BASELINE.json configurations the headline line does not time, each as a run of THIS script in a subprocess on the same
    GPU (the subprocess's own barrier-bracketed wall clock over its timed steps, as the headline), every one in the storage type
    BASELINE.json names for it AND in the other 16-bit build (TCVOM_DTYPE): configs[1] = GCA+TAM forward-only 3x512x512 (bf16);
    configs[4] = FBA+TAM fwd+bwd 3x1088x1920 (fp16; with its own window_mfma_frac); and configs[2] -- the headline, bf16 -- again in
    the fp16 build with the SAME steps / warm-up as the headline run (a co-equal line, not a side note)."""
    import subprocess
    runs = (('config2_gca_tam_fwd_512_bf16', ['--height', '512', '--width', '512', '--forward-only', '--steps', '30', '--warmup', '5'], 'bf16'),
            ('config2_gca_tam_fwd_512_fp16', ['--height', '512', '--width', '512', '--forward-only', '--steps', '30', '--warmup', '5'], 'fp16'),
            ('config5_fba_tam_fwd_bwd_1080p_fp16', ['--config', 'fba', '--steps', '8', '--warmup', '2'], 'fp16'),
            ('config5_fba_tam_fwd_bwd_1080p_bf16', ['--config', 'fba', '--steps', '8', '--warmup', '2'], 'bf16'),
            ('config3_gca_tam_fwd_bwd_1080p_fp16', ['--steps', str(steps), '--warmup', str(warmup)], 'fp16'),
            # the headline again with the attention probabilities NOT flushed to exact zeros below 2^-25 (TCVOM_NO_P_FLUSH=1, DESIGN.md
            # section 0(6)): the dominant GEMM family on dense operands -- the sustained clock, hence `frac`, leans on the zeros of the
            # bench window; this line is the figure without that help (`roofline.frac_dense_operands`)
            ('config3_dense_operands_bf16', ['--steps', '5', '--warmup', '2'], 'bf16'))
    with_roofline = ('config5_fba_tam_fwd_bwd_1080p_fp16', 'config5_fba_tam_fwd_bwd_1080p_bf16', 'config3_dense_operands_bf16')
    out = {}
    for name, flags, dtype in runs:
        t0 = time.time()
        e = dict(os.environ, TCVOM_DTYPE=dtype)
        if name == 'config3_dense_operands_bf16':
            e['TCVOM_NO_P_FLUSH'] = '1'
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
            e.pop(k, None)
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--no-cpu-baseline', '--no-other-configs'] +
                               ([] if name in with_roofline else ['--no-profile']) + flags,
                               env=e, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')]
            if p.returncode != 0 or not line:
                out[name] = {'error': (p.stderr or p.stdout)[-400:]}
                continue
            r = json.loads(line[-1])
            out[name] = {k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'window_mfma_frac', 'final_loss')}
            out[name]['workload'] = r['config']['workload']
            out[name]['forward_only'] = r['config'].get('forward_only', False)
            if name in with_roofline and 'roofline' in r:        # the side run's own dominant kernel against the MFMA peak, event-timed live
                out[name]['roofline'] = {k: r['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'launches_per_step',
                                                                       'avg_launch_ms', 'algorithmic_gflop_per_launch', 'all_igemm', 'aggregate')
                                         if k in r['roofline']}
            out[name]['subprocess_s'] = round(time.time() - t0, 1)
        except Exception as ex:                     # noqa: BLE001 -- the headline line must survive a failing side run
            out[name] = {'error': repr(ex)[-400:]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--height', type=int, default=FULL_H)
    ap.add_argument('--width', type=int, default=FULL_W)
    ap.add_argument('--config', choices=('gca', 'fba', 'index'), default='gca',
                    help='gca: the headline GCA+TAM window (BASELINE.json configs[2]); fba: FBA+TAM (configs[4], the heaviest base); '
                         'index: IndexNet+TAM (not a BASELINE config; 2 clips per step: its ASPP has a BatchNorm over the batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--forward-only', action='store_true',
                    help='time the forward pass alone (train-mode statistics, no_grad): BASELINE.json configs[1] with --height 512 --width 512')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the `other_configs` object of the default 1-GPU run (config 2 and config 5 in both 16-bit builds, config 3 in '
                         'the fp16 build at the headline steps / warm-up: each a run of this script in a subprocess)')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--sync-bn', dest='sync_bn', action='store_true', default=None,
                    help='SyncBatchNorm statistics over the ranks, as train_ddp.py:271-273 converts every non-FBA model: the DEFAULT '
                         'for --gpus > 1 (gca / index).  The [frames][2][C] sums are exchanged inside the BatchNorm finalize kernels '
                         'through hipIpc peer mailboxes (TCVOM_SYNCBN=rccl: one all-reduce per BatchNorm call instead).  With '
                         '--gpus 1 the exchange runs against a one-rank loop-back mailbox (its cost on one GPU)')
    ap.add_argument('--no-sync-bn', dest='sync_bn', action='store_false', help='per-rank BatchNorm statistics (the A/B of --sync-bn)')
    ap.add_argument('--no-ab', action='store_true',
                    help='N > 1 with SyncBatchNorm: skip the short per-rank-BatchNorm A/B (`dist.no_sync_bn_ms_per_step`) that follows the timed steps')
    args = ap.parse_args()

    # storage type of the run, unless TCVOM_DTYPE says otherwise: what BASELINE.json names for the configuration -- bf16 for the
    # GCA+TAM window (configs[1..3], the north star), fp16 for FBA+TAM (configs[4])
    os.environ.setdefault('TCVOM_DTYPE', 'fp16' if args.config == 'fba' else 'bf16')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    # TCVOM_DIST_BACKEND=gloo: dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices; RCCL
    # refuses that).  The driver's multi-GPU runs use the default: one rank per GPU over RCCL.
    backend = os.environ.get('TCVOM_DIST_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        # device_id: RCCL binds this rank's communicator to its GPU up front (no guessing in barrier())
        dist.init_process_group(backend=backend, init_method='env://', **({'device_id': device} if backend == 'nccl' else {}))
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world)

    import tcvom_amd._lib as L
    from tcvom_amd.ddp import GradientAverager, banks_of, broadcast_module_state, convert_sync_batchnorm, sync_batchnorm_info
    from tcvom_amd.facade import train_step_loss
    from tcvom_amd.optim import FusedAdam
    H, W = args.height, args.width
    model, a, fg, bg = build(device, H, W, seed=rank, config=args.config)
    base = {'fba': 'FBA+TAM', 'index': 'IndexNet+TAM'}.get(args.config, 'GCA+TAM')
    clips = 2 if args.config == 'index' else 1
    # train_ddp.py:271-273: every model but FBA (GroupNorm) trains with SyncBatchNorm under DDP
    sync_bn = (world > 1 and args.config != 'fba') if args.sync_bn is None else bool(args.sync_bn)
    if sync_bn:
        if world > 1:
            convert_sync_batchnorm(model)
        else:
            from tcvom_amd.mailbox import PeerMailbox
            convert_sync_batchnorm(model, mailbox=PeerMailbox(loopback=True))
    broadcast_module_state(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, weight_decay=1e-4)
    # the banks finish the flat gradient in layer ranges; each range's all-reduce starts while the next still computes
    averager = GradientAverager(params, banks=banks_of(model))

    ar_events = []           # (before, after) HIP events around the gradient averaging of every timed step (N > 1)

    def step():
        if args.forward_only:
            with torch.no_grad():
                return train_step_loss(model(a, fg, bg))
        out = model(a, fg, bg)
        loss = train_step_loss(out)
        model.zero_grad(set_to_none=True)
        loss.backward()
        if world > 1 and ar_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            averager.average()
            e1.record()
            ar_events.append((e0, e1))
        else:
            averager.average()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    sync_transport, sync_n0 = sync_batchnorm_info(model)
    del ar_events[:]
    averager.host_presence_ms_total, averager.host_presence_calls = 0.0, 0
    mbox = next((getattr(m, 'sync_mailbox', None) for m in model.modules() if getattr(m, 'sync', False)), None)
    if mbox is not None:
        mbox.wait_stats(reset=True)
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
    fence()
    sync_n1 = sync_batchnorm_info(model)[1]
    elapsed = torch.tensor([time.time() - t0], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed)
    final_loss = float(loss.detach())
    # ---- per-rank diagnostics of the N > 1 path (gathered to rank 0; the one-rank loop-back line carries the same fields)
    # exposed all-reduce: time the compute stream spent inside GradientAverager.average() -- waiting for the spans whose all-reduce
    # started during backward, plus the small remainder (BatchNorm arena, biases) -- per step, from HIP events on that stream
    diag = {'rank': rank, 'device': dev_index,
            'allreduce_exposed_ms_per_step': round(sum(e0.elapsed_time(e1) for e0, e1 in ar_events) / max(len(ar_events), 1), 4) if ar_events else 0.0,
            'sync_bn_transport': sync_transport,
            # the one blocking HOST collective of a step: GradientAverager._fill_locally_unused's 2.4 KB gloo all-reduce (wall time)
            'host_presence_allreduce_ms_per_step': round(averager.host_presence_ms_total / max(averager.host_presence_calls, 1), 4)}
    if mbox is not None:
        mx, sm = mbox.wait_stats(reset=True)
        diag.update(mailbox_max_wait_ms=round(mx, 4), mailbox_wait_ms_per_step=round(sm / max(args.steps, 1), 4),
                    mailbox_exchanges_per_step=(sync_n1 - sync_n0) // max(args.steps, 1), mailbox_world=mbox.world,
                    ipc_peer_devices=dict(mbox.peer_devices), ipc_crossed_devices=mbox.crossed_devices(),
                    mailbox_self_test='passed' if mbox.world > 1 else 'loop-back (one rank)')
    ar_events = None
    diags = [diag]
    if world > 1:
        diags = [None] * world
        dist.all_gather_object(diags, diag)

    # ---- N > 1 with SyncBatchNorm: the same job again WITHOUT the statistics exchange (per-rank BatchNorm), a short A/B inside
    # this run, so that one multi-GPU lease separates the cost of SyncBatchNorm from the cost of the gradient all-reduce
    no_sync = None
    if world > 1 and sync_bn and not args.forward_only and not args.no_ab:
        ab_steps, ab_warm = max(2, min(args.steps, 6)), 2
        model_b, a_b, fg_b, bg_b = build(device, H, W, seed=rank, config=args.config)
        broadcast_module_state(model_b)
        params_b = [p for p in model_b.parameters() if p.requires_grad]
        opt_b = FusedAdam(params_b, lr=1e-4, weight_decay=1e-4)
        av_b = GradientAverager(params_b, banks=banks_of(model_b))
        ar_b = []

        def step_b(record):
            out = model_b(a_b, fg_b, bg_b)
            loss_b = train_step_loss(out)
            model_b.zero_grad(set_to_none=True)
            loss_b.backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            av_b.average()
            e1.record()
            if record:
                ar_b.append((e0, e1))
            opt_b.step()
        for _ in range(ab_warm):
            step_b(False)
        fence()
        tb = time.time()
        for _ in range(ab_steps):
            step_b(True)
        fence()
        el_b = torch.tensor([time.time() - tb], dtype=torch.float64, device=device)
        dist.all_reduce(el_b, op=dist.ReduceOp.MAX)
        ar_ms = torch.tensor([sum(e0.elapsed_time(e1) for e0, e1 in ar_b) / ab_steps], dtype=torch.float64, device=device)
        dist.all_reduce(ar_ms, op=dist.ReduceOp.MAX)
        no_sync = {'no_sync_bn_ms_per_step': round(1e3 * float(el_b) / ab_steps, 3), 'no_sync_bn_steps': ab_steps, 'no_sync_bn_warmup': ab_warm,
                   'no_sync_bn_windows_per_s': round(world * clips * ab_steps / float(el_b), 4),
                   'no_sync_bn_allreduce_exposed_ms_per_step_max_over_ranks': round(float(ar_ms), 4)}
        del model_b, opt_b, av_b, params_b
        torch.cuda.empty_cache()

    result = None
    if rank == 0:
        win_per_s = world * clips * args.steps / elapsed
        gflop = window_gflop(H, W, args.config) if args.config != 'index' else None      # (no FLOP count taken for IndexNet)
        mode = 'fwd' if args.forward_only else 'fwd+bwd'
        if args.forward_only and gflop is not None:
            gflop *= (GFLOP_FWD_CONV_1080P + GFLOP_FWD_GCA_1080P) / GFLOP_WINDOW_1080P       # the forward share of the window's work
        result = {
            'metric': ('1080p 3-frame windows/sec (%s) %s' % (mode, base)) if (H, W) == (FULL_H, FULL_W)
                      else '%dx%d 3-frame windows/sec (%s) %s' % (H, W, mode, base),
            'value': round(win_per_s, 4), 'unit': 'windows/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': L.DTYPE_NAME, 'data': 'synthetic',
            'config': {'precision_notes': ('bf16 MFMA operands, stored activations and gradients everywhere EXCEPT the forward of the encoder stem + layer1 + layer2 '
                                          '(18 of 201 conv layers, 9 % of the forward FLOPs): there IEEE fp16 packed weights, IEEE fp16 stored activations '
                                          '(written next to the bf16 copy the backward reads) and IEEE fp16 conv outputs on v_mfma_f32_32x32x16_f16 -- same '
                                          'bytes, same MFMA rate (DESIGN.md section 6); attention probabilities P and softmax-backward T in bf16; TAM p / ds '
                                          'as bf16 pairs; fp32 accumulation, statistics, softmax, scores, losses, master weights, Adam') if (L.DTYPE_NAME == 'bf16' and args.config == 'gca' and os.environ.get('TCVOM_NO_F16_ISLAND') is None) else
                                         ('%s storage, fp32 accumulation / statistics / softmax / losses / master weights' % L.DTYPE_NAME),
                       'workload': ('GCA+TAM (vmn_gca) forward only (train-mode statistics, no_grad; losses computed), one 3-frame %dx%d window (B=1 clip) '
                                    'per step, agg_window 7, dilate_kernel 12, formula-initialised weights; %s storage' % (H, W, L.DTYPE_NAME)) if args.forward_only else
                                   ('GCA+TAM (vmn_gca) fwd+bwd+grad-allreduce+Adam, L_alpha+0.5L_dt+0.25L_att, one 3-frame '
                                    '%dx%d window (B=1 clip) per GPU per step, agg_window 7, dilate_kernel 12, '
                                    'formula-initialised weights, train mode; %s storage' % (H, W, L.DTYPE_NAME)) if args.config == 'gca' else
                                   ('IndexNet+TAM (vmn_index: MobileNetV2 encoder with learned index blocks, ASPP, indexed up-sampling decoder) '
                                    'fwd+bwd+grad-allreduce+Adam, L_alpha+L_comp+L_grad+0.5L_dt+0.25L_att, two 3-frame %dx%d windows per GPU per '
                                    'step (train-mode BatchNorm over the batch in the ASPP), formula-initialised weights' % (H, W)) if args.config == 'index' else
                                   ('FBA+TAM (vmn_fba: ResNet-50 GN+WS dilated os8, PPM, 7-channel head, 11-channel input with the '
                                    '2-scale trimap channels) fwd+bwd+grad-allreduce+Adam, L_alpha_comp+L_lap+L_grad+0.5L_dt+0.25L_att, '
                                    'one 3-frame %dx%d window per GPU per step, formula-initialised weights, train mode; %s storage '
                                    '(BASELINE config 5 names fp16)' % (H, W, L.DTYPE_NAME)),
                       'global_batch_clips': world * clips, 'frames': 3, 'height': H, 'width': W, 'parallelism': 'dp%d' % world, 'sync_bn': bool(sync_bn),
                       'forward_only': bool(args.forward_only)},
            'final_loss': round(final_loss, 6),
            'dist': {'backend': backend if world > 1 else None, 'world_size': dist.get_world_size() if world > 1 else 1,
                     'sync_bn_transport': sync_transport,            # 'mailbox': in-kernel peer exchange; 'allreduce': one collective per BN call
                     'sync_bn_exchanges_per_step': (sync_n1 - sync_n0) // args.steps if sync_n1 is not None else None,
                     'grad_spans_overlapped_with_backward': averager.early_spans,
                     'grad_allreduce_plan': averager.last_plan,
                     'grads_unused_on_every_rank': averager.globally_unused,
                     'visible_devices': torch.cuda.device_count(),
                     'per_rank': diags, **(no_sync or {})},
            'window_mfma_frac': round(gflop * win_per_s / world / 1e3 / MFMA_PEAK_TFLOPS, 5) if gflop is not None else None,
        }
    # ---- roofline of the dominant kernel (event-instrumented extra step on rank 0's stream)
    if not args.no_profile and not args.forward_only:
        agg = igemm_profile(step)
        if rank == 0 and agg:
            dom = max(agg, key=lambda k: agg[k][1])
            n, ms, gf = agg[dom]
            tf = gf / ms                                    # GFLOP / ms == TFLOP/s
            traffic, tsrc = profile_traffic(dom, n)
            result['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(tf, 2), 'peak': MFMA_PEAK_TFLOPS,
                                  'unit': 'TFLOP/s', 'frac': round(tf / MFMA_PEAK_TFLOPS, 4),
                                  'traffic': (round(traffic) if traffic is not None and (H, W) == (FULL_H, FULL_W) else None),
                                  'traffic_source': tsrc,
                                  'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4),
                                  'algorithmic_gflop_per_launch': round(gf / n, 3),
                                  'all_igemm': {k: {'launches': v[0], 'ms': round(v[1], 3), 'tflops': round(v[2] / max(v[1], 1e-9), 1)}
                                                for k, v in sorted(agg.items())},
                                  # every conv / GEMM launch of the step (MFMA kernels of all families) and the implicit-GEMM
                                  # family alone (what is left on igemm_nt / igemm_tt: small-K, strided, 1x1, high-precision layers)
                                  # the Temporal Attention Module against the HBM roofline (algorithmic bytes / event time)
                                  'tam': {k: {'launches': v[0], 'ms': round(v[1], 4), 'GBps': round(v[2] / max(v[1], 1e-9) / 1e6, 1),
                                              'frac_of_8TBps': round(v[2] / max(v[1], 1e-9) / 1e6 / 8000.0, 4)}
                                          for k, v in sorted(getattr(igemm_profile, 'tam', {}).items())},
                                  'aggregate': {name: {'ms': round(sum(v[1] for k, v in agg.items() if sel(k)), 3),
                                                       'tflops': round(sum(v[2] for k, v in agg.items() if sel(k)) /
                                                                       max(sum(v[1] for k, v in agg.items() if sel(k)), 1e-9), 1)}
                                                for name, sel in (('all_conv_gemm', lambda k: True),
                                                                  ('igemm_nt_tt', lambda k: k.startswith('igemm_')))}}
            algo = getattr(igemm_profile, 'algo', {})
            result['roofline']['algorithmic_mib_per_launch'] = {k: round(v[1] / max(v[0], 1) / 2 ** 20, 2) for k, v in sorted(algo.items())}
            # every family against BOTH roofs: the thin-channel layers (halo / sconv / pwconv / igemm_tt<32,128>) are bound by their
            # operand bytes, not by the matrix pipe -- `hbm_frac` = algorithmic bytes / event time / 8 TB/s
            for k, e in result['roofline']['all_igemm'].items():
                if k in algo and e['ms'] > 0:
                    e['hbm_frac'] = round(algo[k][1] / (e['ms'] * 1e-3) / 8e12, 3)
            if args.config == 'gca':
                result['roofline']['tam_all_unknown'] = tam_all_unknown(device, H // 8, W // 8)
    if rank == 0:
        # the window against the HBM roofline (17.8 GB of ideal-fusion traffic per 1080p window) and the counter
        # traffic of the newest committed PMC profile (FETCH_SIZE x2 + WRITE_SIZE over the traced process, per step)
        if args.config == 'gca' and (H, W) == (FULL_H, FULL_W):
            gib, src = profile_step_traffic()
            result['hbm'] = {'algorithmic_gb_per_window': ALGO_GB_WINDOW_1080P, 'peak_tbps': 8.0,
                             'achieved_tbps': round(ALGO_GB_WINDOW_1080P * win_per_s / world / 1e3, 4),
                             'frac': round(ALGO_GB_WINDOW_1080P * win_per_s / world / 1e3 / 8.0, 4),
                             'counter_gib_per_step': gib, 'counter_source': src}
    if world > 1:
        dist.barrier()
    if (rank == 0 and world == 1 and not args.no_other_configs and not args.forward_only and args.config == 'gca'
            and (H, W) == (FULL_H, FULL_W)):
        del model, opt, averager, params
        torch.cuda.empty_cache()
        result['other_configs'] = other_configs(args.steps, args.warmup)
        dense = result['other_configs'].get('config3_dense_operands_bf16', {}).get('roofline')
        if dense and 'roofline' in result and dense.get('kernel') == result['roofline'].get('kernel'):
            result['roofline']['frac_dense_operands'] = dense['frac']
            result['roofline']['frac_dense_operands_note'] = ('the same kernel family in a 5-step side run with TCVOM_NO_P_FLUSH=1: attention probabilities '
                                                              'below 2^-25 kept as denormal-scale values instead of exact zeros (no work is skipped either way)')
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if args.config != 'index':                      # (no CPU line for the extra IndexNet configuration)
                result['cpu_baseline'] = cpu_baseline((H, W)) if args.config == 'gca' else cpu_baseline_fba()
            keep = getattr(cpu_baseline, 'oracle_outputs', None)
            if args.config == 'gca' and keep and 'alpha' in keep:
                # BASELINE.json's metric carries "alpha MSE vs ref": the oracle pass that was just timed is also the CHECKER of this window --
                # a fresh formula-initialised network (the timed one has taken K optimizer steps), same window, train-mode forward, against
                # the oracle's matte; MSE over the unknown region as calc_metric.py:25 defines it, dtSSD-style delta as tests/test_gpu_window.py
                model_p, a_p, fg_p, bg_p = build(device, H, W, seed=0, config='gca')
                with torch.no_grad():
                    out_p = model_p(a_p, fg_p, bg_p)
                al, rl = out_p[7].float().cpu(), keep['alpha']
                um = keep['tris_vis'].isclose(torch.tensor(128.0 / 255.0))
                dd = al - rl
                d_a, d_r = al[:, 1] - al[:, 1].roll(1, -1), rl[:, 1] - rl[:, 1].roll(1, -1)
                result['parity'] = {'checker': 'oracle (pinned by outputs of the reference, tests/golden), the same pass the cpu_baseline timed',
                                    'window': '3x%dx%d, formula weights, train-mode forward' % (H, W),
                                    'alpha_mse_unknown': float((dd[um] ** 2).mean()), 'alpha_mse': float((dd ** 2).mean()),
                                    'dtssd_style_delta': float(torch.sqrt(((d_a - d_r)[um[:, 1]] ** 2).mean())),
                                    'max_abs_diff': float(dd.abs().max()), 'unknown_pixels': int(um.sum()), 'bound_alpha_mse': 1e-4}
                del model_p, out_p
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
