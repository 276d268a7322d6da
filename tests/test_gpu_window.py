"""GPU parity tests, whole window: FullModel_VMD('vmn_gca') on the HIP path vs vectors captured from the
real reference (tests/golden/window_*.npz) and vs the CPU oracle on the same formula weights.

Tolerance (BASELINE.json north_star): alpha-matte MSE vs the reference <= 1e-4; dtSSD-style delta reported.
The HIP path stores activations in bf16 (fp32 accumulation/statistics); losses are compared at 3 % relative.
"""
import numpy as np
import pytest
import torch

from helpers import golden, WINDOW_CASES, FULL_GRADS, assert_close, tol
from tcvom_amd.synthetic import formula_tensor, synthetic_window

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _model(win, dil):
    from models.model import FullModel_VMD
    m = FullModel_VMD('vmn_gca', agg_window=win, dilate_kernel=dil)
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
    return m.to(DEV)


@pytest.mark.parametrize('name', list(WINDOW_CASES))
def test_window_vs_reference_golden(name):
    from tcvom_amd.facade import train_step_loss
    B, S, H, W, dil, win = WINDOW_CASES[name]
    g = golden(name)
    m = _model(win, dil).train()
    a, fg, bg = (t.to(DEV) for t in synthetic_window(B, S, H, W, seed=0))
    out = m(a, fg, bg)
    loss = train_step_loss(out)
    loss.backward()
    torch.cuda.synchronize()
    alphas = out[7].float().cpu().numpy()
    ref = g['alphas']
    mse = float(np.mean((alphas - ref) ** 2))
    unk = out[6].cpu().numpy()  # tris_vis == 128/255 at unknown pixels
    um = np.isclose(unk, 128.0 / 255.0)
    mse_unk = float(np.mean((alphas[um] - ref[um]) ** 2)) if um.any() else 0.0
    print('%s: alpha MSE %.3e (unknown-only %.3e), losses %s vs %s' % (
        name, mse, mse_unk, [float(x) for x in out[:5]], g['losses'].tolist()))
    # The 64-pixel-high goldens have 8..24 elements per channel in the os32 BatchNorms: bf16 storage noise (2^-9 per layer)
    # is amplified by the ill-conditioned statistics -- an all-bf16-storage pipeline emulated in the fp32 oracle reaches only
    # 3e-4 on the unknown pixels there (tests/test_bf16_noise_floor.py) -- so those two are bounded at 1e-3.  From 128 x 160
    # upwards the north-star bound holds against the REFERENCE golden: whole-frame MSE <= 1e-4 (measured 4.3e-5; unknown-only
    # 9.6e-5 against an emulated all-bf16 floor of 1.9e-4, asserted at 2e-4); test_window_north_star_parity asserts the
    # unknown-only bound at 544 x 960 and at the benchmark size.
    # fp16 storage (the default build): 3.8e-6 / 7.9e-6 / 1.9e-6 whole frame, 6.7e-6 / 1.9e-5 / 4.3e-6 on the unknown pixels for the
    # three goldens -- the north-star 1e-4 holds on all of them, 64-pixel windows included, unknown region included.
    # Round 6, bf16 build with the fp16 island (encoder stem + layer1 + layer2 forward in IEEE fp16, ops.F16_ISLAND): measured whole frame /
    # unknown-only 1.2e-5 / 2.1e-5 (64x64), 2.9e-5 / 6.9e-5 (s5 64x96), 2.0e-5 / 4.6e-5 (w5 64x96), 7.4e-6 / 1.6e-5 (128x160): the north-star
    # 1e-4 now holds in bf16 against every REFERENCE-held golden, unknown region included (round 5: 1e-3 / 2e-4 bounds).
    if H * W >= 128 * 160:
        assert mse <= tol(2e-5, 2e-5) and mse_unk <= tol(4e-5, 2e-5), 'alpha MSE vs reference'
    else:
        assert mse <= tol(6e-5, 4e-5) and mse_unk <= tol(1e-4, 1e-4), 'alpha MSE vs reference'
    assert_close(torch.stack([o.detach().float().cpu() for o in out[:5]]), g['losses'], 3e-2, 1e-3, 'losses')
    assert_close(out[8].sum().cpu(), g['comps_sum'], 1e-2, 1.0, 'comps')
    assert_close(out[6].sum().cpu(), g['tris_vis_sum'], 1e-5, 1e-2, 'tris_vis')
    # state updates: running stats and power-iteration vectors
    sd = m.NET.state_dict()
    for k in ('encoder.bn1.running_mean', 'encoder.bn1.running_var', 'encoder.conv1.module.weight_u',
              'decoder.layer1.0.conv1.module.weight_v', 'encoder.bn1.num_batches_tracked'):
        assert_close(sd[k].float().cpu(), g['state:' + k].astype(np.float32), 2e-2, 2e-3, k)
    # gradients: every trainable tensor gets one; norms agree with the reference within bf16 noise on the
    # well-conditioned (large-gradient) tensors
    params = dict(m.NET.named_parameters())
    names = [str(n) for n in g['grad_names']]
    missing = [k for k in names if params[k].grad is None]
    assert not missing, 'no gradient for %s' % missing[:5]
    mine = np.array([float(params[k].grad.double().norm()) for k in names])
    refn = g['grad_norms']
    big = refn > 0.05 * refn.max()
    ratio = mine[big] / refn[big]
    print('grad-norm ratio (top tensors): min %.3f median %.3f max %.3f' % (ratio.min(), float(np.median(ratio)), ratio.max()))
    if H * W >= 128 * 160:      # the 64-pixel-high cases have 8..24-element BatchNorms: backward is ill-conditioned
        # bf16: two identical runs of this window differ by the fp32 order of the atomic partial sums only, yet the largest
        # per-tensor ratio moves between 1.19 and 1.44 (10 runs): the backward map amplifies the bf16 storage noise (DESIGN.md
        # section 6).  fp16: 0.91 .. 1.03.
        # round 6 (fp16 island), bf16: min 0.938 median 0.996 max 1.092
        assert abs(np.median(ratio) - 1) < tol(0.08, 0.05) and tol(0.8, 0.8) < ratio.min() and ratio.max() < tol(1.3, 1.2), 'gradient norms'
    else:
        # 64-pixel windows, 65 runs of window_s5_64x96 (tools/probes/ratio_probe.sh; the same spread with the round-4 kernels switched
        # off): min 0.95 .. 1.00, median ~1.2, max 1.22 .. 1.57 -- the order of the fp32 atomics alone moves the largest per-tensor ratio
        # that much through the 8..24-element BatchNorm backward; the earlier bound of 1.5 sat inside that tail (1 run in ~20 failed)
        # (bf16 with the fp16 island, one run each: 0.747 / 0.994 / 1.089, 0.927 / 1.033 / 1.103, 0.915 / 0.979 / 1.175)
        assert abs(np.median(ratio) - 1) < 0.35 and tol(0.55, 0.75) < ratio.min() and ratio.max() < 2.0, 'gradient norms'


def test_window_large_vs_oracle():
    """A better conditioned case (256x320, B=1) against the CPU oracle: alpha MSE and dtSSD-style delta."""
    import oracle
    from oracle.state_spec import vmn_gca_state_spec
    from tcvom_amd.facade import train_step_loss
    B, S, H, W, dil, win = 1, 3, 256, 320, 12, 7
    m = _model(win, dil).train()
    a, fg, bg = synthetic_window(B, S, H, W, seed=0)
    out = m(a.to(DEV), fg.to(DEV), bg.to(DEV))
    train_step_loss(out).backward()
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
             for k, s in vmn_gca_state_spec().items()}
    with torch.no_grad():
        ro, _ = oracle.window_forward(state, a, fg, bg, window=win, dilate_kernel=dil, training=True)
    al, rl = out[7].float().cpu(), ro[7]
    mse = float(((al - rl) ** 2).mean())
    um = ro[6].isclose(torch.tensor(128.0 / 255.0))
    mse_unk = float(((al - rl)[um] ** 2).mean())
    d_a, d_r = al[:, 1] - al[:, 1].roll(1, -1), rl[:, 1] - rl[:, 1].roll(1, -1)
    dtssd_delta = float(torch.sqrt(((d_a - d_r)[um[:, 1]] ** 2).mean()))
    print('256x320: alpha MSE %.3e (unknown-only %.3e), spatial-gradient RMS delta %.3e ; losses %s vs %s' % (
        mse, mse_unk, dtssd_delta, [float(x) for x in out[:5]], [float(x) for x in ro[:5]]))
    # north star: alpha MSE <= 1e-4 vs the reference path, over the unknown region as calc_metric.py:25 defines it
    # (and a fortiori over the whole frame)
    # bf16 with the fp16 island (round 6): 2.7e-6 / 1.21e-5, delta 9.0e-3 (round 5: 9.1 .. 9.7e-5 on the unknown pixels, 5 % under the
    # bound); the asserted ceilings are 2x the measured values
    assert mse <= tol(6e-6, 1e-5) and mse_unk <= tol(2.5e-5, 2e-5)        # fp16: 8.5e-7 / 3.8e-6
    assert dtssd_delta <= 1.8e-2
    assert_close(torch.stack([o.detach().float().cpu() for o in out[:5]]), torch.stack(list(ro[:5])), tol(1e-2, 3e-3), 1e-3, 'losses')


def _oracle_state(requires_grad=False):
    from oracle.state_spec import vmn_gca_state_spec
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
             for k, s in vmn_gca_state_spec().items()}
    if requires_grad:
        for k, v in state.items():
            if v.is_floating_point() and not any(t in k for t in ('weight_u', 'weight_v', 'running_')):
                v.requires_grad_(True)
    return state


@pytest.mark.parametrize('H,W', [(512, 512), (544, 960), (1088, 1920)])       # 512 x 512 = BASELINE config 2 (forward, 3 frames)
def test_window_north_star_parity(H, W):
    """BASELINE.json north star at the benchmark geometry (config 3: one 3 x 1088 x 1920 window, train mode, dilate_kernel 12)
    and at half of it: forward + losses of the HIP path against the fp32 CPU oracle on identical inputs and formula
    weights.  Bound: alpha MSE over the unknown region (calc_metric.py:25) <= 1e-4; the dtSSD-style delta
    (calc_metric.py:31-34 applied to neighbouring columns, the window has one predicted frame) is reported."""
    import os
    import time
    import oracle
    B, S, dil, win = 1, 3, 12, 7
    m = _model(win, dil).train()
    a, fg, bg = synthetic_window(B, S, H, W, seed=0)
    with torch.no_grad():
        out = m(a.to(DEV), fg.to(DEV), bg.to(DEV))
    torch.cuda.synchronize()
    al = out[7].float().cpu()
    losses = [float(x) for x in out[:5]]
    del out, m
    torch.cuda.empty_cache()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    with torch.no_grad():
        ro, _ = oracle.window_forward(_oracle_state(), a, fg, bg, window=win, dilate_kernel=dil, training=True)
    rl = ro[7]
    um = ro[6].isclose(torch.tensor(128.0 / 255.0))
    d = al - rl
    mse, mse_unk = float((d ** 2).mean()), float((d[um] ** 2).mean())
    d_a, d_r = al[:, 1] - al[:, 1].roll(1, -1), rl[:, 1] - rl[:, 1].roll(1, -1)
    dtssd_delta = float(torch.sqrt(((d_a - d_r)[um[:, 1]] ** 2).mean()))
    print('%dx%d: alpha MSE %.3e (unknown-only %.3e, %d unknown pixels), dtSSD-style delta %.3e, max |d| %.3e; losses %s vs %s; oracle %.0f s'
          % (H, W, mse, mse_unk, int(um.sum()), dtssd_delta, float(d.abs().max()), losses, [float(x) for x in ro[:5]], time.time() - t0))
    # bf16 with the fp16 island (round 6) at 512^2 / 544x960 / 1088x1920: unknown-only 1.04e-5 / 9.98e-6 / 9.87e-6 (round 5: 7.7 / 7.1 / 6.75e-5),
    # whole frame 1.5e-6 / 7.6e-7 / 3.7e-7, dtSSD-style delta 8.3 / 8.2 / 8.1e-3 (2.0e-2), max |d| 0.042 / 0.046 / 0.041 (0.12); ceilings = 2x
    assert mse <= tol(4e-6, 1e-5) and mse_unk <= tol(2.5e-5, 2e-5)        # fp16 at 512^2 / 544x960 / 1088x1920: unknown-only 3.2e-6 / 2.7e-6 / 2.6e-6
    assert dtssd_delta <= 1.7e-2 and float(d.abs().max()) <= 0.09
    assert_close(torch.tensor(losses), torch.stack(list(ro[:5])), tol(1e-2, 3e-3), 1e-3, 'losses')


def test_window_full_size_backward_parity():
    """The benchmarked step itself -- BASELINE config 3: GCA+TAM forward + BACKWARD of one 3 x 1088 x 1920 window, train mode --
    against the fp32 CPU oracle's backward on identical inputs and formula weights (the oracle needs ~25 GB and a minute on 32
    threads).  At this size run the paths no smaller test reaches: the ld % 256 fused-transpose branch of the attention backward,
    the 802 MB score matrix, 32-bit offset limits, the <= 112-problem accumulator-stationary weight-gradient launches.
    Asserted: every trainable parameter gets a finite, non-zero gradient; the gradient norm of the whole network and of every
    module group agrees with the oracle's (bounds from the measured spread, printed); two identical HIP runs agree (they differ
    by the order of fp32 atomic partial sums only)."""
    import os
    import time
    import oracle
    from tcvom_amd.facade import train_step_loss
    H, W = 1088, 1920
    a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
    ad, fd, bd = a.to(DEV), fg.to(DEV), bg.to(DEV)

    def hip_run():
        m = _model(7, 12).train()
        out = m(ad, fd, bd)
        loss = train_step_loss(out)
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.double().cpu() for k, p in m.NET.named_parameters() if p.grad is not None}
        need = [k for k, p in m.NET.named_parameters() if p.requires_grad]
        del out, loss, m
        torch.cuda.empty_cache()
        return g, need
    g1, need = hip_run()
    g2, _ = hip_run()
    missing = [k for k in need if k not in g1]
    assert not missing, 'no gradient for %s' % missing[:5]
    bad = [k for k in need if not bool(torch.isfinite(g1[k]).all()) or float(g1[k].norm()) == 0.0]
    assert not bad, 'zero or non-finite gradient for %s' % bad[:5]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    state = _oracle_state(requires_grad=True)
    out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True)
    oracle.train_step_loss(out).backward()
    go = {k: v.grad.double() for k, v in state.items() if getattr(v, 'grad', None) is not None}
    t_oracle = time.time() - t0
    del out
    norm = lambda gs, ks: float(torch.sqrt(sum((gs[k] ** 2).sum() for k in ks)))
    cos = lambda x, y: float((x * y).sum() / (x.norm() * y.norm() + 1e-300))
    keys = [k for k in need if k in go and float(go[k].norm()) > 0]
    assert len(keys) >= 0.95 * len(need)
    groups = {}
    for k in keys:
        groups.setdefault('.'.join(k.split('.')[:2]), []).append(k)
    total = norm(g1, keys) / norm(go, keys)
    rerun = norm(g2, keys) / norm(g1, keys)
    cat = lambda gs: torch.cat([gs[k].flatten() for k in keys])
    c_oracle, c_rerun = cos(cat(g1), cat(go)), cos(cat(g1), cat(g2))
    rows = [(top, norm(g1, ks) / norm(go, ks), norm(g2, ks) / norm(g1, ks), norm(go, ks), len(ks)) for top, ks in sorted(groups.items())]
    print('1088x1920 backward: total gradient norm HIP / oracle %.3f (rerun / run %.3f), cosine vs oracle %.3f, run vs rerun %.3f; oracle %.0f s'
          % (total, rerun, c_oracle, c_rerun, t_oracle))
    print('\n'.join('%-30s norm ratio %.3f  rerun %.3f  oracle norm %.3e  (%d tensors)' % r for r in rows))
    # fp16 (measured): total 1.001, cosine of the whole-network gradient against the oracle 0.962 (two identical runs: 0.986),
    # every module group 0.977 .. 1.038
    # bf16 with the fp16 island (round 6): total 1.001, cosine against the oracle 0.936 (two identical runs: 0.946 -- the distance to the
    # oracle IS the run-to-run distance of the atomics order, amplified by the backward map), every module group 0.973 .. 1.054
    lo, hi = tol((0.95, 1.05), (0.95, 1.05))
    assert lo <= total <= hi, 'whole-network gradient norm vs the oracle'
    assert 0.9 <= rerun <= 1.1, 'two identical runs'
    assert c_oracle >= tol(0.85, 0.9), 'whole-network gradient direction vs the oracle'
    top = max(r[3] for r in rows)
    glo, ghi = tol((0.9, 1.12), (0.9, 1.12))
    for name, ratio, rr, on, n in rows:
        if on >= 0.02 * top:                      # groups that carry the gradient; tiny groups are dominated by amplified noise
            assert glo <= ratio <= ghi, 'gradient norm of %s: %.3f of the oracle' % (name, ratio)


def test_gradient_fidelity_vs_oracle():
    """Backward parity of the whole window beyond gradient NORMS: cosine similarity of every parameter gradient of the HIP path
    (bf16 activations) against the fp32 CPU oracle at 256 x 320, norm-weighted per module group.  The backward map of this
    60-layer train-mode-BatchNorm net with random weights amplifies last-bit noise (two identical HIP runs differ by the fp32
    order of their atomic partial sums only and already reach just 0.76..1.0 against each other), so the
    bounds are set from that measured floor: decoder groups >= 0.8 (measured 0.89..1.0), encoder groups >= 0.6 (0.67..0.95), the
    concatenated gradient of the whole network >= 0.7 (0.79); op-level gradients are pinned tightly in test_gpu_ops.py."""
    import os
    import oracle
    from tcvom_amd.facade import train_step_loss
    H, W = 256, 320
    m = _model(7, 12).train()
    a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
    train_step_loss(m(a.to(DEV), fg.to(DEV), bg.to(DEV))).backward()
    torch.cuda.synchronize()
    g = {k: p.grad.double().cpu() for k, p in m.NET.named_parameters() if p.grad is not None}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    state = _oracle_state(requires_grad=True)
    out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True)
    oracle.train_step_loss(out).backward()
    go = {k: v.grad.double() for k, v in state.items() if getattr(v, 'grad', None) is not None}
    cos = lambda x, y: float((x * y).sum() / (x.norm() * y.norm() + 1e-300))
    groups = {}
    for k in g:
        if k in go and float(go[k].norm()) > 0:
            top = '.'.join(k.split('.')[:2])
            # the stem's three convs and BatchNorms are ONE group: as groups of 1 - 2 tensors (bn1: weight and bias) their cosines
            # scatter 0.66 .. 0.80 from run to run around the same mean (round 5: 0.657 / 0.659 / 0.695 / 0.705 for bn1 in four runs
            # of one build) and the smallest of six noisy samples met the 0.6 bound only ~7 runs in 8
            if top in ('encoder.conv1', 'encoder.conv2', 'encoder.conv3', 'encoder.bn1', 'encoder.bn2', 'encoder.bn3'):
                top = 'encoder.stem'
            groups.setdefault(top, []).append(k)
    rows = []
    for top, ks in sorted(groups.items()):
        w = [float(go[k].norm()) for k in ks]
        c = [cos(g[k], go[k]) for k in ks]
        rows.append((top, sum(ci * wi for ci, wi in zip(c, w)) / sum(w), min(c), len(ks)))
    print('\n'.join('%-30s weighted cos %.3f  min %.3f  (%d tensors)' % r for r in rows))
    dec = [r[1] for r in rows if r[0].startswith('decoder.')]
    enc = [r[1] for r in rows if r[0].startswith('encoder.')]
    # fp16 storage: decoder groups 0.976 .. 1.0, encoder groups 0.944 .. 0.989 -- most of what the bf16 build loses against the
    # oracle is storage noise amplified by the backward map, not summation order
    # bf16 with the fp16 island (round 6): decoder groups 0.955 .. 1.0, encoder groups 0.919 .. 0.981 (round 5: 0.89.. / 0.67..)
    assert min(dec) >= tol(0.9, 0.95), 'decoder gradient direction'
    assert min(enc) >= tol(0.85, 0.9), 'encoder gradient direction'
    tot_h = torch.cat([g[k].flatten() for ks in groups.values() for k in ks])
    tot_o = torch.cat([go[k].flatten() for ks in groups.values() for k in ks])
    assert cos(tot_h, tot_o) >= tol(0.88, 0.9), 'whole-network gradient direction'       # measured 0.94 (bf16, round 6; 0.79 in round 5)


def _conditioned_gradient(regime, H, W):
    """Two FusedAdam steps of the HIP path from the formula state (`damped_residual_gains`: bn2.weight x 0.15 first), then the THIRD step's
    gradient twice on the HIP path and once in the fp32 CPU oracle from the same state_dict: cosines, norm ratio, per-group norm ratios."""

    import os
    import oracle
    from tcvom_amd.facade import train_step_loss
    from tcvom_amd.optim import FusedAdam
    m = _model(7, 12)
    if regime == 'damped_residual_gains':
        with torch.no_grad():
            for k, p in m.NET.named_parameters():
                if k.endswith('bn2.weight'):
                    p.mul_(0.15)
    m = m.train()
    a, fg, bg = synthetic_window(1, 3, H, W, seed=0)
    ad, fd, bd = a.to(DEV), fg.to(DEV), bg.to(DEV)
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        train_step_loss(m(ad, fd, bd)).backward()
        opt.step()
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in m.NET.state_dict().items()}

    def hip_grad():
        m.NET.load_state_dict(sd)                      # the third step from the SAME state (a forward moves u / v / running stats)
        m.zero_grad(set_to_none=True)
        train_step_loss(m(ad, fd, bd)).backward()
        torch.cuda.synchronize()
        return {k: p.grad.double().cpu() for k, p in m.NET.named_parameters() if p.grad is not None}
    g1, g2 = hip_grad(), hip_grad()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    state = {k: v.detach().cpu().clone() for k, v in sd.items()}
    for k, v in state.items():
        if v.is_floating_point() and not any(t in k for t in ('weight_u', 'weight_v', 'running_')):
            v.requires_grad_(True)
    out, _ = oracle.window_forward(state, a, fg, bg, window=7, dilate_kernel=12, training=True)
    oracle.train_step_loss(out).backward()
    go = {k: v.grad.double() for k, v in state.items() if getattr(v, 'grad', None) is not None}
    keys = [k for k in g1 if k in go and float(go[k].norm()) > 0]
    assert len(keys) >= 0.95 * len(g1)
    cat = lambda gs: torch.cat([gs[k].flatten() for k in keys])
    cos = lambda x, y: float((x * y).sum() / (x.norm() * y.norm() + 1e-300))
    c_oracle, c_rerun = cos(cat(g1), cat(go)), cos(cat(g1), cat(g2))
    nr = float(cat(g1).norm() / cat(go).norm())
    enc = [k for k in keys if k.startswith('encoder.')]
    dec = [k for k in keys if k.startswith('decoder.')]
    sub = lambda gs, ks: torch.cat([gs[k].flatten() for k in ks])
    print('%s, two Adam steps in, %dx%d: whole-network gradient cosine HIP vs oracle %.4f (encoder %.4f, decoder %.4f), run vs rerun %.4f, '
          'norm ratio %.4f' % (regime, H, W, c_oracle, cos(sub(g1, enc), sub(go, enc)), cos(sub(g1, dec), sub(go, dec)), c_rerun, nr))
    groups = {}
    for k in keys:
        groups.setdefault('.'.join(k.split('.')[:2]), []).append(k)
    gn = lambda gs, ks: float(torch.sqrt(sum((gs[k] ** 2).sum() for k in ks)))
    rows = {top: (gn(g1, ks) / gn(go, ks), gn(go, ks)) for top, ks in sorted(groups.items())}
    return c_oracle, c_rerun, nr, rows


@pytest.mark.parametrize('regime', ['formula_gains', 'damped_residual_gains'])
def test_gradient_noise_on_a_conditioned_network(regime):
    """Run-to-run and HIP-vs-oracle agreement of the WHOLE-network gradient on a network that is two optimizer steps into
    training, at 256 x 320.  The only run-to-run difference of the HIP path is the order of fp32
    atomic partial sums (igemm_tt / wgrad_ws / SpectralNorm sums); how far that moves the gradient depends on how strongly the
    backward map of the 60-layer train-mode-BatchNorm stack amplifies last-bit differences, so two regimes are bounded:
      * formula_gains: the formula weights as they are -- every BatchNorm scale 0.6 +- 0.25, INCLUDING every BasicBlock's bn2
        (the reference zero-initialises those, resnet_enc.py:96-98; SURVEY App. A asks for O(1) values in parity tests) -- a
        random network whose residual branches carry as much signal as the identity paths: the ill-conditioned end;
      * damped_residual_gains: bn2.weight x 0.15 (what zero-init-residual training leaves early on): the identity paths dominate
        and the amplification argument no longer applies -- here the kernels themselves are what is measured.
    Both start from two FusedAdam steps of the HIP path (running statistics, power-iteration vectors and weights moved off the
    formula state); the resulting state_dict is loaded into the fp32 CPU oracle, and the third step's gradient is compared."""
    c_oracle, c_rerun, nr, _rows = _conditioned_gradient(regime, 256, 320)
    if regime == 'damped_residual_gains':
        # measured (MI355X, round 5): bf16 0.9988 vs the oracle / 0.9993 run vs rerun, norm ratio 0.9997; fp16 0.9998 / 0.9999 / 0.9996
        # round 6 (fp16 island), bf16: 0.9993 / 0.9995, norm ratio 0.9986
        assert c_oracle >= tol(0.998, 0.999) and c_rerun >= tol(0.998, 0.999), (c_oracle, c_rerun)
        assert abs(nr - 1) <= tol(0.01, 0.01)
    else:
        # measured floor of the ill-conditioned regime: bf16 0.828 vs the oracle with 0.912 between two identical
        # runs; fp16 0.967 / 0.986 -- the rerun distance IS the amplified order of the fp32 atomics, and the distance to the oracle is
        # about twice it (storage rounding amplified the same way)
        # round 6 (fp16 island), bf16: 0.9415 vs the oracle, 0.9458 between two identical runs, norm ratio 0.992
        assert c_oracle >= tol(0.88, 0.93) and c_rerun >= tol(0.88, 0.96), (c_oracle, c_rerun)
        assert abs(nr - 1) <= tol(0.04, 0.03)



def test_full_size_gradient_direction_on_a_conditioned_network():
    """the kernels that ONLY run at the benchmark geometry (the ld % 256 fused-transpose branch of the attention
    backward, the 112-problem accumulator-stationary weight-gradient launches, the heterogeneous geometry tables, 32-bit offset limits) were
    pinned by gradient NORMS within +-25 % and a cosine >= 0.6 on the noise-amplifying formula network.  On the damped network (bn2 x 0.15,
    two Adam steps in) the backward map does not amplify last-bit noise, so the same 3 x 1088 x 1920 window is held to the oracle tightly:
    whole-network gradient cosine >= 0.99, every module group that carries gradient within 5 % in norm."""
    c_oracle, c_rerun, nr, rows = _conditioned_gradient('damped_residual_gains', 1088, 1920)
    print('\n'.join('%-30s norm ratio %.4f  oracle norm %.3e' % (k, v[0], v[1]) for k, v in rows.items()))
    assert c_oracle >= tol(0.99, 0.995) and c_rerun >= tol(0.99, 0.995), (c_oracle, c_rerun)
    assert abs(nr - 1) <= 0.02
    top = max(v[1] for v in rows.values())
    for k, (ratio, on) in rows.items():
        if on >= 0.02 * top:
            assert 0.95 <= ratio <= 1.05, 'gradient norm of %s: %.4f of the oracle' % (k, ratio)


def test_freeze_backbone_keeps_the_feature_extractor_in_eval_mode():
    """freeze_backbone=True (pretrain_ddp.py's TAM pre-training; VMN_model.py:77-103, VMN_GCA.py:18-24): encoder and decoder
    front run in eval mode under no_grad inside a training step -- no gradient, BatchNorm running statistics and
    num_batches_tracked unchanged, SpectralNorm u / v not iterated -- while the decoder tail (TAM, layer3, layer4, conv1,
    conv2) trains; and the frozen front equals the front of the same network in eval mode."""
    from models.model import FullModel_VMD
    from tcvom_amd.facade import train_step_loss
    torch.manual_seed(0)
    a, fg, bg = (t.to(DEV) for t in synthetic_window(1, 3, 96, 128, seed=0))
    # eval-mode features of a randomly initialised net blow up (identity running statistics, un-iterated power iteration):
    # calibrate them with two train-mode passes of the unfrozen network first, as the goldens of EvalModel do
    m0 = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=5)
    m0.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m0.NET.state_dict().items()})
    m0 = m0.to(DEV).train()
    with torch.no_grad():
        m0(a, fg, bg)
        m0(a, fg, bg)
    m = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=5, freeze_backbone=True)
    m.NET.load_state_dict(m0.NET.state_dict())
    m = m.to(DEV).train()
    before = {k: v.clone() for k, v in m.NET.state_dict().items()}
    out = m(a, fg, bg)
    train_step_loss(out).backward()
    torch.cuda.synchronize()
    after = m.NET.state_dict()
    front = lambda k: k.startswith('encoder.') or k.startswith(('decoder.layer1.', 'decoder.layer2.', 'decoder.gca.'))
    for k, p in m.NET.named_parameters():
        if front(k):
            assert p.grad is None, 'frozen parameter %s has a gradient (the reference leaves it None: Adam skips it)' % k
        elif p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    changed_tail = 0
    for k in before:
        if front(k):
            assert torch.equal(before[k], after[k]), 'frozen state %s changed' % k
        elif k.endswith(('running_mean', 'weight_u')):
            changed_tail += int(not torch.equal(before[k], after[k]))
    assert changed_tail > 0, 'the decoder tail must still update its BatchNorm statistics / power iteration'
    # an optimizer step with weight decay (train_ddp.py:296-297: Adam(weight_decay=1e-4)) must not move the frozen backbone
    from tcvom_amd.optim import FusedAdam
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, weight_decay=1e-1)
    opt.step()
    torch.cuda.synchronize()
    moved = 0
    for k, p in m.NET.named_parameters():
        if front(k):
            assert torch.equal(p.detach(), before[k]), 'frozen parameter %s moved in optimizer.step()' % k
        elif p.requires_grad:
            moved += int(not torch.equal(p.detach(), before[k]))
    assert moved > 0
    m.NET.load_state_dict(before)
    m.zero_grad(set_to_none=True)
    # the same weights in eval mode produce the same features -> with the tail's BatchNorm in train mode the alpha differs, so
    # compare through a second frozen run: deterministic, and different from a fully-trainable run of the same step
    out2 = m(a, fg, bg)
    assert torch.isfinite(out2[7]).all()
    m2 = FullModel_VMD('vmn_gca', agg_window=7, dilate_kernel=5)
    m2.NET.load_state_dict(before)
    m2 = m2.to(DEV).train()
    assert not torch.allclose(m2(a, fg, bg)[7], out[7]), 'a frozen backbone must not use batch statistics'


def test_eval_mode_runs_and_is_deterministic():
    m = _model(7, 3)
    a, fg, bg = (t.to(DEV) for t in synthetic_window(1, 3, 64, 64, seed=0))
    m.train()
    with torch.no_grad():
        m(a, fg, bg)
        m(a, fg, bg)
    m.eval()
    with torch.no_grad():
        o1 = m(a, fg, bg)
        o2 = m(a, fg, bg)
    assert torch.equal(o1[7], o2[7])
    assert torch.isfinite(o1[7]).all()


def eval_inputs(B, S, H, W):
    """Same construction as tests/golden/gen_golden.py:eval_inputs."""
    a, fg, bg = synthetic_window(B, S, H, W, seed=3)
    al = a / 255.0
    imgs = torch.round(fg * al + bg * (1 - al))
    tris = torch.where(a <= 0, torch.zeros_like(a), torch.where(a >= 255, torch.full_like(a, 255.0), torch.full_like(a, 128.0)))
    return imgs, tris


@pytest.mark.parametrize('name,shape', [('eval_s3_64x96', (1, 3, 64, 96, 3, 7)), ('eval_s3_128x160', (1, 3, 128, 160, None, 7))])
def test_eval_model_vs_reference_golden(name, shape):
    """EvalModel (frames + user trimaps, eval-mode BatchNorm / SpectralNorm) against the reference's EvalModel.  As in
    the generator, two train-mode calibration windows first give the running statistics something meaningful."""
    from models.model import EvalModel
    B, S, H, W, dil, win = shape
    g = golden(name)
    fm = _model(win, 12).train()
    with torch.no_grad():
        for _ in range(2):
            fm(*(t.to(DEV) for t in synthetic_window(B, S, H, W, seed=0)))
    em = EvalModel('vmn_gca', agg_window=win, dilate_kernel=dil)
    em.NET.load_state_dict(fm.NET.state_dict())
    em = em.to(DEV).eval()
    imgs, tris = eval_inputs(B, S, H, W)
    assert_close(imgs.sum(), g['imgs_sum'], 0, 0.5, 'imgs')
    alphas = em(imgs.to(DEV), tris.to(DEV)).cpu().numpy()
    ref = g['alphas']
    assert alphas.shape == ref.shape
    assert np.array_equal(alphas[:, 0], ref[:, 0]) and np.array_equal(alphas[:, -1], ref[:, -1])      # zeros at the clip ends
    known = (tris.numpy() != 128.0) if dil is None else None
    if known is not None:
        assert np.array_equal(alphas[:, 1][known[:, 1]], ref[:, 1][known[:, 1]])                      # trimap passes through
    mse = float(np.mean((alphas - ref) ** 2))
    print('%s: alpha MSE vs reference EvalModel %.3e' % (name, mse))
    assert mse <= 1e-6                                   # measured 2e-8 (eval mode: no batch statistics to amplify rounding)


def test_evaluation_metrics_on_device():
    """tcvom_amd.metrics.frame_metrics (one fused kernel) against the values of calc_metric.py's own functions."""
    from helpers import metric_inputs
    from tcvom_amd.metrics import frame_metrics
    g = golden('metrics')
    a, gt, tri, ha, hg, flow = (torch.from_numpy(t).to('cuda') for t in metric_inputs())
    out = frame_metrics(a, gt, tri, ha, hg, flow)
    assert out['pixels'] == int(g['pixels'])
    for k, ref in (('SAD', 'sad'), ('MSE', 'mse'), ('SSDA', 'ssda'), ('dtSSD', 'dtssd')):
        assert abs(out[k] - float(g[ref])) <= 1e-5 * abs(float(g[ref])), (k, out[k], float(g[ref]))
    fix, org, valid = out['MESSDdt']
    assert valid == int(g['messd'][2])
    assert abs(fix - g['messd'][0]) <= 1e-4 * g['messd'][0] and abs(org - g['messd'][1]) <= 1e-4 * g['messd'][1]
    only = frame_metrics(a, gt, tri)
    assert only['SAD'] == out['SAD'] and 'dtSSD' not in only


@pytest.mark.parametrize('name,arch,shape', [('gca_single_s1_128x160', 'gca', (1, 1, 128, 160, 5)), ('fba_single_s3_64x64', 'fba', (1, 3, 64, 64, 3))])
def test_single_image_bases_vs_reference_golden(name, arch, shape):
    """FullModel('gca') / FullModel('fba'): the single-image bases without the temporal module (models/model.py:199-246),
    state_dict layout, losses, centre-frame alpha and gradient norm against the reference."""
    from tcvom_amd.facade import FullModel
    from tcvom_amd.synthetic import formula_tensor
    B, S, H, W, dil = shape
    g = golden(name)
    m = FullModel(arch, dilate_kernel=dil)
    sd = m.NET.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['keys']]
    assert [','.join(str(int(d)) for d in v.shape) for v in sd.values()] == [str(s) for s in g['shapes']]
    m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in sd.items()})
    m = m.to(DEV).train()
    a, fg, bg = (t.to(DEV) for t in synthetic_window(B, S, H, W, seed=6))
    out = m(a, fg, bg)
    assert len(out) == 10
    (out[0] + out[1] + out[2]).backward()
    losses = torch.stack([o.detach().float().cpu() for o in out[:3]])
    mse = float(((out[5].float().cpu() - torch.from_numpy(g['alphas'])) ** 2).mean())
    print('%s: alpha MSE %.3e, losses %s vs %s' % (name, mse, losses.tolist(), g['losses'].tolist()))
    assert mse <= tol(1e-4 if arch == 'fba' else 1e-3, 2e-5)       # bf16, GCA at 128x160: small BatchNorms, see test_window_vs_reference_golden; fp16: 3.2e-6 / 6.5e-7
    assert_close(losses, g['losses'], 3e-2, 1e-3, 'losses')
    params = dict(m.NET.named_parameters())
    names = [str(n) for n in g['grad_names']]
    assert all(params[n].grad is not None for n in names)
    got = np.linalg.norm([float(params[n].grad.double().norm()) for n in names])
    want = np.linalg.norm(g['grad_norms'])
    assert abs(got - want) <= 0.25 * want, (got, want)


def test_baseline_facade_accepts_vmn_architectures():
    """FullModel('vmn_gca', ...) (models/model.py:199-246 with a VMN arch): the 10-item list with the single-image losses of
    the interior frames; equals the first three losses and the visualisation tensors of FullModel_VMD on the same weights."""
    from tcvom_amd.facade import FullModel, FullModel_VMD
    from tcvom_amd.synthetic import formula_tensor
    a, fg, bg = (t.to(DEV) for t in synthetic_window(1, 3, 128, 160, seed=0))
    outs = []
    for cls in (FullModel, FullModel_VMD):
        m = cls('vmn_gca', agg_window=7, dilate_kernel=12)
        m.NET.load_state_dict({k: formula_tensor(k, v.shape, v.dtype) for k, v in m.NET.state_dict().items()})
        with torch.no_grad():
            outs.append(m.to(DEV).train()(a, fg, bg))
    base, vmd = outs
    assert len(base) == 10 and len(vmd) == 12
    for i in range(3):
        assert_close(base[i].cpu(), vmd[i].cpu(), 1e-3, 1e-5, 'loss %d' % i)
    # two runs differ by the order of fp32 atomic sums (SpectralNorm sigma, statistics): compare like the golden tests do
    assert float(((base[5] - vmd[7]) ** 2).mean()) <= 1e-4
