"""The A/B switches of tcvom_amd/ops.py select alternative code paths that stay in the product as fall-backs (shapes the fast paths
do not take) -- the NT attention GEMMs on transposed copies + the fp32 score matrix + row softmax (TCVOM_NO_GCA_KMAJOR,
TCVOM_NO_FUSED_SOFTMAX), SpectralNorm's inner product as a pass over the weight gradient, zero-padded instead of row-range
gradients, the end frames of the tail-only branches run on zero gradients (TCVOM_NO_SN_DOT, TCVOM_NO_RANGED, TCVOM_NO_TAIL_SKIP), the
round-5 re-routings (TCVOM_NO_PWCONV, TCVOM_NO_WGRAD_HETERO, TCVOM_NO_WGRAD_GROUP_LAYERS).
One 544 x 960 training window through each set must give the losses and per-group gradient norms of the default paths."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(extra):
    env = dict(os.environ, **extra)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_switch_probe.py')], cwd=REPO, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + '\n' + out.stderr[-1500:]
    line = [l for l in out.stdout.splitlines() if l.startswith('PROBE ')][-1]
    return json.loads(line[6:])


# ---- every TCVOM_* environment variable the product reads, by role.  test_every_switch_is_listed fails when the sources read one that is
# neither in a probe set below (an alternative code path, exercised at 544 x 960 against the default) nor in NOT_CODE_PATHS with a reason.
SETS = {
    'attention': {'TCVOM_NO_GCA_KMAJOR': '1', 'TCVOM_NO_FUSED_SOFTMAX': '1'},
    # (k-major A off alone: the forward O = P V on a transposed copy of V while the backward keeps its k-major operands)
    'attention, V^T copy': {'TCVOM_NO_GCA_KMAJOR_A': '1'},
    'gradients': {'TCVOM_NO_SN_DOT': '1', 'TCVOM_NO_RANGED': '1', 'TCVOM_NO_TAIL_SKIP': '1', 'TCVOM_NO_RES_MASK': '1', 'TCVOM_NO_ADD3': '1'},
    # the weight-stationary conv splits a frame-batched launch into runs of frames when the frames together would reach 2^31
    # elements (fragment-major weights have no other kernel): forced here to one frame per launch
    'wsconv frame runs': {'TCVOM_WS_MAX_FRAMES': '1'},
    # every conv on the implicit GEMM: no halo-tile / weight-stationary / pointwise / strided-halo kernels, row-major weight packs
    'implicit GEMM everywhere': {'TCVOM_NO_SCONV': '1', 'TCVOM_NO_WSCONV': '1', 'TCVOM_NO_PWCONV': '1', 'TCVOM_NO_HALO_S2': '1'},
    # every weight gradient on igemm_tt: no accumulator-stationary / halo / 256-tile TT kernels, no grouped launches
    'igemm_tt weight gradients': {'TCVOM_NO_WGRADWS': '1', 'TCVOM_NO_HALO_WGRAD': '1', 'TCVOM_NO_TT256': '1', 'TCVOM_NO_WGRAD_HETERO': '1',
                                  'TCVOM_NO_WGRAD_GROUP_LAYERS': '1'},
    # round 6: the halo-form weight gradient for the 32 -> 32 layers only (the 8 / 16-channel and stride-2 stem, guidance-head and os1 shortcut
    # layers back on igemm_tt<32,128>)
    'halo weight gradient 32 -> 32 only': {'TCVOM_NO_HALO_WGRAD_THIN': '1'},
    # round 6: the implicit-GEMM TT weight gradients as one launch per layer geometry instead of one per tile shape (tcvom_wgrad_igemm_hetero)
    'igemm_tt one launch per geometry': {'TCVOM_NO_WGRAD_TT_HETERO': '1'},
    # the accumulator-stationary weight gradient without its dilated form (FBA's layers; a no-op for GCA, kept for enumeration)
    'wgrad_ws undilated only': {'TCVOM_WGRADWS_NO_DIL': '1'},
    # the 256-tile GEMM family without its special forms: no K-split tail, no 192-row tiles, no paired dq / dk launch, no statistics epilogue
    'gemm_nt256 plain forms': {'TCVOM_NO_KSPLIT': '1', 'TCVOM_NO_M192': '1', 'TCVOM_NO_GEMM_PAIR': '1', 'TCVOM_NO_G256_STATS': '1'},
    # attention probabilities below 2^-25 kept instead of flushed to exact zeros (bench.py: roofline.frac_dense_operands)
    'dense attention operands': {'TCVOM_NO_P_FLUSH': '1'},
    # the Temporal Attention Module on the one-wave-per-pixel tile kernels (what C != 128 runs), split between both, and with the
    # one-wave-per-key backward pass B
    'TAM vector kernels': {'TCVOM_TAM_DENSE': '65'}, 'TAM split': {'TCVOM_TAM_DENSE': '12'}, 'TAM key pass on VALU': {'TCVOM_TAM_KEY_VALU': '1'},
    # weight packs: one launch per power-iteration call, one thread per packed element
    'weight packs': {'TCVOM_NO_PACK_ALL': '1', 'TCVOM_SN_PACK_TILED': '0'},
}
NOT_CODE_PATHS = {
    'TCVOM_DTYPE': 'selects the library build (bf16 / fp16): tests/test_gpu_dtype_builds.py runs the suite in the other one',
    'TCVOM_HIP_LIB': 'path of the library to load', 'TCVOM_LIB': 'path of the library to load',
    'TCVOM_LOSS_SCALE': 'fp16 build: the constant loss scale (tests/test_gpu_ops.py: LossScaler tests)',
    'TCVOM_NO_OVERFLOW_GUARD': 'fp16 build: overflow guard off (LossScaler tests)',
    'TCVOM_MBOX_TIMEOUT_S': 'SyncBatchNorm mailbox timeout (tests/test_gpu_syncbn.py)',
    'TCVOM_SYNCBN': 'SyncBatchNorm transport, mailbox / rccl: both parametrised in tests/test_gpu_syncbn.py',
    'TCVOM_CONV_TRACE': 'profiling aid (cycle stamps of one wsconv workgroup)',
    'TCVOM_NO_PPM_LINK': 'FBA only (pyramid pooling link): tests/test_gpu_fba.py',
    'TCVOM_NO_NT_T192': 'FBA only (256 x 192 tiles of the K = 256 os8 3x3 layers at 1080p; the GCA window never selects them): the tile itself '
                        'is covered by test_conv_bn_large_tile_configs, the switch is the same-box A/B of bench.py --config fba',
    'TCVOM_NO_F16_ISLAND': 'plain bf16 forward everywhere: the precision A/B of the fp16 island -- test_the_fp16_island_is_what_meets_the_bound below',
}


def _switches_in_sources():
    import glob
    import re
    found = set()
    for f in glob.glob(os.path.join(REPO, 'tcvom_amd', '*.py')) + glob.glob(os.path.join(REPO, 'tcvom_amd', 'csrc', '*.hip')) + \
            glob.glob(os.path.join(REPO, 'tcvom_amd', 'csrc', '*.h')):
        src = open(f).read()
        found |= set(re.findall(r'getenv\("(TCVOM_[A-Z0-9_]+)"', src))
        found |= set(re.findall(r'environ(?:\.get\(|\[)\s*[\'"](TCVOM_[A-Z0-9_]+)[\'"]', src))
    return found


def test_every_switch_is_listed():
    """No environment switch without a test: every TCVOM_* variable the package or the kernels read is either part of a probe set (its
    alternative path is compared with the default one below) or named in NOT_CODE_PATHS with the reason / the test that covers it."""
    found = _switches_in_sources()
    covered = {k for env in SETS.values() for k in env} | set(NOT_CODE_PATHS)
    assert found - covered == set(), 'switches read by the sources but not covered here: %s' % sorted(found - covered)
    assert covered - found == set(), 'listed here but no longer read by the sources: %s' % sorted(covered - found)


def test_the_fp16_island_is_what_meets_the_bound():
    """TCVOM_NO_F16_ISLAND=1 (bf16 build): plain bf16 storage in the encoder stem / layer1 / layer2 too.  The probe's alpha against the
    default run differs by the bf16 storage noise the island removes -- an order of magnitude above the run-to-run distance."""
    from tcvom_amd._lib import DTYPE_NAME
    if os.environ.get('TCVOM_DTYPE_SUBTEST') or DTYPE_NAME != 'bf16':
        pytest.skip('bf16 build only')
    base, again, plain = _probe({}), _probe({}), _probe({'TCVOM_NO_F16_ISLAND': '1'})
    d_run = sum((x - y) ** 2 for x, y in zip(base['alpha_sample'], again['alpha_sample'])) / len(base['alpha_sample'])
    d_isl = sum((x - y) ** 2 for x, y in zip(base['alpha_sample'], plain['alpha_sample'])) / len(base['alpha_sample'])
    print('alpha sample MSE: run vs rerun %.3e, island vs plain bf16 %.3e' % (d_run, d_isl))
    assert d_isl > 3 * d_run and d_isl > 5e-6


def test_alternative_code_paths_agree_with_the_default_ones():
    if os.environ.get('TCVOM_DTYPE_SUBTEST'):
        pytest.skip('default storage type only')
    base = _probe({})
    again = _probe({})
    sets = SETS

    def worst(a, b):
        w = 0.0
        for k, v in a['grad_norm'].items():
            w = max(w, abs(v - b['grad_norm'][k]) / max(abs(v), 1e-12))
        return w

    def total(r):
        return sum(v * v for v in r['grad_norm'].values()) ** 0.5

    # run-to-run: fp32 atomics in the weight gradients / power iteration flip 16-bit roundings; the small-gradient groups of the
    # encoder move by percents between two identical runs (tests/test_gpu_window.py: cosine 0.98 between two runs at 1088 x 1920)
    noise = worst(base, again)
    tnoise = abs(total(base) - total(again)) / total(base)
    lnoise = max(abs(x - y) / max(abs(x), 1e-3) for x, y in zip(base['losses'], again['losses']))
    print('run-to-run gradient-norm difference: worst group %.2e, total %.2e; losses %.2e' % (noise, tnoise, lnoise))
    for name, env in sets.items():
        alt = _probe(env)
        for x, y in zip(base['losses'], alt['losses']):
            # (bf16 storage: two identical runs already differ by ~1e-3 in L_att -- the statistics atomics flip 16-bit roundings)
            assert abs(x - y) <= max(2e-3, 4 * lnoise) * max(abs(x), 1e-3), (name, base['losses'], alt['losses'], lnoise)
        w = worst(base, alt)
        print('%s: worst per-group gradient-norm difference %.2e' % (name, w))
        assert w < max(3e-2, 5 * noise), (name, w)
        t = abs(total(base) - total(alt)) / total(base)
        print('%s: total gradient-norm difference %.2e' % (name, t))
        assert t < max(1e-2, 5 * tnoise), (name, t)
