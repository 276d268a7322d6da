"""Geometry of the implicit-GEMM launches (tcvom_conv_desc) for every conv flavour of the hot path.

A convolution (regular, strided, or ConvTranspose2d k4 s2 p1) and its data-gradient are each
expressed as one or four "phases": a set of output positions on a regular sub-grid plus a tap list
(input offset, weight slot).  Stride-2 data gradients and ConvTranspose forwards are the 4-phase
sub-pixel form, so no zero-insertion and no wasted MACs.  Host-only integer code.
"""
from ._lib import ConvDesc, MAX_TAPS


def _desc(N, H, W, C, OH, OW, K, PH, PW, in_step, out_step, off_h, off_w, taps, wt, ldo=None):
    """taps: list of (dh, dw, wslot).  The reduction runs over ntaps*C elements in 64-deep steps; the tail of the last step
    falls into tap slots past the list, which the kernels treat as zero taps (igemm_nt_kernel: `taps` table) -- a 1x1 conv
    over 24 channels is ONE step, not the 8 zero-padded taps (192 elements) that made ntaps*C a multiple of 64."""
    taps = list(taps)
    assert ((len(taps) * C + 63) // 64 * 64 - 1) // C < MAX_TAPS, 'too many taps'
    d = ConvDesc()
    d.N, d.H, d.W, d.C = N, H, W, C
    d.OH, d.OW, d.K = OH, OW, K
    d.PH, d.PW = PH, PW
    d.in_step, d.out_step, d.out_off_h, d.out_off_w = in_step, out_step, off_h, off_w
    d.ntaps = len(taps)
    for t, (dh, dw, ws) in enumerate(taps):
        d.tap_dh[t], d.tap_dw[t], d.tap_w[t] = dh, dw, ws
    for t in range(len(taps), MAX_TAPS):
        d.tap_w[t] = -1
    d.wt = wt
    d.ldo = K if ldo is None else ldo
    d.act = 0
    d.out_fp32 = 0
    d.stats_group_offset = 0
    d.batch = 1
    d.w_layout = 0
    return d


class ConvGeometry(object):
    """All launches of one conv layer on one input shape."""

    def __init__(self, spec, N, H, W):
        R, S, st, p = spec.R, spec.S, spec.stride, spec.pad
        K, C, Cp = spec.K, spec.C, spec.cpad
        dil = getattr(spec, 'dilation', 1)
        assert dil == 1 or st == 1, 'dilated convs are stride 1 (ResnetDilated, models/FBA/models.py:203-217)'
        if getattr(spec, 'stem', False):
            # 7x7 stride-2 pad-3 stem on the 2x2 space-to-depth input [N][H][W][64] (H, W = half resolution): a 4x4
            # stride-1 kernel with taps (a, b) in -2..1; weight slot (a + 2) * 4 + (b + 2) (csrc/spectral.hip kind 16)
            self.N, self.H, self.W = N, H, W
            taps = [(a, b, (a + 2) * 4 + (b + 2)) for a in range(-2, 2) for b in range(-2, 2)]
            d = _desc(N, H, W, Cp, H, W, K, H, W, 1, 1, 0, 0, taps, 16)
            self.fwd, self.wgrad, self.dgrad = [d], [d], []
            self.OH, self.OW, self.K, self.C = H, W, K, C
            self.out_pixels = self.in_pixels = N * H * W
            return
        self.N, self.H, self.W = N, H, W
        self.fwd, self.dgrad, self.wgrad = [], [], []
        f16 = getattr(spec, 'f16', False)      # fp16 island of the bf16 build: IEEE fp16 operands and results in the FORWARD launch

        def fwd_desc(*a):
            # a = (N, H, W, C, OH, OW, K, PH, PW, in_step, out_step, off_h, off_w, taps, wt)
            self.wgrad.append(_desc(*a))
            if f16:
                d = _desc(*a)                   # (its own descriptor: the kernel choice -- hence the statistics layout -- follows the flags)
                d.in_f16, d.out_fp32 = 1, 2
                self.fwd.append(d)
            else:
                self.fwd.append(self.wgrad[-1])
        if not spec.transposed:
            OH, OW = (H + 2 * p - R) // st + 1, (W + 2 * p - S) // st + 1
            OH, OW = (H + 2 * p - dil * (R - 1) - 1) // st + 1, (W + 2 * p - dil * (S - 1) - 1) // st + 1
            taps = [(r * dil - p, s * dil - p, r * S + s) for r in range(R) for s in range(S)]
            fwd_desc(N, H, W, Cp, OH, OW, K, OH, OW, st, 1, 0, 0, taps, R * S)
            if spec.needs_dgrad:
                # dx[n,h,w,c] = sum_{r,s,k} dy[n,(h+p-r)/st,(w+p-s)/st,k] W[k][c][r][s]   (exact divisions only)
                for ph in range(st):
                    for pw in range(st):
                        PH, PW = (H - ph + st - 1) // st, (W - pw + st - 1) // st
                        tp = [((ph + p - r * dil) // st, (pw + p - s * dil) // st, r * S + s)
                              for r in range(R) for s in range(S)
                              if (ph + p - r * dil) % st == 0 and (pw + p - s * dil) % st == 0]
                        assert tp, 'phase without taps'
                        self.dgrad.append(_desc(N, OH, OW, K, H, W, C, PH, PW, 1, st, ph, pw, tp, R * S, ldo=Cp if Cp > 8 else None))
        else:
            assert R == 4 and S == 4 and st == 2 and p == 1, 'only ConvTranspose2d(k=4, s=2, p=1)'
            OH, OW = 2 * H, 2 * W
            for ph in range(2):
                for pw in range(2):
                    tp = [((ph + 1 - r) // 2, (pw + 1 - s) // 2, r * 4 + s)
                          for r in range(4) for s in range(4)
                          if (ph + 1 - r) % 2 == 0 and (pw + 1 - s) % 2 == 0]
                    fwd_desc(N, H, W, Cp, OH, OW, K, H, W, 1, 2, ph, pw, tp, 16)
            if spec.needs_dgrad:
                taps = [(r - 1, s - 1, r * 4 + s) for r in range(4) for s in range(4)]
                self.dgrad.append(_desc(N, OH, OW, K, H, W, C, H, W, 2, 1, 0, 0, taps, 16))
        self.OH, self.OW, self.K, self.C = OH, OW, K, C
        self.out_pixels = N * OH * OW
        self.in_pixels = N * H * W
        if getattr(spec, 'frag', False):        # packed fragment-major by the WeightBank (weights.py: ConvSpec.frag)
            for d in self.fwd + self.dgrad:
                d.w_layout = 1


def dense_desc(rows_b, rows_a, kred, ld_out, batch=1, in_bstride=0, w_bstride=0, out_bstride=0, vec_bstride=0,
               out_fp32=False):
    """Dense GEMM through the conv engine:  out[n][m] = sum_k B[n][k] * A[m][k]
    (B = `in` operand with rows_b rows, A = `w` operand with rows_a rows, both with row length kred)."""
    assert kred % 64 == 0
    d = _desc(1, 1, rows_b, kred, 1, rows_b, rows_a, 1, rows_b, 1, 1, 0, 0, [(0, 0, 0)], 1, ldo=ld_out)
    d.out_fp32 = 1 if out_fp32 else 0
    d.batch = batch
    d.in_bstride, d.w_bstride, d.out_bstride, d.vec_bstride = in_bstride, w_bstride, out_bstride, vec_bstride
    return d


def dense_tt_desc(rows, cols_a, cols_b):
    """Dense TT GEMM through the weight-gradient engine:  dw[m][n] += sum_p A[p][m] * B[p][n]
    (A = `dy` operand [rows][ldy] using its first cols_a columns, B = `in` operand [rows][cols_b])."""
    return _desc(1, 1, rows, cols_b, 1, rows, cols_a, 1, rows, 1, 1, 0, 0, [(0, 0, 0)], 1)
