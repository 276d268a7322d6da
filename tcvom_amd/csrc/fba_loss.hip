// Losses of the FBA base (models/model.py:129-197, utils/loss_func.py:9-158) as fused kernels, per interior frame:
//   * fba_point_*    refine / F / B selection, the five L1 terms, L1_grad, the visualisation tensors           (1 launch)
//   * excl_*         exclusion_loss(F, B, level=3): per level one reduction (mean |grad|), one for the per-sample terms
//   * lap_*          LapLoss on alpha, F and B: the pyramid is linear, so lap(x) - lap(t) = lap(x - t) and ONE 7-channel
//                    pyramid of the difference serves all three; 5 levels x (down-sample, residual) launches
// and the matching backward kernels (gather form: the transposes of the reflect-padded 5x5 Gaussian, of the zero-
// interleaved up-sampling and of the forward differences are written as gathers, no atomics on the gradient fields).
// All tensors fp32 NCHW.  Reductions end in a block reduction + one atomicAdd per block into small accumulators.
#include "common.h"

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)
static int lgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}
#define EPS_L 1.001e-5f

struct FbaFrame {                      // one interior frame of a window, B samples
    const float* pred;                 // [B][7][HW], sample stride pred_stride
    const float* gt;                   // [B][HW], sample stride frame_stride
    const float* mask;                 // [B][HW], sample stride frame_stride (dilated unknown region, 0 / 1)
    const float* fg;                   // [B][3][HW], sample stride rgb_stride
    const float* bg;
    const float* img;
    int64_t pred_stride, frame_stride, rgb_stride;
    int B, H, W;
};

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float refine_at(const FbaFrame& f, int b, int64_t p) {
    return f.mask[b * f.frame_stride + p] > 0.f ? f.pred[b * f.pred_stride + p] : f.gt[b * f.frame_stride + p];
}
// sqrt(dx^2 + dy^2 + eps) of the forward differences at (y, x) (zero at the last column / row), get_gradient + L1_grad
__device__ __forceinline__ float grad_mag(float c, float r, float d, bool has_r, bool has_d) {
    const float dx = has_r ? r - c : 0.f, dy = has_d ? d - c : 0.f;
    return sqrtf(dx * dx + dy * dy + EPS_L);
}

// acc[0..5] += sum |refine-gt|, sum |F gt + B (1-gt) - img|, sum |fg refine + bg (1-refine) - img|, sum |F-fg|, sum |B-bg|,
// sum |mag(refine) - mag(gt)|.  d0 [B][7][HW] = (refine - gt, F - fg, B - bg); fb [B][6][HW] = (F, B).
__global__ __launch_bounds__(256) void fba_point_fwd_kernel(FbaFrame f, float* __restrict__ d0, float* __restrict__ fb, float* __restrict__ alphas,
                                                            float* __restrict__ comps, float* __restrict__ Fs, float* __restrict__ Bs,
                                                            int64_t vis_frame_stride, int64_t vis_rgb_stride, float* __restrict__ acc) {
    __shared__ float red[4];
    const int64_t HW = (int64_t)f.H * f.W, n = HW * f.B;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    GRID_STRIDE(v, n) {
        const int b = (int)(v / HW);
        const int64_t p = v % HW;
        const int y = (int)(p / f.W), x = (int)(p % f.W);
        const bool m = f.mask[b * f.frame_stride + p] > 0.f;
        const float gt = f.gt[b * f.frame_stride + p];
        const float r = m ? f.pred[b * f.pred_stride + p] : gt;
        alphas[b * vis_frame_stride + p] = r;
        d0[((int64_t)b * 7) * HW + p] = r - gt;
        s[0] += fabsf(r - gt);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float fg = f.fg[b * f.rgb_stride + c * HW + p], bg = f.bg[b * f.rgb_stride + c * HW + p], im = f.img[b * f.rgb_stride + c * HW + p];
            const float cF = m ? f.pred[b * f.pred_stride + (1 + c) * HW + p] : fg;
            const float cB = m ? f.pred[b * f.pred_stride + (4 + c) * HW + p] : bg;
            Fs[b * vis_rgb_stride + c * HW + p] = cF;
            Bs[b * vis_rgb_stride + c * HW + p] = cB;
            comps[b * vis_rgb_stride + c * HW + p] = cF * r + cB * (1.f - r);
            fb[((int64_t)b * 6 + c) * HW + p] = cF;
            fb[((int64_t)b * 6 + 3 + c) * HW + p] = cB;
            d0[((int64_t)b * 7 + 1 + c) * HW + p] = cF - fg;
            d0[((int64_t)b * 7 + 4 + c) * HW + p] = cB - bg;
            s[1] += fabsf(cF * gt + cB * (1.f - gt) - im);
            s[2] += fabsf(fg * r + bg * (1.f - r) - im);
            s[3] += fabsf(cF - fg);
            s[4] += fabsf(cB - bg);
        }
        const bool hr = x + 1 < f.W, hd = y + 1 < f.H;
        const float rr = hr ? refine_at(f, b, p + 1) : 0.f, rd = hd ? refine_at(f, b, p + f.W) : 0.f;
        const float gr = hr ? f.gt[b * f.frame_stride + p + 1] : 0.f, gd = hd ? f.gt[b * f.frame_stride + p + f.W] : 0.f;
        s[5] += fabsf(grad_mag(r, rr, rd, hr, hd) - grad_mag(gt, gr, gd, hr, hd));
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float t = block_sum_256(s[i], red);
        if (threadIdx.x == 0) atomicAdd(acc + i, t);
    }
}

// derivative of sum |mag(refine) - mag(gt)| w.r.t. refine at pixel (y, x): the magnitudes at (y, x), (y, x-1), (y-1, x) use it
__device__ __forceinline__ float dmag_at(const FbaFrame& f, int b, int y, int x, int wy, int wx) {
    // contribution of the magnitude located at (y, x) to the derivative w.r.t. refine at (wy, wx)
    const int64_t p = (int64_t)y * f.W + x;
    const bool hr = x + 1 < f.W, hd = y + 1 < f.H;
    const float c = refine_at(f, b, p), r = hr ? refine_at(f, b, p + 1) : 0.f, d = hd ? refine_at(f, b, p + f.W) : 0.f;
    const float gc = f.gt[b * f.frame_stride + p], gr = hr ? f.gt[b * f.frame_stride + p + 1] : 0.f, gd = hd ? f.gt[b * f.frame_stride + p + f.W] : 0.f;
    const float dx = hr ? r - c : 0.f, dy = hd ? d - c : 0.f;
    const float mag = sqrtf(dx * dx + dy * dy + EPS_L);
    const float s = sgnf(mag - grad_mag(gc, gr, gd, hr, hd)) / mag;
    if (wy == y && wx == x) return s * (-dx - dy);
    if (wy == y && wx == x + 1) return s * dx;
    return s * dy;                                                   // (wy, wx) == (y + 1, x)
}

// coef[0..5] = d loss / d acc[i]; g_d0 [B][7][HW] (from the Laplacian loss), g_fb [B][6][HW] (from the exclusion loss), either may be NULL
__global__ void fba_point_bwd_kernel(FbaFrame f, const float* __restrict__ coef, const float* __restrict__ g_d0, const float* __restrict__ g_fb,
                                     float* __restrict__ dpred) {
    const int64_t HW = (int64_t)f.H * f.W, n = HW * f.B;
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4], c5 = coef[5];
    GRID_STRIDE(v, n) {
        const int b = (int)(v / HW);
        const int64_t p = v % HW;
        float* o = dpred + b * f.pred_stride + p;
        if (!(f.mask[b * f.frame_stride + p] > 0.f)) {
#pragma unroll
            for (int c = 0; c < 7; ++c) o[c * HW] = 0.f;
            continue;
        }
        const int y = (int)(p / f.W), x = (int)(p % f.W);
        const float gt = f.gt[b * f.frame_stride + p];
        const float r = f.pred[b * f.pred_stride + p];
        float gr = c0 * sgnf(r - gt) + (g_d0 ? g_d0[((int64_t)b * 7) * HW + p] : 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float fg = f.fg[b * f.rgb_stride + c * HW + p], bg = f.bg[b * f.rgb_stride + c * HW + p], im = f.img[b * f.rgb_stride + c * HW + p];
            const float cF = f.pred[b * f.pred_stride + (1 + c) * HW + p], cB = f.pred[b * f.pred_stride + (4 + c) * HW + p];
            const float sa = sgnf(cF * gt + cB * (1.f - gt) - im);
            gr += c2 * sgnf(fg * r + bg * (1.f - r) - im) * (fg - bg);
            o[(1 + c) * HW] = c1 * sa * gt + c3 * sgnf(cF - fg) + (g_d0 ? g_d0[((int64_t)b * 7 + 1 + c) * HW + p] : 0.f) +
                              (g_fb ? g_fb[((int64_t)b * 6 + c) * HW + p] : 0.f);
            o[(4 + c) * HW] = c1 * sa * (1.f - gt) + c4 * sgnf(cB - bg) + (g_d0 ? g_d0[((int64_t)b * 7 + 4 + c) * HW + p] : 0.f) +
                              (g_fb ? g_fb[((int64_t)b * 6 + 3 + c) * HW + p] : 0.f);
        }
        float gm = dmag_at(f, b, y, x, y, x);
        if (x > 0) gm += dmag_at(f, b, y, x - 1, y, x);
        if (y > 0) gm += dmag_at(f, b, y - 1, x, y, x);
        o[0] = gr + c5 * gm;
    }
}

// ------------------------------------------------------------------------------------------ exclusion loss
// t = 2 sigmoid(u) - 1 = tanh(u / 2);  f(u) = t^2;  f'(u) = t (1 - t^2)
__device__ __forceinline__ float ex_t(float u) { return 2.f / (1.f + __expf(-u)) - 1.f; }
// forward differences of channel `ch` of lvl [B][6][h][w] at (y, x)
__device__ __forceinline__ void ex_grad(const float* __restrict__ im, int h, int w, int y, int x, float& gx, float& gy) {
    const float c = im[(int64_t)y * w + x];
    gx = x + 1 < w ? im[(int64_t)y * w + x + 1] - c : 0.f;
    gy = y + 1 < h ? im[(int64_t)(y + 1) * w + x] - c : 0.f;
}
// sums[0..3] += sum |gx1|, |gx2|, |gy1|, |gy2| over all samples and the 3 channels
__global__ __launch_bounds__(256) void excl_abs_kernel(const float* __restrict__ lvl, int B, int h, int w, float* __restrict__ sums) {
    __shared__ float red[4];
    const int64_t hw = (int64_t)h * w, n = hw * 3 * B;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    GRID_STRIDE(v, n) {
        const int64_t p = v % hw;
        const int c = (int)((v / hw) % 3), b = (int)(v / (3 * hw));
        const int y = (int)(p / w), x = (int)(p % w);
        float gx1, gy1, gx2, gy2;
        ex_grad(lvl + ((int64_t)b * 6 + c) * hw, h, w, y, x, gx1, gy1);
        ex_grad(lvl + ((int64_t)b * 6 + 3 + c) * hw, h, w, y, x, gx2, gy2);
        s[0] += fabsf(gx1); s[1] += fabsf(gx2); s[2] += fabsf(gy1); s[3] += fabsf(gy2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float t = block_sum_256(s[i], red);
        if (threadIdx.x == 0) atomicAdd(sums + i, t);
    }
}
// terms [B][2] += sum f(gx1) f(ax gx2), sum f(gy1) f(ay gy2) per sample; also (mode 1) dsum[0..1] += d L / d ax, d L / d ay
// = sum_b w[b] sum f(g1) f'(a g2) g2 with w [B][2] = d L / d terms.   grid.y = sample
__global__ __launch_bounds__(256) void excl_terms_kernel(const float* __restrict__ lvl, int h, int w, const float* __restrict__ sums,
                                                         const float* __restrict__ wts, float* __restrict__ out, int mode) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)h * w, n = hw * 3;
    const float N = (float)(hw * 3 * gridDim.y);
    const float ax = 2.f * (sums[0] / N) / (sums[1] / N + EPS_L), ay = 2.f * (sums[2] / N) / (sums[3] / N + EPS_L);
    float sx = 0.f, sy = 0.f;
    GRID_STRIDE(v, n) {
        const int64_t p = v % hw;
        const int c = (int)(v / hw);
        const int y = (int)(p / w), x = (int)(p % w);
        float gx1, gy1, gx2, gy2;
        ex_grad(lvl + ((int64_t)b * 6 + c) * hw, h, w, y, x, gx1, gy1);
        ex_grad(lvl + ((int64_t)b * 6 + 3 + c) * hw, h, w, y, x, gx2, gy2);
        const float t1x = ex_t(gx1), t2x = ex_t(ax * gx2), t1y = ex_t(gy1), t2y = ex_t(ay * gy2);
        if (mode == 0) {
            sx += t1x * t1x * t2x * t2x;
            sy += t1y * t1y * t2y * t2y;
        } else {
            sx += t1x * t1x * t2x * (1.f - t2x * t2x) * gx2;
            sy += t1y * t1y * t2y * (1.f - t2y * t2y) * gy2;
        }
    }
    sx = block_sum_256(sx, red);
    sy = block_sum_256(sy, red);
    if (threadIdx.x == 0) {
        if (mode == 0) { atomicAdd(out + b * 2, sx); atomicAdd(out + b * 2 + 1, sy); }
        else { atomicAdd(out, wts[b * 2] * sx); atomicAdd(out + 1, wts[b * 2 + 1] * sy); }
    }
}
// 2x2 average pooling of [BC][h][w] fp32
__global__ void avgpool2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int h, int w) {
    GRID_STRIDE(v, n) {
        const int ow = w / 2, oh = h / 2;
        const int j = (int)(v % ow), i = (int)((v / ow) % oh);
        const int64_t bc = v / ((int64_t)ow * oh);
        const float* s = x + (bc * h + 2 * i) * w + 2 * j;
        y[v] = 0.25f * (s[0] + s[1] + s[w] + s[w + 1]);
    }
}
// d L / d gx of image 1 (which = 0) or image 2 (which = 1) at (y, x), x-direction (dir = 0) or y-direction (dir = 1)
__device__ __forceinline__ float ex_dg(const float* __restrict__ lvl, int b, int c, int h, int w, int y, int x, int which, int dir, float a,
                                       float wb, float dA, float m1, float m2, float N) {
    const int64_t hw = (int64_t)h * w;
    float gx1, gy1, gx2, gy2;
    ex_grad(lvl + ((int64_t)b * 6 + c) * hw, h, w, y, x, gx1, gy1);
    ex_grad(lvl + ((int64_t)b * 6 + 3 + c) * hw, h, w, y, x, gx2, gy2);
    const float g1 = dir ? gy1 : gx1, g2 = dir ? gy2 : gx2;
    const float t1 = ex_t(g1), t2 = ex_t(a * g2);
    if (which == 0) return wb * t1 * (1.f - t1 * t1) * t2 * t2 + dA * 2.f * sgnf(g1) / (N * (m2 + EPS_L));
    return wb * t1 * t1 * t2 * (1.f - t2 * t2) * a - dA * 2.f * m1 * sgnf(g2) / (N * (m2 + EPS_L) * (m2 + EPS_L));
}
// dlvl [B][6][h][w] = gradient of the level's terms w.r.t. the level image (+ 0.25 * dcoarse of the next level, if any)
__global__ void excl_bwd_kernel(const float* __restrict__ lvl, int B, int h, int w, const float* __restrict__ sums, const float* __restrict__ wts,
                                const float* __restrict__ dsum, const float* __restrict__ dcoarse, float* __restrict__ dlvl) {
    const int64_t hw = (int64_t)h * w, n = hw * 6 * B;
    const float N = (float)(hw * 3 * B);
    const float m1x = sums[0] / N, m2x = sums[1] / N, m1y = sums[2] / N, m2y = sums[3] / N;
    const float ax = 2.f * m1x / (m2x + EPS_L), ay = 2.f * m1y / (m2y + EPS_L);
    GRID_STRIDE(v, n) {
        const int64_t p = v % hw;
        const int ch = (int)((v / hw) % 6), b = (int)(v / (6 * hw));
        const int which = ch / 3, c = ch % 3;
        const int y = (int)(p / w), x = (int)(p % w);
        const float wx = wts[b * 2], wy = wts[b * 2 + 1];
        float g = 0.f;
        if (x + 1 < w) g -= ex_dg(lvl, b, c, h, w, y, x, which, 0, ax, wx, dsum[0], m1x, m2x, N);
        if (x > 0) g += ex_dg(lvl, b, c, h, w, y, x - 1, which, 0, ax, wx, dsum[0], m1x, m2x, N);
        if (y + 1 < h) g -= ex_dg(lvl, b, c, h, w, y, x, which, 1, ay, wy, dsum[1], m1y, m2y, N);
        if (y > 0) g += ex_dg(lvl, b, c, h, w, y - 1, x, which, 1, ay, wy, dsum[1], m1y, m2y, N);
        if (dcoarse) g += 0.25f * dcoarse[(((int64_t)b * 6 + ch) * (h / 2) + y / 2) * (w / 2) + x / 2];
        dlvl[v] = g;
    }
}

// ------------------------------------------------------------------------------------------ Laplacian pyramid
__device__ __forceinline__ int refl(int u, int n) { return u < 0 ? -u : (u >= n ? 2 * (n - 1) - u : u); }
__device__ __forceinline__ float g5(int a) { return a == 2 ? 0.375f : ((a == 1 || a == 3) ? 0.25f : 0.0625f); }    // [1 4 6 4 1] / 16

// down [BC][h/2][w/2] = (5x5 Gaussian with reflect padding of cur [BC][h][w]) at the even pixels
__global__ void lap_down_kernel(const float* __restrict__ cur, float* __restrict__ down, int64_t n, int h, int w) {
    GRID_STRIDE(v, n) {
        const int ow = w / 2, oh = h / 2;
        const int j = (int)(v % ow), i = (int)((v / ow) % oh);
        const float* s = cur + (v / ((int64_t)ow * oh)) * h * w;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const int yy = refl(2 * i + a - 2, h);
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 5; ++b) r += g5(b) * s[(int64_t)yy * w + refl(2 * j + b - 2, w)];
            acc += g5(a) * r;
        }
        down[v] = acc;
    }
}
// residual of the level: pyr = cur - 4 * Gaussian(zero-interleaved down); acc[c] += sum |pyr| (c = channel of 7); sgn = sign(pyr)
__global__ __launch_bounds__(256) void lap_resid_kernel(const float* __restrict__ cur, const float* __restrict__ down, int8_t* __restrict__ sgn,
                                                        int h, int w, float* __restrict__ acc) {
    __shared__ float red[4];
    const int64_t hw = (int64_t)h * w;
    const int bc = blockIdx.y, c = bc % 7;
    const float* s = down + (int64_t)bc * (h / 2) * (w / 2);
    float sum = 0.f;
    GRID_STRIDE(p, hw) {
        const int y = (int)(p / w), x = (int)(p % w);
        float up = 0.f;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const int zy = refl(y + a - 2, h);
            if (zy & 1) continue;
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const int zx = refl(x + b - 2, w);
                if (!(zx & 1)) r += g5(b) * s[(int64_t)(zy / 2) * (w / 2) + zx / 2];
            }
            up += g5(a) * r;
        }
        const float pyr = cur[(int64_t)bc * hw + p] - 4.f * up;
        sum += fabsf(pyr);
        sgn[(int64_t)bc * hw + p] = pyr > 0.f ? 1 : (pyr < 0.f ? -1 : 0);
    }
    sum = block_sum_256(sum, red);
    if (threadIdx.x == 0) atomicAdd(acc + c, sum);
}
// virtual (pre-reflection) positions that map onto index x of an axis of length n padded by 2: x itself, -x, 2(n-1) - x
__device__ __forceinline__ int virt_positions(int x, int n, int* v) {
    int k = 0;
    v[k++] = x;
    if (x >= 1 && x <= 2) v[k++] = -x;
    if (x >= n - 3 && x <= n - 2) v[k++] = 2 * (n - 1) - x;
    return k;
}
// r [BC][h/2][w/2] = gnext (or 0) - (U^T s)[i][j], s = coef[c] * sgn (fine level [h][w]); U = 4 * Gaussian o zero-interleave
__global__ void lap_bwd_coarse_kernel(const int8_t* __restrict__ sgn, const float* __restrict__ coef, const float* __restrict__ gnext,
                                      float* __restrict__ r, int64_t n, int h, int w) {
    GRID_STRIDE(v, n) {
        const int ow = w / 2, oh = h / 2;
        const int j = (int)(v % ow), i = (int)((v / ow) % oh);
        const int64_t bc = v / ((int64_t)ow * oh);
        const int8_t* s = sgn + bc * h * w;
        int vy[3], vx[3];
        const int ny = virt_positions(2 * i, h, vy), nx = virt_positions(2 * j, w, vx);
        float acc = 0.f;
        for (int iy = 0; iy < ny; ++iy)
            for (int a = 0; a < 5; ++a) {
                const int y = vy[iy] - a + 2;
                if (y < 0 || y >= h) continue;
                float row = 0.f;
                for (int ix = 0; ix < nx; ++ix)
                    for (int b = 0; b < 5; ++b) {
                        const int x = vx[ix] - b + 2;
                        if (x >= 0 && x < w) row += g5(b) * (float)s[(int64_t)y * w + x];
                    }
                acc += g5(a) * row;
            }
        r[v] = (gnext ? gnext[v] : 0.f) - 4.f * coef[bc % 7] * acc;
    }
}
// g [BC][h][w] = coef[c] * sgn + (D^T r)[y][x], D = (even sub-sampling) o (reflect-padded Gaussian); r [BC][h/2][w/2]
__global__ void lap_bwd_fine_kernel(const int8_t* __restrict__ sgn, const float* __restrict__ coef, const float* __restrict__ r, float* __restrict__ g,
                                    int64_t n, int h, int w) {
    GRID_STRIDE(v, n) {
        const int x = (int)(v % w), y = (int)((v / w) % h);
        const int64_t bc = v / ((int64_t)w * h);
        const float* rs = r + bc * (h / 2) * (w / 2);
        int vy[3], vx[3];
        const int ny = virt_positions(y, h, vy), nx = virt_positions(x, w, vx);
        float acc = 0.f;
        for (int iy = 0; iy < ny; ++iy)
            for (int a = (vy[iy] & 1); a < 5; a += 2) {                 // 2 i = vy - a + 2 must be even
                const int i = (vy[iy] - a + 2) / 2;
                if (vy[iy] - a + 2 < 0 || i >= h / 2) continue;
                float row = 0.f;
                for (int ix = 0; ix < nx; ++ix)
                    for (int b = (vx[ix] & 1); b < 5; b += 2) {
                        const int j = (vx[ix] - b + 2) / 2;
                        if (vx[ix] - b + 2 >= 0 && j < w / 2) row += g5(b) * rs[(int64_t)i * (w / 2) + j];
                    }
                acc += g5(a) * row;
            }
        g[v] = coef[bc % 7] * (float)sgn[v] + acc;
    }
}

// ------------------------------------------------------------------------------------------ C ABI
static FbaFrame mk_frame(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg, const float* img,
                         int64_t pred_stride, int64_t frame_stride, int64_t rgb_stride, int B, int H, int W) {
    FbaFrame f;
    f.pred = pred; f.gt = gt; f.mask = mask; f.fg = fg; f.bg = bg; f.img = img;
    f.pred_stride = pred_stride; f.frame_stride = frame_stride; f.rgb_stride = rgb_stride;
    f.B = B; f.H = H; f.W = W;
    return f;
}
extern "C" int tcvom_fba_point_fwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg, const float* img,
                                   int64_t pred_stride, int64_t frame_stride, int64_t rgb_stride, float* d0, float* fb, float* alphas,
                                   float* comps, float* Fs, float* Bs, float* acc, int32_t B, int32_t H, int32_t W, void* stream) {
    TCVOM_CHECK_ARG(pred && gt && mask && fg && bg && img && d0 && fb && alphas && comps && Fs && Bs && acc && B > 0 && H > 0 && W > 0, "fba_point_fwd: bad args");
    const FbaFrame f = mk_frame(pred, gt, mask, fg, bg, img, pred_stride, frame_stride, rgb_stride, B, H, W);
    hipLaunchKernelGGL(fba_point_fwd_kernel, dim3(lgrid((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, f, d0, fb, alphas, comps, Fs, Bs,
                       frame_stride, rgb_stride, acc);
    TCVOM_LAUNCH_CHECK("fba_point_fwd");
    return TCVOM_OK;
}
extern "C" int tcvom_fba_point_bwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg, const float* img,
                                   int64_t pred_stride, int64_t frame_stride, int64_t rgb_stride, const float* coef, const float* g_d0,
                                   const float* g_fb, float* dpred, int32_t B, int32_t H, int32_t W, void* stream) {
    TCVOM_CHECK_ARG(pred && gt && mask && fg && bg && img && coef && dpred && B > 0 && H > 0 && W > 0, "fba_point_bwd: bad args");
    const FbaFrame f = mk_frame(pred, gt, mask, fg, bg, img, pred_stride, frame_stride, rgb_stride, B, H, W);
    hipLaunchKernelGGL(fba_point_bwd_kernel, dim3(lgrid((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, f, coef, g_d0, g_fb, dpred);
    TCVOM_LAUNCH_CHECK("fba_point_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_excl_abs(const float* lvl, float* sums, int32_t B, int32_t h, int32_t w, void* stream) {
    TCVOM_CHECK_ARG(lvl && sums && B > 0 && h > 0 && w > 0, "excl_abs: bad args");
    hipLaunchKernelGGL(excl_abs_kernel, dim3(lgrid((int64_t)B * 3 * h * w)), dim3(256), 0, (hipStream_t)stream, lvl, B, h, w, sums);
    TCVOM_LAUNCH_CHECK("excl_abs");
    return TCVOM_OK;
}
extern "C" int tcvom_excl_terms(const float* lvl, const float* sums, const float* wts, float* out, int32_t mode, int32_t B, int32_t h, int32_t w,
                                void* stream) {
    TCVOM_CHECK_ARG(lvl && sums && out && (mode == 0 || wts) && B > 0 && h > 0 && w > 0, "excl_terms: bad args");
    int gx = lgrid((int64_t)3 * h * w);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(excl_terms_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, lvl, h, w, sums, wts, out, mode);
    TCVOM_LAUNCH_CHECK("excl_terms");
    return TCVOM_OK;
}
extern "C" int tcvom_avgpool2_f32(const float* x, float* y, int64_t planes, int32_t h, int32_t w, void* stream) {
    TCVOM_CHECK_ARG(x && y && planes > 0 && h % 2 == 0 && w % 2 == 0, "avgpool2_f32: bad args");
    const int64_t n = planes * (h / 2) * (w / 2);
    hipLaunchKernelGGL(avgpool2_f32_kernel, dim3(lgrid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, h, w);
    TCVOM_LAUNCH_CHECK("avgpool2_f32");
    return TCVOM_OK;
}
extern "C" int tcvom_excl_bwd(const float* lvl, const float* sums, const float* wts, const float* dsum, const float* dcoarse, float* dlvl,
                              int32_t B, int32_t h, int32_t w, void* stream) {
    TCVOM_CHECK_ARG(lvl && sums && wts && dsum && dlvl && B > 0 && h > 0 && w > 0 && (!dcoarse || (h % 2 == 0 && w % 2 == 0)), "excl_bwd: bad args");
    hipLaunchKernelGGL(excl_bwd_kernel, dim3(lgrid((int64_t)B * 6 * h * w)), dim3(256), 0, (hipStream_t)stream, lvl, B, h, w, sums, wts, dsum, dcoarse, dlvl);
    TCVOM_LAUNCH_CHECK("excl_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_lap_down(const float* cur, float* down, int64_t planes, int32_t h, int32_t w, void* stream) {
    TCVOM_CHECK_ARG(cur && down && planes > 0 && h >= 4 && w >= 4 && h % 2 == 0 && w % 2 == 0, "lap_down: bad args");
    const int64_t n = planes * (h / 2) * (w / 2);
    hipLaunchKernelGGL(lap_down_kernel, dim3(lgrid(n)), dim3(256), 0, (hipStream_t)stream, cur, down, n, h, w);
    TCVOM_LAUNCH_CHECK("lap_down");
    return TCVOM_OK;
}
extern "C" int tcvom_lap_resid(const float* cur, const float* down, int8_t* sgn, float* acc, int64_t planes, int32_t h, int32_t w, void* stream) {
    TCVOM_CHECK_ARG(cur && down && sgn && acc && planes > 0 && planes % 7 == 0 && planes < 65536 && h >= 4 && w >= 4, "lap_resid: bad args");
    int gx = lgrid((int64_t)h * w);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(lap_resid_kernel, dim3(gx, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, cur, down, sgn, h, w, acc);
    TCVOM_LAUNCH_CHECK("lap_resid");
    return TCVOM_OK;
}
extern "C" int tcvom_lap_bwd_coarse(const int8_t* sgn, const float* coef, const float* gnext, float* r, int64_t planes, int32_t h, int32_t w,
                                    void* stream) {
    TCVOM_CHECK_ARG(sgn && coef && r && planes > 0 && planes % 7 == 0 && h >= 4 && w >= 4, "lap_bwd_coarse: bad args");
    const int64_t n = planes * (h / 2) * (w / 2);
    hipLaunchKernelGGL(lap_bwd_coarse_kernel, dim3(lgrid(n)), dim3(256), 0, (hipStream_t)stream, sgn, coef, gnext, r, n, h, w);
    TCVOM_LAUNCH_CHECK("lap_bwd_coarse");
    return TCVOM_OK;
}
extern "C" int tcvom_lap_bwd_fine(const int8_t* sgn, const float* coef, const float* r, float* g, int64_t planes, int32_t h, int32_t w,
                                  void* stream) {
    TCVOM_CHECK_ARG(sgn && coef && r && g && planes > 0 && planes % 7 == 0 && h >= 4 && w >= 4, "lap_bwd_fine: bad args");
    const int64_t n = planes * h * w;
    hipLaunchKernelGGL(lap_bwd_fine_kernel, dim3(lgrid(n)), dim3(256), 0, (hipStream_t)stream, sgn, coef, r, g, n, h, w);
    TCVOM_LAUNCH_CHECK("lap_bwd_fine");
    return TCVOM_OK;
}

// ------------------------------------------------------------------------------------------ the scalars between the loss kernels
// acc layout (tcvom_amd/fba_losses.py): [0..5] point sums, [6 + 7 l + c] Laplacian |residual| sums of level l / channel c (5 levels),
// then per exclusion level: sums[4], terms[B][2].  One thread: the ~100 ATen scalar kernels these replace were 0.3 ms of a 1080p step.
//   out[0] = L_alpha_comp, out[1] = L_lap, out[2] = L_grad                     (models/model.py:142-175, normalize = True)
__global__ void fba_loss_finish_kernel(const float* __restrict__ acc, int B, float n1, float n3, float en0, float en1, float en2,
                                       float* __restrict__ out) {
    const float en[3] = {en0, en1, en2};
    float excl = 0.f;
    int off = 6 + 35;
    for (int l = 0; l < 3; ++l) {
        const float* t = acc + off + 4;
        float mx = 0.f, my = 0.f;
        for (int b = 0; b < B; ++b) { mx += powf(t[2 * b] / en[l] + EPS_L, 0.25f); my += powf(t[2 * b + 1] / en[l] + EPS_L, 0.25f); }
        excl += (mx + my) / (float)B;
        off += 4 + 2 * B;
    }
    excl *= 1.f / 3.f;
    float lap = 0.f;
    for (int l = 0; l < 5; ++l) {
        const float pw = (float)(1 << l);
        lap += acc[6 + 7 * l] * (1.f / n1) * pw;
        for (int c = 1; c < 7; ++c) lap += acc[6 + 7 * l + c] * (0.25f / n3) * pw;
    }
    out[0] = acc[0] / n1 + acc[1] / n3 + 0.25f * (acc[2] / n3 + acc[3] / n3 + acc[4] / n3);
    out[1] = lap;
    out[2] = acc[5] / n1 + 0.25f * excl;
}
// backward coefficients: out = coef[6] | wts[3][B][2] (d L / d terms of the exclusion levels) | cf[5][7] (Laplacian level weights)
__global__ void fba_loss_coefs_kernel(const float* __restrict__ acc, const float* __restrict__ g_ac, const float* __restrict__ g_lap,
                                      const float* __restrict__ g_grad, int B, float n1, float n3, float en0, float en1, float en2,
                                      float* __restrict__ out) {
    const float ga = g_ac ? *g_ac : 0.f, gl = g_lap ? *g_lap : 0.f, gg = g_grad ? *g_grad : 0.f;
    out[0] = ga / n1; out[1] = ga / n3; out[2] = 0.25f * ga / n3; out[3] = 0.25f * ga / n3; out[4] = 0.25f * ga / n3; out[5] = gg / n1;
    const float en[3] = {en0, en1, en2};
    int off = 6 + 35;
    float* w = out + 6;
    for (int l = 0; l < 3; ++l) {
        const float* t = acc + off + 4;
        for (int i = 0; i < 2 * B; ++i)
            w[l * 2 * B + i] = gg * (0.25f / 3.f / (float)B) * 0.25f * powf(t[i] / en[l] + EPS_L, -0.75f) / en[l];
        off += 4 + 2 * B;
    }
    float* cf = out + 6 + 6 * B;
    for (int l = 0; l < 5; ++l) {
        const float pw = (float)(1 << l);
        cf[7 * l] = gl * pw / n1;
        for (int c = 1; c < 7; ++c) cf[7 * l + c] = gl * pw * 0.25f / n3;
    }
}
extern "C" int tcvom_fba_loss_finish(const float* acc, int32_t B, float n1, float n3, float excl_n0, float excl_n1, float excl_n2, float* out,
                                     void* stream) {
    TCVOM_CHECK_ARG(acc && out && B > 0 && B <= 64, "fba_loss_finish: bad args");
    hipLaunchKernelGGL(fba_loss_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, B, n1, n3, excl_n0, excl_n1, excl_n2, out);
    TCVOM_LAUNCH_CHECK("fba_loss_finish");
    return TCVOM_OK;
}
extern "C" int tcvom_fba_loss_coefs(const float* acc, const float* g_ac, const float* g_lap, const float* g_grad, int32_t B, float n1, float n3,
                                    float excl_n0, float excl_n1, float excl_n2, float* out /* 6 + 6 B + 35 floats */, void* stream) {
    TCVOM_CHECK_ARG(acc && out && B > 0 && B <= 64, "fba_loss_coefs: bad args");
    hipLaunchKernelGGL(fba_loss_coefs_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, g_ac, g_lap, g_grad, B, n1, n3, excl_n0, excl_n1,
                       excl_n2, out);
    TCVOM_LAUNCH_CHECK("fba_loss_coefs");
    return TCVOM_OK;
}
