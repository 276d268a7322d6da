// DIM base network pieces (models/DIM/vggnet.py, BASELINE.json config 1): max-pool with indices / max-unpool on NHWC
// bf16, the ReLU mask for conv+bias+ReLU layers without BatchNorm, and the composition / gradient losses of
// FullModel.single_image_loss (models/model.py:94-127, utils/loss_func.py:42-59).  HBM-bound streaming kernels:
// 16-byte (8 x bf16) accesses per lane.
#include "common.h"

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

static int dgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- MaxPool2d(2, 2, return_indices) / MaxUnpool2d(2, 2)
// one thread = one output pixel x 8 channels; idx byte per element = dy*2 + dx of the FIRST maximum in scan order
__global__ void maxpool2_idx_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, uint2* __restrict__ idx,
                                    int64_t n, int OH, int OW, int C8) {
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        const int64_t p = v / C8;
        const int ow = (int)(p % OW);
        const int64_t t = p / OW;
        const int oh = (int)(t % OH);
        const int64_t nn = t / OH;
        const int W = OW * 2;
        const int64_t base = ((nn * (OH * 2) + oh * 2) * W + ow * 2) * C8 + c8;
        float best[8];
        unsigned char bi[8];
        unpack8(x[base], best);
#pragma unroll
        for (int k = 0; k < 8; ++k) bi[k] = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            float f[8];
            unpack8(x[base + ((q >> 1) * (int64_t)W + (q & 1)) * C8], f);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (f[k] > best[k] || (f[k] != f[k] && best[k] == best[k])) { best[k] = f[k]; bi[k] = (unsigned char)q; }
        }
        y[v] = pack8(best);
        uint2 o;
        o.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
        o.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
        idx[v] = o;
    }
}
__global__ void unpool2_kernel(const uint4* __restrict__ y, const uint2* __restrict__ idx, uint4* __restrict__ x,
                               int64_t n, int OH, int OW, int C8) {
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        const int64_t p = v / C8;
        const int ow = (int)(p % OW);
        const int64_t t = p / OW;
        const int oh = (int)(t % OH);
        const int64_t nn = t / OH;
        const int W = OW * 2;
        const int64_t base = ((nn * (OH * 2) + oh * 2) * W + ow * 2) * C8 + c8;
        float f[8];
        unpack8(y[v], f);
        const uint2 iv = idx[v];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned b = ((k < 4 ? iv.x : iv.y) >> (8 * (k & 3))) & 0xffu;
                o[k] = b == (unsigned)q ? f[k] : 0.f;
            }
            x[base + ((q >> 1) * (int64_t)W + (q & 1)) * C8] = pack8(o);
        }
    }
}
__global__ void pick2_kernel(const uint4* __restrict__ x, const uint2* __restrict__ idx, uint4* __restrict__ y,
                             int64_t n, int OH, int OW, int C8) {
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        const int64_t p = v / C8;
        const int ow = (int)(p % OW);
        const int64_t t = p / OW;
        const int oh = (int)(t % OH);
        const int64_t nn = t / OH;
        const int W = OW * 2;
        const int64_t base = ((nn * (OH * 2) + oh * 2) * W + ow * 2) * C8 + c8;
        const uint2 iv = idx[v];
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f[8];
            unpack8(x[base + ((q >> 1) * (int64_t)W + (q & 1)) * C8], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned b = ((k < 4 ? iv.x : iv.y) >> (8 * (k & 3))) & 0xffu;
                if (b == (unsigned)q) o[k] = f[k];
            }
        }
        y[v] = pack8(o);
    }
}
__global__ void relu_bwd_kernel(const uint4* __restrict__ dz, const uint4* __restrict__ y, uint4* __restrict__ dy, int64_t n8, float slope) {
    GRID_STRIDE(v, n8) {
        float g[8], a[8];
        unpack8(dz[v], g);
        unpack8(y[v], a);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = a[k] > 0.f ? g[k] : slope * g[k];
        dy[v] = pack8(g);
    }
}

// ---------------------------------------------------------------- im2col / col2im for the 7x7 conv6 (vggnet.py:56)
// 49 taps exceed the implicit-GEMM tap table, and at os32 the layer is a weight-streaming dense GEMM anyway
// (4096 x 25088 weights against a few hundred pixels), so it runs as unfold -> dense GEMM.
//   unfold: u[p][t*C + c] = x[p + off_t][c] (0 outside the image)
//   fold  : dx[q][c] = sum_t du[q - off_t][c*T + t]      (du comes from the [C][T][K] data-gradient weights: c-major)
__global__ void unfold_kernel(const uint4* __restrict__ x, uint4* __restrict__ u, int64_t n, int H, int W, int C8, int KS) {
    const int T = KS * KS, R = KS / 2;
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        const int t = (int)((v / C8) % T);
        const int64_t p = v / ((int64_t)C8 * T);
        const int xw = (int)(p % W), yh = (int)((p / W) % H);
        const int ih = yh + t / KS - R, iw = xw + t % KS - R;
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) val = x[(p + (int64_t)(t / KS - R) * W + (t % KS - R)) * C8 + c8];
        u[v] = val;
    }
}
__global__ void fold_kernel(const h16raw* __restrict__ du, h16raw* __restrict__ dx, int64_t n, int H, int W, int C, int KS) {
    const int T = KS * KS, R = KS / 2;
    GRID_STRIDE(v, n) {
        const int c = (int)(v % C);
        const int64_t q = v / C;
        const int xw = (int)(q % W), yh = (int)((q / W) % H);
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const int ph = yh - (t / KS - R), pw = xw - (t % KS - R);
            if (ph < 0 || ph >= H || pw < 0 || pw >= W) continue;
            const int64_t p = q - (int64_t)(t / KS - R) * W - (t % KS - R);
            acc += h2f(du[p * ((int64_t)C * T) + (int64_t)c * T + t]);
        }
        dx[v] = f2h(acc);
    }
}

// ---------------------------------------------------------------- DIM losses
#define DIM_EPS 1.001e-5f
__device__ __forceinline__ float refine_at(const float* pred, const float* gt, const float* mask, int64_t i) {
    return mask[i] != 0.f ? pred[i] : gt[i];
}
__global__ __launch_bounds__(256) void dim_losses_fwd_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
    const float* __restrict__ fg, const float* __restrict__ bg, const float* __restrict__ img, float* __restrict__ comps,
    float* __restrict__ acc, int B, int H, int W, int64_t p_stride, int64_t frame_stride, int64_t rgb_stride) {
    __shared__ float red[4];
    const int64_t HW = (int64_t)H * W;
    float s_comp = 0.f, s_grad = 0.f, s_cnt = 0.f;
    GRID_STRIDE(v, (int64_t)B * HW) {
        const int64_t b = v / HW, i = v % HW;
        const int x = (int)(i % W), y = (int)(i / W);
        const float* pp = pred + b * p_stride;
        const float* gp = gt + b * frame_stride;
        const float* mp = mask + b * frame_stride;
        const float m = mp[i];
        const float r = m != 0.f ? pp[i] : gp[i];
        float cs = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t o = b * rgb_stride + c * HW + i;
            const float comp = fg[o] * r + bg[o] * (1.f - r);
            if (comps) comps[(b * 3 + c) * HW + i] = fminf(fmaxf(comp, 0.f), 1.f);
            cs += fabsf(comp - img[o]);
        }
        s_comp += cs * m;
        const float rx = x + 1 < W ? refine_at(pp, gp, mp, i + 1) - r : 0.f;
        const float ry = y + 1 < H ? refine_at(pp, gp, mp, i + W) - r : 0.f;
        const float g0 = gp[i];
        const float gx = x + 1 < W ? gp[i + 1] - g0 : 0.f;
        const float gy = y + 1 < H ? gp[i + W] - g0 : 0.f;
        s_grad += fabsf(sqrtf(rx * rx + ry * ry + DIM_EPS) - sqrtf(gx * gx + gy * gy + DIM_EPS)) * m;
        s_cnt += m > DIM_EPS ? 1.f : 0.f;
    }
    s_comp = block_sum_256(s_comp, red);
    s_grad = block_sum_256(s_grad, red);
    s_cnt = block_sum_256(s_cnt, red);
    if (threadIdx.x == 0) {
        atomicAdd(acc + 0, s_comp);         // acc = {comp sum, count, grad sum, count}: two (sum, count) pairs for
        atomicAdd(acc + 1, s_cnt);          // tcvom_loss_finalize
        atomicAdd(acc + 2, s_grad);
        atomicAdd(acc + 3, s_cnt);
    }
}
// d/dpred[i] (mask[i] != 0):  comp term  sum_c sign(comp_c - img_c)(fg_c - bg_c) mask / den_comp
//   grad term: pixel p contributes |mag_r(p) - mag_g(p)| mask(p); r(i) appears in p = i (as the base of both
//   differences), p = i-1 (x difference) and p = i-W (y difference)
__device__ __forceinline__ float grad_term(const float* pp, const float* gp, const float* mp, int64_t p, int x, int y, int W,
                                           int H, float* dbase, float* dxn, float* dyn) {
    // returns mask(p) * sign(mag_r - mag_g) / mag_r pieces:  dbase = -(rx + ry) * k, dxn = rx * k, dyn = ry * k
    const float r = refine_at(pp, gp, mp, p);
    const float rx = x + 1 < W ? refine_at(pp, gp, mp, p + 1) - r : 0.f;
    const float ry = y + 1 < H ? refine_at(pp, gp, mp, p + W) - r : 0.f;
    const float g0 = gp[p];
    const float gx = x + 1 < W ? gp[p + 1] - g0 : 0.f;
    const float gy = y + 1 < H ? gp[p + W] - g0 : 0.f;
    const float mr = sqrtf(rx * rx + ry * ry + DIM_EPS), mg = sqrtf(gx * gx + gy * gy + DIM_EPS);
    const float d = mr - mg;
    const float k = mp[p] * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / mr;
    *dbase = -(rx + ry) * k;
    *dxn = rx * k;
    *dyn = ry * k;
    return k;
}
__global__ __launch_bounds__(256) void dim_losses_bwd_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
    const float* __restrict__ fg, const float* __restrict__ bg, const float* __restrict__ img, const float* __restrict__ acc,
    const float* __restrict__ g_comp, const float* __restrict__ g_grad, float* __restrict__ dpred, int accumulate,
    int B, int H, int W, int64_t p_stride, int64_t frame_stride, int64_t rgb_stride) {
    const int64_t HW = (int64_t)H * W;
    const float cnt = acc[1];
    const float den_comp = fminf(fmaxf(cnt, DIM_EPS), (float)B * 3.f * (float)HW + 1.f);
    const float den_grad = fminf(fmaxf(cnt, DIM_EPS), (float)B * (float)HW + 1.f);
    const float wc = g_comp[0] / den_comp, wg = g_grad[0] / den_grad;
    GRID_STRIDE(v, (int64_t)B * HW) {
        const int64_t b = v / HW, i = v % HW;
        const int x = (int)(i % W), y = (int)(i / W);
        const float* pp = pred + b * p_stride;
        const float* gp = gt + b * frame_stride;
        const float* mp = mask + b * frame_stride;
        float d = 0.f;
        if (mp[i] != 0.f) {
            const float r = pp[i];
            float cs = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int64_t o = b * rgb_stride + c * HW + i;
                const float diff = fg[o] * r + bg[o] * (1.f - r) - img[o];
                cs += (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * (fg[o] - bg[o]);
            }
            d = wc * cs * mp[i];
            float db, dxn, dyn;
            grad_term(pp, gp, mp, i, x, y, W, H, &db, &dxn, &dyn);
            float gsum = db;
            if (x > 0) { grad_term(pp, gp, mp, i - 1, x - 1, y, W, H, &db, &dxn, &dyn); gsum += dxn; }
            if (y > 0) { grad_term(pp, gp, mp, i - W, x, y - 1, W, H, &db, &dxn, &dyn); gsum += dyn; }
            d += wg * gsum;
        }
        float* o = dpred + b * p_stride + i;
        *o = accumulate ? *o + d : d;
    }
}

// ---------------------------------------------------------------- C ABI
#define POOL_ARGS_OK(name) \
    TCVOM_CHECK_ARG(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, name ": bad shape N=%d H=%d W=%d C=%d", N, H, W, C)

extern "C" int tcvom_maxpool2_idx(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && y && idx, "maxpool2_idx: null pointer");
    POOL_ARGS_OK("maxpool2_idx");
    const int64_t n = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_idx_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, (uint2*)idx, n, H / 2, W / 2, C / 8);
    TCVOM_LAUNCH_CHECK("maxpool2_idx");
    return TCVOM_OK;
}
extern "C" int tcvom_unpool2(const void* y, const uint8_t* idx, void* x, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && y && idx, "unpool2: null pointer");
    POOL_ARGS_OK("unpool2");
    const int64_t n = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(unpool2_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)y, (const uint2*)idx, (uint4*)x, n, H / 2, W / 2, C / 8);
    TCVOM_LAUNCH_CHECK("unpool2");
    return TCVOM_OK;
}
extern "C" int tcvom_pick2(const void* x, const uint8_t* idx, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && y && idx, "pick2: null pointer");
    POOL_ARGS_OK("pick2");
    const int64_t n = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(pick2_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint2*)idx, (uint4*)y, n, H / 2, W / 2, C / 8);
    TCVOM_LAUNCH_CHECK("pick2");
    return TCVOM_OK;
}
extern "C" int tcvom_relu_bwd(const void* dz, const void* y, void* dy, int64_t numel, float negative_slope, void* stream) {
    TCVOM_CHECK_ARG(dz && y && dy && numel > 0 && numel % 8 == 0, "relu_bwd: bad args");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(dgrid(numel / 8)), dim3(256), 0, (hipStream_t)stream, (const uint4*)dz, (const uint4*)y, (uint4*)dy, numel / 8,
                       negative_slope);
    TCVOM_LAUNCH_CHECK("relu_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_unfold(const void* x, void* u, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize, void* stream) {
    TCVOM_CHECK_ARG(x && u && N > 0 && H > 0 && W > 0 && C % 8 == 0 && ksize % 2 == 1, "unfold: bad args");
    const int64_t n = (int64_t)N * H * W * ksize * ksize * (C / 8);
    hipLaunchKernelGGL(unfold_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)u, n, H, W, C / 8, ksize);
    TCVOM_LAUNCH_CHECK("unfold");
    return TCVOM_OK;
}
extern "C" int tcvom_fold(const void* du, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize, void* stream) {
    TCVOM_CHECK_ARG(du && dx && N > 0 && H > 0 && W > 0 && C > 0 && ksize % 2 == 1, "fold: bad args");
    const int64_t n = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(fold_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const h16raw*)du, (h16raw*)dx, n, H, W, C, ksize);
    TCVOM_LAUNCH_CHECK("fold");
    return TCVOM_OK;
}
extern "C" int tcvom_dim_losses_fwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg,
                                    const float* img, float* comps, float* acc, int32_t B, int32_t H, int32_t W,
                                    int64_t p_stride, int64_t frame_stride, int64_t rgb_stride, void* stream) {
    TCVOM_CHECK_ARG(pred && gt && mask && fg && bg && img && acc && B > 0 && H > 0 && W > 0, "dim_losses_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, 4 * sizeof(float), st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "dim_losses_fwd: memset");
    hipLaunchKernelGGL(dim_losses_fwd_kernel, dim3(dgrid((int64_t)B * H * W)), dim3(256), 0, st, pred, gt, mask, fg, bg, img, comps, acc,
                       B, H, W, p_stride, frame_stride, rgb_stride);
    TCVOM_LAUNCH_CHECK("dim_losses_fwd");
    return TCVOM_OK;
}
extern "C" int tcvom_dim_losses_bwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg,
                                    const float* img, const float* acc, const float* g_comp, const float* g_grad, float* dpred,
                                    int32_t accumulate, int32_t B, int32_t H, int32_t W, int64_t p_stride, int64_t frame_stride,
                                    int64_t rgb_stride, void* stream) {
    TCVOM_CHECK_ARG(pred && gt && mask && fg && bg && img && acc && g_comp && g_grad && dpred && B > 0 && H > 0 && W > 0,
                    "dim_losses_bwd: bad args");
    hipLaunchKernelGGL(dim_losses_bwd_kernel, dim3(dgrid((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, pred, gt, mask, fg, bg,
                       img, acc, g_comp, g_grad, dpred, accumulate, B, H, W, p_stride, frame_stride, rgb_stride);
    TCVOM_LAUNCH_CHECK("dim_losses_bwd");
    return TCVOM_OK;
}
