// Element-wise kernels of the IndexNet base (NHWC bf16, 16-byte accesses, fp32 math):
//   index_pool : DepthwiseM2OIndexBlock's normalisation + the indexed pooling of the encoder in one pass
//                (models/Index/hlindex.py:149-168, models/Index/net.py:203-205): from the four branch outputs x_k of a 2x2
//                cell, y_k = sigmoid(x_k), z = softmax_k(y); sub-pixel k = (k / 2, k % 2) of the cell (pixel shuffle):
//                    xe = z_k * l        (idx_en * l: the skip feature the decoder reads)
//                    pooled = sum_k xe   (4 * avg_pool2d)
//                    idx_de = y_k        (decoder indices)
//   index_up   : IndexedUpsamlping's input (models/Index/hldecoder.py:128-133): out[..., :C1] = idx_de * nearest_x2(l_encode)
//                (or l_encode itself when the stage has no indices), out[..., C1:] = l_low
// Both with analytic backward kernels (y, z are recomputed from x_k: nothing but the inputs is kept).
#include "common.h"

static int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}
#define GRID_STRIDE(v, n) \
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (n); v += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// one thread = one 2x2 cell x one channel octet
__global__ __launch_bounds__(256) void index_pool_fwd_kernel(const uint4* __restrict__ x1, const uint4* __restrict__ x2,
                                                             const uint4* __restrict__ x3, const uint4* __restrict__ x4,
                                                             const uint4* __restrict__ l, uint4* __restrict__ xe, uint4* __restrict__ pooled,
                                                             uint4* __restrict__ de, int64_t cells, int h2, int w2, int C8) {
    GRID_STRIDE(v, cells * C8) {
        const int c8 = (int)(v % C8);
        const int64_t cell = v / C8;
        const int j = (int)(cell % w2);
        const int i = (int)((cell / w2) % h2);
        const int64_t n = cell / ((int64_t)w2 * h2);
        const uint4* xs[4] = {x1, x2, x3, x4};
        float y[4][8], sum[8], acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sum[e] = 0.f; acc[e] = 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[8];
            unpack8(xs[k][v], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { y[k][e] = sigmoidf_(f[e]); sum[e] += __expf(y[k][e]); }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t p = ((n * 2 * h2 + 2 * i + (k >> 1)) * (2 * w2) + 2 * j + (k & 1)) * C8 + c8;
            float lf[8], o[8];
            unpack8(l[p], lf);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = __expf(y[k][e]) / sum[e] * lf[e];
                // pooled sums what the decoder will read: the bf16-rounded xe would differ from this fp32 sum by rounding only
                acc[e] += o[e];
            }
            xe[p] = pack8(o);
            de[p] = pack8(y[k]);
        }
        pooled[v] = pack8(acc);
    }
}

__global__ __launch_bounds__(256) void index_pool_bwd_kernel(const uint4* __restrict__ x1, const uint4* __restrict__ x2,
                                                             const uint4* __restrict__ x3, const uint4* __restrict__ x4,
                                                             const uint4* __restrict__ l, const uint4* __restrict__ dxe,
                                                             const uint4* __restrict__ dpooled, const uint4* __restrict__ dde,
                                                             uint4* __restrict__ dx1, uint4* __restrict__ dx2, uint4* __restrict__ dx3,
                                                             uint4* __restrict__ dx4, uint4* __restrict__ dl, int64_t cells, int h2, int w2,
                                                             int C8) {
    GRID_STRIDE(v, cells * C8) {
        const int c8 = (int)(v % C8);
        const int64_t cell = v / C8;
        const int j = (int)(cell % w2);
        const int i = (int)((cell / w2) % h2);
        const int64_t n = cell / ((int64_t)w2 * h2);
        const uint4* xs[4] = {x1, x2, x3, x4};
        uint4* dxs[4] = {dx1, dx2, dx3, dx4};
        float y[4][8], z[4][8], sum[8], gp[8], dz[4][8], dot[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sum[e] = 0.f; dot[e] = 0.f; }
        if (dpooled) unpack8(dpooled[v], gp);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gp[e] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[8];
            unpack8(xs[k][v], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { y[k][e] = sigmoidf_(f[e]); z[k][e] = __expf(y[k][e]); sum[e] += z[k][e]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t p = ((n * 2 * h2 + 2 * i + (k >> 1)) * (2 * w2) + 2 * j + (k & 1)) * C8 + c8;
            float lf[8], g[8], o[8];
            unpack8(l[p], lf);
            if (dxe) unpack8(dxe[p], g);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                z[k][e] /= sum[e];
                const float gt = g[e] + gp[e];                     // xe feeds the skip AND the pooled output
                o[e] = gt * z[k][e];
                dz[k][e] = gt * lf[e];
                dot[e] += z[k][e] * dz[k][e];
            }
            dl[p] = pack8(o);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t p = ((n * 2 * h2 + 2 * i + (k >> 1)) * (2 * w2) + 2 * j + (k & 1)) * C8 + c8;
            float gd[8], o[8];
            if (dde) unpack8(dde[p], gd);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) gd[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dy = z[k][e] * (dz[k][e] - dot[e]) + gd[e];
                o[e] = dy * y[k][e] * (1.f - y[k][e]);
            }
            dxs[k][v] = pack8(o);
        }
    }
}

// out [N][H][W][C1 + C2]; enc [N][H/2][W/2][C1] with idx [N][H][W][C1], or (idx == NULL) enc [N][H][W][C1]
__global__ __launch_bounds__(256) void index_up_fwd_kernel(const uint4* __restrict__ enc, const uint4* __restrict__ idx,
                                                           const uint4* __restrict__ low, uint4* __restrict__ out, int64_t pixels, int H,
                                                           int W, int C1_8, int C2_8) {
    const int CT = C1_8 + C2_8;
    GRID_STRIDE(v, pixels * CT) {
        const int c8 = (int)(v % CT);
        const int64_t p = v / CT;
        if (c8 >= C1_8) { out[v] = low[p * C2_8 + (c8 - C1_8)]; continue; }
        if (!idx) { out[v] = enc[p * C1_8 + c8]; continue; }
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const int64_t n = p / ((int64_t)W * H);
        float a[8], b[8];
        unpack8(enc[((n * (H / 2) + h / 2) * (W / 2) + w / 2) * C1_8 + c8], a);
        unpack8(idx[p * C1_8 + c8], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= b[e];
        out[v] = pack8(a);
    }
}

// one thread = one 2x2 cell (or one pixel when idx == NULL) x one channel octet of the enc part; the low part is copied out
__global__ __launch_bounds__(256) void index_up_bwd_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ enc,
                                                           const uint4* __restrict__ idx, uint4* __restrict__ denc, uint4* __restrict__ didx,
                                                           uint4* __restrict__ dlow, int64_t pixels, int H, int W, int C1_8, int C2_8) {
    const int CT = C1_8 + C2_8;
    GRID_STRIDE(v, pixels * CT) {
        const int c8 = (int)(v % CT);
        const int64_t p = v / CT;
        if (c8 >= C1_8) { dlow[p * C2_8 + (c8 - C1_8)] = dout[v]; continue; }
        if (!idx) { denc[p * C1_8 + c8] = dout[v]; continue; }
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        if ((h | w) & 1) continue;                                   // the cell's top-left pixel does the cell
        const int64_t n = p / ((int64_t)W * H);
        float e0[8], acc[8];
        const int64_t pe = ((n * (H / 2) + h / 2) * (W / 2) + w / 2) * C1_8 + c8;
        unpack8(enc[pe], e0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t q = p + (int64_t)(k >> 1) * W + (k & 1);
            float g[8], b[8], o[8];
            unpack8(dout[q * CT + c8], g);
            unpack8(idx[q * C1_8 + c8], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[e] += g[e] * b[e]; o[e] = g[e] * e0[e]; }
            didx[q * C1_8 + c8] = pack8(o);
        }
        denc[pe] = pack8(acc);
    }
}

extern "C" int tcvom_index_pool_fwd(const void* x1, const void* x2, const void* x3, const void* x4, const void* l, void* xe,
                                    void* pooled, void* idx_de, int32_t N, int32_t h2, int32_t w2, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x1 && x2 && x3 && x4 && l && xe && pooled && idx_de && N > 0 && h2 > 0 && w2 > 0 && C >= 8 && C % 8 == 0,
                    "index_pool_fwd: bad args (C=%d)", C);
    const int64_t cells = (int64_t)N * h2 * w2;
    hipLaunchKernelGGL(index_pool_fwd_kernel, dim3(grid_for(cells * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const uint4*)x1,
                       (const uint4*)x2, (const uint4*)x3, (const uint4*)x4, (const uint4*)l, (uint4*)xe, (uint4*)pooled, (uint4*)idx_de,
                       cells, h2, w2, C / 8);
    TCVOM_LAUNCH_CHECK("index_pool_fwd");
    return TCVOM_OK;
}

extern "C" int tcvom_index_pool_bwd(const void* x1, const void* x2, const void* x3, const void* x4, const void* l, const void* dxe,
                                    const void* dpooled, const void* dde, void* dx1, void* dx2, void* dx3, void* dx4, void* dl,
                                    int32_t N, int32_t h2, int32_t w2, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x1 && x2 && x3 && x4 && l && dx1 && dx2 && dx3 && dx4 && dl && N > 0 && h2 > 0 && w2 > 0 && C >= 8 && C % 8 == 0,
                    "index_pool_bwd: bad args (C=%d)", C);
    const int64_t cells = (int64_t)N * h2 * w2;
    hipLaunchKernelGGL(index_pool_bwd_kernel, dim3(grid_for(cells * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const uint4*)x1,
                       (const uint4*)x2, (const uint4*)x3, (const uint4*)x4, (const uint4*)l, (const uint4*)dxe, (const uint4*)dpooled,
                       (const uint4*)dde, (uint4*)dx1, (uint4*)dx2, (uint4*)dx3, (uint4*)dx4, (uint4*)dl, cells, h2, w2, C / 8);
    TCVOM_LAUNCH_CHECK("index_pool_bwd");
    return TCVOM_OK;
}

extern "C" int tcvom_index_up_fwd(const void* enc, const void* idx, const void* low, void* out, int32_t N, int32_t H, int32_t W,
                                  int32_t C1, int32_t C2, void* stream) {
    TCVOM_CHECK_ARG(enc && low && out && N > 0 && H > 0 && W > 0 && C1 >= 8 && C1 % 8 == 0 && C2 >= 8 && C2 % 8 == 0, "index_up_fwd: bad args");
    TCVOM_CHECK_ARG(!idx || (H % 2 == 0 && W % 2 == 0), "index_up_fwd: the indexed stage doubles the resolution (H=%d W=%d)", H, W);
    const int64_t pixels = (int64_t)N * H * W;
    hipLaunchKernelGGL(index_up_fwd_kernel, dim3(grid_for(pixels * ((C1 + C2) / 8))), dim3(256), 0, (hipStream_t)stream, (const uint4*)enc,
                       (const uint4*)idx, (const uint4*)low, (uint4*)out, pixels, H, W, C1 / 8, C2 / 8);
    TCVOM_LAUNCH_CHECK("index_up_fwd");
    return TCVOM_OK;
}

extern "C" int tcvom_index_up_bwd(const void* dout, const void* enc, const void* idx, void* denc, void* didx, void* dlow, int32_t N,
                                  int32_t H, int32_t W, int32_t C1, int32_t C2, void* stream) {
    TCVOM_CHECK_ARG(dout && enc && denc && dlow && N > 0 && H > 0 && W > 0 && C1 >= 8 && C1 % 8 == 0 && C2 >= 8 && C2 % 8 == 0, "index_up_bwd: bad args");
    TCVOM_CHECK_ARG(!idx || (didx && H % 2 == 0 && W % 2 == 0), "index_up_bwd: bad indexed-stage args");
    const int64_t pixels = (int64_t)N * H * W;
    hipLaunchKernelGGL(index_up_bwd_kernel, dim3(grid_for(pixels * ((C1 + C2) / 8))), dim3(256), 0, (hipStream_t)stream, (const uint4*)dout,
                       (const uint4*)enc, (const uint4*)idx, (uint4*)denc, (uint4*)didx, (uint4*)dlow, pixels, H, W, C1 / 8, C2 / 8);
    TCVOM_LAUNCH_CHECK("index_up_bwd");
    return TCVOM_OK;
}

// ---------------------------------------------------------------- pred[1]: nn.Conv2d(1, 1, 5, padding=2, bias=False) on a fp32 map
// (models/Index/net.py:21; the last layer of the decoder: its input is the ONE-channel output of pred[0] = conv + BN + ReLU6).
// x, y: fp32 [N][H][W]; w: fp32 [25].  flip: taps reversed (the data gradient).  One thread per output pixel; the 25 inputs of
// neighbouring threads overlap in L1.
__global__ __launch_bounds__(256) void conv5x5_c1_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                         int64_t pixels, int H, int W, int flip) {
    float wt[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) wt[t] = w[flip ? 24 - t : t];
    GRID_STRIDE(p, pixels) {
        const int xw = (int)(p % W);
        const int yh = (int)((p / W) % H);
        const int64_t base = p - (int64_t)yh * W - xw;               // first pixel of the sample
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const int ih = yh + t / 5 - 2, iw = xw + t % 5 - 2;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) acc += wt[t] * x[base + (int64_t)ih * W + iw];
        }
        y[p] = acc;
    }
}
// dw[t] += sum_p dy[p] * x[p + off_t]: per-thread accumulators, wave + block reduction, one atomic per tap and block
__global__ __launch_bounds__(256) void conv5x5_c1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dw, int64_t pixels, int H, int W) {
    __shared__ float red[4];
    float acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = 0.f;
    GRID_STRIDE(p, pixels) {
        const int xw = (int)(p % W);
        const int yh = (int)((p / W) % H);
        const int64_t base = p - (int64_t)yh * W - xw;
        const float g = dy[p];
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const int ih = yh + t / 5 - 2, iw = xw + t % 5 - 2;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) acc[t] += g * x[base + (int64_t)ih * W + iw];
        }
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const float s = block_sum_256(acc[t], red);
        if (threadIdx.x == 0) atomicAdd(dw + t, s);
    }
}

extern "C" int tcvom_conv5x5_c1(const float* x, const float* w, float* y, int32_t N, int32_t H, int32_t W, int32_t flip, void* stream) {
    TCVOM_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0, "conv5x5_c1: bad args");
    const int64_t pixels = (int64_t)N * H * W;
    hipLaunchKernelGGL(conv5x5_c1_kernel, dim3(grid_for(pixels)), dim3(256), 0, (hipStream_t)stream, x, w, y, pixels, H, W, flip);
    TCVOM_LAUNCH_CHECK("conv5x5_c1");
    return TCVOM_OK;
}

extern "C" int tcvom_conv5x5_c1_wgrad(const float* dy, const float* x, float* dw, int32_t N, int32_t H, int32_t W, void* stream) {
    TCVOM_CHECK_ARG(dy && x && dw && N > 0 && H > 0 && W > 0, "conv5x5_c1_wgrad: bad args");
    const int64_t pixels = (int64_t)N * H * W;
    if (hipMemsetAsync(dw, 0, 25 * sizeof(float), (hipStream_t)stream) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "conv5x5_c1_wgrad: memset");
    int blocks = grid_for(pixels);
    if (blocks > 1024) blocks = 1024;                                 // 25 atomics per block
    hipLaunchKernelGGL(conv5x5_c1_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dw, pixels, H, W);
    TCVOM_LAUNCH_CHECK("conv5x5_c1_wgrad");
    return TCVOM_OK;
}
