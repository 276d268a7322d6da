// Depthwise 3x3 convolution (groups = channels) of the IndexNet / MobileNetV2 blocks, NHWC bf16, stride 1, any dilation:
// models/Index/net.py:38-61 (InvertedResidual: the 3x3 of every block, run UNPADDED on the block input that `fixed_padding`
// enlarged by the dilation on each side) and models/Index/hlaspp.py:38-46 (the dilated 3x3 of the ASPP branches, padding =
// dilation).  One output element needs 9 inputs of its own channel: there is no reduction over channels, hence no MFMA --
// the kernels are HBM / L2 streaming kernels in the thread layout of the BatchNorm kernels (norm.hip): a thread owns ONE
// channel octet (its 9 x 8 fp32 weights live in registers) and walks pixels, every access is a 16-byte load / store, the 9
// neighbours of consecutive pixels hit L1 / L2.
//   forward  : y = conv(x, w) + per-block partial (sum, sum of squares) per channel for the BatchNorm that follows
//              (same [groups][2][C] layout as the conv engine's epilogue statistics)
//   data grad: the same kernel on dy with the taps reversed and the complementary padding (2 * dil - pad)
//   weight grad: dw[t][c] = sum_p dy[p][c] x[p + off_t][c], block partial sums + fp32 atomics
#include "common.h"

// x [F][N][H][W][C], y [F][N][OH][OW][C]; w fp32 [9][C] (tap-major); F = frames of a frame-batched call (blockIdx.y), every
// frame has its own statistics group range.  flip: use tap 8 - t (data gradient).
__global__ __launch_bounds__(256) void dw3x3_kernel(const uint4* __restrict__ x, const float* __restrict__ w, uint4* __restrict__ y,
                                                    float* __restrict__ stats, int N, int H, int W, int OH, int OW, int C8, int C,
                                                    int dil, int pad, int rows_per_block, int flip) {
    extern __shared__ float red[];                                   // [2][256][8] (statistics only)
    const int tid = threadIdx.x;
    const int oct = tid % C8, prow = tid / C8, RP = 256 / C8;
    const int64_t P = (int64_t)N * OH * OW;
    x += (int64_t)blockIdx.y * N * H * W * C8;
    y += (int64_t)blockIdx.y * P * C8;
    float wt[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float* q = w + (flip ? 8 - t : t) * C + oct * 8;
        const float4 a = reinterpret_cast<const float4*>(q)[0], b = reinterpret_cast<const float4*>(q)[1];
        wt[t][0] = a.x; wt[t][1] = a.y; wt[t][2] = a.z; wt[t][3] = a.w; wt[t][4] = b.x; wt[t][5] = b.y; wt[t][6] = b.z; wt[t][7] = b.w;
    }
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    if (prow < RP) {
        for (int64_t p = pbeg + prow; p < pend; p += RP) {
            const int ow = (int)(p % OW);
            const int oh = (int)((p / OW) % OH);
            const int n = (int)(p / ((int64_t)OW * OH));
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            uint4 v[9];
            bool ok[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {                            // all 9 loads in flight together
                const int ih = oh + (t / 3) * dil - pad, iw = ow + (t % 3) * dil - pad;
                ok[t] = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                v[t] = ok[t] ? x[(((int64_t)n * H + ih) * W + iw) * C8 + oct] : uint4{0, 0, 0, 0};
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float f[8];
                unpack8(v[t], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += f[k] * wt[t][k];
            }
            y[p * C8 + oct] = pack8(acc);
            if (stats) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1[k] += acc[k]; s2[k] += acc[k] * acc[k]; }
            }
        }
    }
    if (!stats) return;
    float* r1 = red;
    float* r2 = red + 256 * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { r1[tid * 8 + k] = s1[k]; r2[tid * 8 + k] = s2[k]; }
    __syncthreads();
    if (tid < C8) {
        float a1[8], a2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a1[k] = 0.f; a2[k] = 0.f; }
        for (int r = 0; r < RP; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { a1[k] += r1[(r * C8 + tid) * 8 + k]; a2[k] += r2[(r * C8 + tid) * 8 + k]; }
        }
        float* po = stats + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
        for (int k = 0; k < 8; ++k) { po[tid * 8 + k] = a1[k]; po[C + tid * 8 + k] = a2[k]; }
    }
}

// dw[t][c] += sum over the block's pixels of dy[p][c] * x[p + off_t][c]   (all frames add into the same dw)
__global__ __launch_bounds__(256) void dw3x3_wgrad_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ x, float* __restrict__ dw,
                                                          int N, int H, int W, int OH, int OW, int C8, int C, int dil, int pad,
                                                          int rows_per_block) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const int oct = tid % C8, prow = tid / C8, RP = 256 / C8;
    const int64_t P = (int64_t)N * OH * OW;
    x += (int64_t)blockIdx.y * N * H * W * C8;
    dy += (int64_t)blockIdx.y * P * C8;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    if (prow < RP) {
        for (int64_t p = pbeg + prow; p < pend; p += RP) {
            const int ow = (int)(p % OW);
            const int oh = (int)((p / OW) % OH);
            const int n = (int)(p / ((int64_t)OW * OH));
            float g[8];
            unpack8(dy[p * C8 + oct], g);
            uint4 v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ih = oh + (t / 3) * dil - pad, iw = ow + (t % 3) * dil - pad;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                v[t] = ok ? x[(((int64_t)n * H + ih) * W + iw) * C8 + oct] : uint4{0, 0, 0, 0};
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float f[8];
                unpack8(v[t], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[t][k] += g[k] * f[k];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = acc[t][k];
        __syncthreads();
        if (tid < C8) {
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < RP; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += red[(r * C8 + tid) * 8 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(dw + t * C + tid * 8 + k, a[k]);
        }
    }
}

static int dw_rows_per_block(int64_t pixels, int C) {
    const int rp = 256 / (C / 8);
    int64_t rows = (int64_t)rp * 8;                                  // ~8 pixels per thread
    if ((pixels + rows - 1) / rows > 4096) rows = ((pixels + 4095) / 4096 + rp - 1) / rp * rp;
    return (int)rows;
}

extern "C" int tcvom_dw3x3_stats_groups(int64_t out_pixels, int32_t C) {
    if (out_pixels <= 0 || C < 8 || C % 8 != 0 || C > 2048) return 0;
    return cdiv(out_pixels, dw_rows_per_block(out_pixels, C));
}

extern "C" int tcvom_dw3x3(const void* x, const float* w, void* y, float* stats, int32_t N, int32_t H, int32_t W, int32_t C,
                           int32_t dilation, int32_t pad, int32_t flip, int32_t nframes, void* stream) {
    TCVOM_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0 && nframes >= 1, "dw3x3: bad args");
    TCVOM_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= 2048 && dilation >= 1 && pad >= 0, "dw3x3: C=%d dilation=%d pad=%d", C, dilation, pad);
    TCVOM_CHECK_ARG(((uintptr_t)w % 16) == 0, "dw3x3: weights must be 16-byte aligned");
    const int OH = H + 2 * pad - 2 * dilation, OW = W + 2 * pad - 2 * dilation;
    TCVOM_CHECK_ARG(OH > 0 && OW > 0, "dw3x3: empty output (%d x %d)", OH, OW);
    const int64_t P = (int64_t)N * OH * OW;
    const int rpb = dw_rows_per_block(P, C);
    const dim3 grid(cdiv(P, rpb), nframes);
    hipLaunchKernelGGL(dw3x3_kernel, grid, dim3(256), stats ? 2 * 256 * 8 * sizeof(float) : 0, (hipStream_t)stream, (const uint4*)x, w,
                       (uint4*)y, stats, N, H, W, OH, OW, C / 8, C, dilation, pad, rpb, flip);
    TCVOM_LAUNCH_CHECK("dw3x3");
    return TCVOM_OK;
}

extern "C" int tcvom_dw3x3_wgrad(const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t C,
                                 int32_t dilation, int32_t pad, int32_t nframes, void* stream) {
    TCVOM_CHECK_ARG(dy && x && dw && N > 0 && H > 0 && W > 0 && nframes >= 1, "dw3x3_wgrad: bad args");
    TCVOM_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= 2048 && dilation >= 1 && pad >= 0, "dw3x3_wgrad: C=%d dilation=%d pad=%d", C, dilation, pad);
    const int OH = H + 2 * pad - 2 * dilation, OW = W + 2 * pad - 2 * dilation;
    TCVOM_CHECK_ARG(OH > 0 && OW > 0, "dw3x3_wgrad: empty output (%d x %d)", OH, OW);
    const int64_t P = (int64_t)N * OH * OW;
    const int rp = 256 / (C / 8);
    int64_t rows = (int64_t)rp * 32;                                 // longer runs: every block ends with 9 * C atomics
    if ((P + rows - 1) / rows > 1024) rows = ((P + 1023) / 1024 + rp - 1) / rp * rp;
    if (hipMemsetAsync(dw, 0, sizeof(float) * 9 * C, (hipStream_t)stream) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "dw3x3_wgrad: memset");
    const dim3 grid(cdiv(P, rows), nframes);
    hipLaunchKernelGGL(dw3x3_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dy, (const uint4*)x, dw, N, H, W, OH, OW,
                       C / 8, C, dilation, pad, (int)rows);
    TCVOM_LAUNCH_CHECK("dw3x3_wgrad");
    return TCVOM_OK;
}
