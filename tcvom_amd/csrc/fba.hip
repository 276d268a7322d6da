// Kernels of the FBA base (BASELINE config 5: FullModel_VMD('vmn_fba')) that the conv / norm engine does not cover:
//   * MaxPool2d(3, 2, 1) of the ResNet stem                          models/FBA/resnet_GN_WS.py:101
//   * pyramid pooling: AdaptiveAvgPool2d(1, 2, 3, 6) and the bilinear resize back    models/VMN/VMN_FBA.py:23-31
//   * bilinear x2 up-sampling (align_corners=False) into a slice of a concat buffer   models/VMN/VMN_FBA.py:37-48
//   * the 1x1 head + clamp / sigmoid + fba_fusion                      models/VMN/VMN_FBA.py:50-57, models/FBA/models.py:246-255
//   * the 11-channel network input: trimap_transform (exact Euclidean distance transform + Gaussian "clicks")
//     utils/utils.py:12-39, models/model.py:71-77, written directly in the 2x2 space-to-depth layout of the stem
// Activations are NHWC bf16; losses-side tensors are NCHW fp32 like the rest of the facade.
#include "common.h"

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)
static int dgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------ MaxPool2d(3, 2, 1)
// idx = position (0..8) of the FIRST maximum inside the window in row-major scan order (PyTorch's choice)
__global__ void maxpool3s2_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, uint8_t* __restrict__ idx, int64_t n,
                                  int H, int W, int OH, int OW, int C8) {
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        int64_t t = v / C8;
        const int j = (int)(t % OW); t /= OW;
        const int i = (int)(t % OH);
        const int64_t nb = t / OH;
        float best[8];
        int bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
        for (int r = 0; r < 3; ++r) {
            const int h = 2 * i - 1 + r;
            if (h < 0 || h >= H) continue;
            for (int s = 0; s < 3; ++s) {
                const int w = 2 * j - 1 + s;
                if (w < 0 || w >= W) continue;
                float f[8];
                unpack8(x[((nb * H + h) * W + w) * C8 + c8], f);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (f[k] > best[k]) { best[k] = f[k]; bi[k] = r * 3 + s; }
            }
        }
        y[v] = pack8(best);
        uint8_t* ip = idx + v * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) ip[k] = (uint8_t)bi[k];
    }
}
// gather form of the backward: every input pixel collects from the (at most 4) windows that contain it
__global__ void maxpool3s2_bwd_kernel(const uint4* __restrict__ dy, const uint8_t* __restrict__ idx, uint4* __restrict__ dx,
                                      int64_t n, int H, int W, int OH, int OW, int C8) {
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        int64_t t = v / C8;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int64_t nb = t / H;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = h / 2; i <= (h + 1) / 2; ++i) {
            if (i >= OH) continue;
            const int r = h - (2 * i - 1);
            for (int j = w / 2; j <= (w + 1) / 2; ++j) {
                if (j >= OW) continue;
                const int pos = r * 3 + (w - (2 * j - 1));
                const int64_t o = ((nb * OH + i) * OW + j) * C8 + c8;
                float g[8];
                unpack8(dy[o], g);
                const uint8_t* ip = idx + o * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (ip[k] == pos) acc[k] += g[k];
            }
        }
        dx[v] = pack8(acc);
    }
}

extern "C" int tcvom_maxpool3s2(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && y && idx && N > 0 && H > 1 && W > 1 && C % 8 == 0, "maxpool3s2: bad args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int64_t n = (int64_t)N * OH * OW * (C / 8);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, idx, n, H, W, OH, OW, C / 8);
    TCVOM_LAUNCH_CHECK("maxpool3s2");
    return TCVOM_OK;
}
extern "C" int tcvom_maxpool3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(dy && dx && idx && N > 0 && H > 1 && W > 1 && C % 8 == 0, "maxpool3s2_bwd: bad args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int64_t n = (int64_t)N * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)dy, idx, (uint4*)dx, n, H, W, OH, OW, C / 8);
    TCVOM_LAUNCH_CHECK("maxpool3s2_bwd");
    return TCVOM_OK;
}

// ------------------------------------------------------------------------------------------ adaptive average pooling
// AdaptiveAvgPool2d(s): bin b covers rows floor(b*h/s) .. ceil((b+1)*h/s) - 1 (bins overlap when s does not divide h)
__device__ __forceinline__ int bin_lo(int b, int n, int s) { return (b * n) / s; }
__device__ __forceinline__ int bin_hi(int b, int n, int s) { return ((b + 1) * n + s - 1) / s; }

// grid (N*s*s, row splits); block = 256 threads = C8 channel octets x 256/C8 pixel lanes; out fp32 [N][s][s][C], zeroed
__global__ __launch_bounds__(256) void adaptive_pool_kernel(const uint4* __restrict__ x, float* __restrict__ out, int h, int w, int C8, int s) {
    const int bin = blockIdx.x % (s * s), nb = blockIdx.x / (s * s);
    const int bi = bin / s, bj = bin % s;
    const int h0 = bin_lo(bi, h, s), h1 = bin_hi(bi, h, s), w0 = bin_lo(bj, w, s), w1 = bin_hi(bj, w, s);
    const int lanes = 256 / C8, c8 = threadIdx.x % C8, pl = threadIdx.x / C8;
    const int bw = w1 - w0;
    const int64_t npix = (int64_t)(h1 - h0) * bw;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t p = (int64_t)blockIdx.y * lanes + pl; p < npix; p += (int64_t)gridDim.y * lanes) {
        const int yy = h0 + (int)(p / bw), xx = w0 + (int)(p % bw);
        float f[8];
        unpack8(x[(((int64_t)nb * h + yy) * w + xx) * C8 + c8], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[k];
    }
    const float inv = 1.f / (float)npix;
    float* o = out + ((int64_t)blockIdx.x * C8 + c8) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(o + k, acc[k] * inv);
}
// All scales of the pyramid in ONE pass over x (the per-scale kernel above reads the 400 MB feature map of a 1080p window once per
// scale: 4 x 177 us): the union of every scale's bin boundaries cuts the map into cells, each cell lies inside or outside any bin; a
// block sums one (sample, cell, row split) and adds the sum / |bin| to every bin of every scale that contains the cell.
struct PoolMulti { float* out[4]; int s[4]; int n; int rows[40], cols[40]; int nr, nc; };
// partial (or NULL): [sample][cell][row split][C] fp32 -- the block's cell sum is STORED there and adaptive_pool_combine_kernel adds the
// cells of every bin (with atomics every block added its 2048 channel sums to up to 4 maps: 8.8 M atomics on 0.3 M addresses per 1080p
// window, which -- not the 400 MB read -- was what the pass took: 317 us)
__global__ __launch_bounds__(256) void adaptive_pool_multi_kernel(const uint4* __restrict__ x, const PoolMulti pm, int h, int w, int C8,
                                                                  float* __restrict__ partial) {
    const int ncell = (pm.nr - 1) * (pm.nc - 1);
    const int cell = blockIdx.x % ncell, nb = blockIdx.x / ncell;
    const int ci = cell / (pm.nc - 1), cj = cell % (pm.nc - 1);
    const int h0 = pm.rows[ci], h1 = pm.rows[ci + 1], w0 = pm.cols[cj], w1 = pm.cols[cj + 1];
    const int lanes = 256 / C8, pl = threadIdx.x / C8;
    const int bw = w1 - w0;
    const int64_t npix = (int64_t)(h1 - h0) * bw;
    for (int c8 = threadIdx.x % C8; c8 < C8; c8 += 256) {           // (C8 <= 256: one pass)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // four pixels per trip, their loads in flight together (one dependent 16-byte load per trip left this pass latency bound:
        // 342 us for the 400 MB map of a 1080p window)
        const int64_t pstep = (int64_t)gridDim.y * lanes;
        int64_t p = (int64_t)blockIdx.y * lanes + pl;
        for (; p + 3 * pstep < npix; p += 4 * pstep) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t pu = p + u * pstep;
                const int yy = h0 + (int)(pu / bw), xx = w0 + (int)(pu % bw);
                v[u] = x[(((int64_t)nb * h + yy) * w + xx) * C8 + c8];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += f[k];
            }
        }
        for (; p < npix; p += pstep) {
            const int yy = h0 + (int)(p / bw), xx = w0 + (int)(p % bw);
            float f[8];
            unpack8(x[(((int64_t)nb * h + yy) * w + xx) * C8 + c8], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += f[k];
        }
        if (partial) {
            float* o = partial + ((((int64_t)blockIdx.x) * gridDim.y + blockIdx.y) * C8 + c8) * 8;
            if (lanes > 1) {                                   // (C8 < 256: the block's pixel lanes hold partial sums of the same channels)
                __shared__ float red[256 * 8];
#pragma unroll
                for (int k = 0; k < 8; ++k) red[k * 256 + threadIdx.x] = acc[k];
                __syncthreads();
                if (pl == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float t = 0.f;
                        for (int l = 0; l < lanes; ++l) t += red[k * 256 + l * C8 + c8];
                        o[k] = t;
                    }
                }
            } else {
                *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
            continue;
        }
        for (int q = 0; q < pm.n; ++q) {
            const int sc = pm.s[q];
            for (int bi = 0; bi < sc; ++bi) {
                const int bh0 = bin_lo(bi, h, sc), bh1 = bin_hi(bi, h, sc);
                if (h0 < bh0 || h1 > bh1) continue;
                for (int bj = 0; bj < sc; ++bj) {
                    const int bw0 = bin_lo(bj, w, sc), bw1 = bin_hi(bj, w, sc);
                    if (w0 < bw0 || w1 > bw1) continue;
                    const float inv = 1.f / (float)((bh1 - bh0) * (bw1 - bw0));
                    float* o = pm.out[q] + ((((int64_t)nb * sc + bi) * sc + bj) * C8 + c8) * 8;
#pragma unroll
                    for (int k = 0; k < 8; ++k) atomicAdd(o + k, acc[k] * inv);
                }
            }
        }
    }
}
// second stage: one block per (sample, scale, bin) sums the partial sums of the cells inside the bin, in a fixed order (deterministic)
__global__ __launch_bounds__(256) void adaptive_pool_combine_kernel(const float* __restrict__ partial, const PoolMulti pm, int h, int w, int C,
                                                                    int split, int nbins_total) {
    const int ncell = (pm.nr - 1) * (pm.nc - 1);
    const int nb = blockIdx.x / nbins_total;
    int b = blockIdx.x % nbins_total, q = 0;
    while (b >= pm.s[q] * pm.s[q]) { b -= pm.s[q] * pm.s[q]; ++q; }
    const int sc = pm.s[q], bi = b / sc, bj = b % sc;
    const int bh0 = bin_lo(bi, h, sc), bh1 = bin_hi(bi, h, sc), bw0 = bin_lo(bj, w, sc), bw1 = bin_hi(bj, w, sc);
    const float inv = 1.f / (float)((bh1 - bh0) * (bw1 - bw0));
    const int c = blockIdx.y * 256 + threadIdx.x;              // grid.y = channel chunks of 256
    if (c >= C) return;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int ci = 0; ci < pm.nr - 1; ++ci) {
        if (pm.rows[ci] < bh0 || pm.rows[ci + 1] > bh1) continue;
        for (int cj = 0; cj < pm.nc - 1; ++cj) {
            if (pm.cols[cj] < bw0 || pm.cols[cj + 1] > bw1) continue;
            const float* p = partial + (((int64_t)nb * ncell + ci * (pm.nc - 1) + cj) * split) * C + c;
            int y = 0;
            for (; y + 3 < split; y += 4) {                     // four loads in flight; the order of the sum is fixed
                t0 += p[(int64_t)y * C]; t1 += p[(int64_t)(y + 1) * C]; t2 += p[(int64_t)(y + 2) * C]; t3 += p[(int64_t)(y + 3) * C];
            }
            for (; y < split; ++y) t0 += p[(int64_t)y * C];
        }
    }
    pm.out[q][(((int64_t)nb * sc + bi) * sc + bj) * C + c] = ((t0 + t1) + (t2 + t3)) * inv;
}
// dx = sum over the scales of dout_s[bin] / |bin| for every bin that contains the pixel
// dx = sum over the scales of dout_s[bin] / |bin| for every bin that contains the pixel.  Inside a CELL of the union partition
// (see adaptive_pool_multi_kernel) the set of bins is the same for every pixel, so dx is CONSTANT per channel there: a block computes
// the 8 values of its channel octets once per (sample, cell) and streams them over the cell's pixels -- a pure write pass.  (Per pixel,
// every 16-byte output re-read up to 16 x 32 bytes of pooled gradients through L2: 742 us for the 400 MB gradient of a 1080p window.)
// `add` (or NULL): a second gradient of x, read with a pixel stride of add_ld8 16-byte chunks and added -- the slice of the pyramid
// concat buffer's gradient that belongs to x (VMN_FBA.py:25-31: x feeds the pooling AND the concat); as two autograd gradients that
// cost a strided 400 MB copy and a 400 MB add per 1080p window
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const PoolMulti pm, uint4* __restrict__ dx, int h, int w, int C8,
                                                                const uint4* __restrict__ add, int add_ld8) {
    const int ncell = (pm.nr - 1) * (pm.nc - 1);
    const int cell = blockIdx.x % ncell, nb = blockIdx.x / ncell;
    const int ci = cell / (pm.nc - 1), cj = cell % (pm.nc - 1);
    const int h0 = pm.rows[ci], h1 = pm.rows[ci + 1], w0 = pm.cols[cj], w1 = pm.cols[cj + 1];
    const int lanes = C8 >= 256 ? 1 : 256 / C8, pl = C8 >= 256 ? 0 : threadIdx.x / C8;
    const int bw = w1 - w0;
    const int npix = (h1 - h0) * bw;
    for (int c8 = C8 >= 256 ? threadIdx.x : threadIdx.x % C8; c8 < C8; c8 += 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < pm.n; ++q) {
            const int sc = pm.s[q];
            for (int bi = 0; bi < sc; ++bi) {
                const int bh0 = bin_lo(bi, h, sc), bh1 = bin_hi(bi, h, sc);
                if (h0 < bh0 || h1 > bh1) continue;
                for (int bj = 0; bj < sc; ++bj) {
                    const int bw0 = bin_lo(bj, w, sc), bw1 = bin_hi(bj, w, sc);
                    if (w0 < bw0 || w1 > bw1) continue;
                    const float inv = 1.f / (float)((bh1 - bh0) * (bw1 - bw0));
                    const float* g = pm.out[q] + ((((int64_t)nb * sc + bi) * sc + bj) * C8 + c8) * 8;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += g[k] * inv;
                }
            }
        }
        const uint4 val = pack8(acc);
        for (int p = blockIdx.y * lanes + pl; p < npix; p += gridDim.y * lanes) {
            const int yy = h0 + p / bw, xx = w0 + p % bw;
            const int64_t pix = ((int64_t)nb * h + yy) * w + xx;
            if (add) {
                float a8[8];
                unpack8(add[pix * add_ld8 + c8], a8);
#pragma unroll
                for (int k = 0; k < 8; ++k) a8[k] += acc[k];
                dx[pix * C8 + c8] = pack8(a8);
            } else dx[pix * C8 + c8] = val;
        }
    }
}

// host: the union of every scale's bin boundaries (sorted, unique) -> cells
static int pool_cells(PoolMulti& pm, const int32_t* scales, int nscales, int h, int w) {
    int rows[64], cols[64], nr = 0, nc = 0;
    for (int q = 0; q < nscales; ++q) {
        const int sc = scales[q];
        for (int b = 0; b < sc; ++b) {
            rows[nr++] = (b * h) / sc; rows[nr++] = ((b + 1) * h + sc - 1) / sc;
            cols[nc++] = (b * w) / sc; cols[nc++] = ((b + 1) * w + sc - 1) / sc;
        }
    }
    auto uniq = [](int* v, int n) {
        for (int i = 1; i < n; ++i) { const int t = v[i]; int j = i - 1; while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; } v[j + 1] = t; }
        int m = 0;
        for (int i = 0; i < n; ++i) if (m == 0 || v[m - 1] != v[i]) v[m++] = v[i];
        return m;
    };
    nr = uniq(rows, nr); nc = uniq(cols, nc);
    if (nr > 40 || nc > 40) return -1;
    for (int i = 0; i < nr; ++i) pm.rows[i] = rows[i];
    for (int i = 0; i < nc; ++i) pm.cols[i] = cols[i];
    pm.nr = nr; pm.nc = nc;
    return 0;
}

extern "C" int tcvom_adaptive_avgpool(const void* x, float* out, int32_t N, int32_t h, int32_t w, int32_t C, int32_t s, void* stream) {
    TCVOM_CHECK_ARG(x && out && N > 0 && h >= s && w >= s && s >= 1 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, "adaptive_avgpool: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)N * s * s * C, st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "adaptive_avgpool: memset failed");
    const int64_t npix = (int64_t)((h + s - 1) / s + 1) * ((w + s - 1) / s + 1);
    int split = (int)((npix * (C / 8) + 256 * 64 - 1) / (256 * 64));
    if (split > 64) split = 64;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(adaptive_pool_kernel, dim3(N * s * s, split), dim3(256), 0, st, (const uint4*)x, out, h, w, C / 8, s);
    TCVOM_LAUNCH_CHECK("adaptive_avgpool");
    return TCVOM_OK;
}
// floats of scratch tcvom_adaptive_avgpool_multi_ws needs (0: too many bin boundaries)
static int pool_cells(PoolMulti& pm, const int32_t* scales, int nscales, int h, int w);
static int pool_split(int cells) { int split = (1024 + cells - 1) / cells; return split > 16 ? 16 : split; }
extern "C" int tcvom_adaptive_avgpool_scratch_floats(const int32_t* scales, int32_t nscales, int32_t N, int32_t h, int32_t w, int32_t C) {
    PoolMulti pm;
    if (!scales || nscales < 1 || nscales > 4 || pool_cells(pm, scales, nscales, h, w) != 0) return 0;
    const int cells = (pm.nr - 1) * (pm.nc - 1) * N;
    const long long nfl = (long long)cells * pool_split(cells) * C;
    return nfl < (1ll << 31) ? (int)nfl : 0;
}
static int adaptive_avgpool_multi_impl(const void* x, float* const* outs, const int32_t* scales, int32_t nscales, float* scratch, int32_t N,
                                       int32_t h, int32_t w, int32_t C, void* stream);
extern "C" int tcvom_adaptive_avgpool_multi(const void* x, float* const* outs, const int32_t* scales, int32_t nscales, int32_t N, int32_t h,
                                            int32_t w, int32_t C, void* stream) {
    return adaptive_avgpool_multi_impl(x, outs, scales, nscales, nullptr, N, h, w, C, stream);
}
// two-stage form: per-(cell, row split) sums into `scratch` (tcvom_adaptive_avgpool_scratch_floats floats), then one combine launch:
// no atomics, no memsets of the outputs, deterministic
extern "C" int tcvom_adaptive_avgpool_multi_ws(const void* x, float* const* outs, const int32_t* scales, int32_t nscales, float* scratch,
                                               int32_t N, int32_t h, int32_t w, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(scratch, "adaptive_avgpool_multi_ws: null scratch");
    return adaptive_avgpool_multi_impl(x, outs, scales, nscales, scratch, N, h, w, C, stream);
}
static int adaptive_avgpool_multi_impl(const void* x, float* const* outs, const int32_t* scales, int32_t nscales, float* scratch, int32_t N,
                                       int32_t h, int32_t w, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && outs && scales && nscales >= 1 && nscales <= 4 && N > 0 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0,
                    "adaptive_avgpool_multi: bad args");
    hipStream_t st = (hipStream_t)stream;
    PoolMulti pm;
    pm.n = nscales;
    for (int q = 0; q < 4; ++q) {
        const int i = q < nscales ? q : 0;
        TCVOM_CHECK_ARG(outs[i] && scales[i] >= 1 && scales[i] <= 8 && h >= scales[i] && w >= scales[i], "adaptive_avgpool_multi: scale %d", scales[i]);
        pm.out[q] = outs[i];
        pm.s[q] = scales[i];
        if (!scratch && q < nscales && hipMemsetAsync(outs[i], 0, sizeof(float) * (size_t)N * scales[i] * scales[i] * C, st) != hipSuccess)
            return tcvom_fail(TCVOM_ERR_LAUNCH, "adaptive_avgpool_multi: memset failed");
    }
    TCVOM_CHECK_ARG(pool_cells(pm, scales, nscales, h, w) == 0, "adaptive_avgpool_multi: too many bin boundaries");
    const int nr = pm.nr, nc = pm.nc;
    // ~1024 blocks: row splits per cell
    const int cells = (nr - 1) * (nc - 1) * N;
    const int split = pool_split(cells);
    hipLaunchKernelGGL(adaptive_pool_multi_kernel, dim3((unsigned)cells, (unsigned)split), dim3(256), 0, st, (const uint4*)x, pm, h, w, C / 8, scratch);
    if (scratch) {
        int nbins = 0;
        for (int q = 0; q < nscales; ++q) nbins += scales[q] * scales[q];
        hipLaunchKernelGGL(adaptive_pool_combine_kernel, dim3((unsigned)(N * nbins), (unsigned)cdiv(C, 256)), dim3(256), 0, st, (const float*)scratch, pm, h, w, C, split, nbins);
    }
    TCVOM_LAUNCH_CHECK("adaptive_avgpool_multi");
    return TCVOM_OK;
}
static int adaptive_avgpool_bwd_impl(const float* const* dout, const int32_t* scales, int32_t nscales, void* dx, const void* add,
                                     int32_t add_ld, int32_t N, int32_t h, int32_t w, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(dout && scales && dx && nscales >= 1 && nscales <= 4 && N > 0 && C % 8 == 0 && (256 % (C / 8) == 0 || (C / 8) % 256 == 0),
                    "adaptive_avgpool_bwd: bad args");
    TCVOM_CHECK_ARG((256 % (C / 8) == 0 || (C / 8) % 256 == 0) && C <= 16384, "adaptive_avgpool_bwd: C=%d", C);
    PoolMulti pm;
    pm.n = nscales;
    for (int q = 0; q < 4; ++q) {
        const int i = q < nscales ? q : 0;
        TCVOM_CHECK_ARG(dout[i] && scales[i] >= 1 && scales[i] <= 8 && h >= scales[i] && w >= scales[i], "adaptive_avgpool_bwd: scale %d", scales[i]);
        pm.out[q] = const_cast<float*>(dout[i]);
        pm.s[q] = scales[i];
    }
    TCVOM_CHECK_ARG(pool_cells(pm, scales, nscales, h, w) == 0, "adaptive_avgpool_bwd: too many bin boundaries");
    const int cells = (pm.nr - 1) * (pm.nc - 1) * N;
    int split = (2048 + cells - 1) / cells;
    if (split > 32) split = 32;
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel, dim3((unsigned)cells, (unsigned)split), dim3(256), 0, (hipStream_t)stream, pm, (uint4*)dx, h, w, C / 8,
                       (const uint4*)add, add_ld / 8);
    TCVOM_LAUNCH_CHECK("adaptive_avgpool_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_adaptive_avgpool_bwd(const float* const* dout, const int32_t* scales, int32_t nscales, void* dx, int32_t N, int32_t h,
                                          int32_t w, int32_t C, void* stream) {
    return adaptive_avgpool_bwd_impl(dout, scales, nscales, dx, nullptr, 0, N, h, w, C, stream);
}
// dx = (pooling gradient) + add[..., :C], `add` NHWC with add_ld channels per pixel
extern "C" int tcvom_adaptive_avgpool_bwd_add(const float* const* dout, const int32_t* scales, int32_t nscales, void* dx, const void* add,
                                              int32_t add_ld, int32_t N, int32_t h, int32_t w, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(add && add_ld % 8 == 0 && add_ld >= C && ((uintptr_t)add % 16) == 0, "adaptive_avgpool_bwd_add: bad addend");
    return adaptive_avgpool_bwd_impl(dout, scales, nscales, dx, add, add_ld, N, h, w, C, stream);
}

// ------------------------------------------------------------------------------------------ bilinear resize
// F.interpolate(mode='bilinear', align_corners=False): source coordinate max(0, (dst + 0.5) * in/out - 0.5)
struct Lerp { int i0, i1; float l; };
__device__ __forceinline__ Lerp lerp_of(int d, float scale, int n_in) {
    float s = ((float)d + 0.5f) * scale - 0.5f;
    if (s < 0.f) s = 0.f;
    Lerp r;
    r.i0 = (int)s;
    if (r.i0 > n_in - 1) r.i0 = n_in - 1;
    r.i1 = r.i0 + 1 < n_in ? r.i0 + 1 : n_in - 1;
    r.l = s - (float)r.i0;
    return r;
}
// weight of source index `i` in destination index d
__device__ __forceinline__ float lerp_weight(int d, int i, float scale, int n_in) {
    const Lerp r = lerp_of(d, scale, n_in);
    return (r.i0 == i ? 1.f - r.l : 0.f) + (r.i1 == i ? r.l : 0.f);
}

// src [N][hs][ws][ld_src] (channels c_src..+C) -> dst [N][hd][wd][ld_dst] (channels c_dst..+C)
__global__ void bilinear_kernel(const h16raw* __restrict__ src, h16raw* __restrict__ dst, int64_t n, int hs, int ws, int hd, int wd,
                                int C8, int ld_src, int c_src, int ld_dst, int c_dst, float sh, float sw) {
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        int64_t t = v / C8;
        const int X = (int)(t % wd); t /= wd;
        const int Y = (int)(t % hd);
        const int64_t nb = t / hd;
        const Lerp ly = lerp_of(Y, sh, hs), lx = lerp_of(X, sw, ws);
        const h16raw* sb = src + nb * hs * ws * ld_src + c_src + c8 * 8;
        float a[8], b[8], c[8], d[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i0 * ws + lx.i0) * ld_src), a);
        unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i0 * ws + lx.i1) * ld_src), b);
        unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i1 * ws + lx.i0) * ld_src), c);
        unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i1 * ws + lx.i1) * ld_src), d);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o[k] = (1.f - ly.l) * ((1.f - lx.l) * a[k] + lx.l * b[k]) + ly.l * ((1.f - lx.l) * c[k] + lx.l * d[k]);
        *reinterpret_cast<uint4*>(dst + ((nb * hd + Y) * wd + X) * ld_dst + c_dst + c8 * 8) = pack8(o);
    }
}
// gather form of the x2 backward: source pixel (i, j) collects from destination rows 2i-2 .. 2i+2 (weights from lerp_weight,
// which reproduces the border clamping of the forward)
__global__ void bilinear_up2_bwd_kernel(const h16raw* __restrict__ ddst, h16raw* __restrict__ dsrc, int64_t n, int hs, int ws, int C8,
                                        int ld_dst, int c_dst) {
    const int hd = 2 * hs, wd = 2 * ws;
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        int64_t t = v / C8;
        const int j = (int)(t % ws); t /= ws;
        const int i = (int)(t % hs);
        const int64_t nb = t / hs;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int Y = 2 * i - 2; Y <= 2 * i + 2; ++Y) {
            if (Y < 0 || Y >= hd) continue;
            const float wy = lerp_weight(Y, i, 0.5f, hs);
            if (wy == 0.f) continue;
            for (int X = 2 * j - 2; X <= 2 * j + 2; ++X) {
                if (X < 0 || X >= wd) continue;
                const float wgt = wy * lerp_weight(X, j, 0.5f, ws);
                if (wgt == 0.f) continue;
                float g[8];
                unpack8(*reinterpret_cast<const uint4*>(ddst + ((nb * hd + Y) * wd + X) * ld_dst + c_dst + c8 * 8), g);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += wgt * g[k];
            }
        }
        *reinterpret_cast<uint4*>(dsrc + v * 8) = pack8(acc);
    }
}
// backward onto a SMALL source (the s x s pyramid-pooling maps): grid (N*hs*ws, row splits); a block walks the destination
// rows that touch its source pixel and adds its partial sum to dsrc (fp32 [N][hs][ws][C], zeroed)
__global__ __launch_bounds__(256) void bilinear_small_bwd_kernel(const h16raw* __restrict__ ddst, float* __restrict__ dsrc, int hs, int ws,
                                                                 int hd, int wd, int C8, int ld_dst, int c_dst, float sh, float sw) {
    const int sp = blockIdx.x % (hs * ws), nb = blockIdx.x / (hs * ws);
    const int i = sp / ws, j = sp % ws;
    const int lanes = 256 / C8, c8 = threadIdx.x % C8, pl = threadIdx.x / C8;
    // destination rows / columns whose two source taps can include i / j
    int y0 = (int)floorf(((float)i - 1.f + 0.5f) / sh - 0.5f) - 1, y1 = (int)ceilf(((float)i + 1.f + 0.5f) / sh - 0.5f) + 1;
    int x0 = (int)floorf(((float)j - 1.f + 0.5f) / sw - 0.5f) - 1, x1 = (int)ceilf(((float)j + 1.f + 0.5f) / sw - 0.5f) + 1;
    if (y0 < 0) y0 = 0;
    if (x0 < 0) x0 = 0;
    if (y1 > hd - 1) y1 = hd - 1;
    if (x1 > wd - 1) x1 = wd - 1;
    const int bw = x1 - x0 + 1;
    const int64_t npix = (int64_t)(y1 - y0 + 1) * bw;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t p = (int64_t)blockIdx.y * lanes + pl; p < npix; p += (int64_t)gridDim.y * lanes) {
        const int Y = y0 + (int)(p / bw), X = x0 + (int)(p % bw);
        const float wgt = lerp_weight(Y, i, sh, hs) * lerp_weight(X, j, sw, ws);
        if (wgt == 0.f) continue;
        float g[8];
        unpack8(*reinterpret_cast<const uint4*>(ddst + (((int64_t)nb * hd + Y) * wd + X) * ld_dst + c_dst + c8 * 8), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += wgt * g[k];
    }
    // the `lanes` pixel groups of the block hold partial sums of the SAME channels: summed through LDS first, one atomic per
    // (channel, block) instead of one per thread (with many row splits every thread's atomics hit the same few hundred addresses)
    __shared__ float red[256 * 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k * 256 + threadIdx.x] = acc[k];
    __syncthreads();
    if (pl == 0) {
        float* o = dsrc + ((int64_t)blockIdx.x * C8 + c8) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += red[k * 256 + l * C8 + c8];
            atomicAdd(o + k, t);
        }
    }
}

extern "C" int tcvom_bilinear(const void* src, void* dst, int32_t N, int32_t hs, int32_t ws, int32_t hd, int32_t wd, int32_t C,
                              int32_t ld_src, int32_t c_src, int32_t ld_dst, int32_t c_dst, void* stream) {
    TCVOM_CHECK_ARG(src && dst && N > 0 && hs > 0 && ws > 0 && hd > 0 && wd > 0 && C % 8 == 0, "bilinear: bad args");
    TCVOM_CHECK_ARG(ld_src % 8 == 0 && c_src % 8 == 0 && ld_dst % 8 == 0 && c_dst % 8 == 0 && c_src + C <= ld_src && c_dst + C <= ld_dst, "bilinear: bad channel slices");
    const int64_t n = (int64_t)N * hd * wd * (C / 8);
    hipLaunchKernelGGL(bilinear_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const h16raw*)src, (h16raw*)dst, n, hs, ws, hd, wd,
                       C / 8, ld_src, c_src, ld_dst, c_dst, (float)hs / (float)hd, (float)ws / (float)wd);
    TCVOM_LAUNCH_CHECK("bilinear");
    return TCVOM_OK;
}
// cat(bilinear x2 of x, skip) zero-padded to ld channels, every output row written once and whole (VMN_FBA.py:37-48): the
// up-sampling kernel + a strided copy of the skip tensor + a strided zero fill were three passes over the concat buffer
__global__ void up2_concat_kernel(const h16raw* __restrict__ x, const h16raw* __restrict__ skip, h16raw* __restrict__ dst, int64_t n,
                                  int hs, int ws, int Cx8, int Cs8, int ld8) {
    const int hd = 2 * hs, wd = 2 * ws;
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % ld8);
        int64_t t = v / ld8;
        const int X = (int)(t % wd); t /= wd;
        const int Y = (int)(t % hd);
        const int64_t nb = t / hd;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (c8 < Cx8) {
            const Lerp ly = lerp_of(Y, 0.5f, hs), lx = lerp_of(X, 0.5f, ws);
            const h16raw* sb = x + nb * hs * ws * (Cx8 * 8) + c8 * 8;
            float a[8], b[8], c[8], d[8], r[8];
            unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i0 * ws + lx.i0) * (Cx8 * 8)), a);
            unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i0 * ws + lx.i1) * (Cx8 * 8)), b);
            unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i1 * ws + lx.i0) * (Cx8 * 8)), c);
            unpack8(*reinterpret_cast<const uint4*>(sb + ((int64_t)ly.i1 * ws + lx.i1) * (Cx8 * 8)), d);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                r[k] = (1.f - ly.l) * ((1.f - lx.l) * a[k] + lx.l * b[k]) + ly.l * ((1.f - lx.l) * c[k] + lx.l * d[k]);
            o = pack8(r);
        } else if (c8 < Cx8 + Cs8) {
            o = *reinterpret_cast<const uint4*>(skip + (((nb * hd + Y) * wd + X) * Cs8 + (c8 - Cx8)) * 8);
        }
        *reinterpret_cast<uint4*>(dst + (((nb * hd + Y) * wd + X) * ld8 + c8) * 8) = o;
    }
}
extern "C" int tcvom_up2_concat(const void* x, const void* skip, void* dst, int32_t N, int32_t hs, int32_t ws, int32_t Cx, int32_t Cs,
                                int32_t ld, void* stream) {
    TCVOM_CHECK_ARG(x && skip && dst && N > 0 && hs > 0 && ws > 0 && Cx % 8 == 0 && Cs % 8 == 0 && ld % 8 == 0 && Cx + Cs <= ld, "up2_concat: bad args");
    const int64_t n = (int64_t)N * 4 * hs * ws * (ld / 8);
    hipLaunchKernelGGL(up2_concat_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const h16raw*)x, (const h16raw*)skip, (h16raw*)dst, n,
                       hs, ws, Cx / 8, Cs / 8, ld / 8);
    TCVOM_LAUNCH_CHECK("up2_concat");
    return TCVOM_OK;
}
extern "C" int tcvom_bilinear_up2_bwd(const void* ddst, void* dsrc, int32_t N, int32_t hs, int32_t ws, int32_t C, int32_t ld_dst, int32_t c_dst,
                                      void* stream) {
    TCVOM_CHECK_ARG(ddst && dsrc && N > 0 && hs > 0 && ws > 0 && C % 8 == 0 && ld_dst % 8 == 0 && c_dst % 8 == 0 && c_dst + C <= ld_dst, "bilinear_up2_bwd: bad args");
    const int64_t n = (int64_t)N * hs * ws * (C / 8);
    hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const h16raw*)ddst, (h16raw*)dsrc, n, hs, ws,
                       C / 8, ld_dst, c_dst);
    TCVOM_LAUNCH_CHECK("bilinear_up2_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_bilinear_small_bwd(const void* ddst, float* dsrc, int32_t N, int32_t hs, int32_t ws, int32_t hd, int32_t wd, int32_t C,
                                        int32_t ld_dst, int32_t c_dst, void* stream) {
    TCVOM_CHECK_ARG(ddst && dsrc && N > 0 && hs > 0 && ws > 0 && hd >= hs && wd >= ws && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, "bilinear_small_bwd: bad args");
    TCVOM_CHECK_ARG(ld_dst % 8 == 0 && c_dst % 8 == 0 && c_dst + C <= ld_dst, "bilinear_small_bwd: bad channel slice");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dsrc, 0, sizeof(float) * (size_t)N * hs * ws * C, st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "bilinear_small_bwd: memset failed");
    const int64_t npix = (int64_t)(2 * hd / hs + 4) * (2 * wd / ws + 4);
    int split = (int)((npix * (C / 8) + 256 * 64 - 1) / (256 * 64));
    if (split > 64) split = 64;
    // the 1 x 1 and 2 x 2 maps have 1 .. 4 source pixels per sample: 64 row splits left the launch at 192 .. 768 workgroups walking a
    // 50 MB slice each (106 .. 215 us per map at 1080p); at least ~1024 workgroups, each still >= 4 pixel rows per lane group
    const int64_t per_split_min = (int64_t)(256 / (C / 8)) * 4;
    const int want = (1024 + N * hs * ws - 1) / (N * hs * ws);
    if (split < want) split = want;
    if ((int64_t)split * per_split_min > npix) split = (int)(npix / per_split_min);
    if (split > 1024) split = 1024;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(bilinear_small_bwd_kernel, dim3(N * hs * ws, split), dim3(256), 0, st, (const h16raw*)ddst, dsrc, hs, ws, hd, wd, C / 8,
                       ld_dst, c_dst, (float)hs / (float)hd, (float)ws / (float)wd);
    TCVOM_LAUNCH_CHECK("bilinear_small_bwd");
    return TCVOM_OK;
}

// ------------------------------------------------------------------------------------------ head: 1x1 conv 16 -> 7, clamp / sigmoid, fusion
struct FbaPix { float o[7], a, F0[3], B0[3], F1[3], B1[3], Fc[3], Bc[3], num, den, r; };
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ float in01(float x) { return (x >= 0.f && x <= 1.f) ? 1.f : 0.f; }      // torch.clamp passes the gradient at the bounds

__device__ __forceinline__ void fba_pixel(const float* x16, const float* wb /* [7][16] + [7] */, const float* I, FbaPix& p) {
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        float a = wb[112 + m];
#pragma unroll
        for (int c = 0; c < 16; ++c) a += wb[m * 16 + c] * x16[c];
        p.o[m] = a;
    }
    const float a = clamp01(p.o[0]);
    p.a = a;
    float num = a * 0.1f, den = 0.1f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.F0[c] = sigmoidf(p.o[1 + c]);
        p.B0[c] = sigmoidf(p.o[4 + c]);
        p.F1[c] = a * I[c] + (1.f - a * a) * p.F0[c] - a * (1.f - a) * p.B0[c];
        p.B1[c] = (1.f - a) * I[c] + (2.f * a - a * a) * p.B0[c] - a * (1.f - a) * p.F1[c];
        p.Fc[c] = clamp01(p.F1[c]);
        p.Bc[c] = clamp01(p.B1[c]);
        const float D = p.Fc[c] - p.Bc[c];
        num += (I[c] - p.Bc[c]) * D;
        den += D * D;
    }
    p.num = num; p.den = den; p.r = num / den;
}

// x [N][HW][16] bf16; img fp32 [N][3][HW] with image stride img_stride; pred fp32 [N][7][HW] with stride pred_stride
__global__ void fba_head_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ img,
                                    float* __restrict__ pred, int64_t n, int64_t HW, int64_t img_stride, int64_t pred_stride) {
    __shared__ float wb[119];
    if (threadIdx.x < 112) wb[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < 7) wb[112 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    GRID_STRIDE(v, n) {
        const int64_t nb = v / HW, p = v % HW;
        float x16[16], I[3];
        unpack8(x[v * 2], x16);
        unpack8(x[v * 2 + 1], x16 + 8);
#pragma unroll
        for (int c = 0; c < 3; ++c) I[c] = img[nb * img_stride + c * HW + p];
        FbaPix px;
        fba_pixel(x16, wb, I, px);
        float* o = pred + nb * pred_stride + p;
        o[0] = clamp01(px.r);
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[(1 + c) * HW] = px.Fc[c]; o[(4 + c) * HW] = px.Bc[c]; }
    }
}
// dpred fp32 [N][7][HW] -> dx bf16 [N][HW][16], dw [replicas][7][16], db [replicas][7] (atomic, zeroed by the caller)
__global__ __launch_bounds__(256) void fba_head_bwd_kernel(const uint4* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ img, const float* __restrict__ dpred, uint4* __restrict__ dx,
                                                           float* __restrict__ dw, float* __restrict__ db, int64_t n, int64_t HW,
                                                           int64_t img_stride, int64_t pred_stride, int replicas) {
    __shared__ float wb[119];
    __shared__ float red[4][119];
    if (threadIdx.x < 112) wb[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < 7) wb[112 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    float gw[112], gb[7];
#pragma unroll
    for (int i = 0; i < 112; ++i) gw[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) gb[i] = 0.f;
    GRID_STRIDE(v, n) {
        const int64_t nb = v / HW, p = v % HW;
        float x16[16], I[3];
        unpack8(x[v * 2], x16);
        unpack8(x[v * 2 + 1], x16 + 8);
#pragma unroll
        for (int c = 0; c < 3; ++c) I[c] = img[nb * img_stride + c * HW + p];
        FbaPix px;
        fba_pixel(x16, wb, I, px);
        const float* g = dpred + nb * pred_stride + p;
        const float a = px.a;
        const float g_r = g[0] * in01(px.r);
        const float g_num = g_r / px.den, g_den = -g_r * px.num / (px.den * px.den);
        float g_a = g_num * 0.1f, go[7];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float D = px.Fc[c] - px.Bc[c];
            const float gFc = g[(1 + c) * HW] + g_num * (I[c] - px.Bc[c]) + g_den * 2.f * D;
            const float gBc = g[(4 + c) * HW] + g_num * (-D - (I[c] - px.Bc[c])) - g_den * 2.f * D;
            float gF1 = gFc * in01(px.F1[c]);
            const float gB1 = gBc * in01(px.B1[c]);
            g_a += gB1 * (-I[c] + (2.f - 2.f * a) * px.B0[c] - (1.f - 2.f * a) * px.F1[c]);
            float gB0 = gB1 * (2.f * a - a * a);
            gF1 += gB1 * (-a * (1.f - a));
            g_a += gF1 * (I[c] - 2.f * a * px.F0[c] - (1.f - 2.f * a) * px.B0[c]);
            const float gF0 = gF1 * (1.f - a * a);
            gB0 += gF1 * (-a * (1.f - a));
            go[1 + c] = gF0 * px.F0[c] * (1.f - px.F0[c]);
            go[4 + c] = gB0 * px.B0[c] * (1.f - px.B0[c]);
        }
        go[0] = g_a * in01(px.o[0]);
        float d16[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 7; ++m) s += wb[m * 16 + c] * go[m];
            d16[c] = s;
        }
        dx[v * 2] = pack8(d16);
        dx[v * 2 + 1] = pack8(d16 + 8);
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            gb[m] += go[m];
#pragma unroll
            for (int c = 0; c < 16; ++c) gw[m * 16 + c] += go[m] * x16[c];
        }
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 112; ++i) { const float s = wave_sum(gw[i]); if ((threadIdx.x & 63) == 0) red[wave][i] = s; }
#pragma unroll
    for (int i = 0; i < 7; ++i) { const float s = wave_sum(gb[i]); if ((threadIdx.x & 63) == 0) red[wave][112 + i] = s; }
    __syncthreads();
    if (threadIdx.x < 119) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int rep = blockIdx.x % replicas;
        if (threadIdx.x < 112) atomicAdd(dw + rep * 112 + threadIdx.x, s);
        else atomicAdd(db + rep * 7 + (threadIdx.x - 112), s);
    }
}

extern "C" int tcvom_fba_head_fwd(const void* x, const float* w, const float* b, const float* img, float* pred, int32_t N, int64_t HW,
                                  int64_t img_stride, int64_t pred_stride, void* stream) {
    TCVOM_CHECK_ARG(x && w && b && img && pred && N > 0 && HW > 0, "fba_head_fwd: bad args");
    const int64_t n = (int64_t)N * HW;
    hipLaunchKernelGGL(fba_head_fwd_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, w, b, img, pred, n, HW, img_stride, pred_stride);
    TCVOM_LAUNCH_CHECK("fba_head_fwd");
    return TCVOM_OK;
}
extern "C" int tcvom_fba_head_bwd(const void* x, const float* w, const float* b, const float* img, const float* dpred, void* dx, float* dw,
                                  float* db, int32_t replicas, int32_t N, int64_t HW, int64_t img_stride, int64_t pred_stride, void* stream) {
    TCVOM_CHECK_ARG(x && w && b && img && dpred && dx && dw && db && N > 0 && HW > 0 && replicas >= 1, "fba_head_bwd: bad args");
    const int64_t n = (int64_t)N * HW;
    int grid = dgrid(n);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fba_head_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, w, b, img, dpred, (uint4*)dx, dw, db,
                       n, HW, img_stride, pred_stride, replicas);
    TCVOM_LAUNCH_CHECK("fba_head_bwd");
    return TCVOM_OK;
}

// ------------------------------------------------------------------------------------------ network input: 8-channel trimap + image
// class maps (models/model.py:71-75): tri1 = dilated unknown ? 255 : alpha (after the eps snapping); bg = (tri1 == 0), fg = (tri1 == 1).
// trimap_transform (utils/utils.py:25-39): d = Euclidean distance to the nearest pixel of the class (exact two-pass EDT),
// clicks = exp(-d^2 / (2 (s * 320)^2)), s = 0.02, 0.08, 0.16.  No pixel of the class in the frame -> d = inf -> 0.
#define EDT_INF 1.0e30f
__device__ __forceinline__ int fba_class(float alpha, uint8_t dil, float eps) {   // 0 bg, 1 fg, 2 neither
    if (dil) return 2;
    float a = alpha < eps ? 0.f : alpha;
    a = a > 1.f - eps ? 1.f : a;
    return a == 0.f ? 0 : (a == 1.f ? 1 : 2);
}
// pass 1: per column, squared vertical distance to the nearest class pixel; g2 [frames][2][H][W].
// A column is a recurrence along H; one thread per column (the first form of this kernel) is 2 x 1088 DEPENDENT global accesses on
// 90 waves: 875 us at 1088 x 1920 x 3.  Here a block owns 64 columns x the whole height, thread (column, segment) owns EDT_SEG rows:
// it loads them at once (independent loads), keeps the class membership of its rows as BITS (one 64-bit word per class), publishes the
// first / last class row of its segment in LDS, finds the nearest class row above / below its segment among the other segments' entries
// and resolves every row with count-leading / trailing-zeros on the masks: no dependent memory access anywhere.
#define EDT_SEG 64
#define EDT_MAXSEG 32            // H <= 2048: a thread owns the segments seg, seg + 16
__global__ __launch_bounds__(1024) void edt_columns_kernel(const float* __restrict__ gts, const uint8_t* __restrict__ dil, float* __restrict__ g2,
                                                           int H, int W, float eps) {
    __shared__ int first[2][EDT_MAXSEG][64], last[2][EDT_MAXSEG][64];   // per class, segment, column: first / last class row (-1: none)
    const int cx = threadIdx.x & 63, nseg = (H + EDT_SEG - 1) / EDT_SEG;
    const int x = blockIdx.x * 64 + cx;
    const int64_t f = blockIdx.y;
    unsigned long long m[2][2] = {{0ull, 0ull}, {0ull, 0ull}};      // [own segment 0 / 1][class]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int seg = (threadIdx.x >> 6) + 16 * q, y0 = seg * EDT_SEG;
        if (x < W && seg < nseg) {
            const float* a = gts + f * H * W + x;
            const uint8_t* d = dil ? dil + f * H * W + x : nullptr;
#pragma unroll 16
            for (int j = 0; j < EDT_SEG; ++j) {
                const int y = y0 + j;
                if (y < H) {
                    const int c = fba_class(a[(int64_t)y * W], d ? d[(int64_t)y * W] : (uint8_t)0, eps);
                    if (c < 2) m[q][c] |= 1ull << j;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            first[k][seg][cx] = m[q][k] ? y0 + __builtin_ctzll(m[q][k]) : -1;
            last[k][seg][cx] = m[q][k] ? y0 + 63 - __builtin_clzll(m[q][k]) : -1;
        }
    }
    __syncthreads();
    if (x >= W) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int seg = (threadIdx.x >> 6) + 16 * q, y0 = seg * EDT_SEG;
        if (seg >= nseg) break;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int above = -1, below = -1;                              // nearest class row before / after this segment
            for (int s2 = seg - 1; s2 >= 0 && above < 0; --s2) above = last[k][s2][cx];
            for (int s2 = seg + 1; s2 < nseg && below < 0; ++s2) below = first[k][s2][cx];
            float* o = g2 + ((f * 2 + k) * H) * W + x;
            const unsigned long long mk = m[q][k];
            for (int j = 0; j < EDT_SEG; ++j) {
                const int y = y0 + j;
                if (y >= H) break;
                const unsigned long long lo = mk & (j == 63 ? ~0ull : ((1ull << (j + 1)) - 1ull)), hi = mk >> j;
                const int up = lo ? j - (63 - __builtin_clzll(lo)) : (above >= 0 ? y - above : -1);
                const int dn = hi ? __builtin_ctzll(hi) : (below >= 0 ? below - y : -1);
                const int dmin = up < 0 ? dn : (dn < 0 ? up : (up < dn ? up : dn));
                o[(int64_t)y * W] = dmin < 0 ? EDT_INF : (float)dmin * (float)dmin;
            }
        }
    }
}
// pass 2: one block per (frame, class, row): d2[x] = min_x' (x - x')^2 + g2[x'], then the three click maps, written into the
// space-to-depth network input x2 [frames][H/2][W/2][64] (channel 16 * (2 (h&1) + (w&1)) + 3 + 3 k + sigma) and, optionally, tris
__global__ __launch_bounds__(256) void edt_rows_kernel(const float* __restrict__ g2, h16raw* __restrict__ x2, float* __restrict__ tris, int H, int W) {
    extern __shared__ float row[];                     // [W] g2 of this row, then [ceil(W / 16)] minima of its 16-column blocks
    const int h = blockIdx.x % H, k = (blockIdx.x / H) % 2;
    const int64_t f = blockIdx.x / (2 * H);
    const float* g = g2 + ((f * 2 + k) * H + h) * W;
    const int nblk = (W + 15) >> 4;
    float* bmin = row + W;
    for (int x = threadIdx.x; x < W; x += 256) row[x] = g[x];
    __syncthreads();
    for (int b = threadIdx.x; b < nblk; b += 256) {
        float mval = EDT_INF;
        for (int i = b * 16; i < min(W, b * 16 + 16); ++i) mval = fminf(mval, row[i]);
        bmin[b] = mval;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        // d2[x] = min over x' of (x - x')^2 + g2[x'], exactly: the candidates are visited in 16-column blocks outwards from x; a block
        // whose best case -- its nearest edge's distance squared plus its minimum -- cannot beat `best` is skipped.  (The plain
        // outward scan visits 2 sqrt(best) candidates per pixel: 1376 us at 1088 x 1920 x 3, where most of the background lies
        // hundreds of pixels from the foreground.)
        const int xb = x >> 4;
        float best = EDT_INF;
        for (int i = xb * 16; i < min(W, xb * 16 + 16); ++i) { const float dx = (float)(x - i); best = fminf(best, dx * dx + row[i]); }
        for (int bd = 1; bd < nblk; ++bd) {
            const float edge = (float)((bd - 1) * 16 + 1);             // a block bd away starts at least this far from x
            if (__all(edge * edge >= best)) break;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int b = side ? xb + bd : xb - bd;
                if (b < 0 || b >= nblk) continue;
                const int lo = b * 16, hi = min(W, lo + 16) - 1;
                const float de = (float)(side ? lo - x : x - hi);      // distance to the nearest column of the block (>= 1)
                if (de * de + bmin[b] >= best) continue;
                for (int i = lo; i <= hi; ++i) { const float dx = (float)(x - i); best = fminf(best, dx * dx + row[i]); }
            }
        }
        const int sub = (h & 1) * 2 + (x & 1);
        h16raw* o = x2 + (((f * (H / 2) + h / 2) * (W / 2) + x / 2) * 64) + sub * 16 + 3 + 3 * k;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float sg = (s == 0 ? 0.02f : (s == 1 ? 0.08f : 0.16f)) * 320.f;
            const float c = best < EDT_INF ? __expf(-best / (2.f * sg * sg)) : 0.f;
            o[s] = f2h(c);
            if (tris) tris[((f * 8 + 3 * k + s) * H + h) * W + x] = c;
        }
    }
}
// pass 3: image and indicator channels.  x2 channels 0..2 (normalised RGB), 9 (bg), 10 (fg) of every sub-pixel; extras
// [frames][H][W][8] = (normalised RGB, RGB, bg, fg) for the last decoder stage; imgs fp32 [frames][3][H][W] (scaled RGB)
__global__ void fba_input_kernel(const float* __restrict__ gts, const uint8_t* __restrict__ dil, const float* __restrict__ imgs, h16raw* __restrict__ x2,
                                 uint4* __restrict__ extras, float* __restrict__ tris, int64_t n, int H, int W, float eps) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
    GRID_STRIDE(v, n) {
        const int w = (int)(v % W);
        const int h = (int)((v / W) % H);
        const int64_t f = v / ((int64_t)H * W);
        const int cls = fba_class(gts[v], dil ? dil[v] : (uint8_t)0, eps);
        float e[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float px = imgs[(f * 3 + c) * H * W + (int64_t)h * W + w];
            e[c] = (px - mean[c]) * istd[c];
            e[3 + c] = px;
        }
        e[6] = cls == 0 ? 1.f : 0.f;
        e[7] = cls == 1 ? 1.f : 0.f;
        extras[v] = pack8(e);
        h16raw* o = x2 + (((f * (H / 2) + h / 2) * (W / 2) + w / 2) * 64) + ((h & 1) * 2 + (w & 1)) * 16;
        o[0] = f2h(e[0]); o[1] = f2h(e[1]); o[2] = f2h(e[2]);
        o[9] = f2h(e[6]); o[10] = f2h(e[7]);
        if (tris) {
            tris[((f * 8 + 6) * H + h) * W + w] = e[6];
            tris[((f * 8 + 7) * H + h) * W + w] = e[7];
        }
    }
}

extern "C" int tcvom_fba_input(const float* gts, const uint8_t* unk_dil, const float* imgs, void* x2, void* extras, float* tris,
                               float* edt_scratch, int64_t frames, int32_t H, int32_t W, float eps, void* stream) {
    TCVOM_CHECK_ARG(gts && imgs && x2 && extras && edt_scratch && frames > 0 && H % 2 == 0 && W % 2 == 0 && W <= 8192, "fba_input: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(x2, 0, sizeof(h16raw) * (size_t)frames * (H / 2) * (W / 2) * 64, st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "fba_input: memset failed");
    TCVOM_CHECK_ARG(H <= EDT_MAXSEG * EDT_SEG && frames < 65536, "fba_input: H=%d (at most %d rows)", H, EDT_MAXSEG * EDT_SEG);
    hipLaunchKernelGGL(edt_columns_kernel, dim3((unsigned)cdiv(W, 64), (unsigned)frames), dim3(1024), 0, st, gts, unk_dil, edt_scratch, H, W, eps);
    hipLaunchKernelGGL(edt_rows_kernel, dim3((unsigned)(frames * 2 * H)), dim3(256), sizeof(float) * (W + (W + 15) / 16), st, edt_scratch, (h16raw*)x2, tris, H, W);
    const int64_t n = frames * (int64_t)H * W;
    hipLaunchKernelGGL(fba_input_kernel, dim3(dgrid(n)), dim3(256), 0, st, gts, unk_dil, imgs, (h16raw*)x2, (uint4*)extras, tris, n, H, W, eps);
    TCVOM_LAUNCH_CHECK("fba_input");
    return TCVOM_OK;
}

// ------------------------------------------------------------------------------------------ evaluation metrics (calc_metric.py:22-46)
// One pass over a frame: SAD / MSE / SSDA sums over the unknown region of the trimap, dtSSD against the adjacent frame and the
// flow-warped MESSDdt (utils/utils.py:88-123: bilinear grid_sample, align_corners=True, zero padding).  acc (double[8], zeroed
// by the caller): 0 unknown pixels, 1 sum |a-g|, 2 sum (a-g)^2, 3 sum ((a-ha)-(g-hg))^2, 4 sum |(a-g)-(pa-pg)|,
// 5 sum |(a-g)^2-(pa-pg)^2|, 6 pixels with valid flow inside the unknown region.
__device__ __forceinline__ float sample_zero(const float* __restrict__ im, int H, int W, float x, float y) {
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float lx = x - x0f, ly = y - y0f;
    float v = 0.f;
    if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) v += (1.f - ly) * (1.f - lx) * im[(int64_t)y0 * W + x0];
        if (x0 + 1 >= 0 && x0 + 1 < W) v += (1.f - ly) * lx * im[(int64_t)y0 * W + x0 + 1];
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) v += ly * (1.f - lx) * im[(int64_t)(y0 + 1) * W + x0];
        if (x0 + 1 >= 0 && x0 + 1 < W) v += ly * lx * im[(int64_t)(y0 + 1) * W + x0 + 1];
    }
    return v;
}
__global__ __launch_bounds__(256) void matting_metrics_kernel(const float* __restrict__ a, const float* __restrict__ g, const uint8_t* __restrict__ tri,
                                                              const float* __restrict__ ha, const float* __restrict__ hg,
                                                              const float* __restrict__ flow, double* __restrict__ acc, int H, int W) {
    __shared__ float red[4];
    float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t n = (int64_t)H * W;
    GRID_STRIDE(v, n) {
        const uint8_t t = tri[v];
        if (t == 0 || t == 255) continue;
        const float d = a[v] - g[v];
        s[0] += 1.f;
        s[1] += fabsf(d);
        s[2] += d * d;
        if (ha) {
            const float dd = (a[v] - ha[v]) - (g[v] - hg[v]);
            s[3] += dd * dd;
        }
        if (flow) {
            const float fx = flow[v], fy = flow[n + v];
            if (fx == fx && fy == fy) {                                  // NaN marks an invalid flow vector
                const int y = (int)(v / W), x = (int)(v % W);
                const float e = sample_zero(ha, H, W, (float)x + fx, (float)y + fy) - sample_zero(hg, H, W, (float)x + fx, (float)y + fy);
                s[4] += fabsf(d - e);
                s[5] += fabsf(d * d - e * e);
                s[6] += 1.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float r = block_sum_256(s[i], red);
        if (threadIdx.x == 0 && r != 0.f) atomicAdd(acc + i, (double)r);
    }
}

extern "C" int tcvom_matting_metrics(const float* a, const float* g, const uint8_t* tri, const float* ha, const float* hg, const float* flow,
                                     double* acc, int32_t H, int32_t W, void* stream) {
    TCVOM_CHECK_ARG(a && g && tri && acc && H > 0 && W > 0 && (!flow || (ha && hg)) && ((ha == nullptr) == (hg == nullptr)), "matting_metrics: bad args");
    int grid = dgrid((int64_t)H * W);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(matting_metrics_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, g, tri, ha, hg, flow, acc, H, W);
    TCVOM_LAUNCH_CHECK("matting_metrics");
    return TCVOM_OK;
}
