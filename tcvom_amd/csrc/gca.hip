// Guided Contextual Attention (models/GCA/ops.py:106-229) — the data-movement kernels around the
// two big MFMA GEMMs (scores S = G G^T and O = P V, run through igemm_nt with ntaps == 1).
//
//   G  [B][N][576]   raw reflect-padded 3x3x64 patches of the os16 guidance map (queries AND keys)
//   c_j = (mm_j ? s0 : s1) / max(||G_j||, 1e-4)      per-key scale  (ops.py:139-143,174-175,186)
//   d_j = 1e4 * mm_j                                  self-mask      (ops.py:159-161,188)
//   S'[i][j] = c_j <G_i, G_j> - d_j [i == j];   P = softmax_j S'   (ops.py:177-190)
//   V  [B][N][2048]  reflect-padded 4x4 stride-2 patches of the os8 feature (ops.py:115-119)
//   O = P V ;  y = fold_{4x4,s2,p1}(O) / 4                          (ops.py:204)
// and the matching backward pieces.  Patch column order is (tap, channel) everywhere.
#include "common.h"

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// ------------------------------------------------------------------ unknown-area statistics -> [B][2] scales
__global__ __launch_bounds__(256) void gca_scale_kernel(const unsigned char* __restrict__ unk8, float* __restrict__ scales,
                                                        int h8, int w8) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int h = h8 / 2, w = w8 / 2;
    float a = 0.f;
    for (int i = threadIdx.x; i < h * w; i += 256)
        a += unk8[((int64_t)b * h8 + 2 * (i / w)) * w8 + 2 * (i % w)] ? 1.f : 0.f;
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) {
        const float um = a / (float)(h * w), km = 1.f - um;
        scales[b * 2 + 0] = fminf(fmaxf(sqrtf(um / km), 0.1f), 10.f);
        scales[b * 2 + 1] = fminf(fmaxf(sqrtf(km / um), 0.1f), 10.f);
    }
}

// ------------------------------------------------------------------ guidance patches + per-key vectors
// g8: [B,h8,w8,64] bf16 (guidance_conv output at os8; os16 grid = even positions).  one wave per key.
__global__ __launch_bounds__(256) void gca_patches_kernel(const h16raw* __restrict__ g8, const unsigned char* __restrict__ unk8,
                                                          const float* __restrict__ scales, h16raw* __restrict__ G,
                                                          float* __restrict__ cvec, float* __restrict__ dvec,
                                                          float* __restrict__ nrm, int B, int h8, int w8, int CG) {
    const int h = h8 / 2, w = w8 / 2, N = h * w;
    const int lane = threadIdx.x & 63;
    const int64_t key = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= (int64_t)B * N) return;
    const int b = (int)(key / N), j = (int)(key % N), jy = j / w, jx = j % w;
    float ss = 0.f;
    bool any = false;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = refl(jy + t / 3 - 1, h), xx = refl(jx + t % 3 - 1, w);
        const int64_t src = ((int64_t)b * h8 + 2 * yy) * w8 + 2 * xx;
        any = any || unk8[src] != 0;
        for (int c = lane; c < CG; c += 64) {
            const h16raw val = g8[src * CG + c];
            G[(key * 9 + t) * CG + c] = val;
            const float f = h2f(val);
            ss += f * f;
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) {
        const float n = sqrtf(ss);
        const float sc = any ? scales[b * 2] : scales[b * 2 + 1];
        cvec[key] = sc / fmaxf(n, 1e-4f);
        dvec[key] = any ? 1e4f : 0.f;
        nrm[key] = n;
    }
}

// ------------------------------------------------------------------ row softmax: fp32 [rows][ld] -> bf16 [rows][ldp]
// One block per row.  Fast path (rows up to 8192 columns, 16-byte aligned): a thread owns 8 consecutive columns of each
// 2048-column chunk, keeps its <= 32 logits in registers (ONE pass over S: 2 x float4 loads per chunk) and writes
// 8 bf16 per 16-byte store.  The generic path (longer rows / odd strides) re-reads the row from L2.
template <int NCHUNK>
__global__ __launch_bounds__(256) void row_softmax_reg_kernel(const float* __restrict__ S, h16raw* __restrict__ P, int ncols,
                                                              int64_t ld, int64_t ldp) {
    __shared__ float red[4];
    const float* s = S + (int64_t)blockIdx.x * ld;
    h16raw* p = P + (int64_t)blockIdx.x * ldp;
    float v[NCHUNK][8];
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int j0 = c * 2048 + threadIdx.x * 8;
        if (j0 + 8 <= ncols) {
            const float4 a = *reinterpret_cast<const float4*>(s + j0), b = *reinterpret_cast<const float4*>(s + j0 + 4);
            v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w; v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[c][k] = j0 + k < ncols ? s[j0 + k] : -3.0e38f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) mx = fmaxf(mx, v[c][k]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[c][k] = __expf(v[c][k] - mx);         // exp(-3e38 - mx) == 0 for the padding
            den += v[c][k];
        }
    den = block_sum_256(den, red);
    const float r = 1.f / den;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int j0 = c * 2048 + threadIdx.x * 8;
        if (j0 < ldp) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = v[c][k] * r;
            *reinterpret_cast<uint4*>(p + j0) = pack8(o);
        }
    }
}
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ S, h16raw* __restrict__ P, int ncols,
                                                          int64_t ld, int64_t ldp) {
    __shared__ float red[4];
    const float* s = S + (int64_t)blockIdx.x * ld;
    h16raw* p = P + (int64_t)blockIdx.x * ldp;
    float mx = -3.0e38f;
    for (int j = threadIdx.x; j < ncols; j += 256) mx = fmaxf(mx, s[j]);
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float den = 0.f;
    for (int j = threadIdx.x; j < ncols; j += 256) den += __expf(s[j] - mx);
    den = block_sum_256(den, red);
    const float r = 1.f / den;
    for (int j = threadIdx.x; j < ldp; j += 256) p[j] = j < ncols ? f2h(__expf(s[j] - mx) * r) : (h16raw)0;
}

// dS'[i][j] = P (dP - sum_j P dP);  T = dS' * c_j  (bf16, padded columns zero)
template <int NCHUNK>
__global__ __launch_bounds__(256) void row_softmax_bwd_reg_kernel(const h16raw* __restrict__ P, const float* __restrict__ dP,
                                                                  const float* __restrict__ cvec, h16raw* __restrict__ T,
                                                                  int ncols, int64_t ld, int64_t ldp, int rows_per_batch) {
    __shared__ float red[4];
    cvec += (int64_t)(blockIdx.x / rows_per_batch) * ncols;      // per-key scales are [B][N]
    const h16raw* p = P + (int64_t)blockIdx.x * ldp;
    const float* d = dP + (int64_t)blockIdx.x * ld;
    h16raw* t = T + (int64_t)blockIdx.x * ldp;
    const bool cvec4 = (ncols & 3) == 0 && (((uintptr_t)cvec) & 15) == 0;
    // pass 1 keeps P * (the row's probabilities) and dP in registers as w = P and g = dP; only 8 + 8 live values per
    // chunk are needed twice, so the products are stored: pd = P*dP (for the sum) and the second pass needs P and dP
    float pv[NCHUNK][8], dv[NCHUNK][8];
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int j0 = c * 2048 + threadIdx.x * 8;
        if (j0 + 8 <= ncols) {
            unpack8(*reinterpret_cast<const uint4*>(p + j0), pv[c]);
            const float4 x = *reinterpret_cast<const float4*>(d + j0), y = *reinterpret_cast<const float4*>(d + j0 + 4);
            dv[c][0] = x.x; dv[c][1] = x.y; dv[c][2] = x.z; dv[c][3] = x.w; dv[c][4] = y.x; dv[c][5] = y.y; dv[c][6] = y.z; dv[c][7] = y.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                pv[c][k] = j0 + k < ncols ? h2f(p[j0 + k]) : 0.f;
                dv[c][k] = j0 + k < ncols ? d[j0 + k] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a += pv[c][k] * dv[c][k];
    }
    a = block_sum_256(a, red);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int j0 = c * 2048 + threadIdx.x * 8;
        if (j0 < ldp) {
            float cv[8];
            if (cvec4 && j0 + 8 <= ncols) {
                const float4 x = *reinterpret_cast<const float4*>(cvec + j0), y = *reinterpret_cast<const float4*>(cvec + j0 + 4);
                cv[0] = x.x; cv[1] = x.y; cv[2] = x.z; cv[3] = x.w; cv[4] = y.x; cv[5] = y.y; cv[6] = y.z; cv[7] = y.w;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) cv[k] = j0 + k < ncols ? cvec[j0 + k] : 0.f;
            }
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = pv[c][k] * (dv[c][k] - a) * cv[k];       // pv == 0 beyond ncols
            *reinterpret_cast<uint4*>(t + j0) = pack8(o);
        }
    }
}
__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const h16raw* __restrict__ P, const float* __restrict__ dP,
                                                              const float* __restrict__ cvec, h16raw* __restrict__ T,
                                                              int ncols, int64_t ld, int64_t ldp, int rows_per_batch) {
    __shared__ float red[4];
    cvec += (int64_t)(blockIdx.x / rows_per_batch) * ncols;      // per-key scales are [B][N]
    const h16raw* p = P + (int64_t)blockIdx.x * ldp;
    const float* d = dP + (int64_t)blockIdx.x * ld;
    h16raw* t = T + (int64_t)blockIdx.x * ldp;
    float a = 0.f;
    for (int j = threadIdx.x; j < ncols; j += 256) a += h2f(p[j]) * d[j];
    a = block_sum_256(a, red);
    for (int j = threadIdx.x; j < ldp; j += 256)
        t[j] = j < ncols ? f2h(h2f(p[j]) * (d[j] - a) * cvec[j]) : (h16raw)0;
}

// ------------------------------------------------------------------ value patches V[key][(ky*4+kx)*C + c]
__global__ __launch_bounds__(256) void gca_value_patches_kernel(const uint4* __restrict__ alpha, uint4* __restrict__ V,
                                                                int B, int h8, int w8, int C8) {
    const int h = h8 / 2, w = w8 / 2, N = h * w;
    const int64_t total = (int64_t)B * N * 16 * C8;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(v % C8);
        const int t = (int)((v / C8) % 16);
        const int64_t key = v / ((int64_t)C8 * 16);
        const int b = (int)(key / N), j = (int)(key % N);
        const int yy = refl(2 * (j / w) - 1 + t / 4, h8), xx = refl(2 * (j % w) - 1 + t % 4, w8);
        V[v] = alpha[(((int64_t)b * h8 + yy) * w8 + xx) * C8 + c];
    }
}
// dalpha[y][x][c] = sum over (key, tap) whose reflected source is (y,x) of dV[key][tap][c]   (dV fp32)
__global__ __launch_bounds__(256) void gca_value_patches_bwd_kernel(const float4* __restrict__ dV, uint2* __restrict__ dalpha,
                                                                    int B, int h8, int w8, int C4) {
    // one thread = one pixel x 4 channels (16-byte loads of the fp32 patch gradients, 8-byte bf16 store)
    const int h = h8 / 2, w = w8 / 2, N = h * w;
    const int64_t total = (int64_t)B * h8 * w8 * C4;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(v % C4);
        int64_t t = v / C4;
        const int x = (int)(t % w8); t /= w8;
        const int y = (int)(t % h8);
        const int b = (int)(t / h8);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // the (patch, tap) pairs whose reflected source row is y: source rows s = y, plus the out-of-range rows that reflect onto
        // y (-1 -> 1, h8 -> h8 - 2); row s is tap ky of patch jy = (s + 1 - ky) / 2 for the two ky of the parity of s + 1.
        // (Enumerated directly: the search over all (patch, tap) candidates with a reflection test each -- up to 256 per thread --
        // was the kernel: 92 us for 200 MB of gradients.)
        int sy[3], sx[3], ny = 1, nx = 1;
        sy[0] = y; sx[0] = x;
        if (y == 1) sy[ny++] = -1;
        if (y == h8 - 2) sy[ny++] = h8;
        if (x == 1) sx[nx++] = -1;
        if (x == w8 - 2) sx[nx++] = w8;
        for (int a = 0; a < ny; ++a)
#pragma unroll
            for (int pa = 0; pa < 2; ++pa) {
                const int ky = ((sy[a] + 1) & 1) + 2 * pa, jy = (sy[a] + 1 - ky) / 2;
                if (sy[a] + 1 - ky < 0 || jy >= h) continue;
                for (int bb = 0; bb < nx; ++bb)
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        const int kx = ((sx[bb] + 1) & 1) + 2 * pb, jx = (sx[bb] + 1 - kx) / 2;
                        if (sx[bb] + 1 - kx < 0 || jx >= w) continue;
                        const float4 g = dV[(((int64_t)b * N + jy * w + jx) * 16 + ky * 4 + kx) * C4 + c];
                        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
                    }
            }
        dalpha[v] = make_uint2(pack2h(acc.x, acc.y), pack2h(acc.z, acc.w));
    }
}

// ------------------------------------------------------------------ fold: y = fold(O)/4 and its adjoint
template <bool F32>                              // F32: O is fp32 (the backward's <dO, O> row sums want the unrounded P V)
__global__ __launch_bounds__(256) void gca_fold_kernel(const uint4* __restrict__ O, uint4* __restrict__ Y, int B, int h8, int w8, int C8) {
    const int h = h8 / 2, w = w8 / 2, N = h * w;
    const int64_t total = (int64_t)B * h8 * w8 * C8;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(v % C8);
        int64_t t = v / C8;
        const int x = (int)(t % w8); t /= w8;
        const int y = (int)(t % h8);
        const int b = (int)(t / h8);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int iy = (y + 1) / 2 - a, ky = y + 1 - 2 * iy;
            if (iy < 0 || iy >= h || ky < 0 || ky > 3) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ix = (x + 1) / 2 - e, kx = x + 1 - 2 * ix;
                if (ix < 0 || ix >= w || kx < 0 || kx > 3) continue;
                const int64_t oi = (((int64_t)b * N + iy * w + ix) * 16 + ky * 4 + kx) * C8 + c;
                if (F32) {
                    const float4 lo = reinterpret_cast<const float4*>(O)[2 * oi], hi = reinterpret_cast<const float4*>(O)[2 * oi + 1];
                    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
                } else unpack8(O[oi], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += f[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= 0.25f;
        Y[v] = pack8(acc);
    }
}
// dO[i][(ky*4+kx)*C + c] = dY[2iy-1+ky][2ix-1+kx][c] / 4  (zero outside)
__global__ __launch_bounds__(256) void gca_unfold_kernel(const uint4* __restrict__ dY, uint4* __restrict__ dO, int B, int h8, int w8, int C8) {
    const int h = h8 / 2, w = w8 / 2, N = h * w;
    const int64_t total = (int64_t)B * N * 16 * C8;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(v % C8);
        const int t = (int)((v / C8) % 16);
        const int64_t key = v / ((int64_t)C8 * 16);
        const int b = (int)(key / N), i = (int)(key % N);
        const int yy = 2 * (i / w) - 1 + t / 4, xx = 2 * (i % w) - 1 + t % 4;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (yy >= 0 && yy < h8 && xx >= 0 && xx < w8) {
            float f[8];
            unpack8(dY[(((int64_t)b * h8 + yy) * w8 + xx) * C8 + c], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] *= 0.25f;
            q = pack8(f);
        }
        dO[v] = q;
    }
}

// ------------------------------------------------------------------ patch gradient -> guidance map gradient
// dWp[j] = dWq[j] + M'[j] - coef_j G[j],  coef_j = <M'_j, G_j>/n_j^2 (n_j > 1e-4)   [one wave per key, in place in dWq]
__global__ __launch_bounds__(256) void gca_patch_grad_kernel(float* __restrict__ dWq, const float* __restrict__ Mp,
                                                             const h16raw* __restrict__ G, const float* __restrict__ nrm,
                                                             int64_t keys, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t key = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= keys) return;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += Mp[key * D + d] * h2f(G[key * D + d]);
    dot = wave_sum(dot);
    const float n = nrm[key];
    const float coef = n > 1e-4f ? dot / (n * n) : 0.f;
    for (int d = lane; d < D; d += 64) dWq[key * D + d] += Mp[key * D + d] - coef * h2f(G[key * D + d]);
}
// dg8[b][2y][2x][c] = sum over (key, tap) with reflected source (y,x) of dWp[key][tap*CG + c]; zero at odd positions
__global__ __launch_bounds__(256) void gca_patches_bwd_kernel(const float* __restrict__ dWp, h16raw* __restrict__ dg8,
                                                              int B, int h8, int w8, int CG) {
    const int h = h8 / 2, w = w8 / 2, N = h * w;
    const int64_t total = (int64_t)B * h8 * w8 * CG;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(v % CG);
        int64_t t = v / CG;
        const int x8 = (int)(t % w8); t /= w8;
        const int y8 = (int)(t % h8);
        const int b = (int)(t / h8);
        float acc = 0.f;
        if ((y8 & 1) == 0 && (x8 & 1) == 0) {
            const int y = y8 / 2, x = x8 / 2;
            // source rows s = y, plus -1 (reflects onto 1) / h (onto h - 2); row s is tap ky of patch jy = s + 1 - ky
            int sy[3], sx[3], ny = 1, nx = 1;
            sy[0] = y; sx[0] = x;
            if (y == 1) sy[ny++] = -1;
            if (y == h - 2) sy[ny++] = h;
            if (x == 1) sx[nx++] = -1;
            if (x == w - 2) sx[nx++] = w;
            for (int a = 0; a < ny; ++a)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int jy = sy[a] + 1 - ky;
                    if (jy < 0 || jy >= h) continue;
                    for (int bb = 0; bb < nx; ++bb)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int jx = sx[bb] + 1 - kx;
                            if (jx < 0 || jx >= w) continue;
                            acc += dWp[(((int64_t)b * N + jy * w + jx) * 9 + ky * 3 + kx) * CG + c];
                        }
                }
        }
        dg8[v] = f2h(acc);
    }
}

static int sgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int tcvom_gca_prepare(const void* g8, const uint8_t* unk8, void* G, float* scales, float* cvec, float* dvec,
                                 float* nrm, int32_t B, int32_t h8, int32_t w8, int32_t CG, void* stream) {
    TCVOM_CHECK_ARG(g8 && unk8 && G && scales && cvec && dvec && nrm, "gca_prepare: null pointer");
    TCVOM_CHECK_ARG(h8 % 2 == 0 && w8 % 2 == 0 && h8 >= 4 && w8 >= 4, "gca_prepare: os8 grid %dx%d must be even and >= 4", h8, w8);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gca_scale_kernel, dim3(B), dim3(256), 0, st, unk8, scales, h8, w8);
    const int64_t keys = (int64_t)B * (h8 / 2) * (w8 / 2);
    hipLaunchKernelGGL(gca_patches_kernel, dim3(cdiv(keys, 4)), dim3(256), 0, st, (const h16raw*)g8, unk8, scales,
                       (h16raw*)G, cvec, dvec, nrm, B, h8, w8, CG);
    TCVOM_LAUNCH_CHECK("gca_prepare");
    return TCVOM_OK;
}
extern "C" int tcvom_row_softmax(const float* S, void* P, int32_t rows, int32_t ncols, int64_t ld, int64_t ldp, void* stream) {
    TCVOM_CHECK_ARG(S && P && rows > 0 && ncols > 0 && ld >= ncols && ldp >= ncols, "row_softmax: bad args");
    const bool fast = ldp <= 8192 && ld % 4 == 0 && ldp % 8 == 0 && ((uintptr_t)S % 16 == 0) && ((uintptr_t)P % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (fast && ldp <= 2048) hipLaunchKernelGGL(row_softmax_reg_kernel<1>, dim3(rows), dim3(256), 0, st, S, (h16raw*)P, ncols, ld, ldp);
    else if (fast && ldp <= 4096) hipLaunchKernelGGL(row_softmax_reg_kernel<2>, dim3(rows), dim3(256), 0, st, S, (h16raw*)P, ncols, ld, ldp);
    else if (fast) hipLaunchKernelGGL(row_softmax_reg_kernel<4>, dim3(rows), dim3(256), 0, st, S, (h16raw*)P, ncols, ld, ldp);
    else hipLaunchKernelGGL(row_softmax_kernel, dim3(rows), dim3(256), 0, st, S, (h16raw*)P, ncols, ld, ldp);
    TCVOM_LAUNCH_CHECK("row_softmax");
    return TCVOM_OK;
}
extern "C" int tcvom_row_softmax_bwd(const void* P, const float* dP, const float* cvec, void* T, int32_t rows,
                                     int32_t ncols, int64_t ld, int64_t ldp, int32_t rows_per_batch, void* stream) {
    TCVOM_CHECK_ARG(P && dP && cvec && T && rows > 0 && ncols > 0 && rows_per_batch > 0, "row_softmax_bwd: bad args");
    const bool fast = ldp <= 8192 && ld % 4 == 0 && ldp % 8 == 0 && ((uintptr_t)dP % 16 == 0) && ((uintptr_t)P % 16 == 0) &&
                      ((uintptr_t)T % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
#define SM_BWD_ARGS (const h16raw*)P, dP, cvec, (h16raw*)T, ncols, ld, ldp, rows_per_batch
    if (fast && ldp <= 2048) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<1>, dim3(rows), dim3(256), 0, st, SM_BWD_ARGS);
    else if (fast && ldp <= 4096) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<2>, dim3(rows), dim3(256), 0, st, SM_BWD_ARGS);
    else if (fast) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<4>, dim3(rows), dim3(256), 0, st, SM_BWD_ARGS);
    else hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3(rows), dim3(256), 0, st, SM_BWD_ARGS);
#undef SM_BWD_ARGS
    TCVOM_LAUNCH_CHECK("row_softmax_bwd");
    return TCVOM_OK;
}
// delta[r] = <a[r], b[r]> over `cols` bf16 columns (one wave per row): the row sums sum_j P dP of the softmax backward
// as <dO_i, O_i>, which needs no N x N operand
template <bool BF32>                             // BF32: b is fp32
__global__ __launch_bounds__(256) void rowdot_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, float* __restrict__ out,
                                                          int64_t rows, int cols8) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int c = lane; c < cols8; c += 64) {
        float x[8], y[8];
        unpack8(a[r * cols8 + c], x);
        if (BF32) {
            const float4 lo = reinterpret_cast<const float4*>(b)[2 * (r * cols8 + c)], hi = reinterpret_cast<const float4*>(b)[2 * (r * cols8 + c) + 1];
            y[0] = lo.x; y[1] = lo.y; y[2] = lo.z; y[3] = lo.w; y[4] = hi.x; y[5] = hi.y; y[6] = hi.z; y[7] = hi.w;
        } else unpack8(b[r * cols8 + c], y);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += x[k] * y[k];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[r] = acc;
}
extern "C" int tcvom_rowdot_bf16(const void* a, const void* b, int32_t b_fp32, float* out, int64_t rows, int32_t cols, void* stream) {
    TCVOM_CHECK_ARG(a && b && out && rows > 0 && cols > 0 && cols % 8 == 0, "rowdot_bf16: bad args");
    if (b_fp32) hipLaunchKernelGGL(rowdot_bf16_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, out, rows, cols / 8);
    else hipLaunchKernelGGL(rowdot_bf16_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, out, rows, cols / 8);
    TCVOM_LAUNCH_CHECK("rowdot_bf16");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_value_patches(const void* alpha, void* V, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(alpha && V && C % 8 == 0, "gca_value_patches: bad args");
    hipLaunchKernelGGL(gca_value_patches_kernel, dim3(sgrid((int64_t)B * h8 * w8 / 4 * 16 * C / 8)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4*)alpha, (uint4*)V, B, h8, w8, C / 8);
    TCVOM_LAUNCH_CHECK("gca_value_patches");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_value_patches_bwd(const float* dV, void* dalpha, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(dV && dalpha && C % 4 == 0, "gca_value_patches_bwd: bad args");
    hipLaunchKernelGGL(gca_value_patches_bwd_kernel, dim3(sgrid((int64_t)B * h8 * w8 * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)dV, (uint2*)dalpha, B, h8, w8, C / 4);
    TCVOM_LAUNCH_CHECK("gca_value_patches_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_fold(const void* O, void* Y, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(O && Y && C % 8 == 0, "gca_fold: bad args");
    hipLaunchKernelGGL(gca_fold_kernel<false>, dim3(sgrid((int64_t)B * h8 * w8 * C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)O, (uint4*)Y, B, h8, w8, C / 8);
    TCVOM_LAUNCH_CHECK("gca_fold");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_fold_f32(const float* O, void* Y, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(O && Y && C % 8 == 0, "gca_fold_f32: bad args");
    hipLaunchKernelGGL(gca_fold_kernel<true>, dim3(sgrid((int64_t)B * h8 * w8 * C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)O, (uint4*)Y, B, h8, w8, C / 8);
    TCVOM_LAUNCH_CHECK("gca_fold_f32");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_unfold(const void* dY, void* dO, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(dY && dO && C % 8 == 0, "gca_unfold: bad args");
    hipLaunchKernelGGL(gca_unfold_kernel, dim3(sgrid((int64_t)B * h8 * w8 / 4 * 16 * C / 8)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4*)dY, (uint4*)dO, B, h8, w8, C / 8);
    TCVOM_LAUNCH_CHECK("gca_unfold");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_patches_bwd(float* dWq, const float* Mp, const void* G, const float* nrm, void* dg8, int32_t B,
                                     int32_t h8, int32_t w8, int32_t CG, void* stream) {
    TCVOM_CHECK_ARG(dWq && Mp && G && nrm && dg8, "gca_patches_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t keys = (int64_t)B * (h8 / 2) * (w8 / 2);
    hipLaunchKernelGGL(gca_patch_grad_kernel, dim3(cdiv(keys, 4)), dim3(256), 0, st, dWq, Mp, (const h16raw*)G, nrm, keys, 9 * CG);
    hipLaunchKernelGGL(gca_patches_bwd_kernel, dim3(sgrid((int64_t)B * h8 * w8 * CG)), dim3(256), 0, st, dWq, (h16raw*)dg8, B, h8, w8, CG);
    TCVOM_LAUNCH_CHECK("gca_patches_bwd");
    return TCVOM_OK;
}
