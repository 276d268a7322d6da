// Implicit-GEMM convolution engine on bf16 MFMA (v_mfma_f32_32x32x16_bf16) for gfx950.
//
//   igemm_nt : out[pixel][k] = sum_{tap,c} in[gather(pixel,tap)][c] * w[k][tap][c]
//              (conv fwd, data-gradient via re-packed weights, transposed-conv phases, 1x1,
//               and with ntaps==1 a dense  C^T[n][m] = sum_k A[m][k] B[n][k]  GEMM)
//   igemm_tt : dw[k][tap][c] += sum_pixel dy[pixel][k] * in[gather(pixel,tap)][c]
//              (weight gradient; both operands are pixel-major, so 8x8 blocks are transposed
//               in registers on their way to LDS; with ntaps==1 a dense "TT" GEMM)
//
// MFMA roles: rows (M) = output channels, columns (N) = pixels, so that a lane's 4 consecutive
// accumulator registers are 4 consecutive channels of ONE pixel -> 8-byte NHWC stores, and the
// BatchNorm statistics of a channel are a reduction across lanes.
// LDS tiles are [row][32 k + 8 pad] bf16 (80-byte rows): ds_read_b128 of 16 consecutive rows hits
// 16 distinct 4-bank slots (20*r mod 64), i.e. conflict-free for the MFMA fragment reads.
#include "common.h"

#define NT_LDS_STRIDE 40   // bf16 elements per LDS row in igemm_nt (32 + 8 pad)
#define TT_LDS_STRIDE 72   // bf16 elements per LDS row in igemm_tt (64 + 8 pad)

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_nt_kernel(
    const bf16raw* __restrict__ in, const bf16raw* __restrict__ wgt, void* __restrict__ outp,
    const float* __restrict__ bias, const float* __restrict__ mscale, const float* __restrict__ mdiag,
    float* __restrict__ stats, const tcvom_conv_desc d)
{
    constexpr int WAVES_N = TN / WN;
    constexpr int WAVES_M = TM / WM;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_IT = (TM * 4 + 255) / 256;
    constexpr int B_IT = (TN * 4) / 256;

    __shared__ __attribute__((aligned(16))) bf16raw lds[2 * (TM + TN) * NT_LDS_STRIDE + 8 * TCVOM_MAX_TAPS];
    bf16raw* As = lds;                                   // [2][TM][40]
    bf16raw* Bs = lds + 2 * TM * NT_LDS_STRIDE;          // [2][TN][40]
    int* taps = reinterpret_cast<int*>(lds + 2 * (TM + TN) * NT_LDS_STRIDE);   // [16][3]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    if (d.batch > 1) {
        const int64_t z = blockIdx.z;
        in += z * d.in_bstride;
        wgt += z * d.w_bstride;
        if (bias) bias += z * d.vec_bstride;
        if (mscale) mscale += z * d.vec_bstride;
        if (mdiag) mdiag += z * d.vec_bstride;
    }
    if (tid < TCVOM_MAX_TAPS) {
        taps[tid * 3 + 0] = d.tap_dh[tid];
        taps[tid * 3 + 1] = d.tap_dw[tid];
        taps[tid * 3 + 2] = d.tap_w[tid];
    }
    const int C = d.C, H = d.H, W = d.W, K = d.K, WT = d.wt;
    const int cshift = (d.ntaps == 1) ? 31 : __builtin_ctz(C);
    const int cmask = (d.ntaps == 1) ? 0x7fffffff : (C - 1);
    const int Ptot = d.N * d.PH * d.PW;
    const int p0 = blockIdx.x * TN;
    const int m0 = blockIdx.y * TM;

    // per-thread pixel rows of the B tile (fixed for the whole reduction loop)
    int b_ih0[B_IT], b_iw0[B_IT], b_nb[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int row = (tid + it * 256) >> 2;
        const int p = p0 + row;
        if (p < Ptot) {
            const int j = p % d.PW;
            const int t = p / d.PW;
            const int i = t % d.PH;
            const int n = t / d.PH;
            b_ih0[it] = i * d.in_step;
            b_iw0[it] = j * d.in_step;
            b_nb[it] = n * H;
        } else {
            b_ih0[it] = -(1 << 28);
            b_iw0[it] = 0;
            b_nb[it] = 0;
        }
    }
    __syncthreads();

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nstage = (d.ntaps * C) >> 5;
    uint4 areg[A_IT], breg[B_IT];

#define NT_LOAD_STAGE(s)                                                                        \
    {                                                                                           \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                   \
            const int q = tid + it * 256;                                                       \
            const int row = q >> 2, kk = (s) * 32 + (q & 3) * 8;                                \
            uint4 v = make_uint4(0, 0, 0, 0);                                                   \
            if (TM * 4 >= 256 || q < TM * 4) {                                                  \
                const int tap = kk >> cshift, c0 = kk & cmask;                                  \
                const int ws = taps[tap * 3 + 2];                                               \
                const int m = m0 + row;                                                         \
                if (ws >= 0 && m < K)                                                           \
                    v = *reinterpret_cast<const uint4*>(wgt + ((int64_t)m * WT + ws) * C + c0); \
            }                                                                                   \
            areg[it] = v;                                                                       \
        }                                                                                       \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                   \
            const int q = tid + it * 256;                                                       \
            const int kk = (s) * 32 + (q & 3) * 8;                                              \
            const int tap = kk >> cshift, c0 = kk & cmask;                                      \
            const int ih = b_ih0[it] + taps[tap * 3 + 0], iw = b_iw0[it] + taps[tap * 3 + 1];   \
            uint4 v = make_uint4(0, 0, 0, 0);                                                   \
            if (ih >= 0 && ih < H && iw >= 0 && iw < W)                                         \
                v = *reinterpret_cast<const uint4*>(in + ((int64_t)(b_nb[it] + ih) * W + iw) * C + c0); \
            breg[it] = v;                                                                       \
        }                                                                                       \
    }
#define NT_STORE_STAGE(buf)                                                                     \
    {                                                                                           \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                   \
            const int q = tid + it * 256;                                                       \
            if (TM * 4 >= 256 || q < TM * 4)                                                    \
                *reinterpret_cast<uint4*>(As + ((buf) * TM + (q >> 2)) * NT_LDS_STRIDE + (q & 3) * 8) = areg[it]; \
        }                                                                                       \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                   \
            const int q = tid + it * 256;                                                       \
            *reinterpret_cast<uint4*>(Bs + ((buf) * TN + (q >> 2)) * NT_LDS_STRIDE + (q & 3) * 8) = breg[it]; \
        }                                                                                       \
    }

    NT_LOAD_STAGE(0);
    NT_STORE_STAGE(0);
    __syncthreads();

    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) NT_LOAD_STAGE(s + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[MI], bfr[NI];
            const int kofs = kk * 16 + (lane >> 5) * 8;
#pragma unroll
            for (int a = 0; a < MI; ++a)
                af[a] = *reinterpret_cast<const bf16x8_t*>(
                    As + (buf * TM + wm * WM + a * 32 + (lane & 31)) * NT_LDS_STRIDE + kofs);
#pragma unroll
            for (int b = 0; b < NI; ++b)
                bfr[b] = *reinterpret_cast<const bf16x8_t*>(
                    Bs + (buf * TN + wn * WN + b * 32 + (lane & 31)) * NT_LDS_STRIDE + kofs);
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (s + 1 < nstage) NT_STORE_STAGE(buf ^ 1);
        __syncthreads();
    }
#undef NT_LOAD_STAGE
#undef NT_STORE_STAGE

    // ------------------------------------------------------------------ epilogue
    int64_t out_off[NI];
    bool pvalid[NI];
    int pglob[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int p = p0 + wn * WN + b * 32 + (lane & 31);
        pvalid[b] = p < Ptot;
        pglob[b] = p;
        const int pp = pvalid[b] ? p : 0;
        const int j = pp % d.PW;
        const int t = pp / d.PW;
        const int i = t % d.PH;
        const int n = t / d.PH;
        out_off[b] = ((int64_t)(n * d.OH + i * d.out_step + d.out_off_h) * d.OW + j * d.out_step + d.out_off_w) * d.ldo;
        if (d.batch > 1) out_off[b] += (int64_t)blockIdx.z * d.out_bstride;
    }
    const bool do_stats = stats != nullptr;
#pragma unroll
    for (int a = 0; a < MI; ++a) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int mrow = m0 + wm * WM + a * 32 + 8 * g + 4 * (lane >> 5);   // first of 4 rows
            float bs[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (mrow + r < K) {
                    if (bias) bs[r] = bias[mrow + r];
                    if (mscale) sc[r] = mscale[mrow + r];
                    if (mdiag) dg[r] = mdiag[mrow + r];
                }
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[a][b][g * 4 + r] * sc[r] + bs[r];
                    if (mdiag && (mrow + r) == pglob[b]) x -= dg[r];
                    if (d.act == 1) x = fmaxf(x, 0.f);
                    v[r] = x;
                    if (pvalid[b]) { s1[r] += x; s2[r] += x * x; }
                }
                if (pvalid[b] && mrow < K) {
                    if (d.out_fp32) {
                        float* o = reinterpret_cast<float*>(outp) + out_off[b] + mrow;
                        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        bf16raw* o = reinterpret_cast<bf16raw*>(outp) + out_off[b] + mrow;
                        *reinterpret_cast<uint2*>(o) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    }
                }
            }
            if (do_stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) {
                        s1[r] += __shfl_xor(s1[r], o, 64);
                        s2[r] += __shfl_xor(s2[r], o, 64);
                    }
                }
                if ((lane & 31) == 0 && mrow < K) {
                    const int64_t grp = d.stats_group_offset + (int64_t)blockIdx.x * WAVES_N + wn;
                    float* sp = stats + grp * 2 * K;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sp[mrow + r] = s1[r];
                        sp[K + mrow + r] = s2[r];
                    }
                }
            }
        }
    }
}

// Tile configuration.  The ResNet layers all have the same FLOP count but very different pixel counts: at
// os16/os32 (8160 / 2040 pixels at 1080p) a 128x128 tiling yields only 64-128 workgroups for 256 CUs, so those
// layers switch to 64x64 tiles (4x the workgroups, ~2.5x the co-resident workgroups per CU).
struct NtCfg { int tm, tn, waves_n; };
static NtCfg nt_config(const tcvom_conv_desc* d) {
    const long long P = (long long)d->N * d->PH * d->PW;
    const int nb = d->batch > 1 ? d->batch : 1;
    if (d->K >= 128) {
        const long long wgs = (long long)cdiv(P, 128) * cdiv(d->K, 128) * nb;
        if (wgs < 512) return {64, 64, 2};
        return {128, 128, 2};
    }
    if (d->K > 32) return {64, 256, 4};
    return {32, 256, 4};
}

extern "C" int tcvom_conv_stats_groups(const tcvom_conv_desc* d) {
    const NtCfg c = nt_config(d);
    const long long P = (long long)d->N * d->PH * d->PW;
    return cdiv(P, c.tn) * c.waves_n;
}

extern "C" int tcvom_conv_igemm(const void* in, const void* w, void* out, const float* bias,
                                const float* mscale, const float* mdiag, float* stats_partial,
                                const tcvom_conv_desc* d, void* stream) {
    TCVOM_CHECK_ARG(in && w && out && d, "conv_igemm: null pointer");
    TCVOM_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= TCVOM_MAX_TAPS, "conv_igemm: ntaps=%d", d->ntaps);
    TCVOM_CHECK_ARG(((long long)d->ntaps * d->C) % 32 == 0, "conv_igemm: ntaps*C=%d not a multiple of 32", d->ntaps * d->C);
    TCVOM_CHECK_ARG(d->ntaps == 1 || (d->C >= 8 && (d->C & (d->C - 1)) == 0), "conv_igemm: C=%d must be a power of two >= 8", d->C);
    TCVOM_CHECK_ARG(d->K % 4 == 0 && d->ldo % 4 == 0, "conv_igemm: K=%d ldo=%d must be multiples of 4", d->K, d->ldo);
    TCVOM_CHECK_ARG(d->C % 8 == 0, "conv_igemm: C=%d must be a multiple of 8", d->C);
    const long long P = (long long)d->N * d->PH * d->PW;
    TCVOM_CHECK_ARG(P > 0 && P < (1ll << 31), "conv_igemm: bad pixel count %lld", P);
    const int nb = d->batch > 1 ? d->batch : 1;
    hipStream_t st = (hipStream_t)stream;
    const bf16raw* ip = (const bf16raw*)in;
    const bf16raw* wp = (const bf16raw*)w;
    const NtCfg c = nt_config(d);
    dim3 grid(cdiv(P, c.tn), cdiv(d->K, c.tm), nb);
    if (c.tm == 128)
        hipLaunchKernelGGL((igemm_nt_kernel<128, 128, 64, 64>), grid, dim3(256), 0, st, ip, wp, out, bias, mscale, mdiag, stats_partial, *d);
    else if (c.tm == 64 && c.tn == 64)
        hipLaunchKernelGGL((igemm_nt_kernel<64, 64, 32, 32>), grid, dim3(256), 0, st, ip, wp, out, bias, mscale, mdiag, stats_partial, *d);
    else if (c.tm == 64)
        hipLaunchKernelGGL((igemm_nt_kernel<64, 256, 64, 64>), grid, dim3(256), 0, st, ip, wp, out, bias, mscale, mdiag, stats_partial, *d);
    else
        hipLaunchKernelGGL((igemm_nt_kernel<32, 256, 32, 64>), grid, dim3(256), 0, st, ip, wp, out, bias, mscale, mdiag, stats_partial, *d);
    TCVOM_LAUNCH_CHECK("conv_igemm");
    return TCVOM_OK;
}

// =====================================================================================
// igemm_tt: weight gradient / dense TT GEMM.
//   columns of the B operand are (tap, c) pairs: col = t*C + c  ->  dw[(m*WT + wslot[t])*C + c]
// =====================================================================================
__device__ __forceinline__ void transpose8x8_store(const uint4* rows, bf16raw* dst /* &X[ch0][pix0] */) {
    // rows[p] = 8 channels of pixel p (4 dwords); write 8 LDS rows (one per channel) of 8 pixels.
    const unsigned* rw = reinterpret_cast<const unsigned*>(rows);   // rw[p*4 + d]
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        uint4 ev, od;
        unsigned* e = reinterpret_cast<unsigned*>(&ev);
        unsigned* o = reinterpret_cast<unsigned*>(&od);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned lo = rw[(2 * q) * 4 + dd], hi = rw[(2 * q + 1) * 4 + dd];
            e[q] = (lo & 0xffffu) | (hi << 16);
            o[q] = (lo >> 16) | (hi & 0xffff0000u);
        }
        *reinterpret_cast<uint4*>(dst + (2 * dd) * TT_LDS_STRIDE) = ev;
        *reinterpret_cast<uint4*>(dst + (2 * dd + 1) * TT_LDS_STRIDE) = od;
    }
}

template <int TM, int TN, int WM, int WN, int KS>
__global__ __launch_bounds__(256) void igemm_tt_kernel(
    const bf16raw* __restrict__ dy, const bf16raw* __restrict__ in, float* __restrict__ dw,
    const tcvom_conv_desc d, const int ldy, const int pchunk)
{
    constexpr int WAVES_M = TM / WM, WAVES_N = TN / WN;
    static_assert(WAVES_M * WAVES_N * KS == 4, "4 waves per workgroup");
    static_assert(TM + TN <= 256, "one loader task per thread");
    constexpr int MI = WM / 32, NI = WN / 32;

    __shared__ __attribute__((aligned(16))) bf16raw lds[2 * (TM + TN) * TT_LDS_STRIDE + 8 * TCVOM_MAX_TAPS];
    bf16raw* As = lds;                                  // [2][TM][72]
    bf16raw* Bs = lds + 2 * TM * TT_LDS_STRIDE;         // [2][TN][72]
    int* taps = reinterpret_cast<int*>(lds + 2 * (TM + TN) * TT_LDS_STRIDE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (WAVES_M * WAVES_N);
    const int wmn = wave % (WAVES_M * WAVES_N);
    const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;

    if (tid < TCVOM_MAX_TAPS) {
        taps[tid * 3 + 0] = d.tap_dh[tid];
        taps[tid * 3 + 1] = d.tap_dw[tid];
        taps[tid * 3 + 2] = d.tap_w[tid];
    }
    __syncthreads();

    const int C = d.C, H = d.H, W = d.W, K = d.K;
    const int cshift = (d.ntaps == 1) ? 31 : __builtin_ctz(C);
    const int cmask = (d.ntaps == 1) ? 0x7fffffff : (C - 1);
    const int ncols = d.ntaps * C;
    const int Ptot = d.N * d.PH * d.PW;
    const int pbeg = blockIdx.x * pchunk;
    const int pend = min(Ptot, pbeg + pchunk);
    const int n0 = blockIdx.y * TN;
    const int m0 = blockIdx.z * TM;

    // loader role of this thread
    const bool isA = tid < TM;
    const bool isB = !isA && tid < TM + TN;
    const int lt = isA ? tid : tid - TM;
    const int po = lt & 7, co = lt >> 3;
    int tdh = 0, tdw = 0, c0 = 0;
    bool colok = false;
    if (isA) {
        colok = (m0 + co * 8 + 8) <= ldy;
    } else if (isB) {
        const int col = n0 + co * 8;
        if (col < ncols) {
            const int tap = col >> cshift;
            c0 = col & cmask;
            tdh = taps[tap * 3 + 0];
            tdw = taps[tap * 3 + 1];
            colok = taps[tap * 3 + 2] >= 0;
        }
    }

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    uint4 rows[8];
    const int nstage = (pend - pbeg + 63) >> 6;

#define TT_LOAD_STAGE(s)                                                                         \
    {                                                                                            \
        int p = pbeg + (s) * 64 + po * 8;                                                        \
        int j = p % d.PW;                                                                        \
        int tq = p / d.PW;                                                                       \
        int i = tq % d.PH;                                                                       \
        int n = tq / d.PH;                                                                       \
        _Pragma("unroll") for (int pp = 0; pp < 8; ++pp) {                                       \
            uint4 v = make_uint4(0, 0, 0, 0);                                                    \
            if (colok && p < pend) {                                                             \
                if (isA) {                                                                       \
                    const int64_t off = ((int64_t)(n * d.OH + i * d.out_step + d.out_off_h) * d.OW \
                                         + j * d.out_step + d.out_off_w) * ldy + m0 + co * 8;    \
                    v = *reinterpret_cast<const uint4*>(dy + off);                               \
                } else if (isB) {                                                                \
                    const int ih = i * d.in_step + tdh, iw = j * d.in_step + tdw;                \
                    if (ih >= 0 && ih < H && iw >= 0 && iw < W)                                  \
                        v = *reinterpret_cast<const uint4*>(in + ((int64_t)(n * H + ih) * W + iw) * C + c0); \
                }                                                                                \
            }                                                                                    \
            rows[pp] = v;                                                                        \
            ++p; ++j;                                                                            \
            if (j == d.PW) { j = 0; ++i; if (i == d.PH) { i = 0; ++n; } }                        \
        }                                                                                        \
    }
#define TT_STORE_STAGE(buf)                                                                      \
    {                                                                                            \
        if (isA) transpose8x8_store(rows, As + ((buf) * TM + co * 8) * TT_LDS_STRIDE + po * 8);  \
        else if (isB) transpose8x8_store(rows, Bs + ((buf) * TN + co * 8) * TT_LDS_STRIDE + po * 8); \
    }

    if (nstage > 0) {
        TT_LOAD_STAGE(0);
        TT_STORE_STAGE(0);
    }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) TT_LOAD_STAGE(s + 1);
#pragma unroll
        for (int kq = 0; kq < 4 / KS; ++kq) {
            const int ks = (KS == 1) ? kq : wk;
            const int kofs = ks * 16 + (lane >> 5) * 8;
            bf16x8_t af[MI], bfr[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a)
                af[a] = *reinterpret_cast<const bf16x8_t*>(
                    As + (buf * TM + wm * WM + a * 32 + (lane & 31)) * TT_LDS_STRIDE + kofs);
#pragma unroll
            for (int b = 0; b < NI; ++b)
                bfr[b] = *reinterpret_cast<const bf16x8_t*>(
                    Bs + (buf * TN + wn * WN + b * 32 + (lane & 31)) * TT_LDS_STRIDE + kofs);
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        if (s + 1 < nstage) TT_STORE_STAGE(buf ^ 1);
        __syncthreads();
    }
#undef TT_LOAD_STAGE
#undef TT_STORE_STAGE

#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int col = n0 + wn * WN + b * 32 + (lane & 31);
        if (col >= ncols) continue;
        const int tap = col >> cshift, cc = col & cmask;
        const int ws = taps[tap * 3 + 2];
        if (ws < 0) continue;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < K) atomicAdd(dw + ((int64_t)m * d.wt + ws) * C + cc, acc[a][b][r]);
            }
        }
    }
}

extern "C" int tcvom_wgrad_igemm(const void* dy, const void* in, float* dw, const tcvom_conv_desc* d,
                                 int32_t ldy, void* stream) {
    TCVOM_CHECK_ARG(dy && in && dw && d, "wgrad_igemm: null pointer");
    TCVOM_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= TCVOM_MAX_TAPS, "wgrad_igemm: ntaps=%d", d->ntaps);
    TCVOM_CHECK_ARG(d->ntaps == 1 || (d->C >= 8 && (d->C & (d->C - 1)) == 0), "wgrad_igemm: C=%d must be a power of two >= 8", d->C);
    TCVOM_CHECK_ARG(d->C % 8 == 0 && ldy % 8 == 0, "wgrad_igemm: C=%d ldy=%d must be multiples of 8", d->C, ldy);
    const long long P = (long long)d->N * d->PH * d->PW;
    TCVOM_CHECK_ARG(P > 0 && P < (1ll << 31), "wgrad_igemm: bad pixel count %lld", P);
    hipStream_t st = (hipStream_t)stream;
    const int ncols = d->ntaps * d->C;
    const bf16raw* a = (const bf16raw*)dy;
    const bf16raw* b = (const bf16raw*)in;
    // tile: wide-N tiles for the small-channel layers so that dy is re-read ncols/128 (not ncols/32) times
    int tm, tn;
    if (d->K >= 128 && ncols >= 128) { tm = tn = 128; }
    else if (d->K > 32) { tm = 64; tn = ncols >= 128 ? 128 : 64; }
    else { tm = 32; tn = ncols >= 128 ? 128 : 32; }
    const int mt = cdiv(d->K, tm), nt = cdiv(ncols, tn);
    // pixel chunks: every workgroup ends with tm*tn atomic adds, so chunks must be long enough to amortise
    // them (>= 512 pixels) while still giving ~3 workgroups per CU
    long long want = 768 / ((long long)mt * nt);
    if (want < 1) want = 1;
    long long pchunk = ((P + want - 1) / want + 63) / 64 * 64;
    if (pchunk < 512) pchunk = 512;
    const int chunks = cdiv(P, pchunk);
    dim3 grid(chunks, nt, mt);
    if (tm == 128)
        hipLaunchKernelGGL((igemm_tt_kernel<128, 128, 64, 64, 1>), grid, dim3(256), 0, st, a, b, dw, *d, ldy, (int)pchunk);
    else if (tm == 64 && tn == 128)
        hipLaunchKernelGGL((igemm_tt_kernel<64, 128, 32, 64, 1>), grid, dim3(256), 0, st, a, b, dw, *d, ldy, (int)pchunk);
    else if (tm == 64)
        hipLaunchKernelGGL((igemm_tt_kernel<64, 64, 32, 32, 1>), grid, dim3(256), 0, st, a, b, dw, *d, ldy, (int)pchunk);
    else if (tn == 128)
        hipLaunchKernelGGL((igemm_tt_kernel<32, 128, 32, 32, 1>), grid, dim3(256), 0, st, a, b, dw, *d, ldy, (int)pchunk);
    else
        hipLaunchKernelGGL((igemm_tt_kernel<32, 32, 32, 32, 4>), grid, dim3(256), 0, st, a, b, dw, *d, ldy, (int)pchunk);
    TCVOM_LAUNCH_CHECK("wgrad_igemm");
    return TCVOM_OK;
}
