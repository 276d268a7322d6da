// Implicit-GEMM convolution engine on bf16 MFMA (v_mfma_f32_32x32x16_bf16) for gfx950.
//
//   igemm_nt : out[pixel][k] = sum_{tap,c} in[gather(pixel,tap)][c] * w[k][tap][c]
//              (conv fwd, data-gradient via re-packed weights, transposed-conv phases, 1x1,
//               and with ntaps==1 a dense  C^T[n][m] = sum_k A[m][k] B[n][k]  GEMM)
//   igemm_tt : dw[k][tap][c] += sum_pixel dy[pixel][k] * in[gather(pixel,tap)][c]
//              (weight gradient; both operands are pixel-major: tiles are DMA'd as they lie and the
//               k-contiguous MFMA fragments come from the LDS transpose read ds_read_b64_tr_b16;
//               with ntaps==1 a dense "TT" GEMM; up to 8 same-shape problems per launch)
//   Descriptors with batch > 1 run `batch` independent problems per launch (the dense GEMMs of GCA, and
//   the S frames of a window through one layer with a per-frame weight copy); up to 4 phase descriptors
//   (sub-pixel phases of a ConvTranspose / stride-2 data gradient) share a launch as well.  Shapes served
//   by the halo-tile direct conv (halo.hip) are dispatched from conv_igemm_launch.
//
// MFMA roles: rows (M) = output channels, columns (N) = pixels, so that a lane's 4 consecutive
// accumulator registers are 4 consecutive channels of ONE pixel -> 8-byte NHWC stores, and the
// BatchNorm statistics of a channel are a reduction across lanes.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include <mutex>

struct TcvomPhases { tcvom_conv_desc d[4]; int n; };


// igemm_nt main loop: 64-deep k-steps, operands DMA'd global->LDS with global_load_lds_dwordx4
// (no VGPR round trip), NST-slot LDS ring (2..4), ONE barrier per k-step:
//     vmcnt(stages still in flight); barrier;  issue DMA of step s+NST-1;  MFMAs on step s
// LDS rows are 128 B (64 bf16) and unpadded (the DMA writes 1 KiB = 8 rows per wave-instruction, lane-linear), so
// bank conflicts are removed by an XOR swizzle applied on the SOURCE side: LDS 16-byte slot `cpos` of row r holds
// k-chunk cpos ^ ((r>>1)&7); a ds_read_b128 lane group (16 rows covering all residues mod 16) then touches 16
// distinct 4-bank quads.  Out-of-image taps and dummy taps read a 16-byte zero page instead of branching.
// p / d and p % d for 0 <= p < 2^24 (exact in fp32): one multiply by the reciprocal and a +-1 fix-up instead of the
// ~40-instruction integer division sequence; the tile prologues and epilogues do several of these per row.
__device__ __forceinline__ void divmod24(int p, int dv, float rcp, int& q, int& r) {
    q = (int)((float)p * rcp);
    r = p - q * dv;
    if (r < 0) { r += dv; --q; }
    else if (r >= dv) { r -= dv; ++q; }
}
__device__ __forceinline__ void divmod_any(int p, int dv, float rcp, bool small, int& q, int& r) {
    if (small) divmod24(p, dv, rcp, q, r);
    else { q = p / dv; r = p - q * dv; }
}
// bits b = 0..15 set where 0 <= x0 + (b - 8) < n
__device__ __forceinline__ unsigned range_bits16(int x0, int n) {
    int lo = 8 - x0, hi = n + 7 - x0;
    lo = lo < 0 ? 0 : lo;
    hi = hi > 15 ? 15 : hi;
    if (hi < lo) return 0u;
    return ((2u << hi) - 1u) & ~((1u << lo) - 1u);
}

#ifdef NT_TRACE
__device__ unsigned long long tcvom_trace_buf[8192];
extern "C" int tcvom_trace_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tcvom_trace_buf), n * sizeof(unsigned long long));
}
#define TRACE(i) if (trace_on && (s) < 256) tcvom_trace_buf[((s) * 4 + (i)) + 4 * 256 * trace_w] = __builtin_readcyclecounter()
#else
#define TRACE(i)
#endif
#ifndef NT_DBG
#define NT_DBG 0            // kernel study builds only: 1 = no LDS reads / MFMAs, 2 = no DMA, 3 = no epilogue
#endif
// XF = 1: IEEE fp16 operands and IEEE fp16 16-bit results whatever the build stores (tcvom_conv_desc.in_f16, common.h)
template <int TM, int TN, int WM, int WN, int NST, int XF = 0>
__global__ __launch_bounds__(((TM / WM) * (TN / WN)) * 64) void igemm_nt_kernel(
    const h16raw* __restrict__ in, const h16raw* __restrict__ wgt, void* __restrict__ outp,
    const float* __restrict__ bias, const float* __restrict__ mscale, const float* __restrict__ mdiag,
    float* __restrict__ stats, const h16raw* __restrict__ zero_page, const TcvomPhases ps)
{
    // up to 4 phases (sub-pixel phases of a transposed conv / stride-2 data gradient) share ONE launch: blockIdx.z
    // selects the phase, so their small grids fill the chip together
#ifdef NT_TRACE
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    const int bcount = ps.d[0].batch > 1 ? ps.d[0].batch : 1;        // blockIdx.z = phase * batch + batch element
    const int phase = ps.n > 1 ? blockIdx.z / bcount : 0;
    const tcvom_conv_desc& d = ps.d[phase];
    const int bz = ps.n > 1 ? blockIdx.z - phase * bcount : blockIdx.z;
    constexpr int WAVES_N = TN / WN;
    constexpr int WAVES_M = TM / WM;
    constexpr int NW = WAVES_M * WAVES_N;              // 4 or 8 waves per workgroup
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_IT = TM / (8 * NW);                // DMA instructions per wave per k-step (8 rows each)
    constexpr int B_IT = TN / (8 * NW);
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the wave count");
    constexpr int SLOT = (TM + TN) * 64;               // bf16 elements per ring slot

    static_assert(NST >= 2 && NST <= 4, "2..4 ring slots");
    __shared__ __attribute__((aligned(16))) h16raw lds[NST * SLOT + 8 * TCVOM_MAX_TAPS];
    int4* taps = reinterpret_cast<int4*>(lds + NST * SLOT);   // per tap: input offset, weight offset (-1: dummy), mask bits

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int Ptot = d.N * d.PH * d.PW;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed, speed only); give every XCD a contiguous run
    // of pixel tiles so that the 3x3 halo rows shared by neighbouring tiles are served by ONE L2 (bijective for any
    // grid size)
    int bx;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (bx * TN >= Ptot) return;                       // phases may have fewer tiles than the launch grid
    if (d.batch > 1) {
        const int64_t z = bz;
        in += z * d.in_bstride;
        wgt += z * d.w_bstride;
        if (bias) bias += z * d.vec_bstride;
        if (mscale) mscale += z * d.vec_bstride;
        if (mdiag) mdiag += z * d.vec_bstride;
    }
    const int C = d.C, H = d.H, W = d.W, K = d.K, WT = d.wt;
    if (tid < TCVOM_MAX_TAPS) {
        const int dh = d.tap_dh[tid], dw = d.tap_dw[tid];
        const bool ok = tid < d.ntaps && d.tap_w[tid] >= 0;
        // bit positions of this tap in a row's validity word: rows bit dh+8, columns bit 16+dw+8 (|dh|,|dw| <= 7)
        taps[tid] = make_int4((dh * W + dw) * C, ok ? d.tap_w[tid] * C : -1, dh + 8, ok ? dw + 8 : 31);
    }
    // tap = kk / C through a multiply-high (exact for kk < 2^32 / C; one tap: magic 0 -> tap 0): C is any multiple of 8
    const unsigned cmagic = (d.ntaps == 1 && (C & 63) == 0) ? 0u : (unsigned)((0x100000000ull + (unsigned)C - 1) / (unsigned)C);
    const int p0 = bx * TN;
    const int m0 = blockIdx.y * TM;
    const bool small_p = Ptot < (1 << 24);
    const float rcp_pw = 1.0f / (float)d.PW, rcp_ph = 1.0f / (float)d.PH;

    // this lane's k-chunk within a 64-deep step (same for all of its DMA instructions, see header comment)
    const int kc8 = (((lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7)) << 3);

    // per-thread rows: A rows (weights) and B rows (pixels) are fixed for the whole reduction loop
    int a_off[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + (it * NW + wave) * 8 + (lane >> 3);
        a_off[it] = m < K ? m * WT * C : -1;
    }
    // b_valid: bit t set when tap t of this pixel row falls inside the image (and is not a dummy tap)
    int b_off[B_IT];
    unsigned b_valid[B_IT], b_hm[B_IT], b_wm[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int p = p0 + (it * NW + wave) * 8 + (lane >> 3);
        b_off[it] = 0;
        b_valid[it] = b_hm[it] = b_wm[it] = 0u;
        if (p < Ptot) {
            int t, j, n, i;
            divmod_any(p, d.PW, rcp_pw, small_p, t, j);
            divmod_any(t, d.PH, rcp_ph, small_p, n, i);
            const int ih0 = i * d.in_step, iw0 = j * d.in_step;
            b_off[it] = ((n * H + ih0) * W + iw0) * C;
            b_hm[it] = range_bits16(ih0, H);
            b_wm[it] = range_bits16(iw0, W);
        }
    }
    __syncthreads();
    {
        const int nt = d.ntaps;
#pragma unroll 4
        for (int tp = 0; tp < nt; ++tp) {
            const int4 tq = taps[tp];
#pragma unroll
            for (int it = 0; it < B_IT; ++it) b_valid[it] |= (((b_hm[it] >> tq.z) & (b_wm[it] >> tq.w)) & 1u) << tp;
        }
    }

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nstage = (d.ntaps * C + 63) >> 6;          // (a partial last step ends in zero-tap slots)
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

#define NT_ISSUE_STAGE(s, slot)                                                                            \
    {                                                                                                      \
        const int kk = (s) * 64 + kc8;                                                                     \
        const int tap = (int)__umulhi((unsigned)kk, cmagic), c0 = kk - tap * C;                            \
        const int tin = reinterpret_cast<const int*>(taps)[tap * 4 + 0];                                   \
        const int tw = reinterpret_cast<const int*>(taps)[tap * 4 + 1];                                    \
        h16raw* abase = lds + (slot) * SLOT;                                                              \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                              \
            const h16raw* src = (a_off[it] >= 0 && tw >= 0) ? wgt + (a_off[it] + tw + c0) : zero_page;    \
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abase + (it * NW + wave) * 512), 16, 0, 0); \
        }                                                                                                  \
        h16raw* bbase = abase + TM * 64;                                                                  \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                              \
            const h16raw* src = ((b_valid[it] >> tap) & 1u) ? in + ((int64_t)b_off[it] + tin + c0) : zero_page; \
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bbase + (it * NW + wave) * 512), 16, 0, 0); \
        }                                                                                                  \
    }

    // fragment read addressing: row r, k-chunk kch -> byte (r*128 + ((kch ^ ((r>>1)&7)) << 4)); (r>>1)&7 is the
    // same for every 32-row fragment of this lane because fragment bases are multiples of 16 rows
    const int a_row = wm * WM + (lane & 31), b_row = wn * WN + (lane & 31);
    const int a_swz = (a_row >> 1) & 7, b_swz = (b_row >> 1) & 7;

    constexpr int LPS = A_IT + B_IT;                    // DMA instructions per wave per stage
    // NST-slot ring: stages s+1 .. s+NST-2 stay in flight across the barrier (counted vmcnt), which hides the
    // DMA latency for the layers that are latency- rather than throughput-bound (small grids, small K)
#if NT_DBG != 2
#pragma unroll
    for (int ps = 0; ps < NST - 1; ++ps)
        if (ps < nstage) NT_ISSUE_STAGE(ps, ps);
#endif
    int slot = 0, islot = NST - 1;
#ifdef NT_TRACE
    const bool trace_on = blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == NW - 1);
    const int trace_w = wave == 0 ? 0 : 1;
#endif
#ifdef NT_TRACE
    if (trace_on) { tcvom_trace_buf[4096 + trace_w * 8 + 0] = t_entry; tcvom_trace_buf[4096 + trace_w * 8 + 1] = __builtin_readcyclecounter(); }
#endif
    for (int s = 0; s < nstage; ++s) {
        TRACE(0);
        if (s + NST - 2 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE(1);
        __builtin_amdgcn_s_barrier();
        TRACE(2);
#if NT_DBG != 2
        if (s + NST - 1 < nstage) NT_ISSUE_STAGE(s + NST - 1, islot);
#endif
        TRACE(3);
        const h16raw* As = lds + slot * SLOT;
        const h16raw* Bs = As + TM * 64;
#if NT_DBG != 1
#pragma unroll
#endif
        for (int kk = 0; kk < (NT_DBG == 1 ? 0 : 4); ++kk) {
            h16x8_t af[MI], bfr[NI];
            const int kch = kk * 2 + (lane >> 5);
#pragma unroll
            for (int a = 0; a < MI; ++a)
                af[a] = *reinterpret_cast<const h16x8_t*>(As + (a_row + a * 32) * 64 + ((kch ^ a_swz) << 3));
#pragma unroll
            for (int b = 0; b < NI; ++b)
                bfr[b] = *reinterpret_cast<const h16x8_t*>(Bs + (b_row + b * 32) * 64 + ((kch ^ b_swz) << 3));
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = mfma16x<XF>(af[a], bfr[b], acc[a][b]);
        }
        slot = slot + 1 == NST ? 0 : slot + 1;
        islot = islot + 1 == NST ? 0 : islot + 1;
    }
#ifdef NT_TRACE
    if (trace_on) tcvom_trace_buf[4096 + trace_w * 8 + 2] = __builtin_readcyclecounter();
#endif
#undef NT_ISSUE_STAGE

#if NT_DBG == 3
    if (acc[0][0][0] != 123.25f && acc[MI - 1][NI - 1][15] != 7.f && acc[0][NI - 1][3] != 1.f && acc[MI - 1][0][9] != 2.f) return;
#endif
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
        int t, j, n, i;
        divmod_any(pp, d.PW, rcp_pw, small_p, t, j);
        divmod_any(t, d.PH, rcp_ph, small_p, n, i);
        out_off[b] = ((int64_t)(n * d.OH + i * d.out_step + d.out_off_h) * d.OW + j * d.out_step + d.out_off_w) * d.ldo;
        if (d.batch > 1) out_off[b] += (int64_t)bz * d.out_bstride;
    }
    // Straight-line form: the activation as max(x, slope * x); the per-row vectors as 16-byte loads (K % 4 == 0: a lane's 4 rows
    // are valid together); the diagonal term and the output type decided once per tile, not per element; the 8 channel sums of a
    // row group cross the 32 pixel lanes with DPP adds.  (With per-element `if (act == ..)`, `if (mdiag ..)`, scalar coefficient
    // loads and a ds_bpermute butterfly per group this took 6.3 k cycles per workgroup -- a quarter of an os16 layer's workgroup.)
    const bool do_stats = stats != nullptr;
    const float slope = d.act == 1 ? 0.f : d.act == 3 ? 0.01f : 1.f;
    const bool vec_ok = (((uintptr_t)bias | (uintptr_t)mscale | (uintptr_t)mdiag) & 15) == 0;     // 16-byte loads of the row vectors
    const int64_t sgrp = do_stats ? d.stats_group_offset + bz * d.stats_bstride + (int64_t)bx * WAVES_N + wn : 0;
    auto emit = [&](auto diag_, auto f32_) {
        constexpr bool DIAG = decltype(diag_)::value, F32 = decltype(f32_)::value;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int mrow = m0 + wm * WM + a * 32 + 8 * g + 4 * (lane >> 5);   // first of 4 rows
                float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), dg4 = bs4;
                if (mrow < K) {
                    if (vec_ok) {
                        if (bias) bs4 = *reinterpret_cast<const float4*>(bias + mrow);
                        if (mscale) sc4 = *reinterpret_cast<const float4*>(mscale + mrow);
                        if (DIAG) dg4 = *reinterpret_cast<const float4*>(mdiag + mrow);
                    } else {
                        if (bias) bs4 = make_float4(bias[mrow], bias[mrow + 1], bias[mrow + 2], bias[mrow + 3]);
                        if (mscale) sc4 = make_float4(mscale[mrow], mscale[mrow + 1], mscale[mrow + 2], mscale[mrow + 3]);
                        if (DIAG) dg4 = make_float4(mdiag[mrow], mdiag[mrow + 1], mdiag[mrow + 2], mdiag[mrow + 3]);
                    }
                }
                const float bs[4] = {bs4.x, bs4.y, bs4.z, bs4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
                float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[a][b][g * 4 + r] * sc[r] + bs[r];
                        if (DIAG) x -= (mrow + r) == pglob[b] ? dg[r] : 0.f;
                        x = fmaxf(x, x * slope);
                        v[r] = x;
                        const float xs = pvalid[b] ? x : 0.f;
                        t8[r] += xs;
                        t8[4 + r] = fmaf(xs, xs, t8[4 + r]);
                    }
                    if (pvalid[b] && mrow < K) {
                        if (F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + out_off[b] + mrow) = make_float4(v[0], v[1], v[2], v[3]);
                        else *reinterpret_cast<uint2*>(reinterpret_cast<h16raw*>(outp) + out_off[b] + mrow) = make_uint2(pack2x<XF>(v[0], v[1]), pack2x<XF>(v[2], v[3]));
                    }
                }
                if (do_stats) {
                    reduce8_store(t8, lane, stats + sgrp * 2 * K, K, mrow, mrow < K);                  // (common.h: halving butterfly)
                }
            }
        }
    };
    if constexpr (XF) {                  // (the host admits out_fp32 == 2 only: fp16 results, no diagonal term)
        emit(std::false_type{}, std::false_type{});
    } else if (mdiag) {
        if (d.out_fp32) emit(std::true_type{}, std::true_type{}); else emit(std::true_type{}, std::false_type{});
    } else {
        if (d.out_fp32) emit(std::false_type{}, std::true_type{}); else emit(std::false_type{}, std::false_type{});
    }
#ifdef NT_TRACE
    if (blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == NW - 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tcvom_trace_buf[4096 + (wave == 0 ? 0 : 1) * 8 + 3] = __builtin_readcyclecounter();
    }
#endif
}

// Tile configuration.  The ResNet layers all have the same FLOP count but very different pixel counts: at
// os16/os32 (8160 / 2040 pixels at 1080p) a 128x128 tiling yields only 64-128 workgroups for 256 CUs, so those
// layers switch to 64x64 tiles (4x the workgroups, ~2.5x the co-resident workgroups per CU).
struct NtCfg { int tm, tn, waves_n; int nst = 0; };

// 256 zero bytes per device: the source of every out-of-image / dummy tap (allocated once, on first use —
// before any graph capture — and never freed; the only allocation the library ever makes)
static const h16raw* zero_page_for_current_device() {
    static const h16raw* pages[64] = {nullptr};
    static std::mutex mtx;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(mtx);
    if (!pages[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = (const h16raw*)p;
    }
    return pages[dev];
}
const h16raw* tcvom_zero_page(void) { return zero_page_for_current_device(); }
static NtCfg nt_config(const tcvom_conv_desc* d, int nphase) {
    const long long P = (long long)d->N * d->PH * d->PW;
    const int nb = nphase * (d->batch > 1 ? d->batch : 1);
    if (d->K >= 128) {
        const long long wgs = (long long)cdiv(P, 128) * cdiv(d->K, 128) * nb;
        // the dense attention GEMMs of GCA (8160 x 8160 x 576 / 8160 x 2048 x 8160 at 1080p): 256x256 tiles halve the
        // L2->LDS bytes per MFMA, which is what bounds the 128x128 loop (measured 818 -> 1012 TFLOP/s on P.V)
        // 128x128 tiles from 256 workgroups on (3-frame launches of the os16 / os32 layers: 16 vs 18 us per frame in isolation, the
        // step barely moves: 30.02 -> 29.85 ms)
        constexpr int t128 = 256;
        // (measured and dropped, round 3: 256 (channels) x 128 (pixels) tiles for the 256-channel os16 layers -- 192 workgroups, one
        //  per CU, 25 % fewer operand bytes per MAC than two co-resident 128 x 128 workgroups: 25.57 -> 25.80 ms per step; the
        //  128 x 128 threshold lowered to 150 workgroups for the 512-channel os32 layers: no change)
        // (measured, round 3: 4x as many 128 x 128 tiles where the 256 x 256 tile count leaves the last round of 256 mostly empty
        //  -- 384 tiles -> 1536 -- is neutral: 24.19 vs 24.20 ms)
        // (round 4, FBA's fifteen K = 256 os8 launches -- 384 tiles of 256 x 256, 1.5 rounds: as 1530 tiles of 128 x 128 the FBA+TAM step is
        //  53.3 -> 53.5 ms, same box: the better balance does not pay for twice the operand bytes per MAC)
        if (d->K >= 256 && wgs >= 1024) {
            // 256 (channels) x 192 (pixels) tiles where the 256 x 256 grid ends in a half-empty round: FBA's K = 256 os8 layers at 1080p are
            // 128 x 3 = 384 tiles of 256 x 256 -- two rounds on 256 CUs, the second half empty -- but 170 x 3 = 510 of 256 x 192: two rounds
            // of 3/4 the work each (the 128 x 96 rule above, one tile size up; same operand bytes per MAC along the channel side)
            static const bool no192 = getenv("TCVOM_NO_NT_T192") != nullptr;          // A/B switch
            const long long t256 = (long long)cdiv(P, 256) * cdiv(d->K, 256) * nb, t192 = (long long)cdiv(P, 192) * cdiv(d->K, 256) * nb;
            const double c256 = (double)((t256 + 255) / 256), c192 = 0.75 * (double)((t192 + 255) / 256);
            // (3 x 3 / multi-tap layers only: a 1 x 1 conv that falls back from gemm_nt256 to this kernel -- a caller with a bias -- keeps the
            //  256 x 256 tiling, whose statistics-group count equals gemm_nt256's, tcvom_conv_stats_groups)
            if (!no192 && d->ntaps > 1 && c192 < 0.9 * c256) return {256, 192, 2};
            return {256, 256, 4};
        }
        // 128 (channels) x 96 (pixels) tiles, 4 waves of 32 x 96, where they spread evenly over the chip and the 128 x 128 ones do
        // not: the 256-channel os16 layers at 1080p are 64 x 2 x 3 = 384 workgroups of 128 x 128 (half the CUs run two, half one)
        // but 85 x 2 x 3 = 510 of 128 x 96 (two per CU on 255 CUs)
        if (wgs >= t128 && wgs < 1024) {
            const long long w96 = (long long)cdiv(P, 96) * cdiv(d->K, 128) * nb;
            // (both tiles run two workgroups per CU: 512 slots)
            const long long slots = 512;
            auto eff = [slots](long long w) { const long long per = (w + slots - 1) / slots; return (double)w / (double)(per * slots); };
            if (eff(w96) > eff(wgs) + 0.1) return {128, 96, 1};
        }
        if (wgs >= t128) return {128, 128, 4};
        if ((long long)cdiv(P, 64) * cdiv(d->K, 128) * nb >= 400) return {128, 64, 2};
        return {64, 64, 2};
    }
    // K <= 64: short reductions (1 .. 9 K-steps), bound by the latency of ONE workgroup times the rounds of co-resident workgroups:
    // a 2-slot ring (49 KB: three workgroups per CU instead of two) measured +0.1 windows/s over the 3-slot form.
    // (Measured without effect, round 4: 32 x 128 tiles for K <= 32 -- twice the workgroups per CU -- and whole-row stores of the
    //  16-bit outputs through LDS: these layers re-fetch every input pixel once per tap through the L2 -> LDS DMA path, which bounds
    //  them at ~3.5 TB/s of DMA traffic -- ConvTranspose 32 -> 32 at 1088 x 1920: 535 MB through the DMA path for 33 MB of input.)
    constexpr int k64nst = 2;
    if (d->K > 32) return {64, 128, 2, k64nst};
    return {32, 256, 4};
}

// halo.hip: direct conv for the full-resolution small-channel layers
int halo_conv_stats_groups(const tcvom_conv_desc* d, int nphase);
int halo_conv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                         float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream);
// sconv.hip: halo-tile direct conv for the K <= 64 transposed / stride-2-gradient / 64 <-> 32 channel layers
int sconv_stats_groups(const tcvom_conv_desc* d, int nphase);
const char* sconv_variant(const tcvom_conv_desc* d, int nphase);
int sconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                     float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream);
// wsconv.hip: weight-stationary 3x3 conv for the 64 / 128-channel stride-1 layers
int wsconv_stats_groups(const tcvom_conv_desc* d, int nphase);
int wsconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                      float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream);
int gemm_tt256_try_launch(const void* dy, const void* x, float* dw, const tcvom_conv_desc* d, int ldy, void* stream, int nb,
                          long long dy_stride, long long x_stride, long long dw_stride);
int gemm_tt256_takes(const tcvom_conv_desc* d);
// pwconv.hip: weight-stationary streaming 1x1 conv for the reduction-poor pointwise layers (C <= 512)
int pwconv_stats_groups(const tcvom_conv_desc* d, int nphase);
const char* pwconv_variant(const tcvom_conv_desc* d, int nphase);
int pwconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                      float* stats, const tcvom_conv_desc* d, int nphase, void* stream);
// gemm256.hip: staggered two-group 256x256 dense GEMM for the attention GEMMs of GCA
int gemm_nt256_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                          const tcvom_conv_desc* d, const h16raw* zero_page, void* stream, const void* in2, void* out2,
                          long long in2_bstride, float* cstats);
int gemm_nt256_takes(const tcvom_conv_desc* d);
int gemm_nt256_takes_stats(const tcvom_conv_desc* d);
int gemm_nt256_stats_groups(const tcvom_conv_desc* d);
// halo.hip: weight gradient of the 32 -> 32 channel full-resolution layers from LDS-resident x halo / dy tiles
const char* halo_wgrad_variant(const tcvom_conv_desc* d, int ldy);
int halo_wgrad_try_launch(const void* const* dys, const void* const* ins, float* const* dws, int nbatch, const tcvom_conv_desc* d,
                          int nphase, int ldy, const h16raw* zero_page, void* stream);
// wgradws.hip: accumulator-stationary weight gradient of the stride-1 3x3 layers with 64 / 128-multiple channel counts
int wgradws_try_launch(const void* const* dys, const void* const* ins, float* const* dws, int nbatch, const tcvom_conv_desc* d,
                       int nphase, int ldy, void* stream);
const char* wgradws_variant(const tcvom_conv_desc* d, int ldy);

extern "C" int tcvom_conv_stats_groups(const tcvom_conv_desc* d, int32_t nphase) {
    const int hg = halo_conv_stats_groups(d, nphase);
    if (hg > 0) return hg;
    const int sg = sconv_stats_groups(d, nphase);
    if (sg > 0) return sg;
    const int wg = wsconv_stats_groups(d, nphase);
    if (wg > 0) return wg;
    const int pg = pwconv_stats_groups(d, nphase);
    if (pg > 0) return pg;
    // (a 1 x 1 conv on the dense 256-tile GEMM writes its statistics itself; a caller that also passes a bias gets the implicit GEMM,
    //  whose 256 x 256 tiling has the same group count)
    if (nphase == 1 && d->w_layout == 0 && gemm_nt256_takes_stats(d)) return gemm_nt256_stats_groups(d);
    const NtCfg c = nt_config(d, nphase);
    const long long P = (long long)d->N * d->PH * d->PW;
    return cdiv(P, c.tn) * c.waves_n;
}

// name of the kernel instantiation tcvom_conv_igemm(_phases) launches for this shape (profiling / bench labels)
extern "C" const char* tcvom_conv_igemm_variant(const tcvom_conv_desc* d, int32_t nphase) {
    if (halo_conv_stats_groups(d, nphase) > 0) return d->C == 8 ? "halo_conv<8>" : "halo_conv<32>";
    if (const char* sv = sconv_variant(d, nphase)) return sv;
    if (wsconv_stats_groups(d, nphase) > 0) return d->C == 64 ? "wsconv<64>" : "wsconv<128>";
    if (const char* pv = pwconv_variant(d, nphase)) return pv;
    if (nphase == 1 && gemm_nt256_takes(d)) return "gemm_nt256";
    const NtCfg c = nt_config(d, nphase);
    if (c.tm == 256 && c.tn == 192) return "igemm_nt<256,192,64,96,2>";
    if (c.tm == 256) return "igemm_nt<256,256,128,64,2>";
    if (c.tm == 128 && c.tn == 128) return "igemm_nt<128,128,64,32,2>";
    if (c.tm == 128 && c.tn == 96) return "igemm_nt<128,96,32,96,2>";
    if (c.tm == 128) return "igemm_nt<128,64,32,32,3>";
    if (c.tm == 64 && c.tn == 64) return "igemm_nt<64,64,32,32,4>";
    if (c.tm == 64) return "igemm_nt<64,128,32,64,2>";
    return "igemm_nt<32,256,32,64,2>";
}

static int conv_igemm_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale,
                             const float* mdiag, float* stats_partial, const tcvom_conv_desc* descs, int nphase,
                             void* stream) {
    TCVOM_CHECK_ARG(in && w && out && descs, "conv_igemm: null pointer");
    TCVOM_CHECK_ARG(nphase >= 1 && nphase <= 4, "conv_igemm: %d phases (1..4)", nphase);
    TcvomPhases ps;
    ps.n = nphase;
    long long Pmax = 0;
    for (int i = 0; i < nphase; ++i) {
        const tcvom_conv_desc* d = descs + i;
        TCVOM_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= TCVOM_MAX_TAPS, "conv_igemm: ntaps=%d", d->ntaps);
        for (int t = 0; t < d->ntaps; ++t)
            TCVOM_CHECK_ARG(d->tap_dh[t] >= -8 && d->tap_dh[t] <= 7 && d->tap_dw[t] >= -8 && d->tap_dw[t] <= 7,
                            "conv_igemm: tap %d offset (%d,%d) outside [-8,7]", t, d->tap_dh[t], d->tap_dw[t]);
        // (the last 64-deep step may reach past the tap list: those slots count as zero taps and must exist in the table)
        TCVOM_CHECK_ARG((((long long)d->ntaps * d->C + 63) / 64 * 64 - 1) / d->C < TCVOM_MAX_TAPS, "conv_igemm: ntaps*C=%d needs tap slots past the table", d->ntaps * d->C);
        TCVOM_CHECK_ARG(d->ntaps == 1 || (d->C >= 8 && d->C % 8 == 0 && d->C <= 32768), "conv_igemm: C=%d must be a multiple of 8 (8..32768)", d->C);
        TCVOM_CHECK_ARG(d->K % 4 == 0 && d->ldo % 4 == 0, "conv_igemm: K=%d ldo=%d must be multiples of 4", d->K, d->ldo);
        TCVOM_CHECK_ARG(d->C % 8 == 0, "conv_igemm: C=%d must be a multiple of 8", d->C);
        TCVOM_CHECK_ARG(nphase == 1 || (d->batch == descs[0].batch && d->K == descs[0].K), "conv_igemm: phases must share K and the batch count");
        const long long P = (long long)d->N * d->PH * d->PW;
        TCVOM_CHECK_ARG(P > 0 && P < (1ll << 31), "conv_igemm: bad pixel count %lld", P);
        TCVOM_CHECK_ARG((long long)d->N * d->H * d->W * d->C < (1ll << 31) && (long long)d->K * d->wt * d->C < (1ll << 31),
                        "conv_igemm: operand too large for 32-bit element offsets");
        if (P > Pmax) Pmax = P;
        ps.d[i] = *d;
    }
    const tcvom_conv_desc* d0 = descs;
    const int nb = nphase * (d0->batch > 1 ? d0->batch : 1);
    hipStream_t st = (hipStream_t)stream;
    const h16raw* ip = (const h16raw*)in;
    const h16raw* wp = (const h16raw*)w;
    const NtCfg c = nt_config(d0, nphase);
    const h16raw* zp = zero_page_for_current_device();
    TCVOM_CHECK_ARG(zp != nullptr, "conv_igemm: could not allocate the zero page");
    // IEEE fp16 operands in the bf16 build (the fp16 island, tcvom_conv_desc.in_f16): fp16 results; halo_conv / wsconv where their plans
    // take the shape, else the implicit GEMM's XF instantiations
    TCVOM_CHECK_ARG(!(TCVOM_BUILD_F16 && d0->in_f16 != 0), "conv_igemm: in_f16 = 1 in the fp16 build (it has no island: every layer already is IEEE fp16)");
    const bool xf = d0->in_f16 != 0 && !TCVOM_BUILD_F16;
    if (xf) {
        for (int i = 0; i < nphase; ++i)
            TCVOM_CHECK_ARG(descs[i].in_f16 != 0 && descs[i].out_fp32 == 2, "conv_igemm: in_f16 needs out_fp32 == 2 (fp16 results) in every phase");
        TCVOM_CHECK_ARG(!mscale && !mdiag, "conv_igemm: in_f16 with a column scale / diagonal term is not built");
    }
    TCVOM_CHECK_ARG(xf || d0->out_fp32 != 2, "conv_igemm: out_fp32 = 2 (IEEE fp16 results) comes with in_f16 = 1 in the bf16 build only (the fp16 island)");
    for (int i = 0; i < nphase; ++i) TCVOM_CHECK_ARG(xf || descs[i].out_fp32 == 0 || descs[i].out_fp32 == 1, "conv_igemm: out_fp32 = %d", descs[i].out_fp32);
    {
        const int r = halo_conv_try_launch(in, w, out, bias, mscale, mdiag, stats_partial, d0, nphase, zp, stream);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    {
        const int r = sconv_try_launch(in, w, out, bias, mscale, mdiag, stats_partial, d0, nphase, zp, stream);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    {
        const int r = wsconv_try_launch(in, w, out, bias, mscale, mdiag, stats_partial, d0, nphase, zp, stream);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    {
        const int r = pwconv_try_launch(in, w, out, bias, mscale, mdiag, stats_partial, d0, nphase, stream);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    TCVOM_CHECK_ARG(d0->w_layout == 0, "conv_igemm: fragment-major weights (w_layout = 1) are only served by the weight-stationary kernel");
    TCVOM_CHECK_ARG(!(stats_partial && (mscale || mdiag) && nphase == 1 && gemm_nt256_takes_stats(d0)),
                    "conv_igemm: statistics together with a column scale / diagonal term are not built for this shape");
    if (nphase == 1 && (!stats_partial || gemm_nt256_takes_stats(d0))) {
        const int r = gemm_nt256_try_launch(in, w, out, bias, mscale, mdiag, d0, zp, stream, nullptr, nullptr, 0, stats_partial);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    dim3 grid(cdiv(Pmax, c.tn), cdiv(d0->K, c.tm), nb);
#define NT_LAUNCH0(threads, ...)                                                                                         \
    hipLaunchKernelGGL((igemm_nt_kernel<__VA_ARGS__>), grid, dim3(threads), 0, st, ip, wp, out, bias, mscale, mdiag,     \
                       stats_partial, zp, ps)
#define NT_LAUNCH(threads, ...)                                                                                          \
    do {                                                                                                                 \
        if (xf) NT_LAUNCH0(threads, __VA_ARGS__, 1);                                                                     \
        else NT_LAUNCH0(threads, __VA_ARGS__);                                                                           \
    } while (0)
    TCVOM_CHECK_ARG(!(xf && c.tm == 256), "conv_igemm: in_f16 is not built for the 256 x 256 tile (K >= 256 with >= 1024 tiles)");
    if (c.tm == 256 && c.tn == 192) NT_LAUNCH0(512, 256, 192, 64, 96, 2);
    else if (c.tm == 256) NT_LAUNCH0(512, 256, 256, 128, 64, 2);
    else if (c.tm == 128 && c.tn == 128) NT_LAUNCH(512, 128, 128, 64, 32, 2);
    // (a 3- or 4-slot ring for this tile -- 84 / 112 KB of LDS, one workgroup per CU -- measured +0.45 ms per step: two co-resident
    //  workgroups hide the DMA latency better than one with a deeper ring)
    else if (c.tm == 128 && c.tn == 96) NT_LAUNCH(256, 128, 96, 32, 96, 2);
    else if (c.tm == 128) NT_LAUNCH(512, 128, 64, 32, 32, 3);
    else if (c.tm == 64 && c.tn == 64) {
        // 4 ring slots = 64 KB: two workgroups per CU (512 on the chip); 3 slots = 48 KB: three.  A grid of 513 .. 768 workgroups
        // (the 512-channel os32 layers at 1080p: 32 x 8 x 3) is ONE round with three per CU instead of a full and a half one.
        const long long nwg = (long long)grid.x * grid.y * grid.z;
        if (nwg > 512 && nwg <= 768) NT_LAUNCH(256, 64, 64, 32, 32, 3);
        else NT_LAUNCH(256, 64, 64, 32, 32, 4);
    }
    else if (c.tm == 64) NT_LAUNCH(256, 64, 128, 32, 64, 2);
    else NT_LAUNCH(256, 32, 256, 32, 64, 2);
#undef NT_LAUNCH
#undef NT_LAUNCH0
    TCVOM_LAUNCH_CHECK("conv_igemm");
    return TCVOM_OK;
}

// Two dense products against the same weight operand, out1 = in1 x w^T and out2 = in2 x w^T (one descriptor), in ONE launch
// where the 256-tile GEMM takes the shape: the d(query) / d(key) GEMMs of GuidedCxtAtten's backward (M = 576, K = 8192, 3 frames)
// are 288 workgroups each -- 1.125 rounds on 256 CUs, i.e. two rounds of time; together 576 = 2.25 -> three rounds for both.
extern "C" int tcvom_gemm_pair(const void* in1, const void* in2, const void* w, void* out1, void* out2,
                               const tcvom_conv_desc* desc, int64_t in2_bstride, void* stream) {
    TCVOM_CHECK_ARG(in1 && in2 && w && out1 && out2 && desc, "gemm_pair: null pointer");
    static const bool unpaired = getenv("TCVOM_NO_GEMM_PAIR") != nullptr;          // A/B switch
    if (!unpaired && gemm_nt256_takes(desc)) {
        const h16raw* zp = zero_page_for_current_device();
        TCVOM_CHECK_ARG(zp != nullptr, "gemm_pair: could not allocate the zero page");
        TCVOM_CHECK_ARG(desc->w_layout == 0, "gemm_pair: plain weight layout only");
        const int r = gemm_nt256_try_launch(in1, w, out1, nullptr, nullptr, nullptr, desc, zp, stream, in2, out2, in2_bstride, nullptr);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    const int r1 = conv_igemm_launch(in1, w, out1, nullptr, nullptr, nullptr, nullptr, desc, 1, stream);
    if (r1 != TCVOM_OK) return r1;
    tcvom_conv_desc d2 = *desc;
    d2.in_bstride = in2_bstride;
    return conv_igemm_launch(in2, w, out2, nullptr, nullptr, nullptr, nullptr, &d2, 1, stream);
}

extern "C" int tcvom_conv_igemm(const void* in, const void* w, void* out, const float* bias,
                                const float* mscale, const float* mdiag, float* stats_partial,
                                const tcvom_conv_desc* d, void* stream) {
    return conv_igemm_launch(in, w, out, bias, mscale, mdiag, stats_partial, d, 1, stream);
}

extern "C" int tcvom_conv_igemm_phases(const void* in, const void* w, void* out, const float* bias, float* stats_partial,
                                       const tcvom_conv_desc* descs, int32_t nphase, void* stream) {
    return conv_igemm_launch(in, w, out, bias, nullptr, nullptr, stats_partial, descs, nphase, stream);
}

// =====================================================================================
// igemm_tt: weight gradient / dense TT GEMM.
//   columns of the B operand are (tap, c) pairs: col = t*C + c  ->  dw[(m*WT + wslot[t])*C + c]
// =====================================================================================
// v2: both operand tiles are DMA'd pixel-major ([64 pixels][channels], global_load_lds_dwordx4, no VGPR staging)
// and the k-contiguous MFMA fragments are produced by the gfx950 LDS transpose read ds_read_b64_tr_b16
// (measured semantics, tools/probes/tr_probe.hip: in a 16-lane group lane i loads 4 consecutive bf16 from ITS OWN
// address; lane l then receives element (l & 3) of the loads of lanes 4e + (l >> 2), e = 0..3).  Pointing lane i
// at row k0 + (i >> 2), columns c0 + 4 (i & 3) .. +3 of a [k][c] tile gives lane l the 4 k-values of column c0 + l.
// Bank conflicts between the 4 rows of such a read are removed by XOR-ing the 64-byte column group with the row
// (source-side swizzle, the DMA destination stays lane-linear).
template <int TC>
struct TtTile {                                  // geometry of a [64 pixels][TC columns] bf16 LDS tile
    static constexpr int LPR = TC / 8;           // 16-byte slots (= DMA lanes) per pixel row
    static constexpr int RPI = 64 / LPR;         // pixel rows per DMA wave-instruction (1 KiB)
    static constexpr int NI = 64 / RPI;          // DMA instructions per 64-pixel stage
    static constexpr int IT4 = (NI + 3) / 4;     // ... per wave of a 4-wave workgroup
    static constexpr int IT8 = (NI + 7) / 8;     // ... of an 8-wave workgroup
    static constexpr int G = (TC * 2) / 64 > 0 ? (TC * 2) / 64 : 1;   // 64-byte column groups per row
    static constexpr int RPL = 256 / (TC * 2) > 0 ? 256 / (TC * 2) : 1;   // rows per 256-byte bank line
    __device__ static __forceinline__ int swz(int row) { return (row / RPL) % G; }
};

// up to TT_MAX_BATCH problems of identical shape (the S calls of one layer in a window: same descriptors, different
// dy / input / dw buffers) share one launch: 3x the workgroups, so the pixel reduction is split 3x less and the
// atomic epilogue shrinks accordingly
#define TT_MAX_BATCH 8
struct TtBatch {
    const h16raw* dy[TT_MAX_BATCH];
    const h16raw* in[TT_MAX_BATCH];
    float* dw[TT_MAX_BATCH];
    int n;
};

// the body of one (problem, phase, pixel chunk, column tile, row tile) workgroup; `d` lies in the kernel-argument segment (uniform
// launches) or in device memory (heterogeneous launches: a descriptor table)
template <int TM, int TN, int WM, int WN, int KS>
__device__ __forceinline__ void igemm_tt_body(const h16raw* __restrict__ dy, const h16raw* __restrict__ in, float* __restrict__ dw,
                                              const tcvom_conv_desc& d, const h16raw* __restrict__ zero_page, const int ldy,
                                              const int chunk, const int pchunk, const int ntile, const int mtile)
{
#ifdef NT_TRACE
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    if (chunk * pchunk >= d.N * d.PH * d.PW) return;
    constexpr int WAVES_M = TM / WM, WAVES_N = TN / WN;
    constexpr int NW = WAVES_M * WAVES_N * KS;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int A_IT = NW == 4 ? TtTile<TM>::IT4 : TtTile<TM>::IT8;
    constexpr int B_IT = NW == 4 ? TtTile<TN>::IT4 : TtTile<TN>::IT8;
    constexpr int MI = WM / 32, NI = WN / 32;
    typedef TtTile<TM> TA;
    typedef TtTile<TN> TB;
    constexpr int SLOT = 64 * (TM + TN);

    __shared__ __attribute__((aligned(16))) h16raw lds[2 * SLOT + 8 * TCVOM_MAX_TAPS];
    int* taps = reinterpret_cast<int*>(lds + 2 * SLOT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // tap = kk / C through a multiply-high (exact for kk < 2^32 / C; one tap: magic 0 -> tap 0): C is any multiple of 8
    const unsigned cmagic = (d.ntaps == 1 && (C & 63) == 0) ? 0u : (unsigned)((0x100000000ull + (unsigned)C - 1) / (unsigned)C);
    const int ncols = d.ntaps * C;
    const int Ptot = d.N * d.PH * d.PW;
    const int pbeg = chunk * pchunk;
    const int pend = min(Ptot, pbeg + pchunk);
    const int n0 = ntile * TN;
    const int m0 = mtile * TM;
    const int PW = d.PW, PH = d.PH;
    const bool small_p = Ptot + 64 * 4 < (1 << 24);     // rows past the chunk end are divided too
    const float rcp_pw = 1.0f / (float)PW, rcp_ph = 1.0f / (float)PH;

    // ---- per-lane DMA state: A = dy tile [64][TM], B = gathered input tile [64][TN]
    int a_col[A_IT], a_n[A_IT], a_i[A_IT], a_j[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int ii = it * NW + wave;
        const int row = ii * TA::RPI + lane / TA::LPR;
        const int cs = (lane % TA::LPR) ^ (TA::swz(row) << 2);
        a_col[it] = m0 + cs * 8;
        a_ok[it] = ii < TA::NI && (a_col[it] + 8) <= ldy;
        const int p = pbeg + row;
        int t;
        divmod_any(p, PW, rcp_pw, small_p, t, a_j[it]);
        divmod_any(t, PH, rcp_ph, small_p, a_n[it], a_i[it]);
    }
    int b_c0[B_IT], b_dh[B_IT], b_dw[B_IT], b_n[B_IT], b_i[B_IT], b_j[B_IT];
    bool b_ok[B_IT], b_live[B_IT];            // b_live: the lane's column exists (columns past ntaps * C of a wide tile are never loaded,
                                              // their accumulator columns never stored)
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int ii = it * NW + wave;
        const int row = ii * TB::RPI + lane / TB::LPR;
        const int cs = (lane % TB::LPR) ^ (TB::swz(row) << 2);
        const int col = n0 + cs * 8;
        b_ok[it] = false;
        b_live[it] = ii < TB::NI && col < ncols;
        b_c0[it] = b_dh[it] = b_dw[it] = 0;
        if (ii < TB::NI && col < ncols) {
            const int tap = (int)__umulhi((unsigned)col, cmagic);
            b_c0[it] = col - tap * C;
            b_dh[it] = taps[tap * 3 + 0];
            b_dw[it] = taps[tap * 3 + 1];
            b_ok[it] = taps[tap * 3 + 2] >= 0;
        }
        const int p = pbeg + row;
        int t;
        divmod_any(p, PW, rcp_pw, small_p, t, b_j[it]);
        divmod_any(t, PH, rcp_ph, small_p, b_n[it], b_i[it]);
    }

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int nstage = (pend - pbeg + 63) >> 6;

    // issue the DMA of stage s (pixels pbeg + 64 s ...) into ring slot `slot`, then advance the pixel counters
    // Pixel counters advance by 64 per stage: one conditional carry when the row is at least 64 pixels wide,
    // otherwise (tiny images) they are recomputed from the pixel index.  All branch-free, 32-bit element offsets.
    const bool wide = PW >= 64;
    const int a_os = d.out_step, a_oh = d.out_off_h, a_ow = d.out_off_w, OHd = d.OH, OWd = d.OW, b_is = d.in_step;
#define TT_ADVANCE(n_, i_, j_, pnext)                                                                            \
    if (wide) {                                                                                                  \
        j_ += 64;                                                                                                \
        const bool c1 = j_ >= PW;                                                                                \
        j_ = c1 ? j_ - PW : j_;                                                                                  \
        i_ += c1 ? 1 : 0;                                                                                        \
        const bool c2 = i_ >= PH;                                                                                \
        i_ = c2 ? 0 : i_;                                                                                        \
        n_ += c2 ? 1 : 0;                                                                                        \
    } else {                                                                                                     \
        int t_;                                                                                                  \
        divmod_any((pnext), PW, rcp_pw, small_p, t_, j_);                                                        \
        divmod_any(t_, PH, rcp_ph, small_p, n_, i_);                                                             \
    }
#define TT_ISSUE_STAGE(s, slot)                                                                                  \
    {                                                                                                            \
        h16raw* abase = lds + (slot) * SLOT;                                                                    \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                                  \
            const int ii = it * NW + wave;                                                                        \
            if (ii < TA::NI) {                                                                                   \
                const int p = pbeg + (s) * 64 + ii * TA::RPI + lane / TA::LPR;                                   \
                const int off = ((a_n[it] * OHd + a_i[it] * a_os + a_oh) * OWd + a_j[it] * a_os + a_ow) * ldy + a_col[it]; \
                const h16raw* src = (a_ok[it] && p < pend) ? dy + off : zero_page;                              \
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abase + ii * 512), 16, 0, 0);             \
                TT_ADVANCE(a_n[it], a_i[it], a_j[it], p + 64)                                                    \
            }                                                                                                    \
        }                                                                                                        \
        h16raw* bbase = abase + 64 * TM;                                                                        \
        _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                                  \
            const int ii = it * NW + wave;                                                                        \
            if (ii < TB::NI && b_live[it]) {                                                                     \
                const int p = pbeg + (s) * 64 + ii * TB::RPI + lane / TB::LPR;                                   \
                const int ih = b_i[it] * b_is + b_dh[it], iw = b_j[it] * b_is + b_dw[it];                        \
                const bool okb = b_ok[it] && p < pend && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W; \
                const int off = ((b_n[it] * H + ih) * W + iw) * C + b_c0[it];                                    \
                const h16raw* src = okb ? in + off : zero_page;                                                 \
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bbase + ii * 512), 16, 0, 0);             \
                TT_ADVANCE(b_n[it], b_i[it], b_j[it], p + 64)                                                    \
            }                                                                                                    \
        }                                                                                                        \
    }

    // ---- transpose-read addressing (element offsets inside a tile), constant over the reduction loop
    const int tr_krow = (lane >> 5) * 8 + ((lane & 15) >> 2);          // + ks*16 + h*4
    const int tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;        // + wave/fragment column base

    if (nstage > 0) TT_ISSUE_STAGE(0, 0);
#ifdef NT_TRACE
    const bool trace_on = blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == NW - 1);
    const int trace_w = wave == 0 ? 0 : 1;
    if (trace_on) { tcvom_trace_buf[4096 + trace_w * 8 + 0] = t_entry; tcvom_trace_buf[4096 + trace_w * 8 + 1] = __builtin_readcyclecounter(); }
#endif
    for (int s = 0; s < nstage; ++s) {
        const int slot = s & 1;
        TRACE(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE(1);
        __builtin_amdgcn_s_barrier();
        TRACE(2);
        if (s + 1 < nstage) TT_ISSUE_STAGE(s + 1, slot ^ 1);
        TRACE(3);
        const h16raw* As = lds + slot * SLOT;
        const h16raw* Bs = As + 64 * TM;
        // fragments of k-step kq + 1 are requested before the MFMAs of k-step kq (two register sets)
        TrFrag fa[2][MI], fb[2][NI];
        constexpr int NKQ = 4 / KS;
#define TT_READ(set, kq)                                                                                         \
        {                                                                                                        \
            const int ks_ = (KS == 1) ? (kq) : wk;                                                               \
            const int r0 = ks_ * 16 + tr_krow, r1 = r0 + 4;                                                      \
            _Pragma("unroll") for (int a = 0; a < MI; ++a) {                                                     \
                const int col = wm * WM + a * 32 + tr_col;                                                       \
                const int sl = col >> 3, e = col & 7;                                                            \
                tr_issue(fa[set][a], As + r0 * TM + ((sl ^ (TA::swz(r0) << 2)) << 3) + e,                        \
                         As + r1 * TM + ((sl ^ (TA::swz(r1) << 2)) << 3) + e);                                   \
            }                                                                                                    \
            _Pragma("unroll") for (int b = 0; b < NI; ++b) {                                                     \
                const int col = wn * WN + b * 32 + tr_col;                                                       \
                const int sl = col >> 3, e = col & 7;                                                            \
                tr_issue(fb[set][b], Bs + r0 * TN + ((sl ^ (TB::swz(r0) << 2)) << 3) + e,                        \
                         Bs + r1 * TN + ((sl ^ (TB::swz(r1) << 2)) << 3) + e);                                   \
            }                                                                                                    \
        }
        TT_READ(0, 0)
#pragma unroll
        for (int kq = 0; kq < NKQ; ++kq) {
            const int set = kq & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int a = 0; a < MI; ++a) tr_fence(fa[set][a]);
#pragma unroll
            for (int b = 0; b < NI; ++b) tr_fence(fb[set][b]);
            if (kq + 1 < NKQ) TT_READ(set ^ 1, kq + 1)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = mfma16(tr_value(fa[set][a]), tr_value(fb[set][b]), acc[a][b], 0, 0, 0);
        }
#undef TT_READ
    }
#undef TT_ISSUE_STAGE
#undef TT_ADVANCE
#ifdef NT_TRACE
    if (trace_on) tcvom_trace_buf[4096 + trace_w * 8 + 2] = __builtin_readcyclecounter();
#endif

#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int col = n0 + wn * WN + b * 32 + (lane & 31);
        if (col >= ncols) continue;
        const int tap = (int)__umulhi((unsigned)col, cmagic), cc = col - tap * C;
        const int ws = taps[tap * 3 + 2];
        if (ws < 0) continue;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#ifdef TT_SKIP_ATOMICS
                if (m < K && acc[a][b][r] == 12345.678f) dw[((int64_t)m * d.wt + ws) * C + cc] = 1.f;
#else
                if (m < K) atomicAdd(dw + ((int64_t)m * d.wt + ws) * C + cc, acc[a][b][r]);
#endif
            }
        }
    }
#ifdef NT_TRACE
    if (trace_on) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tcvom_trace_buf[4096 + trace_w * 8 + 3] = __builtin_readcyclecounter();
    }
#endif
}

template <int TM, int TN, int WM, int WN, int KS>
__global__ __launch_bounds__((TM / WM) * (TN / WN) * KS * 64) void igemm_tt_kernel(
    const TtBatch bt, const h16raw* __restrict__ zero_page, const TcvomPhases ps, const int ldy, const int pchunk,
    const int chunks_per_phase)
{
    const int per_problem = chunks_per_phase * ps.n;
    const int prob = blockIdx.x / per_problem;
    const int rem = blockIdx.x - prob * per_problem;
    const int phase = rem / chunks_per_phase;
    igemm_tt_body<TM, TN, WM, WN, KS>(bt.dy[prob], bt.in[prob], bt.dw[prob], ps.d[phase], zero_page, ldy, rem - phase * chunks_per_phase,
                                      pchunk, blockIdx.y, blockIdx.z);
}

// Problems of DIFFERENT descriptors in one launch (tcvom_wgrad_igemm_hetero): the weight gradients that stay on this kernel -- the
// 1 x 1, stride-2, transposed and small-channel convs of the trunk -- are 20 - 100 us launches of a few dozen workgroups each, bound by
// the latency of ONE workgroup (prologue, 3.4 us per 64-pixel step, atomic epilogue) while most of the chip idles: 21 launches per
// 1080p step at 4 - 10 % of the MFMA peak.  Here a workgroup reads its (problem, descriptor, chunk, tile) from a work table in device
// memory and its descriptor from a descriptor table next to it (both built once per network geometry by the caller, see
// tcvom_wgrad_igemm_hetero_plan); the operand pointers of the problems travel in the kernel arguments.
#define TT_HET_MAX 64
struct TtHetBatch {
    const h16raw* dy[TT_HET_MAX];
    const h16raw* in[TT_HET_MAX];
    float* dw[TT_HET_MAX];
};
template <int TM, int TN, int WM, int WN, int KS>
__global__ __launch_bounds__((TM / WM) * (TN / WN) * KS * 64) void igemm_tt_hetero_kernel(
    const TtHetBatch bt, const h16raw* __restrict__ zero_page, const tcvom_conv_desc* __restrict__ dtab, const int4* __restrict__ work)
{
    // work item: {problem | descriptor << 8, chunk, column tile | row tile << 16, ldy}, then {pchunk, -, -, -}
    const int4 w0 = work[2 * blockIdx.x], w1 = work[2 * blockIdx.x + 1];
    const int prob = __builtin_amdgcn_readfirstlane(w0.x & 0xff), di = __builtin_amdgcn_readfirstlane(w0.x >> 8);
    igemm_tt_body<TM, TN, WM, WN, KS>(bt.dy[prob], bt.in[prob], bt.dw[prob], dtab[di], zero_page, __builtin_amdgcn_readfirstlane(w0.w),
                                      __builtin_amdgcn_readfirstlane(w0.y), __builtin_amdgcn_readfirstlane(w1.x),
                                      __builtin_amdgcn_readfirstlane(w0.z & 0xffff), __builtin_amdgcn_readfirstlane(w0.z >> 16));
}

// tile: wide-N tiles for the small-channel layers so that dy is re-read ncols/128 (not ncols/32) times
static void tt_tile(const tcvom_conv_desc* d, int* tm, int* tn) {
    const int ncols = d->ntaps * d->C;
    if (d->K >= 128 && ncols >= 128) { *tm = *tn = 128; }
    else if (d->K > 32) { *tm = 64; *tn = ncols >= 128 ? 128 : 64; }
    else {
        // K <= 32: one 128-column tile as soon as the columns (taps x C) do not fit a 32-column one -- the C = 8 layers (72 columns)
        // read dy once instead of three times (os1 8 -> 32: 130 -> 90 us, the two stride-2 layers on 8-channel inputs 81 -> 73 and
        // 73 -> 67 us; what is left is the 9-fold gather of x through the L2 -> LDS path, ~4.8 TB/s)
        *tm = 32;
        *tn = ncols > 32 ? 128 : 32;
    }
}
extern "C" const char* tcvom_wgrad_igemm_variant(const tcvom_conv_desc* d) {
    if (const char* v = halo_wgrad_variant(d, d->K)) return v;
    if (const char* v = wgradws_variant(d, d->K)) return v;
    if (gemm_tt256_takes(d)) return "gemm_tt256";
    int tm, tn;
    tt_tile(d, &tm, &tn);
    if (tm == 128) return "igemm_tt<128,128,64,32,1>";
    if (tm == 64 && tn == 128) return "igemm_tt<64,128,32,32,1>";
    if (tm == 64) return "igemm_tt<64,64,32,32,1>";
    if (tn == 128) return "igemm_tt<32,128,32,32,1>";
    return "igemm_tt<32,32,32,32,4>";
}

static int wgrad_igemm_launch(const void* const* dys, const void* const* ins, float* const* dws, int nbatch,
                              const tcvom_conv_desc* descs, int nphase, int32_t ldy, void* stream) {
    TCVOM_CHECK_ARG(dys && ins && dws && descs, "wgrad_igemm: null pointer");
    TCVOM_CHECK_ARG(nbatch >= 1 && nbatch <= TT_MAX_BATCH, "wgrad_igemm: batch of %d (1..%d)", nbatch, TT_MAX_BATCH);
    TtBatch bt;
    bt.n = nbatch;
    for (int i = 0; i < TT_MAX_BATCH; ++i) {
        const int j = i < nbatch ? i : 0;
        TCVOM_CHECK_ARG(dys[j] && ins[j] && dws[j], "wgrad_igemm: null pointer in batch entry %d", j);
        bt.dy[i] = (const h16raw*)dys[j];
        bt.in[i] = (const h16raw*)ins[j];
        bt.dw[i] = dws[j];
    }
    TCVOM_CHECK_ARG(nphase >= 1 && nphase <= 4, "wgrad_igemm: %d phases (1..4)", nphase);
    TcvomPhases ps;
    ps.n = nphase;
    long long P = 0;
    for (int i = 0; i < nphase; ++i) {
        const tcvom_conv_desc* d = descs + i;
        TCVOM_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= TCVOM_MAX_TAPS, "wgrad_igemm: ntaps=%d", d->ntaps);
        TCVOM_CHECK_ARG(d->ntaps == 1 || (d->C >= 8 && d->C % 8 == 0 && d->C <= 32768), "wgrad_igemm: C=%d must be a multiple of 8 (8..32768)", d->C);
        TCVOM_CHECK_ARG(d->C % 8 == 0 && ldy % 8 == 0, "wgrad_igemm: C=%d ldy=%d must be multiples of 8", d->C, ldy);
        TCVOM_CHECK_ARG(d->K == descs[0].K && d->C == descs[0].C && d->ntaps == descs[0].ntaps,
                        "wgrad_igemm: phases must share K, C and the tap count");
        const long long Pi = (long long)d->N * d->PH * d->PW;
        TCVOM_CHECK_ARG(Pi > 0 && Pi < (1ll << 31), "wgrad_igemm: bad pixel count %lld", Pi);
        TCVOM_CHECK_ARG((long long)d->N * d->OH * d->OW * ldy < (1ll << 31) && (long long)d->N * d->H * d->W * d->C < (1ll << 31),
                        "wgrad_igemm: operand too large for 32-bit element offsets");
        if (Pi > P) P = Pi;
        ps.d[i] = *d;
    }
    const tcvom_conv_desc* d = descs;
    hipStream_t st = (hipStream_t)stream;
    const int ncols = d->ntaps * d->C;
    const h16raw* zp = zero_page_for_current_device();
    TCVOM_CHECK_ARG(zp != nullptr, "wgrad_igemm: could not allocate the zero page");
    {
        const int r = halo_wgrad_try_launch(dys, ins, dws, nbatch, descs, nphase, ldy, zp, stream);
        if (r != 0) return r < 0 ? tcvom_fail(TCVOM_ERR_LAUNCH, "wgrad_igemm: halo launch failed") : TCVOM_OK;
    }
    {
        const int r = wgradws_try_launch(dys, ins, dws, nbatch, descs, nphase, ldy, stream);
        if (r != 0) return r < 0 ? r : TCVOM_OK;
    }
    bool tt256 = nphase == 1 && gemm_tt256_takes(d) && ldy % 8 == 0 && ldy >= d->K &&
                 (long long)d->N * d->PH * d->PW * (long long)(ldy > d->C ? ldy : d->C) < (1ll << 31);
    for (int i = 0; i < nbatch && tt256; ++i) tt256 = ((((uintptr_t)dys[i]) | ((uintptr_t)ins[i]) | ((uintptr_t)dws[i])) & 15) == 0;
    if (tt256) {
        // dense 1 x 1 convs with large K x C: split-K TT GEMMs on the k-major path of gemm_nt256 (gemm256.hip) -- one launch when the
        // problems lie at uniform strides (the frames of a frame-batched layer), else one per problem
        bool uniform = nbatch > 1;
        long long sdy = 0, sx = 0, sdw = 0;
        if (uniform) {
            sdy = (const char*)dys[1] - (const char*)dys[0]; sx = (const char*)ins[1] - (const char*)ins[0]; sdw = (const char*)dws[1] - (const char*)dws[0];
            for (int i = 2; i < nbatch; ++i)
                uniform = uniform && (const char*)dys[i] - (const char*)dys[i - 1] == sdy && (const char*)ins[i] - (const char*)ins[i - 1] == sx &&
                          (const char*)dws[i] - (const char*)dws[i - 1] == sdw;
            // (sdw == 0: the calls of a layer whose weight does not change between frames add into ONE gradient -- atomics)
            uniform = uniform && sdy > 0 && sx > 0 && sdw >= 0 && sdy % 16 == 0 && sx % 16 == 0 && sdw % 16 == 0;
        }
        int r = 1;
        if (uniform) r = gemm_tt256_try_launch(dys[0], ins[0], dws[0], d, ldy, stream, nbatch, sdy / 2, sx / 2, sdw / 4);
        else for (int i = 0; i < nbatch && r == 1; ++i) r = gemm_tt256_try_launch(dys[i], ins[i], dws[i], d, ldy, stream, 1, 0, 0, 0);
        if (r < 0) return r;
        TCVOM_CHECK_ARG(r == 1, "wgrad_igemm: the dense weight-gradient GEMM refused a shape it was asked about");
        return TCVOM_OK;
    }
    int tm, tn;
    tt_tile(d, &tm, &tn);
    // (measured, round 3: 64 x 128 tiles -- 49 KB, three workgroups per CU -- for the launches of 513 .. 768 workgroups of 128 x 128
    //  (the 4-phase transposed convs of os32 -> os16: a full round and a half): neutral, 23.72 vs 23.73 ms)
    const int mt = cdiv(d->K, tm), nt = cdiv(ncols, tn);
    // pixel chunks: every workgroup ends with tm*tn atomic adds, so chunks must be long enough to amortise them
    // (>= 512 pixels), and the whole grid should fit in ONE round of co-resident workgroups (LDS-limited occupancy
    // x 256 CUs): a partial second round costs as much as a full one
    const int lds_bytes = 2 * 64 * (tm + tn) * 2 + 256;
    int occ = (160 * 1024) / lds_bytes;
    // (the 32 x 32 tile would fit 8 workgroups per CU; planning for 8 measured 130 -> 175 us on the os1 8 -> 32 layer: half as long
    //  chunks, twice the atomic epilogues)
    if (occ > 4) occ = 4;
    long long want = (256ll * occ) / ((long long)mt * nt * nphase * nbatch);
    if (want < 1) want = 1;
    long long pchunk = ((P + want - 1) / want + 63) / 64 * 64;
    if (pchunk < 512) pchunk = 512;
    const int chunks = cdiv(P, pchunk);
    dim3 grid(chunks * nphase * nbatch, nt, mt);
    if (tm == 128)
        hipLaunchKernelGGL((igemm_tt_kernel<128, 128, 64, 32, 1>), grid, dim3(512), 0, st, bt, zp, ps, ldy, (int)pchunk, chunks);
    else if (tm == 64 && tn == 128)
        hipLaunchKernelGGL((igemm_tt_kernel<64, 128, 32, 32, 1>), grid, dim3(512), 0, st, bt, zp, ps, ldy, (int)pchunk, chunks);
    else if (tm == 64)
        hipLaunchKernelGGL((igemm_tt_kernel<64, 64, 32, 32, 1>), grid, dim3(256), 0, st, bt, zp, ps, ldy, (int)pchunk, chunks);
    else if (tn == 128)
        hipLaunchKernelGGL((igemm_tt_kernel<32, 128, 32, 32, 1>), grid, dim3(256), 0, st, bt, zp, ps, ldy, (int)pchunk, chunks);
    else
        hipLaunchKernelGGL((igemm_tt_kernel<32, 32, 32, 32, 4>), grid, dim3(256), 0, st, bt, zp, ps, ldy, (int)pchunk, chunks);
    TCVOM_LAUNCH_CHECK("wgrad_igemm");
    return TCVOM_OK;
}

// ---- heterogeneous launches (igemm_tt_hetero_kernel)
static int tt_het_check(const tcvom_conv_desc* d, int nphase, int ldy, int tm, int tn) {
    TCVOM_CHECK_ARG(nphase >= 1 && nphase <= 4, "wgrad_igemm_hetero: %d phases (1..4)", nphase);
    for (int i = 0; i < nphase; ++i) {
        const tcvom_conv_desc* e = d + i;
        TCVOM_CHECK_ARG(e->ntaps >= 1 && e->ntaps <= TCVOM_MAX_TAPS, "wgrad_igemm_hetero: ntaps=%d", e->ntaps);
        TCVOM_CHECK_ARG(e->C >= 8 && e->C % 8 == 0 && e->C <= 32768 && ldy % 8 == 0, "wgrad_igemm_hetero: C=%d ldy=%d must be multiples of 8", e->C, ldy);
        TCVOM_CHECK_ARG(e->K == d->K && e->C == d->C && e->ntaps == d->ntaps, "wgrad_igemm_hetero: phases must share K, C and the tap count");
        const long long Pi = (long long)e->N * e->PH * e->PW;
        TCVOM_CHECK_ARG(Pi > 0 && Pi < (1ll << 31), "wgrad_igemm_hetero: bad pixel count %lld", Pi);
        TCVOM_CHECK_ARG((long long)e->N * e->OH * e->OW * ldy < (1ll << 31) && (long long)e->N * e->H * e->W * e->C < (1ll << 31),
                        "wgrad_igemm_hetero: operand too large for 32-bit element offsets");
    }
    int a, b;
    tt_tile(d, &a, &b);
    TCVOM_CHECK_ARG(a == tm && b == tn, "wgrad_igemm_hetero: a problem of the %d x %d tile in a %d x %d launch", a, b, tm, tn);
    return TCVOM_OK;
}

extern "C" int tcvom_wgrad_igemm_hetero_plan(const tcvom_conv_desc* descs, const int32_t* nphase, const int32_t* ldy, int32_t nprob,
                                             int32_t tm, int32_t tn, int32_t* work, int32_t max_items) {
    TCVOM_CHECK_ARG(descs && nphase && ldy && nprob >= 1 && nprob <= TT_HET_MAX, "wgrad_igemm_hetero_plan: %d problems (1..%d)", nprob, TT_HET_MAX);
    TCVOM_CHECK_ARG((tm == 128 && tn == 128) || (tm == 64 && (tn == 128 || tn == 64)) || (tm == 32 && (tn == 128 || tn == 32)),
                    "wgrad_igemm_hetero_plan: no %d x %d tile", tm, tn);
    // pixel chunks: the whole launch should fit in ONE round of co-resident workgroups (LDS-limited occupancy x 256 CUs), and every
    // workgroup ends with tm * tn atomic adds: chunks of at least 512 pixels (the rule of the uniform launches, over all problems)
    const int lds_bytes = 2 * 64 * (tm + tn) * 2 + 256;
    int occ = (160 * 1024) / lds_bytes;
    if (occ > 4) occ = 4;
    double units = 0.0;                                  // sum over (problem, phase, tile) of its pixels
    int di = 0;
    for (int i = 0; i < nprob; ++i) {
        const tcvom_conv_desc* d = descs + di;
        if (int e = tt_het_check(d, nphase[i], ldy[i], tm, tn)) return e;
        const int tiles = cdiv(d->K, tm) * cdiv(d->ntaps * d->C, tn);
        for (int ph = 0; ph < nphase[i]; ++ph) units += (double)d[ph].N * d[ph].PH * d[ph].PW * tiles;
        di += nphase[i];
    }
    TCVOM_CHECK_ARG(di < (1 << 20), "wgrad_igemm_hetero_plan: too many descriptors");
    long long chunk_px = ((long long)(units / (256.0 * occ)) + 63) / 64 * 64;
    if (chunk_px < 512) chunk_px = 512;
    int n = 0;
    di = 0;
    for (int i = 0; i < nprob; ++i) {
        const tcvom_conv_desc* d = descs + di;
        const int mt = cdiv(d->K, tm), nt = cdiv(d->ntaps * d->C, tn);
        TCVOM_CHECK_ARG(mt < (1 << 15) && nt < (1 << 16), "wgrad_igemm_hetero_plan: too many tiles");
        for (int ph = 0; ph < nphase[i]; ++ph) {
            const long long P = (long long)d[ph].N * d[ph].PH * d[ph].PW;
            const int chunks = cdiv(P, chunk_px);
            for (int c = 0; c < chunks; ++c)
                for (int y = 0; y < nt; ++y)
                    for (int z = 0; z < mt; ++z) {
                        if (work) {
                            TCVOM_CHECK_ARG(n < max_items, "wgrad_igemm_hetero_plan: more than %d work items", max_items);
                            int32_t* w = work + (size_t)n * 8;
                            w[0] = i | ((di + ph) << 8); w[1] = c; w[2] = y | (z << 16); w[3] = ldy[i];
                            w[4] = (int32_t)chunk_px; w[5] = w[6] = w[7] = 0;
                        }
                        ++n;
                    }
        }
        di += nphase[i];
    }
    return n;
}

extern "C" int tcvom_wgrad_igemm_hetero(const void* const* dys, const void* const* ins, float* const* dws, int32_t nprob,
                                        const tcvom_conv_desc* desc_table, const int32_t* work, int32_t nwork, int32_t tm, int32_t tn,
                                        void* stream) {
    TCVOM_CHECK_ARG(dys && ins && dws && desc_table && work && nwork >= 1, "wgrad_igemm_hetero: null pointer / empty work list");
    TCVOM_CHECK_ARG(nprob >= 1 && nprob <= TT_HET_MAX, "wgrad_igemm_hetero: %d problems (1..%d)", nprob, TT_HET_MAX);
    TCVOM_CHECK_ARG((((uintptr_t)work) & 15) == 0, "wgrad_igemm_hetero: the work table must be 16-byte aligned");
    TtHetBatch bt;
    for (int i = 0; i < TT_HET_MAX; ++i) {
        const int j = i < nprob ? i : 0;
        TCVOM_CHECK_ARG(dys[j] && ins[j] && dws[j], "wgrad_igemm_hetero: null pointer in problem %d", j);
        bt.dy[i] = (const h16raw*)dys[j];
        bt.in[i] = (const h16raw*)ins[j];
        bt.dw[i] = dws[j];
    }
    const h16raw* zp = zero_page_for_current_device();
    TCVOM_CHECK_ARG(zp != nullptr, "wgrad_igemm_hetero: could not allocate the zero page");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(nwork);
    const int4* wk = reinterpret_cast<const int4*>(work);
    if (tm == 128 && tn == 128) hipLaunchKernelGGL((igemm_tt_hetero_kernel<128, 128, 64, 32, 1>), grid, dim3(512), 0, st, bt, zp, desc_table, wk);
    else if (tm == 64 && tn == 128) hipLaunchKernelGGL((igemm_tt_hetero_kernel<64, 128, 32, 32, 1>), grid, dim3(512), 0, st, bt, zp, desc_table, wk);
    else if (tm == 64 && tn == 64) hipLaunchKernelGGL((igemm_tt_hetero_kernel<64, 64, 32, 32, 1>), grid, dim3(256), 0, st, bt, zp, desc_table, wk);
    else if (tm == 32 && tn == 128) hipLaunchKernelGGL((igemm_tt_hetero_kernel<32, 128, 32, 32, 1>), grid, dim3(256), 0, st, bt, zp, desc_table, wk);
    else if (tm == 32 && tn == 32) hipLaunchKernelGGL((igemm_tt_hetero_kernel<32, 32, 32, 32, 4>), grid, dim3(256), 0, st, bt, zp, desc_table, wk);
    else return tcvom_fail(TCVOM_ERR_ARG, "wgrad_igemm_hetero: no %d x %d tile", tm, tn);
    TCVOM_LAUNCH_CHECK("wgrad_igemm_hetero");
    return TCVOM_OK;
}
extern "C" int32_t tcvom_wgrad_igemm_hetero_max_problems(void) { return TT_HET_MAX; }

extern "C" int tcvom_wgrad_igemm(const void* dy, const void* in, float* dw, const tcvom_conv_desc* d,
                                 int32_t ldy, void* stream) {
    return wgrad_igemm_launch(&dy, &in, &dw, 1, d, 1, ldy, stream);
}

extern "C" int tcvom_wgrad_igemm_phases(const void* dy, const void* in, float* dw, const tcvom_conv_desc* descs,
                                        int32_t nphase, int32_t ldy, void* stream) {
    return wgrad_igemm_launch(&dy, &in, &dw, 1, descs, nphase, ldy, stream);
}

extern "C" int tcvom_wgrad_igemm_batched(const void* const* dy, const void* const* in, float* const* dw, int32_t nbatch,
                                         const tcvom_conv_desc* descs, int32_t nphase, int32_t ldy, void* stream) {
    return wgrad_igemm_launch(dy, in, dw, nbatch, descs, nphase, ldy, stream);
}
