// STUDY KERNEL, not part of the library (round 3; driven by hb_probe.hip).  Result: the same 47 - 48 us per 256-channel os16 layer as
// igemm_nt in every form tried -- (a) 8 waves of 64 x 32 on a 128 x (8 x 16) tile, 2-slot weight ring, double-buffered halo: 47.4 us
// (40 % fewer DMA bytes than igemm_nt); (b) the same with a 3-slot weight ring and a single halo buffer: 50.4; (c) 4 waves of
// 64 x 64 (one fragment read per MFMA): 45.9; (d) (c) with a 3-slot ring: 49.8; (e) the DMA instructions behind the MFMA groups
// instead of in front: the 450-cycle issue block moves into the MFMA phase, the step stays 2440 cycles; (f) this file: 8 waves of
// 64 x 64 on a 128 x (8 x 32) tile, one workgroup per CU: 47.8.  Inside a k-step (cycle stamps): ~300 wait + 90 barrier + 110 - 500
// issue + 1200 - 1700 for 16 MFMAs per wave with two waves per SIMD (1024 cycles of matrix-pipe time): the matrix pipe is busy for
// about half of a step whatever is done about DMA volume, DMA depth or LDS reads.
//
// Halo-B implicit GEMM for the many-channel 3x3 stride-1 layers (the 256-channel os16 layers of vmn_gca: BasicBlock convs
// resnet_enc.py:33-49 / resnet_dec.py:43-59 and their data gradients): taps within +-1, same-size output, C % 64 == 0, K % 128 == 0.
//
// Why: a k-step of igemm_nt on these layers takes ~1.3 us whatever its tile, ring depth or co-residency (47 us for the 3 frames of an
// os16 layer).  tools/probes/hb_probe.hip (cycle stamps inside the k-loop of this kernel's first forms) shows what a step is made
// of: every wave runs fragment reads -> MFMAs -> DMA instructions IN ORDER, a global_load_lds instruction holds the wave's issue
// for ~100 cycles, and with 4 - 7 of them per wave and step plus the LDS latency a step is ~1400 cycles even for a workgroup that
// has its CU to itself, against 512 cycles of MFMA issue.  Fewer DMA bytes alone (a halo form with the 128 x 128 tile: -40 %), a
// 3-slot weight ring, or issuing the DMA behind the MFMAs each changed nothing.  What helps is less DMA and LDS work PER MFMA and
// per wave:
//   * a workgroup owns 128 output channels x an 8 x 32 pixel tile of one sample: 8 waves of 64 x 64 (2 x 2 MFMA tiles per k16
//     step: one 1 KB fragment read per MFMA; igemm_nt's 64 x 32 / 32 x 96 wave tiles read 1.5 / 1.33), two waves per SIMD;
//   * the reduction runs chunk-major: for every 64-channel chunk of the input, the (8 + 2) x (32 + 2) pixel halo of the tile is
//     DMA'd ONCE (double buffered: the next chunk's halo rides behind the first taps of the current one) and all taps read their
//     B fragments from it with shifted addresses;
//   * the weights [128 k][64 c] of one (tap, chunk) are DMA'd per k-step into a 2-slot ring: 2 instructions per wave.
// L2 -> LDS bytes per k-step: 16 KB + 43 KB / taps = 20.8 KB for 128 x 256 x 64 MACs (igemm_nt 128 x 96: 28 KB for 128 x 96 x 64).
//
// Waves (8): wm = wave >> 2 = 64-row block of output channels, wn = wave & 3 = tile rows 2 wn, 2 wn + 1; pixel fragment j = columns
// 16 j .. 16 j + 15 of those rows (lane & 15 = column, (lane >> 4) & 1 = row).
// LDS: A ring 2 x 16 KB, rows 128 B, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) (DMA source side, as igemm_nt);
//      halo 2 x 43 KB, pixel-major 128 B per pixel, chunk c of halo pixel p at position c ^ ((p >> 1) & 7): the 16 lanes of a
//      ds_read_b128 group read 16 consecutive pixels of one halo row -- 8 even, 8 odd, i.e. both 128-byte halves of the bank line,
//      8 distinct positions each; + 1 KB where the padding DMA instructions of a halo land.  One workgroup per CU.
#include <cstdlib>
#include "common.h"

#define HB_TH 8
#define HB_TW 32
#define HB_HW (HB_TW + 2)
#define HB_HH (HB_TH + 2)
#define HB_HPIX (HB_HW * HB_HH)                 // 340
#define HB_HINST 43                             // 340 pixels x 128 B = 43520 B -> 43 DMA instructions of 1 KB
#define HB_HALO_B (HB_HINST * 1024)
#define HB_A_B (16 * 1024)
#define HB_LDS (2 * HB_A_B + 2 * HB_HALO_B + 1024)

struct HbArgs {
    const h16raw* in;
    const h16raw* wgt;
    h16raw* out;
    const float* bias;
    float* stats;
    const h16raw* zero_page;
    int N, H, W, C, K, ldo, wt, act;
    int tiles_x, tiles_y, tiles_per_frame;
    int ntaps;                                  // compacted: real taps only
    int tap_dh[9], tap_dw[9], tap_w[9];
    long long in_bstride, w_bstride, out_bstride, stats_bstride;
    int stats_group_offset;
};

#ifdef HB_TRACE                         // tools/probes/hb_probe.hip: cycle stamps of workgroup 8 / wave 0, 5 per k-step
__device__ unsigned long long hb_trace_buf[4096];
#define HB_STAMP(i) if (tracing && s < 800) hb_trace_buf[s * 5 + (i)] = __builtin_readcyclecounter()
#else
#define HB_STAMP(i)
#endif
#define HB_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xF, true))
__device__ __forceinline__ void hb_reduce8(float (&t)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HB_DPP(t[r], 0xB1, 0xF);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HB_DPP(t[r], 0x4E, 0xF);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HB_DPP(t[r], 0x141, 0xF);     // row_half_mirror
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HB_DPP(t[r], 0x140, 0xF);     // row_mirror
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HB_DPP(t[r], 0x142, 0xA);     // row_bcast15 into rows 1 and 3
}

__global__ __launch_bounds__(512) void hbconv_kernel(const HbArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const Abuf = lds;                                            // [2][16 KB]
    char* const Hbuf = lds + 2 * HB_A_B;                               // [2][43 KB]
    char* const dump = Hbuf + 2 * HB_HALO_B;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-contiguous tile order (neighbouring tiles share halo columns / rows: one L2)
    int tile;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int frame = blockIdx.z, m0 = blockIdx.y * 128;
    const int n = tile / (a.tiles_x * a.tiles_y), trem = tile - n * (a.tiles_x * a.tiles_y);
    const int ty0 = (trem / a.tiles_x) * HB_TH, tx0 = (trem % a.tiles_x) * HB_TW;
    const h16raw* in = a.in + frame * a.in_bstride + (long long)n * a.H * a.W * a.C;
    const h16raw* wgt = a.wgt + frame * a.w_bstride;
    const int C = a.C, H = a.H, W = a.W;

    // ---- A DMA: instruction it of this wave = rows (it * 8 + wave) * 8 + lane / 8 of the 128-row slice
    int a_off[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (it * 8 + wave) * 8 + (lane >> 3);
        const int m = m0 + row;
        a_off[it] = m < a.K ? m * a.wt * C + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : -1;
    }
    // ---- halo DMA: instruction j = it * 8 + wave (it = 0..5) covers units j * 64 .. j * 64 + 63; unit u = pixel u >> 3, position u & 7
    int h_off[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int u = (it * 8 + wave) * 64 + lane, hp = u >> 3;
        const int hy = hp / HB_HW, hx = hp - hy * HB_HW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = hp < HB_HPIX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        h_off[it] = ok ? (y * W + x) * C + ((((u & 7) ^ ((hp >> 1) & 7))) << 3) : -1;
    }
#define HB_ISSUE_A1(wo, slot, it)                                                                           \
    {                                                                                                        \
        const h16raw* src_ = a_off[it] >= 0 ? wgt + a_off[it] + (wo) : a.zero_page;                          \
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(Abuf + (slot) * HB_A_B + ((it) * 8 + wave) * 1024), 16, 0, 0); \
    }
#define HB_ISSUE_H(ck, it)                                                                                   \
    {                                                                                                        \
        const h16raw* src_ = h_off[it] >= 0 ? in + h_off[it] + (ck) * 64 : a.zero_page;                      \
        const int j_ = (it) * 8 + wave;                                                                      \
        char* dst_ = j_ < HB_HINST ? Hbuf + ((ck) & 1) * HB_HALO_B + j_ * 1024 : dump;                       \
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)dst_, 16, 0, 0);                              \
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing
    const int khalf = lane >> 5;
    const int a_row = wm * 64 + (lane & 31), a_swz = (a_row >> 1) & 7;          // rows +32: the same swizzle
    const int prow = 2 * wn + ((lane >> 4) & 1), pcol = lane & 15;              // this lane's pixel of fragment 0 (fragment 1: 16 columns right)
    const int hp0 = (prow + 1) * HB_HW + pcol + 1;                              // its halo pixel for the centre tap

    const int nchunk = C >> 6, nstep = nchunk * a.ntaps;
#pragma unroll
    for (int it = 0; it < 6; ++it) HB_ISSUE_H(0, it)
    {
        const int wo0 = a.tap_w[0] * C;
        HB_ISSUE_A1(wo0, 0, 0) HB_ISSUE_A1(wo0, 0, 1)
    }
    int ck = 0, tp = 0;
#ifdef HB_TRACE
    const bool tracing = blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
#endif
    for (int s = 0; s < nstep; ++s) {
        HB_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HB_STAMP(1);
        __builtin_amdgcn_s_barrier();
        HB_STAMP(2);
        // the DMA instructions of the next k-step -- 2 of the weights, and one of the next chunk's halo behind each of the first 6
        // taps -- go out between the MFMA groups of this one
        const bool more_a = s + 1 < nstep;
        int ntp = tp + 1, nck = ck;
        if (ntp == a.ntaps) { ntp = 0; ++nck; }
        const int wo = a.tap_w[ntp] * C + nck * 64, islot = (s + 1) & 1;
        const bool more_h = tp < 6 && ck + 1 < nchunk;
        HB_STAMP(3);
        const char* As = Abuf + (s & 1) * HB_A_B;
        const int hp = hp0 + a.tap_dh[tp] * HB_HW + a.tap_dw[tp], hq = hp + 16;
        const char* Bp0 = Hbuf + (ck & 1) * HB_HALO_B + hp * 128;
        const char* Bp1 = Hbuf + (ck & 1) * HB_HALO_B + hq * 128;
        const int b_swz0 = (hp >> 1) & 7, b_swz1 = (hq >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kch = kk * 2 + khalf;
            const h16x8_t fa0 = *reinterpret_cast<const h16x8_t*>(As + a_row * 128 + ((kch ^ a_swz) << 4));
            const h16x8_t fa1 = *reinterpret_cast<const h16x8_t*>(As + (a_row + 32) * 128 + ((kch ^ a_swz) << 4));
            const h16x8_t fb0 = *reinterpret_cast<const h16x8_t*>(Bp0 + ((kch ^ b_swz0) << 4));
            const h16x8_t fb1 = *reinterpret_cast<const h16x8_t*>(Bp1 + ((kch ^ b_swz1) << 4));
            acc[0][0] = mfma16(fa0, fb0, acc[0][0], 0, 0, 0);
            acc[1][0] = mfma16(fa1, fb0, acc[1][0], 0, 0, 0);
            acc[0][1] = mfma16(fa0, fb1, acc[0][1], 0, 0, 0);
            acc[1][1] = mfma16(fa1, fb1, acc[1][1], 0, 0, 0);
            if (kk < 2) { if (more_a) HB_ISSUE_A1(wo, islot, kk) }
            else if (kk == 2 && more_h) {
                if (tp == 0) HB_ISSUE_H(ck + 1, 0) else if (tp == 1) HB_ISSUE_H(ck + 1, 1) else if (tp == 2) HB_ISSUE_H(ck + 1, 2)
                else if (tp == 3) HB_ISSUE_H(ck + 1, 3) else if (tp == 4) HB_ISSUE_H(ck + 1, 4) else HB_ISSUE_H(ck + 1, 5)
            }
        }
#ifdef HB_TRACE
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][1][15]));      // the stamp below waits for the last MFMA
#endif
        HB_STAMP(4);
        if (++tp == a.ntaps) { tp = 0; ++ck; }
    }
#undef HB_ISSUE_A1
#undef HB_ISSUE_H

    // ---- epilogue: a lane holds 4 consecutive channels (x 4 groups x 2 channel fragments) of ONE pixel per pixel fragment
    const float slope = a.act == 1 ? 0.f : a.act == 3 ? 0.01f : 1.f;
    const long long sgrp = a.stats ? a.stats_group_offset + frame * a.stats_bstride + (long long)tile * 4 + wn : 0;
    const int y = ty0 + prow;
    bool pvalid[2];
    h16raw* op[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = tx0 + pcol + 16 * j;
        pvalid[j] = y < H && x < W;
        op[j] = a.out + frame * a.out_bstride + ((long long)(n * H + (pvalid[j] ? y : 0)) * W + (pvalid[j] ? x : 0)) * a.ldo;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int mrow = m0 + wm * 64 + i * 32 + 8 * g + 4 * khalf;
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && mrow < a.K) bs = *reinterpret_cast<const float4*>(a.bias + mrow);
            const float bv[4] = {bs.x, bs.y, bs.z, bs.w};
            float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xv = acc[i][j][g * 4 + r] + bv[r];
                    xv = fmaxf(xv, xv * slope);
                    v[r] = xv;
                    const float xs = pvalid[j] ? xv : 0.f;
                    t8[r] += xs;
                    t8[4 + r] = fmaf(xs, xs, t8[4 + r]);
                }
                if (pvalid[j] && mrow < a.K) *reinterpret_cast<uint2*>(op[j] + mrow) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
            }
            if (a.stats) {
                hb_reduce8(t8);                          // lanes 16..31 / 48..63 hold the totals of the two lane halves
                if ((lane & 31) == 16 && mrow < a.K) {
                    float* sp = a.stats + sgrp * 2 * a.K + mrow;
                    *reinterpret_cast<float4*>(sp) = make_float4(t8[0], t8[1], t8[2], t8[3]);
                    *reinterpret_cast<float4*>(sp + a.K) = make_float4(t8[4], t8[5], t8[6], t8[7]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
struct HbPlan { bool ok; int ntaps; int dh[9], dw[9], tw[9]; int tiles_x, tiles_y; };

static HbPlan hb_plan(const tcvom_conv_desc* d, int nphase) {
    HbPlan p;
    p.ok = false;
    static const int mode = getenv("TCVOM_HB") ? atoi(getenv("TCVOM_HB")) : 1;      // 0: off (A/B switch)
    if (!mode || nphase != 1) return p;
    if (d->in_step != 1 || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return p;
    if (d->H != d->OH || d->W != d->OW || d->PH != d->H || d->PW != d->W) return p;
    if (d->C % 64 != 0 || d->K % 128 != 0 || d->C < 256 || d->w_layout != 0 || d->out_fp32 || d->ldo % 4 != 0) return p;
    if (d->ntaps < 2 || d->ntaps > 9) return p;
    p.ntaps = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        if (d->tap_w[t] < 0) continue;
        if (d->tap_dh[t] < -1 || d->tap_dh[t] > 1 || d->tap_dw[t] < -1 || d->tap_dw[t] > 1) return p;
        p.dh[p.ntaps] = d->tap_dh[t]; p.dw[p.ntaps] = d->tap_dw[t]; p.tw[p.ntaps] = d->tap_w[t];
        ++p.ntaps;
    }
    if (p.ntaps < 3) return p;
    p.tiles_x = (d->W + HB_TW - 1) / HB_TW;
    p.tiles_y = (d->H + HB_TH - 1) / HB_TH;
    // enough workgroups to fill the chip, and not so much tile overhang that the padding eats the gain
    const int nb = d->batch > 1 ? d->batch : 1;
    const long long wgs = (long long)d->N * p.tiles_x * p.tiles_y * (d->K / 128) * nb;
    const double cover = (double)d->H * d->W / ((double)p.tiles_x * HB_TW * p.tiles_y * HB_TH);
    if (wgs < 160 || cover < 0.8) return p;
    if ((long long)d->N * d->H * d->W * d->C >= (1ll << 31) || (long long)d->K * d->wt * d->C >= (1ll << 31)) return p;
    p.ok = true;
    return p;
}

int hbconv_stats_groups(const tcvom_conv_desc* d, int nphase) {
    const HbPlan p = hb_plan(d, nphase);
    return p.ok ? d->N * p.tiles_x * p.tiles_y * 4 : 0;
}

// returns 1 when the conv was launched here, 0 when the caller should use another kernel, < 0 on error
int hbconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                      float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream) {
    if (mscale || mdiag) return 0;
    const HbPlan p = hb_plan(d, nphase);
    if (!p.ok) return 0;
    if (bias && ((uintptr_t)bias & 15)) return 0;
    HbArgs a;
    a.in = (const h16raw*)in; a.wgt = (const h16raw*)w; a.out = (h16raw*)out; a.bias = bias; a.stats = stats; a.zero_page = zero_page;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.ldo = d->ldo; a.wt = d->wt; a.act = d->act;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.tiles_per_frame = d->N * p.tiles_x * p.tiles_y;
    a.ntaps = p.ntaps;
    for (int t = 0; t < 9; ++t) { a.tap_dh[t] = t < p.ntaps ? p.dh[t] : 0; a.tap_dw[t] = t < p.ntaps ? p.dw[t] : 0; a.tap_w[t] = t < p.ntaps ? p.tw[t] : 0; }
    const int nb = d->batch > 1 ? d->batch : 1;
    a.in_bstride = nb > 1 ? d->in_bstride : 0;
    a.w_bstride = nb > 1 ? d->w_bstride : 0;
    a.out_bstride = nb > 1 ? d->out_bstride : 0;
    a.stats_bstride = nb > 1 ? d->stats_bstride : 0;
    a.stats_group_offset = d->stats_group_offset;
    if (stats && nb > 1 && d->stats_bstride < (long long)a.tiles_per_frame * 4)
        return tcvom_fail(TCVOM_ERR_ARG, "hbconv: stats_bstride %lld < groups per frame", (long long)d->stats_bstride);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)hbconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS) != hipSuccess)
            return tcvom_fail(TCVOM_ERR_LAUNCH, "hbconv: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const dim3 grid((unsigned)a.tiles_per_frame, (unsigned)(d->K / 128), (unsigned)nb);
    hipLaunchKernelGGL(hbconv_kernel, grid, dim3(512), HB_LDS, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "hbconv: %s", hipGetErrorString(e));
    return 1;
}
