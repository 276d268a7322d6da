// Halo-tile direct convolution for the K <= 64, C in {32, 64} layers that the 32 -> 32 kernel of halo.hip does not take and the
// implicit GEMM serves badly: the transposed 4x4 stride-2 convs of the decoder (resnet_dec.py:23-41: conv1 32 -> 32 at os2 -> os1,
// layer4 64 -> 64 at os4 -> os2; four sub-pixel phases of 2 x 2 taps each), the 3x3 convs between 64 and 32 channels at os2
// (layer4 conv2 and its data gradient) and the data gradients of the stride-2 3x3 convs (encoder conv3, guidance head: four
// phases of 1 / 2 / 2 / 4 taps, padded to 4).
//
// Why: these layers are HBM-bound, but igemm_nt re-fetches every input pixel once PER TAP through the L2 -> LDS DMA path, and that
// path (~3.5 TB/s) bounds them: ConvTranspose 32 -> 32 at 1088 x 1920 moves 535 MB through it for 33 MB of input, 150 us against a
// 40 us HBM floor (profiles/r04_conv_launches_1080p.txt).  As in halo.hip a persistent workgroup DMAs the (8 + 2) x (32 + 2) pixel
// halo of a tile ONCE, all taps read it from LDS, the weights of the workgroup's phase stay in LDS, the next halo is in flight
// while the current one is used.  Differences from halo_conv_kernel: up to 64 output channels (KB blocks of 32 rows), 32 or 64 input
// channels, a PHASE per workgroup (its own tap list and output offset; the output of phase (a, b) is pixel (2 i + a, 2 j + b)).
//
//   MFMA 32x32x16:  A = weights [32 out-channels][16 k],  B = pixels [32 pixels of one tile row][16 k],  k = (tap, channel) in
//   chunks of 16;  wave w owns tile rows 2w, 2w+1.  LDS halo image: pixel-major, 16-byte channel chunk c of halo pixel p at slot
//   c ^ ((hx >> 2) & 3) (C = 32: 64-byte pixels) / c ^ ((hx >> 1) & 7) (C = 64: 128-byte pixels): the 16 consecutive pixels of a
//   ds_read_b128 group cover 16 distinct bank groups (hx = the pixel's column in the halo, so the second tile row of a wave and the
//   chunks of a tap are immediate offsets / XOR constants of ONE address register per tap).  Weight rows are padded by 16 bytes.
// (Measured and dropped: issuing a tile's 16-bit stores one tile late -- packed in registers, behind the next halo's DMA, so that the
//  loop-top vmcnt(0) does not wait for stores issued a moment earlier: 54 / 46 / 73 / 51 / 41 / 80 us -> 58 / 49 / 80 / 54 / 45 / 86 us
//  on the six launches of a 1080p step.  The store round trip is not what a tile waits for.)
#include <cstdlib>
#include <type_traits>
#include "common.h"

#define SC_TH 8
#define SC_TW 32
#define SC_HW (SC_TW + 2)
#define SC_HH (SC_TH + 2)
#define SC_PIX (SC_HW * SC_HH)          // 340
#define SC_MAXT 9
#ifndef SC_ABL
#define SC_ABL 0          // kernel ablations for timing (1: no output stores, 2: no halo DMA after the first tile, 3: no MFMAs / fragment reads;
#endif                    // study builds: tcvom_amd/lib/study via TCVOM_LIB); 0 in the product

struct SconvArgs {
    const h16raw* in;
    const h16raw* wgt;
    void* out;
    const float* bias;
    float* stats;
    const h16raw* zero_page;
    int H, W, K, ldo, wt, act, out_fp32;
    int PH, PW;                                                          // phase grid (tiles cover it; input pixels outside H x W read zero)
    int OH, OW, ostep, nphase;
    int off_h[4], off_w[4], stats_group_offset[4];
    int tap_dh[4][SC_MAXT], tap_dw[4][SC_MAXT], tap_w[4][SC_MAXT];      // tap_w < 0: zero tap (padding of a short phase)
    int tiles_x, tiles_y, tiles_per_wg, wgs_per_fp;                      // per (frame, phase)
    int spf;                                                             // samples per frame
    long long stats_bstride, w_bstride;
};

#define SC_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xF, true))
__device__ __forceinline__ void sc_reduce8(float (&t)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += SC_DPP(t[r], 0xB1, 0xF);
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += SC_DPP(t[r], 0x4E, 0xF);
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += SC_DPP(t[r], 0x141, 0xF);
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += SC_DPP(t[r], 0x140, 0xF);
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += SC_DPP(t[r], 0x142, 0xA);
}
typedef __attribute__((ext_vector_type(4))) unsigned int sc_u32x4_t;
struct ScFrag { sc_u32x4_t v; };
template <int OFF>
__device__ __forceinline__ void sc_read(ScFrag& f, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void sc_fence(ScFrag& f) { asm volatile("" : "+v"(f.v)); }
__device__ __forceinline__ h16x8_t sc_value(const ScFrag& f) { return __builtin_bit_cast(h16x8_t, f.v); }
template <int B, int E, class F>
__device__ __forceinline__ void sc_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sc_static_for<B + 1, E>(f);
    }
}

// C input channels (32 / 64), KB blocks of 32 output channels, NCH = k-chunks of 16 per phase (taps * C / 16)
template <int C, int KB, int NCH>
__global__ __launch_bounds__(256) void sconv_kernel(const SconvArgs a) {
    constexpr int CU = C / 8;
    constexpr int UNITS = SC_PIX * CU;
    constexpr int NDMA = (UNITS + 63) / 64;
    constexpr int DMA_IT = (NDMA + 3) / 4;
    constexpr int SLOT = NDMA * 512;                    // 16-bit elements per halo slot
    constexpr int WROW = NCH * 16 + 8;                  // padded weight row (elements)
    constexpr int NTAPS = NCH * 16 / C;
    constexpr int CPT = C / 16;                         // k-chunks per tap
    constexpr int ROWB = SC_HW * CU * 16;               // bytes between halo rows
    constexpr int PF = 2;                               // fragment reads issued PF chunks ahead of their MFMAs
    extern __shared__ __attribute__((aligned(1024))) h16raw sc_lds[];      // (chunk XORs act on absolute addresses)
    h16raw* halo = sc_lds;                              // [2][SLOT]
    h16raw* wl = sc_lds + 2 * SLOT;                     // [KB * 32][WROW]
    float* bl = reinterpret_cast<float*>(wl + KB * 32 * WROW);      // [KB * 32] bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int H = a.H, W = a.W;

    int v;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_per_frame = a.spf * a.tiles_x * a.tiles_y;
    // the phases of one run of tiles are neighbours in v, hence in one XCD at about the same time: they read the same input
    const int per_frame = a.wgs_per_fp * a.nphase;
    const int frame = v / per_frame, rem = v - frame * per_frame;
    const int wslot = rem / a.nphase, phase = rem - wslot * a.nphase;
    const int t_begin = wslot * a.tiles_per_wg;
    const int t_end = min(tiles_per_frame, t_begin + a.tiles_per_wg);
    const int n_base = frame * a.spf;                   // first sample of this frame

    // ---- per-lane constants
    int d_rel[DMA_IT], d_yx[DMA_IT];
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) {
        const int q = (it * 4 + wave) * 64 + lane;
        const int p = q / CU, sl = q % CU;
        const int hy = p / SC_HW, hx = p - hy * SC_HW;
        const int c16 = CU == 4 ? (sl ^ ((hx >> 2) & 3)) : (sl ^ ((hx >> 1) & 7));
        d_rel[it] = ((hy - 1) * W + (hx - 1)) * C + c16 * 8;
        d_yx[it] = (q < UNITS) ? ((hy << 16) | hx) : -1;
    }
    // B (pixels): one address per tap = halo pixel (2 wave + dh + 1, col + dw + 1), chunk `half`; chunk cc of the tap is ^ (cc << 5),
    // the wave's second tile row + ROWB.  A (weights): row col, immediate offsets per chunk and channel block.
    unsigned tapb[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int hx = col + a.tap_dw[phase][t] + 1, hy = 2 * wave + a.tap_dh[phase][t] + 1;
        const int sw = CU == 4 ? ((hx >> 2) & 3) : ((hx >> 1) & 7);
        tapb[t] = (unsigned)(((hy * SC_HW + hx) * CU + (half ^ sw)) * 16);
    }
    typedef __attribute__((address_space(3))) void* sc_lptr_t;
    typedef const __attribute__((address_space(1))) void* sc_gptr_t;
    const unsigned halo_lds = (unsigned)(uintptr_t)(sc_lptr_t)halo;
    const unsigned a_base = (unsigned)(uintptr_t)(sc_lptr_t)wl + (unsigned)((col * WROW + half * 8) * 2);
    const int64_t zoff = reinterpret_cast<const char*>(a.zero_page) - reinterpret_cast<const char*>(a.in);
    unsigned unit_ok = 0;
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) unit_ok |= (d_yx[it] >= 0 ? 1u : 0u) << it;
#define SC_ISSUE(tx_, ty_, n_, slot)                                                                        \
    {                                                                                                       \
        const int y0_ = (ty_) * SC_TH, x0_ = (tx_) * SC_TW;                                                 \
        const int64_t base_ = ((((int64_t)(n_)) * H + y0_) * W + x0_) * C;                                   \
        const bool inner_ = (ty_) > 0 && ((ty_) + 1) * SC_TH < H && (tx_) > 0 && ((tx_) + 1) * SC_TW < W;   \
        _Pragma("unroll") for (int it = 0; it < DMA_IT; ++it) {                                             \
            if ((it * 4 + wave) < NDMA) {                                                                   \
                bool ok_ = (unit_ok >> it) & 1u;                                                            \
                if (!inner_) {                                                                              \
                    const int hy_ = d_yx[it] >> 16, hx_ = d_yx[it] & 0xffff;                                \
                    ok_ = ok_ && (unsigned)(y0_ + hy_ - 1) < (unsigned)H && (unsigned)(x0_ + hx_ - 1) < (unsigned)W; \
                }                                                                                           \
                const int64_t off_ = ok_ ? (base_ + d_rel[it]) * 2 : zoff;                                  \
                __builtin_amdgcn_global_load_lds((sc_gptr_t)(reinterpret_cast<const char*>(a.in) + off_),   \
                                                 (sc_lptr_t)(halo + (slot) * SLOT + (it * 4 + wave) * 512), 16, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    }
    int ntx, nty, nn;
    {
        const int t0 = t_begin < t_end ? t_begin : 0;
        ntx = t0 % a.tiles_x; nty = (t0 / a.tiles_x) % a.tiles_y; nn = n_base + t0 / (a.tiles_x * a.tiles_y);
    }
#define SC_NEXT() { if (++ntx == a.tiles_x) { ntx = 0; if (++nty == a.tiles_y) { nty = 0; ++nn; } } }

    const float slope = a.act == 1 ? 0.f : a.act == 3 ? 0.01f : 1.f;
    const int ostep = a.ostep, ooh = a.off_h[phase], oow = a.off_w[phase];

    if (t_begin < t_end) SC_ISSUE(ntx, nty, nn, 0);
    // ---- weights of this (frame, phase) -> LDS: wl[k][t*C + c] = wgt[frame][(k*wt + slot(t))*C + c], behind the first halo's DMA.
    // All loads of a thread are issued before its first store (as a load -> wait -> store loop, with the lane-indexed tap table read
    // in front of every load, the 9-tap C = 64 variant spent 18 dependent round trips here before its first tile: a fixed ~10 us per
    // launch); padding taps and rows past K read the zero page.
    {
        constexpr int WUNITS = KB * 32 * NTAPS * CU, WIT = (WUNITS + 255) / 256;
        const h16raw* wsrc = a.wgt + (int64_t)frame * a.w_bstride;
        int wsl[WIT];
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int u = tid + i * 256, t = (u / CU) % NTAPS;
            wsl[i] = a.tap_w[phase][t];
        }
        uint4 wv[WIT];
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int u = tid + i * 256;
            const int cu = u % CU, t = (u / CU) % NTAPS, k = u / (CU * NTAPS);
            const bool ok = u < WUNITS && wsl[i] >= 0 && k < a.K;
            const h16raw* src = ok ? wsrc + ((int64_t)k * a.wt + wsl[i]) * C + cu * 8 : a.zero_page;
            wv[i] = *reinterpret_cast<const uint4*>(src);
        }
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int u = tid + i * 256;
            const int cu = u % CU, t = (u / CU) % NTAPS, k = u / (CU * NTAPS);
            if (u < WUNITS) *reinterpret_cast<uint4*>(wl + k * WROW + t * C + cu * 8) = wv[i];
        }
        if (tid < KB * 32) bl[tid] = (a.bias && tid < a.K) ? a.bias[tid] : 0.f;
    }
    int ctx = ntx, cty = nty, cn = nn;
    int slot = 0;
    float s1[KB][4][4], s2[KB][4][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[kb][g][r] = 0.f; s2[kb][g][r] = 0.f; }

    for (int tile = t_begin; tile < t_end; ++tile) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SC_NEXT();
#if SC_ABL != 2
        if (tile + 1 < t_end) SC_ISSUE(ntx, nty, nn, slot ^ 1);
#endif
        f32x16_t acc[KB][2];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kb][j][r] = 0.f;
        const unsigned hs = halo_lds + (unsigned)slot * (SLOT * 2);
        ScFrag fa[PF + 1][KB], fb[PF + 1][2];
        auto issue = [&](auto c_) {
            constexpr int c = decltype(c_)::value;
            constexpr int n = c % (PF + 1), tap = c / CPT, cc = c % CPT;
            sc_static_for<0, KB>([&](auto kb_) {
                constexpr int kb = decltype(kb_)::value;
                sc_read<c * 32 + kb * 32 * WROW * 2>(fa[n][kb], a_base);
            });
            const unsigned ba = (hs + tapb[tap]) ^ (unsigned)(cc << 5);
            sc_read<0>(fb[n][0], ba);
            sc_read<ROWB>(fb[n][1], ba);
        };
#if SC_ABL < 3
        sc_static_for<0, (PF < NCH ? PF : NCH)>(issue);
        sc_static_for<0, NCH>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            if constexpr (c + PF < NCH) issue(std::integral_constant<int, c + PF>{});
            constexpr int ahead = (NCH - 1 - c < PF ? NCH - 1 - c : PF) * (KB + 2);      // reads issued after chunk c's
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(ahead) : "memory");
            constexpr int k = c % (PF + 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) sc_fence(fa[k][kb]);
            sc_fence(fb[k][0]); sc_fence(fb[k][1]);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                acc[kb][0] = mfma16(sc_value(fa[k][kb]), sc_value(fb[k][0]), acc[kb][0], 0, 0, 0);
                acc[kb][1] = mfma16(sc_value(fa[k][kb]), sc_value(fb[k][1]), acc[kb][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#endif

        // ---- epilogue: bias, activation, store at the phase's output pixels, running channel sums
        const int tx = ctx, ty = cty, n = cn;
        ctx = ntx; cty = nty; cn = nn;
        const int y0 = ty * SC_TH + 2 * wave, x = tx * SC_TW + col;
        const bool xin = x < a.PW;
        const int64_t o0 = ((int64_t)(n * a.OH + y0 * ostep + ooh) * a.OW + x * ostep + oow) * a.ldo + 4 * half;
        const int64_t orow = (int64_t)a.OW * a.ldo * ostep;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool pin = xin && (y0 + j) < a.PH;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bl + kb * 32 + 8 * g + 4 * half);
                    const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float xv = acc[kb][j][g * 4 + r] + bv[r];
                        xv = fmaxf(xv, xv * slope);
                        acc[kb][j][g * 4 + r] = xv;
                        const float xs = pin ? xv : 0.f;
                        s1[kb][g][r] += xs;
                        s2[kb][g][r] += xs * xs;
                    }
                }
            }
#if SC_ABL == 1
        if (a.K == 12345) {
#else
        {
#endif
        if (a.out_fp32) {
            float* op = reinterpret_cast<float*>(a.out) + o0;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (xin && (y0 + j) < a.PH && kb * 32 + 8 * g + 4 * half < a.K)
                            *reinterpret_cast<float4*>(op + j * orow + kb * 32 + 8 * g) =
                                make_float4(acc[kb][j][g * 4], acc[kb][j][g * 4 + 1], acc[kb][j][g * 4 + 2], acc[kb][j][g * 4 + 3]);
        } else {
            h16raw* op = reinterpret_cast<h16raw*>(a.out) + o0;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (xin && (y0 + j) < a.PH && kb * 32 + 8 * g + 4 * half < a.K)
                            *reinterpret_cast<uint2*>(op + j * orow + kb * 32 + 8 * g) =
                                make_uint2(pack2h(acc[kb][j][g * 4], acc[kb][j][g * 4 + 1]), pack2h(acc[kb][j][g * 4 + 2], acc[kb][j][g * 4 + 3]));
        }
        }
        slot ^= 1;
    }
#undef SC_ISSUE
#undef SC_NEXT
    if (a.stats) {
        // BatchNorm partial statistics: one group per workgroup (per frame and phase), as in halo_conv_kernel
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = reinterpret_cast<float*>(sc_lds);      // [4 waves][KB][4 groups][2 halves][8]
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t8[8] = {s1[kb][g][0], s1[kb][g][1], s1[kb][g][2], s1[kb][g][3], s2[kb][g][0], s2[kb][g][1], s2[kb][g][2], s2[kb][g][3]};
                sc_reduce8(t8);
                if ((lane & 31) == 16) {
                    float* rp = red + (((wave * KB + kb) * 4 + g) * 2 + half) * 8;
                    *reinterpret_cast<float4*>(rp) = make_float4(t8[0], t8[1], t8[2], t8[3]);
                    *reinterpret_cast<float4*>(rp + 4) = make_float4(t8[4], t8[5], t8[6], t8[7]);
                }
            }
        __syncthreads();
        if (wave == 0 && (lane & 31) == 16) {
            const int64_t grp = a.stats_group_offset[phase] + (int64_t)frame * a.stats_bstride + wslot;
            float* sp = a.stats + grp * 2 * a.K + 4 * half;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (kb * 32 + 8 * g + 4 * half >= a.K) continue;
                    float4 u1 = make_float4(0.f, 0.f, 0.f, 0.f), u2 = u1;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float* rp = red + (((w * KB + kb) * 4 + g) * 2 + half) * 8;
                        const float4 x1 = *reinterpret_cast<const float4*>(rp), x2 = *reinterpret_cast<const float4*>(rp + 4);
                        u1.x += x1.x; u1.y += x1.y; u1.z += x1.z; u1.w += x1.w;
                        u2.x += x2.x; u2.y += x2.y; u2.z += x2.z; u2.w += x2.w;
                    }
                    *reinterpret_cast<float4*>(sp + kb * 32 + 8 * g) = u1;
                    *reinterpret_cast<float4*>(sp + a.K + kb * 32 + 8 * g) = u2;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
struct SconvPlan { bool ok; int C, KB, nch, ntaps; };

static SconvPlan sconv_plan(const tcvom_conv_desc* d, int nphase) {
    SconvPlan p;
    p.ok = false;
    static const bool off = getenv("TCVOM_NO_SCONV") != nullptr;                   // A/B switch
    if (off || !(nphase == 1 || nphase == 4)) return p;
    if (d->in_f16 && !TCVOM_BUILD_F16) return p;          // IEEE fp16 operands in the bf16 build: halo_conv / wsconv / igemm_nt
    const tcvom_conv_desc* d0 = d;
    if ((d0->C != 32 && d0->C != 64) || d0->K > 64 || d0->K % 4 != 0 || d0->ldo % 4 != 0) return p;
    int maxt = 0;
    for (int i = 0; i < nphase; ++i) {
        const tcvom_conv_desc* e = d + i;
        if (e->C != d0->C || e->K != d0->K || e->H != d0->H || e->W != d0->W || e->N != d0->N || e->OH != d0->OH || e->OW != d0->OW ||
            e->ldo != d0->ldo || e->wt != d0->wt || e->batch != d0->batch)
            return p;
        // the phase grid is the input grid, or reaches past it (data gradient onto a reflection-padded input: one more row / column)
        if (e->in_step != 1 || e->PH != d0->PH || e->PW != d0->PW || e->PH < e->H || e->PW < e->W || e->PH > e->H + 8 || e->PW > e->W + 8) return p;
        if (e->out_step != (nphase == 4 ? 2 : 1) || e->out_step * e->PH != e->OH || e->out_step * e->PW != e->OW) return p;
        if (e->out_off_h < 0 || e->out_off_h >= e->out_step || e->out_off_w < 0 || e->out_off_w >= e->out_step) return p;
        if (e->w_layout != 0) return p;
        int n = 0;
        for (int t = 0; t < e->ntaps; ++t) {
            if (e->tap_w[t] < 0) continue;
            if (e->tap_dh[t] < -1 || e->tap_dh[t] > 1 || e->tap_dw[t] < -1 || e->tap_dw[t] > 1) return p;
            ++n;
        }
        if (n == 0 || n > SC_MAXT) return p;
        if (n > maxt) maxt = n;
    }
    if (d0->batch > 1) {
        if (d0->in_bstride != (long long)d0->N * d0->H * d0->W * d0->C || d0->out_bstride != (long long)d0->N * d0->OH * d0->OW * d0->ldo) return p;
        if (d0->vec_bstride != 0) return p;
    }
    const int nb = d0->batch > 1 ? d0->batch : 1;
    if ((long long)d0->N * nb * d0->H * d0->W * d0->C >= (1ll << 31) || (long long)d0->N * nb * d0->OH * d0->OW * d0->ldo >= (1ll << 31)) return p;
    if (d0->W < 32 || d0->H < SC_TH) return p;
    // the single-phase 32 -> 32 3x3 layers belong to halo.hip, the 64 -> 64 / 128 -> 128 ones to wsconv.hip
    if (nphase == 1 && d0->C == d0->K) return p;
    // 1x1 convs would pay for 4 padded taps (measured 19.4 us against 18.0 us on the implicit GEMM): 4-tap tables are for phases
    if (nphase == 1 && maxt <= 4) return p;
    const int taps = maxt <= 4 ? 4 : 9;
    p.C = d0->C;
    p.KB = d0->K > 32 ? 2 : 1;
    p.ntaps = taps;
    p.nch = taps * d0->C / 16;
    // instantiated: (C, KB, NCH) = (32, 1, 8) (32, 2, 8) (32, 2, 18) (64, 1, 16) (64, 2, 16) (64, 1, 36)
    const int key = p.C * 1000 + p.KB * 100 + p.nch;
    if (!(key == 32108 || key == 32208 || key == 32218 || key == 64116 || key == 64216 || key == 64136)) return p;
    p.ok = true;
    return p;
}

static size_t sconv_lds(const SconvPlan& p) {
    const int cu = p.C / 8;
    const int ndma = (SC_PIX * cu + 63) / 64;
    return (size_t)2 * ndma * 1024 + (size_t)p.KB * 32 * (p.nch * 16 + 8) * 2 + (size_t)p.KB * 32 * 4;
}

static int sconv_grid(const tcvom_conv_desc* d, const SconvPlan& p, int nphase, int* tiles_per_wg, int* wgs_per_fp) {
    int occ = (int)((160 * 1024) / sconv_lds(p));
    if (occ > 4) occ = 4;
    if (occ < 1) occ = 1;
    const int nb = d->batch > 1 ? d->batch : 1;
    const int tpf = d->N * ((d->PH + SC_TH - 1) / SC_TH) * ((d->PW + SC_TW - 1) / SC_TW);
    int wpf = 256 * occ / (nb * nphase);
    if (wpf < 1) wpf = 1;
    if (wpf > tpf) wpf = tpf;
    *tiles_per_wg = (tpf + wpf - 1) / wpf;
    *wgs_per_fp = (tpf + *tiles_per_wg - 1) / *tiles_per_wg;
    return *wgs_per_fp * nb * nphase;
}

// statistics groups ONE phase of one frame writes (one per workgroup), or 0 when the shape is not handled here
int sconv_stats_groups(const tcvom_conv_desc* d, int nphase) {
    const SconvPlan p = sconv_plan(d, nphase);
    if (!p.ok) return 0;
    int tpw, wpf;
    sconv_grid(d, p, nphase, &tpw, &wpf);
    return wpf;
}

const char* sconv_variant(const tcvom_conv_desc* d, int nphase) {
    const SconvPlan p = sconv_plan(d, nphase);
    if (!p.ok) return nullptr;
    return p.C == 32 ? (p.KB == 1 ? "sconv<32,1>" : "sconv<32,2>") : (p.KB == 1 ? "sconv<64,1>" : "sconv<64,2>");
}

// returns 1 when the conv was launched here, 0 when the caller should use another kernel, < 0 on error
int sconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                     float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream) {
    if (mscale || mdiag) return 0;
    const SconvPlan p = sconv_plan(d, nphase);
    if (!p.ok) return 0;
    SconvArgs a;
    a.in = (const h16raw*)in;
    a.wgt = (const h16raw*)w;
    a.out = out;
    a.bias = bias;
    a.stats = stats;
    a.zero_page = zero_page;
    a.H = d->H; a.W = d->W; a.K = d->K; a.ldo = d->ldo; a.wt = d->wt; a.act = d->act; a.out_fp32 = d->out_fp32;
    a.OH = d->OH; a.OW = d->OW; a.ostep = d->out_step; a.nphase = nphase;
    const int nb = d->batch > 1 ? d->batch : 1;
    a.spf = d->N;
    a.w_bstride = nb > 1 ? d->w_bstride : 0;
    a.stats_bstride = nb > 1 ? d->stats_bstride : 0;
    for (int i = 0; i < 4; ++i) {
        const tcvom_conv_desc* e = d + (i < nphase ? i : 0);
        a.off_h[i] = e->out_off_h; a.off_w[i] = e->out_off_w;
        a.stats_group_offset[i] = e->stats_group_offset;
        int n = 0;
        for (int t = 0; t < e->ntaps && n < SC_MAXT; ++t) {
            if (e->tap_w[t] < 0) continue;
            a.tap_dh[i][n] = e->tap_dh[t]; a.tap_dw[i][n] = e->tap_dw[t]; a.tap_w[i][n] = e->tap_w[t];
            ++n;
        }
        for (; n < SC_MAXT; ++n) { a.tap_dh[i][n] = 0; a.tap_dw[i][n] = 0; a.tap_w[i][n] = -1; }
    }
    a.PH = d->PH; a.PW = d->PW;
    a.tiles_x = (d->PW + SC_TW - 1) / SC_TW;
    a.tiles_y = (d->PH + SC_TH - 1) / SC_TH;
    const int grid = sconv_grid(d, p, nphase, &a.tiles_per_wg, &a.wgs_per_fp);
    if (stats && nb > 1 && d->stats_bstride < (long long)a.wgs_per_fp * nphase)
        return tcvom_fail(TCVOM_ERR_ARG, "sconv: stats_bstride %lld < groups per frame", (long long)d->stats_bstride);
    const size_t lds_bytes = sconv_lds(p);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define SC_LAUNCH(...)                                                                                      \
    {                                                                                                       \
        static bool attr_ = false;                                                                          \
        if (!attr_) { e = hipFuncSetAttribute((const void*)sconv_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr_ = true; } \
        hipLaunchKernelGGL((sconv_kernel<__VA_ARGS__>), dim3(grid), dim3(256), lds_bytes, st, a);            \
    }
    const int key = p.C * 1000 + p.KB * 100 + p.nch;
    if (key == 32108) SC_LAUNCH(32, 1, 8)
    else if (key == 32208) SC_LAUNCH(32, 2, 8)
    else if (key == 32218) SC_LAUNCH(32, 2, 18)
    else if (key == 64116) SC_LAUNCH(64, 1, 16)
    else if (key == 64216) SC_LAUNCH(64, 2, 16)
    else SC_LAUNCH(64, 1, 36)
#undef SC_LAUNCH
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "sconv: %s", hipGetErrorString(e));
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "sconv: %s", hipGetErrorString(e2));
    return 1;
}
