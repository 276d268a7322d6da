// Halo-tile direct convolution for the full-resolution, small-channel layers (os1 / os2 of vmn_gca: the shortcut
// branches res_gca_enc.py:47-55, encoder conv2 resnet_enc.py:72, decoder layer4 resnet_dec.py:86 and their data
// gradients): stride-1 same-size convs with <= 32 input and 32 output channels and taps within +-1 pixel.
//
// These layers are HBM-bound (2M pixels x 32 channels in, the same out, 9 taps of reuse), but the implicit-GEMM
// kernel re-fetches every input pixel once PER TAP through the L2->LDS DMA path, which then bounds it (measured
// 190 us against a ~55 us HBM floor for 32->32 at 1088x1920).  Here a workgroup DMAs the (8+2) x (32+2) pixel
// halo of its 8x32 output tile into LDS ONCE and all taps read it from there; the weights live in LDS for the
// lifetime of the (persistent) workgroup, the halo of the next tile is in flight while the current one is used.
//
//   MFMA 32x32x16 bf16:  A = weights [32 out-channels][16 k],  B = pixels [32 pixels of one tile row][16 k],
//   k = (tap, channel) in chunks of 16;  wave w owns tile rows 2w, 2w+1.
// LDS halo image: pixel-major, 16-byte channel chunk c of halo pixel p stored at slot c ^ ((p >> 2) & 3) (C = 32;
// applied on the DMA source side, the LDS write stays lane-linear) so that the 16 lanes of a ds_read_b128 group --
// 16 consecutive pixels, 64 B apart -- cover 16 distinct bank groups.  Weight rows are padded by 16 B for the same
// reason.
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifndef HALO_ABL
#define HALO_ABL 0          // kernel ablations for timing (1: no output stores, 2: no statistics); 0 in the product
#endif
#define HALO_TH 8
#define HALO_TW 32
#define HALO_HW (HALO_TW + 2)
#define HALO_HH (HALO_TH + 2)
#define HALO_PIX (HALO_HW * HALO_HH)          // 340
#define HALO_MAX_TAPS 10

struct HaloArgs {
    const h16raw* in;
    const h16raw* wgt;
    void* out;
    const float* bias;
    float* stats;
    const h16raw* zero_page;
    int N, H, W, K, ldo, wt, act, out_fp32, stats_group_offset;
    int org;                     // 1: taps 0..2 on an input that carries its own (reflection) padding ring -- the tile origin moves by (1, 1)
    int tiles_x, tiles_y, ntiles, tiles_per_wg;
    int wgs_per_frame;           // a workgroup stays inside ONE frame: tiles [f*tpf + w*tiles_per_wg, ..) of frame f = v / wgs_per_frame
    long long stats_bstride;     // statistics groups between frames; a frame has wgs_per_frame groups (one per workgroup)
    int spf;                     // samples per frame (weights change every spf samples: w_bstride elements further)
    long long w_bstride;
    int tap_dh[HALO_MAX_TAPS + 2], tap_dw[HALO_MAX_TAPS + 2], tap_w[HALO_MAX_TAPS + 2];   // compacted; tap_w < 0: zero tap
};

#ifdef HALO_TRACE                      // tools/halo_trace.py: cycle stamps of workgroup 8 / wave 0, 5 per tile
__device__ unsigned long long halo_trace_buf[1024];
extern "C" int tcvom_halo_trace_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(halo_trace_buf), sizeof(unsigned long long) * 1024) == hipSuccess ? 0 : -1;
}
#define HALO_STAMP(i) if (tracing && tix < 200) halo_trace_buf[tix * 5 + (i)] = __builtin_readcyclecounter()
#else
#define HALO_STAMP(i)
#endif
// 8 sums over the 32 pixel lanes of each half wave: within quads, half rows, rows, then row 0 -> row 1 / row 2 -> row 3
// (row_bcast15): lanes 16..31 and 48..63 hold the totals.  (mov_dpp without `old`: one v_add_f32_dpp per step.)
#define HALO_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xF, true))
__device__ __forceinline__ void halo_reduce8(float (&t)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HALO_DPP(t[r], 0xB1, 0xF);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HALO_DPP(t[r], 0x4E, 0xF);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HALO_DPP(t[r], 0x141, 0xF);     // row_half_mirror
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HALO_DPP(t[r], 0x140, 0xF);     // row_mirror
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += HALO_DPP(t[r], 0x142, 0xA);     // row_bcast15 into rows 1 and 3
}
// LDS fragment reads the compiler does not see (see the MFMA section of halo_conv_kernel)
#define HALO_PF 3
typedef __attribute__((ext_vector_type(4))) unsigned int halo_u32x4_t;
struct HaloFrag { halo_u32x4_t v; };
__device__ __forceinline__ void halo_read(HaloFrag& f, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(f.v) : "v"(addr)); }
__device__ __forceinline__ void halo_fence(HaloFrag& f) { asm volatile("" : "+v"(f.v)); }
__device__ __forceinline__ h16x8_t halo_value(const HaloFrag& f) { return __builtin_bit_cast(h16x8_t, f.v); }
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// S = input stride (1, or 2: the stride-2 3x3 convs on the 8-channel full-resolution inputs, resnet_enc.py:68 conv1 and the
// guidance head res_gca_enc.py:20-28): the output tile stays 8 x 32 pixels, its halo covers (S*8 + 2) x (S*32 + 2) input pixels.
// XF = 1: IEEE fp16 operands whatever the build stores (tcvom_conv_desc.in_f16: the fp16 island of the bf16 build, common.h)
template <int C, int NCH, int S, int XF = 0>
__global__ __launch_bounds__(256) void halo_conv_kernel(const HaloArgs a) {
    constexpr int CU = C / 8;                           // 16-byte units per pixel
    constexpr int HW_ = S * HALO_TW + 2, HH_ = S * HALO_TH + 2, PIX_ = HW_ * HH_;
    static_assert(S == 1 || CU == 1, "the strided form is instantiated for 8-channel inputs only (no chunk swizzle)");
    constexpr int UNITS = PIX_ * CU;
    constexpr int NDMA = (UNITS + 63) / 64;             // DMA wave-instructions per halo
    constexpr int DMA_IT = (NDMA + 3) / 4;
    constexpr int SLOT = NDMA * 512;                    // bf16 elements per halo slot
    constexpr int WROW = NCH * 16 + 8;                  // padded weight row (elements)
    constexpr int NTAPS = NCH * 16 / C;
    extern __shared__ __attribute__((aligned(16))) h16raw lds[];
    h16raw* halo = lds;                                // [2][SLOT]
    h16raw* wl = lds + 2 * SLOT;                       // [32][WROW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int H = a.H, W = a.W, OH = (H - 2 * a.org) / S, OW = (W - 2 * a.org) / S;

    // XCD-contiguous workgroup order: neighbouring tile runs (which share halo rows) go to one L2
    int v;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_per_frame = a.spf * a.tiles_x * a.tiles_y;
    const int frame = v / a.wgs_per_frame, wslot = v - frame * a.wgs_per_frame;
    const int t_begin = frame * tiles_per_frame + wslot * a.tiles_per_wg;
    const int t_end = min((frame + 1) * tiles_per_frame, t_begin + a.tiles_per_wg);      // (may be empty: the statistics are still written)

    // ---- weights -> LDS, once per workgroup: wl[k][t*C + c] = wgt[frame][(k*wt + slot(t))*C + c]  (every frame of a
    // frame-batched call has its own SpectralNorm'd copy)
    {
        const h16raw* wsrc = a.wgt + (int64_t)frame * a.w_bstride;
        for (int u = tid; u < 32 * NTAPS * CU; u += 256) {
            const int cu = u % CU, t = (u / CU) % NTAPS, k = u / (CU * NTAPS);
            const int ws = a.tap_w[t];
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (ws >= 0 && k < a.K) val = *reinterpret_cast<const uint4*>(wsrc + ((int64_t)k * a.wt + ws) * C + cu * 8);
            *reinterpret_cast<uint4*>(wl + k * WROW + t * C + cu * 8) = val;
        }
    }

    // ---- per-lane constants
    // DMA: unit q = (it*4 + wave)*64 + lane -> halo pixel p = q / CU, stored slot q % CU holds chunk slot ^ swz(p)
    int d_rel[DMA_IT], d_yx[DMA_IT];
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) {
        const int q = (it * 4 + wave) * 64 + lane;
        const int p = q / CU, sl = q % CU;
        const int c16 = (CU == 4) ? (sl ^ ((p >> 2) & 3)) : sl;
        const int hy = p / HW_, hx = p - hy * HW_;
        d_rel[it] = ((hy - 1) * W + (hx - 1)) * C + c16 * 8;
        d_yx[it] = (q < UNITS) ? ((hy << 16) | hx) : -1;
    }
    // B fragments: byte offset inside a halo slot for (chunk, tile row j of this wave)
    int b_addr[NCH][2];
    int a_addr[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int kk = ch * 16 + half * 8;
        const int c16 = (kk % C) >> 3;
        // the two lane halves of a chunk may belong to different taps (C = 8): uniform table reads + select
        constexpr int dummy = 0;
        const int tap0 = (ch * 16) / C, tap1 = (ch * 16 + 8) / C;
        const int dh = half ? a.tap_dh[tap1] : a.tap_dh[tap0], dw = half ? a.tap_dw[tap1] : a.tap_dw[tap0];
        (void)dummy;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = (S * (2 * wave + j) + dh + 1) * HW_ + (S * col + dw + 1);
            const int sl = (CU == 4) ? (c16 ^ ((p >> 2) & 3)) : c16;
            b_addr[ch][j] = (p * CU + sl) * 16;
        }
        a_addr[ch] = (col * WROW + kk) * 2;
    }
    typedef __attribute__((address_space(3))) void* halo_lptr_t;
    const unsigned halo_lds = (unsigned)(uintptr_t)(halo_lptr_t)halo, wl_lds = (unsigned)(uintptr_t)(halo_lptr_t)wl;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // The halo DMA of tile (tx_, ty_, n_) into `slot`.  Interior tiles (wave-uniform test) skip the per-unit bounds tests; the
    // source is selected arithmetically (byte offset from `in`, or the offset of the zero page) so that no unit costs a branch;
    // the tile coordinates advance incrementally.  (Issuing these instructions behind MFMA pairs instead, as wsconv does, was
    // measured slower here: 6.8 k -> 8.6 k cycles per tile -- the fetch starts later and the tile waits for it.)
    const int64_t zoff = reinterpret_cast<const char*>(a.zero_page) - reinterpret_cast<const char*>(a.in);
    unsigned unit_ok = 0;                               // bit it: unit it of this lane exists (q < UNITS)
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) unit_ok |= (d_yx[it] >= 0 ? 1u : 0u) << it;
#define HALO_ISSUE(tx_, ty_, n_, slot)                                                                      \
    {                                                                                                       \
        const int y0_ = S * (ty_) * HALO_TH + a.org, x0_ = S * (tx_) * HALO_TW + a.org;   /* input coordinates of the tile origin */ \
        const int base_ = (((n_) * H + y0_) * W + x0_) * C;                                                 \
        const bool inner_ = (ty_) > 0 && (ty_) + 1 < a.tiles_y && (tx_) > 0 && (tx_) + 1 < a.tiles_x;       \
        _Pragma("unroll") for (int it = 0; it < DMA_IT; ++it) {                                             \
            if ((it * 4 + wave) < NDMA) {                                                                   \
                bool ok_ = (unit_ok >> it) & 1u;                                                            \
                if (!inner_) {                                                                              \
                    const int hy_ = d_yx[it] >> 16, hx_ = d_yx[it] & 0xffff;                                \
                    ok_ = ok_ && (unsigned)(y0_ + hy_ - 1) < (unsigned)H && (unsigned)(x0_ + hx_ - 1) < (unsigned)W; \
                }                                                                                           \
                const int64_t off_ = ok_ ? (int64_t)(base_ + d_rel[it]) * 2 : zoff;                         \
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(a.in) + off_),      \
                                                 (lptr_t)(halo + (slot) * SLOT + (it * 4 + wave) * 512), 16, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    }
    // tile coordinates of the NEXT tile to fetch, advanced incrementally
    int ntx, nty, nn;
    {
        const int t0 = t_begin < t_end ? t_begin : 0;
        ntx = t0 % a.tiles_x; nty = (t0 / a.tiles_x) % a.tiles_y; nn = t0 / (a.tiles_x * a.tiles_y);
    }
#define HALO_NEXT() { if (++ntx == a.tiles_x) { ntx = 0; if (++nty == a.tiles_y) { nty = 0; ++nn; } } }

    // epilogue constants: this lane's 16 bias values (channels 8 g + 4 half + r) and the activation slope
    float bsv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) bsv[g][r] = (a.bias && 8 * g + 4 * half + r < a.K) ? a.bias[8 * g + 4 * half + r] : 0.f;
    const float slope = a.act == 1 ? 0.f : a.act == 3 ? 0.01f : 1.f;

    if (t_begin < t_end) HALO_ISSUE(ntx, nty, nn, 0);
    int ctx = ntx, cty = nty, cn = nn;                  // ... and of the tile being computed
    int slot = 0;
    // channel sums of this wave over its whole tile run: [group g][channel r] of channels 8 g + 4 half + r
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[g][r] = 0.f; s2[g][r] = 0.f; }
#ifdef HALO_TRACE
    const bool tracing = blockIdx.x == 8 && tid == 0;
#endif
    for (int tile = t_begin; tile < t_end; ++tile) {
#ifdef HALO_TRACE
        const int tix = tile - t_begin;
#endif
        HALO_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HALO_STAMP(1);
        HALO_NEXT();
        if (tile + 1 < t_end) HALO_ISSUE(ntx, nty, nn, slot ^ 1);
        HALO_STAMP(2);
#if HALO_ABL == 3
        if (a.K != 12345) { slot ^= 1; continue; }
#endif
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // The fragment reads are inline asm with hand-counted lgkmcnt waits, HALO_PF chunks ahead of their MFMAs.  As plain LDS
        // loads the compiler (a) waited for lgkmcnt(0) in front of every MFMA pair -- 2 x NCH exposed LDS latencies per tile -- and
        // (b) put s_waitcnt vmcnt(0) in front of the first read, because the halo DMA of the NEXT tile issued above might alias
        // what the loads read: the tile then waited for the DMA it should have overlapped (measured at 1088x1920, 32 -> 32
        // channels, 3 frames: 260 us per launch, of which 100 us is the DMA loop alone and 160 us this section + epilogue).
        const unsigned hs = halo_lds + (unsigned)slot * (SLOT * 2);
        HaloFrag fa[HALO_PF + 1], fb[HALO_PF + 1][2];
        static_for<0, (HALO_PF < NCH ? HALO_PF : NCH)>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            halo_read(fa[c], wl_lds + a_addr[c]);
            halo_read(fb[c][0], hs + b_addr[c][0]);
            halo_read(fb[c][1], hs + b_addr[c][1]);
        });
        static_for<0, NCH>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            if constexpr (c + HALO_PF < NCH) {
                constexpr int n = (c + HALO_PF) % (HALO_PF + 1);
                halo_read(fa[n], wl_lds + a_addr[c + HALO_PF]);
                halo_read(fb[n][0], hs + b_addr[c + HALO_PF][0]);
                halo_read(fb[n][1], hs + b_addr[c + HALO_PF][1]);
            }
            constexpr int ahead = (NCH - 1 - c < HALO_PF ? NCH - 1 - c : HALO_PF) * 3;      // reads issued after chunk c's
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(ahead) : "memory");
            constexpr int k = c % (HALO_PF + 1);
            halo_fence(fa[k]); halo_fence(fb[k][0]); halo_fence(fb[k][1]);
            acc[0] = mfma16x<XF>(halo_value(fa[k]), halo_value(fb[k][0]), acc[0]);
            acc[1] = mfma16x<XF>(halo_value(fa[k]), halo_value(fb[k][1]), acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        HALO_STAMP(3);

        // ---- epilogue: bias, activation, store, running channel sums.  Straight-line code:
        // the activation as max(x, slope * x), the bias in registers, one wave-uniform branch on the output type for the whole
        // tile (per-element `if (act == ..)` / `if (bias)` / `if (out_fp32)` made this 1500 instructions = 6.7 k of the 12 k
        // cycles of a tile).
        const int tx = ctx, ty = cty, n = cn;
        ctx = ntx; cty = nty; cn = nn;
        const int y0 = ty * HALO_TH + 2 * wave, x = tx * HALO_TW + col;
        const int64_t o0 = ((int64_t)(n * OH + y0) * OW + x) * a.ldo + 4 * half;
        const int64_t ostep = (int64_t)OW * a.ldo;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xv = acc[j][g * 4 + r] + bsv[g][r];
                    xv = fmaxf(xv, xv * slope);
                    acc[j][g * 4 + r] = xv;
                    s1[g][r] += xv;
                    s2[g][r] += xv * xv;
                }
            }
        }
#if HALO_ABL != 1
        if (a.out_fp32 == 2) {                   // IEEE fp16 whatever the build stores (the fp16 island of the bf16 build)
            h16raw* op = reinterpret_cast<h16raw*>(a.out) + o0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (8 * g + 4 * half < a.K)
                        *reinterpret_cast<uint2*>(op + j * ostep + 8 * g) = make_uint2(pack2_ieee(acc[j][g * 4], acc[j][g * 4 + 1]), pack2_ieee(acc[j][g * 4 + 2], acc[j][g * 4 + 3]));
        } else {
            h16raw* op = reinterpret_cast<h16raw*>(a.out) + o0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (8 * g + 4 * half < a.K)
                        *reinterpret_cast<uint2*>(op + j * ostep + 8 * g) = make_uint2(pack2h(acc[j][g * 4], acc[j][g * 4 + 1]), pack2h(acc[j][g * 4 + 2], acc[j][g * 4 + 3]));
        }
#endif
        HALO_STAMP(4);
        slot ^= 1;
    }
#undef HALO_ISSUE
#undef HALO_NEXT
    if (a.stats && HALO_ABL != 2) {
        // BatchNorm partial statistics: ONE group per workgroup and frame -- the sums ride in registers over the tile run, cross the
        // 32 pixel lanes of each half wave once with DPP adds (per tile that reduction was 256 of the epilogue's 600 instructions) and
        // the four waves through LDS (170 groups per frame at 1080p instead of 680: the finalize is one launch, no bn_partial_reduce)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // the halo / weight arrays are free: every wave is past its last tile
        float* red = reinterpret_cast<float*>(lds);      // [4 waves][4 groups][2 halves][8]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t8[8] = {s1[g][0], s1[g][1], s1[g][2], s1[g][3], s2[g][0], s2[g][1], s2[g][2], s2[g][3]};
            halo_reduce8(t8);
            if ((lane & 31) == 16) {
                float* rp = red + ((wave * 4 + g) * 2 + half) * 8;
                *reinterpret_cast<float4*>(rp) = make_float4(t8[0], t8[1], t8[2], t8[3]);
                *reinterpret_cast<float4*>(rp + 4) = make_float4(t8[4], t8[5], t8[6], t8[7]);
            }
        }
        __syncthreads();
        if (wave == 0 && (lane & 31) == 16) {
            const int64_t grp = a.stats_group_offset + (int64_t)frame * a.stats_bstride + wslot;
            float* sp = a.stats + grp * 2 * a.K + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (8 * g + 4 * half >= a.K) continue;
                float4 u1 = make_float4(0.f, 0.f, 0.f, 0.f), u2 = u1;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* rp = red + ((w * 4 + g) * 2 + half) * 8;
                    const float4 x1 = *reinterpret_cast<const float4*>(rp), x2 = *reinterpret_cast<const float4*>(rp + 4);
                    u1.x += x1.x; u1.y += x1.y; u1.z += x1.z; u1.w += x1.w;
                    u2.x += x2.x; u2.y += x2.y; u2.z += x2.z; u2.w += x2.w;
                }
                *reinterpret_cast<float4*>(sp + 8 * g) = u1;
                *reinterpret_cast<float4*>(sp + a.K + 8 * g) = u2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
struct HaloPlan { bool ok, xf; int C, nch, ntaps, S, org; int taps[HALO_MAX_TAPS + 2]; };

static HaloPlan halo_plan(const tcvom_conv_desc* d, int nphase) {
    HaloPlan p;
    p.ok = false;
    p.xf = false;
    if (nphase != 1) return p;
    if (d->batch > 1) {           // frame-batched call: frames must be contiguous sample groups
        if (d->in_bstride != (long long)d->N * d->H * d->W * d->C || d->out_bstride != (long long)d->N * d->OH * d->OW * d->ldo) return p;
        if (d->vec_bstride != 0) return p;
    }
    const int S = d->in_step;
    static const bool no_strided = getenv("TCVOM_NO_HALO_S2") != nullptr;          // A/B switch
    if ((S != 1 && !(S == 2 && d->C == 8 && !no_strided)) || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return p;
    // org = 1: every tap in 0..2 on an input with its own padding ring (ReflectionPad2d(1) + Conv2d(padding=0), res_gca_enc.py:20-28)
    int org = d->ntaps > 0 ? 1 : 0;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->tap_w[t] >= 0 && (d->tap_dh[t] < 0 || d->tap_dw[t] < 0)) org = 0;
    if (d->PH != d->OH || d->PW != d->OW || d->OH * S != d->H - 2 * org || d->OW * S != d->W - 2 * org) return p;
    if (d->OH % HALO_TH != 0 || d->OW % HALO_TW != 0) return p;
    if (d->K > 32 || d->K % 4 != 0 || (d->C != 8 && d->C != 32)) return p;
    if ((long long)d->N * (d->batch > 1 ? d->batch : 1) * d->H * d->W * d->C >= (1ll << 31)) return p;
    int n = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        if (d->tap_w[t] < 0) continue;
        if (d->tap_dh[t] - org < -1 || d->tap_dh[t] - org > 1 || d->tap_dw[t] - org < -1 || d->tap_dw[t] - org > 1) return p;
        if (n >= HALO_MAX_TAPS) return p;
        p.taps[n++] = t;
    }
    if (n == 0) return p;
    const int per = 16 / (d->C < 16 ? d->C : 16);       // taps per 16-deep chunk (C = 8: 2)
    int padded = (n + per - 1) / per * per;
    int nch = padded * d->C / 16;
    // instantiated shapes: C=8 with 9 taps (5 chunks), C=32 with 9 taps (18 chunks); 16-bit results
    if (!((d->C == 8 && nch == 5) || (d->C == 32 && nch == 18))) return p;
    if (d->out_fp32 == 1) return p;
    // IEEE fp16 operands in the bf16 build (tcvom_conv_desc.in_f16): instantiated for the two stem layers of the fp16 island, with
    // fp16 results; other shapes go to the implicit GEMM (the plan -- not the launch -- declines: the statistics layout follows it)
    p.xf = d->in_f16 != 0 && !TCVOM_BUILD_F16;
    if (p.xf && !((S == 2 && nch == 5) || (S == 1 && d->C == 32 && nch == 18))) return p;
    if (p.xf != (d->out_fp32 == 2)) return p;           // IEEE fp16 results come with IEEE fp16 operands, and only with them
    for (int t = n; t < padded; ++t) p.taps[t] = -1;
    p.ok = true;
    p.C = d->C;
    p.nch = nch;
    p.ntaps = padded;
    p.S = S;
    p.org = org;
    return p;
}

static int halo_grid(const tcvom_conv_desc* d, const HaloPlan& p, int* tiles_per_wg, int* wgs_per_frame, size_t* lds_bytes) {
    const int cu = p.C / 8;
    const int ndma = ((p.S * HALO_TW + 2) * (p.S * HALO_TH + 2) * cu + 63) / 64;
    *lds_bytes = (size_t)2 * ndma * 1024 + (size_t)32 * (p.nch * 16 + 8) * 2;
    int occ = (int)((160 * 1024) / *lds_bytes);
    if (occ > 4) occ = 4;
    if (occ < 1) occ = 1;
    // persistent workgroups, every frame the same number (a workgroup never crosses into another frame: its weights and its
    // statistics group belong to one)
    const int nb = d->batch > 1 ? d->batch : 1;
    const int tpf = d->N * (d->OH / HALO_TH) * (d->OW / HALO_TW);
    int wpf = 256 * occ / nb;
    if (wpf < 1) wpf = 1;
    if (wpf > tpf) wpf = tpf;
    *tiles_per_wg = (tpf + wpf - 1) / wpf;
    *wgs_per_frame = (tpf + *tiles_per_wg - 1) / *tiles_per_wg;
    return *wgs_per_frame * nb;
}

// number of statistics groups the halo kernel writes for `d`, or 0 when the shape is not handled here
int halo_conv_stats_groups(const tcvom_conv_desc* d, int nphase) {
    const HaloPlan p = halo_plan(d, nphase);
    if (!p.ok) return 0;
    int tpw, wpf;
    size_t lds;
    halo_grid(d, p, &tpw, &wpf, &lds);
    return wpf;                                                // per batch element (frame): one per workgroup
}

// returns 1 when the conv was launched here, 0 when the caller should use the implicit GEMM, < 0 on error
int halo_conv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                         float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream) {
    if (mscale || mdiag) return 0;
    const HaloPlan p = halo_plan(d, nphase);
    if (!p.ok) return 0;
    HaloArgs a;
    a.in = (const h16raw*)in;
    a.wgt = (const h16raw*)w;
    a.out = out;
    a.bias = bias;
    a.stats = stats;
    a.zero_page = zero_page;
    a.N = d->N; a.H = d->H; a.W = d->W; a.K = d->K; a.ldo = d->ldo; a.wt = d->wt; a.act = d->act; a.out_fp32 = d->out_fp32;
    a.stats_group_offset = d->stats_group_offset;
    a.org = p.org;
    a.tiles_x = d->OW / HALO_TW;
    a.tiles_y = d->OH / HALO_TH;
    const int nb = d->batch > 1 ? d->batch : 1;
    a.N = d->N * nb;
    a.spf = d->N;
    a.w_bstride = nb > 1 ? d->w_bstride : 0;
    a.ntiles = a.N * a.tiles_x * a.tiles_y;
    for (int t = 0; t < HALO_MAX_TAPS + 2; ++t) {
        const int src = t < p.ntaps ? p.taps[t] : -1;
        a.tap_dh[t] = src >= 0 ? d->tap_dh[src] - p.org : 0;
        a.tap_dw[t] = src >= 0 ? d->tap_dw[src] - p.org : 0;
        a.tap_w[t] = src >= 0 ? d->tap_w[src] : -1;
    }
    size_t lds_bytes;
    const int grid = halo_grid(d, p, &a.tiles_per_wg, &a.wgs_per_frame, &lds_bytes);
    a.stats_bstride = nb > 1 ? d->stats_bstride : 0;
    if (stats && nb > 1 && d->stats_bstride < (long long)a.wgs_per_frame)
        return tcvom_fail(TCVOM_ERR_ARG, "halo_conv: stats_bstride %lld < groups per frame", (long long)d->stats_bstride);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define HALO_LAUNCH(...)                                                                                    \
    {                                                                                                       \
        static bool attr_ = false;                                                                          \
        if (!attr_) { e = hipFuncSetAttribute((const void*)halo_conv_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_ = true; } \
        hipLaunchKernelGGL((halo_conv_kernel<__VA_ARGS__>), dim3(grid), dim3(256), lds_bytes, st, a);        \
    }
    if (p.xf) {
        // fp16 island: encoder conv1 (8 -> 32, stride 2) and conv2 (32 -> 32); IEEE fp16 in, IEEE fp16 out (halo_plan admits no other)
        if (p.S == 2) HALO_LAUNCH(8, 5, 2, 1)
        else HALO_LAUNCH(32, 18, 1, 1)
    }
    else if (p.S == 2) HALO_LAUNCH(8, 5, 2)
    else if (p.C == 8) HALO_LAUNCH(8, 5, 1)
    else HALO_LAUNCH(32, 18, 1)
#undef HALO_LAUNCH
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "halo_conv: %s", hipGetErrorString(e));
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "halo_conv: %s", hipGetErrorString(e2));
    return 1;
}

// ------------------------------------------------------------------------------------------ weight gradient, halo form
// dW[k][tap][c] += sum_p dy[p][k] * x[S p + tap][c] for the same layers (<= 32 input channels, <= 32 output channels, 3 x 3 taps,
// stride 1 or 2): the implicit-GEMM TT kernel streams x once per TAP and dy once per column tile through L2 -> LDS; here a
// persistent workgroup DMAs the x halo and the dy tile of an 8x32 output-pixel tile ONCE, forms all 9 tap products from LDS and
// keeps the accumulators in registers over its whole tile run: HBM traffic = x + dy once.
//   MFMA 32x32x16: A = dy^T [32 out-ch][16 pixels], B = x^T [32 (tap, in-ch) columns][16 pixels] -- 32 / C taps per MFMA (C = 32:
//   nine products per 16 pixels, C = 16: five, C = 8: three; the columns past tap 8 repeat tap 8 and are never stored); the
//   reduction runs over pixels, both operands are pixel-major in LDS and are read with ds_read_b64_tr_b16 (a 16-lane group reads
//   4 pixels x 16 columns; stride 2 = every second halo pixel).  K = 16: the upper 16 rows of A repeat the lower ones, never stored.
//   The reads are inline asm (common.h: the builtin form makes the compiler drain the next tile's DMA first).
//   Wave w owns tile rows 2w, 2w+1 (4 k-steps of 16 pixels per tile).  At the end the 4 waves' accumulators are summed
//   through LDS and added to dw with fp32 atomics, once per workgroup.
struct HaloWgArgs {
    const h16raw* dy[8];
    const h16raw* in[8];
    float* dw[8];
    const h16raw* zero_page;
    int N, H, W, OH, OW, wt, org;             // H, W: the input (with its padding ring when org = 1); OH, OW: the dy grid
    int tiles_x, tiles_y, ntiles, tiles_per_wg;
    int tap_dh[9], tap_dw[9], tap_w[9];       // offsets relative to S * (output pixel) + org: -1 .. 1
};

template <int C, int S, int KO>
__device__ __forceinline__ void halo_wgrad_body(const HaloWgArgs& a) {
    constexpr int CU = C / 8, KU = KO / 8;                                        // 16-byte units per x / dy pixel
    constexpr int IW = S * HALO_TW + 2, IH = S * HALO_TH + 2;                     // the x halo (as halo_conv_kernel's)
    constexpr int XU = IW * IH * CU, YU = HALO_TH * HALO_TW * KU;                 // 16-byte units of the x halo / the dy tile
    constexpr int XI = (XU + 63) / 64, YI = YU / 64;                              // DMA wave-instructions
    constexpr int SLOT = (XI + YI) * 512;                                         // 16-bit elements per tile buffer
    constexpr int TPM = 32 / C, NM = (9 + TPM - 1) / TPM, G0 = (NM + 1) / 2;      // taps per MFMA, MFMAs per 16 pixels, first read group
    static_assert(2 * NM * 4 * 64 * 16 <= 2 * SLOT * 2, "reduction buffers must fit the tile buffers");
    __shared__ __attribute__((aligned(16))) h16raw lds[2 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prob = blockIdx.y;
    const h16raw* __restrict__ dy = a.dy[prob];
    const h16raw* __restrict__ in = a.in[prob];
    const int H = a.H, W = a.W, OH = a.OH, OW = a.OW;
    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define HWG_ISSUE(tile, slot)                                                                                \
    {                                                                                                        \
        const int tx_ = (tile) % a.tiles_x, ty_ = ((tile) / a.tiles_x) % a.tiles_y, n_ = (tile) / (a.tiles_x * a.tiles_y); \
        const int y0_ = ty_ * HALO_TH, x0_ = tx_ * HALO_TW;                                                  \
        for (int ii = wave; ii < XI + YI; ii += 4) {                                                         \
            const h16raw* src_;                                                                             \
            if (ii < XI) {                                                                                   \
                const int q_ = ii * 64 + lane, p_ = q_ / CU, hy_ = p_ / IW, hx_ = p_ - hy_ * IW;             \
                const int y_ = y0_ * S + a.org + hy_ - 1, x_ = x0_ * S + a.org + hx_ - 1;                    \
                const bool ok_ = q_ < XU && (unsigned)y_ < (unsigned)H && (unsigned)x_ < (unsigned)W;        \
                src_ = ok_ ? in + ((int64_t)(n_ * H + y_) * W + x_) * C + (q_ % CU) * 8 : a.zero_page;       \
            } else {                                                                                         \
                const int q_ = (ii - XI) * 64 + lane, p_ = q_ / KU;                                          \
                src_ = dy + ((int64_t)(n_ * OH + y0_ + p_ / HALO_TW) * OW + x0_ + p_ % HALO_TW) * KO + (q_ % KU) * 8; \
            }                                                                                                \
            __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(lds + (slot) * SLOT + ii * 512), 16, 0, 0); \
        }                                                                                                    \
    }

    f32x16_t acc[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // transposing reads: lane -> pixel (lane >> 5) * 8 + ((lane & 15) >> 2) (+4 for the hi half), columns
    // cq = ((lane >> 4) & 1) * 16 + (lane & 3) * 4 ..+3 of the 32: out-channel cq (mod KO) of dy; tap j TPM + cq / C, channel cq % C of x.
    // Byte addresses inside a tile buffer; the k-step moves them by immediates.
    const int tr_p = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int cq = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const unsigned y_rel = XI * 1024 + ((2 * wave * HALO_TW + tr_p) * KO + (cq % KO)) * 2;
    unsigned x_rel[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int tap = min(j * TPM + cq / C, 8);            // (columns past tap 8 repeat it)
        int dh = 0, dw = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) { dh = tap == t ? a.tap_dh[t] : dh; dw = tap == t ? a.tap_dw[t] : dw; }
        x_rel[j] = (((2 * wave * S + dh + 1) * IW + tr_p * S + dw + 1) * C + (cq % C)) * 2;
    }
    constexpr int XROW = S * IW * C * 2, YROW = HALO_TW * KO * 2, XHALF = 16 * S * C * 2, YHALF = 16 * KO * 2, XHI = 4 * S * C * 2, YHI = 4 * KO * 2;

    // k-step ks = tile row 2 wave + (ks >> 1), pixels (ks & 1) * 16 ..+15.  The NM products of a k-step are read in two groups
    // so that the reads of one group are in flight under the MFMAs of the other.
    TrFrag fy[2], fx[NM];
#define HWG_READ_Y(ks, set) tr_issue_imm<((ks) >> 1) * YROW + ((ks) & 1) * YHALF, YHI>(fy[set], ya)
#define HWG_READ_X(ks, t) tr_issue_imm<((ks) >> 1) * XROW + ((ks) & 1) * XHALF, XHI>(fx[t], xa[t])
#define HWG_STEP(ks)                                                                                          \
    {                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        tr_fence(fy[(ks) & 1]);                                                                               \
        _Pragma("unroll") for (int t = 0; t < G0; ++t) tr_fence(fx[t]);                                       \
        _Pragma("unroll") for (int t = G0; t < NM; ++t) HWG_READ_X(ks, t);                                    \
        _Pragma("unroll") for (int t = 0; t < G0; ++t)                                                        \
            acc[t] = mfma16(tr_value(fy[(ks) & 1]), tr_value(fx[t]), acc[t], 0, 0, 0);                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        _Pragma("unroll") for (int t = G0; t < NM; ++t) tr_fence(fx[t]);                                      \
        if ((ks) < 3) {                                                                                       \
            HWG_READ_Y((ks) + 1, ((ks) + 1) & 1);                                                             \
            _Pragma("unroll") for (int t = 0; t < G0; ++t) HWG_READ_X((ks) + 1, t);                           \
        }                                                                                                     \
        _Pragma("unroll") for (int t = G0; t < NM; ++t)                                                       \
            acc[t] = mfma16(tr_value(fy[(ks) & 1]), tr_value(fx[t]), acc[t], 0, 0, 0);                        \
    }

    HWG_ISSUE(t_begin, 0);
    int slot = 0;
    for (int tile = t_begin; tile < t_end; ++tile) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + 1 < t_end) HWG_ISSUE(tile + 1, slot ^ 1);
        const unsigned sb = lds0 + slot * (SLOT * 2);
        const unsigned ya = sb + y_rel;
        unsigned xa[NM];
#pragma unroll
        for (int t = 0; t < NM; ++t) xa[t] = sb + x_rel[t];
        HWG_READ_Y(0, 0);
#pragma unroll
        for (int t = 0; t < G0; ++t) HWG_READ_X(0, t);
        HWG_STEP(0)
        HWG_STEP(1)
        HWG_STEP(2)
        HWG_STEP(3)
        slot ^= 1;
    }
#undef HWG_ISSUE
#undef HWG_STEP
#undef HWG_READ_X
#undef HWG_READ_Y
    // ---- accumulators -> dw: rows k = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 = (tap, c)
    // The sum over the 4 waves is formed in LDS with plain stores / loads (waves 2, 3 -> waves 0, 1 -> wave 0; LDS float
    // atomics measured far slower), then wave 0 alone issues the fp32 global atomics: 4x fewer than one set per wave, and
    // they were the tail of the kernel (all workgroups finish together and hit the same few KB of dw).
    float* __restrict__ dw = a.dw[prob];
    __builtin_amdgcn_s_barrier();                         // every wave is done with the tile buffers
    f32x4_t* red = reinterpret_cast<f32x4_t*>(lds);       // [2 regions][NM products][4 quads][64 lanes] float4
#define HWG_PUT(region)                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NM; ++t)                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                        \
            red[(((region) * NM + t) * 4 + q) * 64 + lane] =                                                 \
                f32x4_t{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
#define HWG_GET(region)                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NM; ++t)                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
            const f32x4_t v = red[(((region) * NM + t) * 4 + q) * 64 + lane];                                \
            acc[t][4 * q] += v[0]; acc[t][4 * q + 1] += v[1]; acc[t][4 * q + 2] += v[2]; acc[t][4 * q + 3] += v[3]; \
        }
    if (wave >= 2) HWG_PUT(wave - 2)
    __syncthreads();
    if (wave < 2) HWG_GET(wave)
    __syncthreads();
    if (wave == 1) HWG_PUT(0)
    __syncthreads();
    if (wave != 0) return;
    HWG_GET(0)
#undef HWG_PUT
#undef HWG_GET
    const int cc = (lane & 31) % C;
#pragma unroll
    for (int t = 0; t < NM; ++t) {
        const int tap = t * TPM + (lane & 31) / C;        // (the column's own tap: taps past 8 exist in the last product only)
        if (tap > 8) continue;
        int ws = 0;
#pragma unroll
        for (int u = 0; u < 9; ++u) ws = tap == u ? a.tap_w[u] : ws;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (k < KO) atomicAdd(dw + ((int64_t)k * a.wt + ws) * C + cc, acc[t][r]);
        }
    }
}

// (two co-resident workgroups per CU, i.e. 2 waves per SIMD with the full register budget -- except 16 -> 32 stride 2, whose two
// 51 KB tile buffers leave room for one)
template <int C, int S, int KO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void halo_wgrad_kernel(const HaloWgArgs a) { halo_wgrad_body<C, S, KO>(a); }
template <int C, int S, int KO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void halo_wgrad_kernel_lds(const HaloWgArgs a) { halo_wgrad_body<C, S, KO>(a); }

// the instantiated (input channels, stride, output channels) shapes: 32 -> 32 (os1 / os2 shortcut, stem conv2), 8 -> 32 (os1 shortcut),
// 8 -> 32 / 8 -> 16 stride 2 (stem conv1, guidance head conv1), 16 -> 32 stride 2 (guidance head conv2)
static int halo_wgrad_shape(const tcvom_conv_desc* d, int nphase, int ldy, int* org_out) {
    static const bool disabled = getenv("TCVOM_NO_HALO_WGRAD") != nullptr;      // A/B switches for tools/igemm_bench.py
    static const bool thin_off = getenv("TCVOM_NO_HALO_WGRAD_THIN") != nullptr; // ... the shapes other than 32 -> 32 (round 6)
    if (disabled || nphase != 1) return 0;
    const int S = d->in_step;
    int shape = 0;
    if (d->C == 32 && d->K == 32 && S == 1) shape = 1;
    else if (thin_off) return 0;
    else if (d->C == 8 && d->K == 32 && S == 1) shape = 2;
    else if (d->C == 8 && d->K == 32 && S == 2) shape = 3;
    else if (d->C == 8 && d->K == 16 && S == 2) shape = 4;
    else if (d->C == 16 && d->K == 32 && S == 2) shape = 5;
    if (!shape || ldy != d->K) return 0;                    // (d->batch is a forward-only field)
    if (d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return 0;
    int org = d->ntaps > 0 ? 1 : 0, nt = 0;                 // org = 1: taps 0..2 on an input with its own padding ring (as halo_plan)
    for (int t = 0; t < d->ntaps; ++t)
        if (d->tap_w[t] >= 0 && (d->tap_dh[t] < 0 || d->tap_dw[t] < 0)) org = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        if (d->tap_w[t] < 0) continue;
        if (d->tap_dh[t] - org < -1 || d->tap_dh[t] - org > 1 || d->tap_dw[t] - org < -1 || d->tap_dw[t] - org > 1) return 0;
        ++nt;
    }
    if (nt != 9) return 0;                                  // the kernel is written for the full 3x3 stencil
    if (d->PH != d->OH || d->PW != d->OW || d->OH * S != d->H - 2 * org || d->OW * S != d->W - 2 * org) return 0;
    if (d->OH % HALO_TH != 0 || d->OW % HALO_TW != 0) return 0;
    if ((long long)d->N * d->H * d->W * d->C >= (1ll << 31) || (long long)d->N * d->OH * d->OW * d->K >= (1ll << 31)) return 0;
    if (org_out) *org_out = org;
    return shape;
}
const char* halo_wgrad_variant(const tcvom_conv_desc* d, int ldy) {
    static const char* const names[] = {nullptr, "halo_wgrad<32>", "halo_wgrad<8,1,32>", "halo_wgrad<8,2,32>", "halo_wgrad<8,2,16>", "halo_wgrad<16,2,32>"};
    return names[halo_wgrad_shape(d, 1, ldy, nullptr)];
}

// 1: launched, 0: not a shape for this kernel, -1: launch error
int halo_wgrad_try_launch(const void* const* dys, const void* const* ins, float* const* dws, int nbatch, const tcvom_conv_desc* d,
                          int nphase, int ldy, const h16raw* zero_page, void* stream) {
    int org = 0;
    const int shape = halo_wgrad_shape(d, nphase, ldy, &org);
    if (!shape || nbatch < 1 || nbatch > 8) return 0;
    HaloWgArgs a;
    int nt = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        if (d->tap_w[t] < 0) continue;
        a.tap_dh[nt] = d->tap_dh[t] - org; a.tap_dw[nt] = d->tap_dw[t] - org; a.tap_w[nt] = d->tap_w[t];
        ++nt;
    }
    for (int i = 0; i < 8; ++i) {
        const int j = i < nbatch ? i : 0;
        a.dy[i] = (const h16raw*)dys[j]; a.in[i] = (const h16raw*)ins[j]; a.dw[i] = dws[j];
    }
    a.zero_page = zero_page;
    a.N = d->N; a.H = d->H; a.W = d->W; a.OH = d->OH; a.OW = d->OW; a.wt = d->wt; a.org = org;
    a.tiles_x = d->OW / HALO_TW; a.tiles_y = d->OH / HALO_TH;
    a.ntiles = d->N * a.tiles_x * a.tiles_y;
    int wgs = (shape == 5 ? 256 : 512) / nbatch;          // co-resident workgroups over all problems (16 -> 32 stride 2: 102 KB of LDS, one per CU)
    if (wgs < 1) wgs = 1;
    if (wgs > a.ntiles) wgs = a.ntiles;
    a.tiles_per_wg = (a.ntiles + wgs - 1) / wgs;
    wgs = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    const dim3 grid(wgs, nbatch), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (shape) {
        case 1: hipLaunchKernelGGL((halo_wgrad_kernel<32, 1, 32>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((halo_wgrad_kernel<8, 1, 32>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((halo_wgrad_kernel<8, 2, 32>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((halo_wgrad_kernel<8, 2, 16>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((halo_wgrad_kernel_lds<16, 2, 32>), grid, block, 0, st, a); break;
    }
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
