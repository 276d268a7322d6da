// Batched SpectralNorm power iteration + weight packing for ALL conv layers of the net in a
// handful of launches (the reference runs ~1.2 k tiny mv/div/norm kernels per window:
// models/GCA/ops.py:25-45,74-80, one power iteration per forward CALL of each wrapped conv).
//
// A device-resident table describes every conv weight ("layer"): fp32 master weight in PyTorch
// layout, optional u/v vectors (SpectralNorm), and where its bf16 packed copies live:
//     fwd pack  [K][T][Cpad]   (A operand of the forward igemm;   Cpad = max(C, 8))
//     bwd pack  [C][T][K]      (A operand of the data-gradient igemm; absent for image-fed layers)
// For ConvTranspose2d weights ([Cin][Kout][R][S], matrix height = Cin, ops.py:30) K means Kout
// and C means Cin in both packs.
//
// Backward: the igemm_tt weight-gradient kernels accumulate dW~ (gradient w.r.t. the normalised,
// packed weight) into an fp32 arena; sn_backward turns that into the gradient of weight_bar:
//     dW_bar = dW~/sigma - (<dW~, W_bar>/sigma^2) * u v^T        (u, v constants, ops.py:32-36)
#include <type_traits>
#include "common.h"

#define SN_WORDS 24
// layer record (int64 words)
enum {
    SN_W = 0, SN_U, SN_V, SN_H, SN_WD, SN_KIND, SN_K, SN_C, SN_T, SN_CPAD,
    SN_FWD_OFF, SN_BWD_OFF, SN_T_OFF, SN_S_OFF, SN_DW_OFF, SN_GRAD_OFF, SN_NUMEL, SN_WS_STATS
};
// kind bits: 64 = forward pack in IEEE fp16 (bf16 build: the fp16 island), 32 = fragment-major packs (sn_frag_index), 1 = ConvTranspose layout, 2 = plain (no spectral norm),
//            8 = weight-standardised (FBA base, models/FBA/layers_WS.py:13-23): packed value = (w - mean_row) * inv_row
//                with the row statistics at SN_WS_STATS (float4 per output channel: mean, 1/(std + 1e-5), std, unused),
//           16 = 7x7 stride-2 stem in space-to-depth form: the packed weight is the 4x4 stride-1 kernel over the 2x2
//                space-to-depth input (T = 16 taps (a, b) in -2..1, Cpad = 64 channels (2p + q) * 16 + c); tap
//                (a, b), sub-pixel (p, q) holds w[k][c][2a + p + 3][2b + q + 3] (zero outside the 7x7 window)

// 7x7 stem: packed (t, c') -> element offset inside row k of the [C][7][7] kernel, or -1
__device__ __forceinline__ int stem_src(int t, int cp, int C) {
    const int a = (t >> 2) - 2, b = (t & 3) - 2, sub = cp >> 4, ch = cp & 15;
    const int u = 2 * a + (sub >> 1) + 3, v = 2 * b + (sub & 1) + 3;
    return (ch < C && u >= 0 && u < 7 && v >= 0 && v < 7) ? (ch * 7 + u) * 7 + v : -1;
}
// ... and back: element (ch, u, v) -> packed (t, c') as t * 64 + c'
__device__ __forceinline__ int stem_dst(int ch, int u, int v) {
    const int uu = u - 3, vv = v - 3;
    const int a = (uu - (uu & 1)) / 2, p = uu & 1, b = (vv - (vv & 1)) / 2, q = vv & 1;
    return ((a + 2) * 4 + (b + 2)) * 64 + (p * 2 + q) * 16 + ch;
}

struct SnScratch {
    float* tvec;      // [sum wd]      W^T u   (atomic sums: zero on entry; sn_finalize_kernel zeroes a layer's slice after its last read)
    float* svec;      // [sum h]       W t
    float* sigma;     // [ncalls][L]
    float* uhist;     // [ncalls][sum h]
    float* vhist;     // [ncalls][sum wd]
    int64_t sum_h, sum_wd;
    int L;
};

__global__ __launch_bounds__(256) void sn_wt_u_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work,
                                                      float* __restrict__ tvec)
{
    const int layer = work[blockIdx.x * 2], r0 = work[blockIdx.x * 2 + 1];
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const float* W = reinterpret_cast<const float*>(L[SN_W]);
    const float* u = reinterpret_cast<const float*>(L[SN_U]);
    const int h = (int)L[SN_H], wd = (int)L[SN_WD];
    float* t = tvec + L[SN_T_OFF];
    __shared__ float us[16];
    if (threadIdx.x < 16) us[threadIdx.x] = (r0 + threadIdx.x < h) ? u[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    const int nr = min(16, h - r0);
    if ((wd & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
        // 16 rows x 4 columns per thread: the 16 float4 loads are issued together (the kernel is bound by the bytes in flight;
        // one 4-byte load per row and iteration: 60 us per call for 102 MB of weights)
        const float4* W4 = reinterpret_cast<const float4*>(W + (int64_t)r0 * wd);
        const int wd4 = wd >> 2;
        for (int c4 = threadIdx.x; c4 < wd4; c4 += 256) {
            float4 w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = r < nr ? W4[(int64_t)r * wd4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) { a.x += w[r].x * us[r]; a.y += w[r].y * us[r]; a.z += w[r].z * us[r]; a.w += w[r].w * us[r]; }
            atomicAdd(t + c4 * 4 + 0, a.x); atomicAdd(t + c4 * 4 + 1, a.y); atomicAdd(t + c4 * 4 + 2, a.z); atomicAdd(t + c4 * 4 + 3, a.w);
        }
        return;
    }
    for (int c = threadIdx.x; c < wd; c += 256) {
        float a = 0.f;
        for (int r = 0; r < nr; ++r) a += W[(int64_t)(r0 + r) * wd + c] * us[r];
        atomicAdd(t + c, a);
    }
}

// mode 0: s = W t (t = tvec, train);  mode 1: s = W v (stored v, eval)
__global__ __launch_bounds__(256) void sn_w_v_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work,
                                                     const float* __restrict__ tvec, float* __restrict__ svec, int mode)
{
    const int layer = work[blockIdx.x * 2], r0 = work[blockIdx.x * 2 + 1];
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const float* W = reinterpret_cast<const float*>(L[SN_W]);
    const int h = (int)L[SN_H], wd = (int)L[SN_WD];
    const float* t = mode == 0 ? tvec + L[SN_T_OFF] : reinterpret_cast<const float*>(L[SN_V]);
    const int r = r0 + (threadIdx.x >> 6);
    if (r >= h) return;
    float a = 0.f;
    if ((wd & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(t)) & 15) == 0) {
        const float4* w4 = reinterpret_cast<const float4*>(W + (int64_t)r * wd);
        const float4* t4 = reinterpret_cast<const float4*>(t);
        const int wd4 = wd >> 2;
        int c = threadIdx.x & 63;
        for (; c + 192 < wd4; c += 256) {                       // 4 x 16 bytes of the row in flight per lane
            const float4 w0 = w4[c], w1 = w4[c + 64], w2 = w4[c + 128], w3 = w4[c + 192];
            const float4 x0 = t4[c], x1 = t4[c + 64], x2 = t4[c + 128], x3 = t4[c + 192];
            a += w0.x * x0.x + w0.y * x0.y + w0.z * x0.z + w0.w * x0.w + w1.x * x1.x + w1.y * x1.y + w1.z * x1.z + w1.w * x1.w
               + w2.x * x2.x + w2.y * x2.y + w2.z * x2.z + w2.w * x2.w + w3.x * x3.x + w3.y * x3.y + w3.z * x3.z + w3.w * x3.w;
        }
        for (; c < wd4; c += 64) { const float4 w0 = w4[c], x0 = t4[c]; a += w0.x * x0.x + w0.y * x0.y + w0.z * x0.z + w0.w * x0.w; }
    } else {
        for (int c = threadIdx.x & 63; c < wd; c += 64) a += W[(int64_t)r * wd + c] * t[c];
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) svec[L[SN_S_OFF] + r] = a;
}

// one block per spectral layer
__global__ __launch_bounds__(256) void sn_finalize_kernel(const int64_t* __restrict__ tab, const int* __restrict__ layers,
                                                          SnScratch sc, int call, int mode)
{
    __shared__ float red[4];
    const int layer = layers[blockIdx.x];
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    float* u = reinterpret_cast<float*>(L[SN_U]);
    float* v = reinterpret_cast<float*>(L[SN_V]);
    const int h = (int)L[SN_H], wd = (int)L[SN_WD];
    float* t = sc.tvec + L[SN_T_OFF];
    const float* s = sc.svec + L[SN_S_OFF];
    float* uh = sc.uhist + (int64_t)call * sc.sum_h + L[SN_S_OFF];
    float* vh = sc.vhist + (int64_t)call * sc.sum_wd + L[SN_T_OFF];
    const float eps = 1e-12f;
    if (mode == 0) {
        float a = 0.f;
        for (int c = threadIdx.x; c < wd; c += 256) a += t[c] * t[c];
        const float nt = sqrtf(block_sum_256(a, red));
        float b = 0.f;
        for (int r = threadIdx.x; r < h; r += 256) b += s[r] * s[r];
        const float ns_raw = sqrtf(block_sum_256(b, red));
        const float inv_t = 1.f / (nt + eps);
        const float ns = ns_raw * inv_t;               // ||W v||
        const float inv_s = inv_t / (ns + eps);        // u = (s_raw*inv_t)/(ns+eps)
        // (t is an accumulator of atomics: left zero for the next training iteration -- no memset launch in front of every call)
        for (int c = threadIdx.x; c < wd; c += 256) { const float x = t[c] * inv_t; v[c] = x; vh[c] = x; t[c] = 0.f; }
        for (int r = threadIdx.x; r < h; r += 256) { const float x = s[r] * inv_s; u[r] = x; uh[r] = x; }
        if (threadIdx.x == 0) sc.sigma[(int64_t)call * sc.L + layer] = ns * ns / (ns + eps);
    } else {
        float a = 0.f;
        for (int r = threadIdx.x; r < h; r += 256) { a += u[r] * s[r]; uh[r] = u[r]; }
        for (int c = threadIdx.x; c < wd; c += 256) vh[c] = v[c];
        a = block_sum_256(a, red);
        if (threadIdx.x == 0) sc.sigma[(int64_t)call * sc.L + layer] = a;
    }
}

// Tiled pack (work rows with which = 2 / 3): a block takes a 32 (k) x 64 (c) x T tile of one layer, reads it in the
// weight's own order (contiguous runs of 64 T or 32 T floats), keeps the scaled bf16 values in LDS and writes the forward
// pack [K][T][Cpad] (128-byte runs) and, for which = 3, the data-gradient pack [C][T][K] (64-byte runs) from there.  The
// one-thread-per-output form below reads the fp32 weights with a stride of T floats (forward pack) or C T floats (backward
// pack): 793 MiB fetched per call for 102 MB of weights (profiles/r01_k_hbm_traffic_pmc.md), 208 us per call.
// kind bit 32: "fragment-major" packs for the weight-stationary conv (csrc/wsconv.hip, tcvom_conv_desc.w_layout = 1): element
// (row, slot, col) -- (k, t, c) of the forward pack, (c, t, k) of the data-gradient pack -- goes to the position below, so
// that every 32 x 16 MFMA A fragment is one contiguous 1 KiB block in lane order
__device__ __forceinline__ int64_t sn_frag_index(int row, int slot, int col, int T, int ncols) {
    return ((((int64_t)(row >> 5) * T + slot) * (ncols >> 4) + (col >> 4)) * 64 + ((col >> 3) & 1) * 32 + (row & 31)) * 8 + (col & 7);
}
constexpr int SNP_TK = 32, SNP_TC = 64, SNP_RB = 16;
constexpr int SNP_SLOTS = 18;                    // T <= 9 with the (build type, IEEE fp16) pair of an fp16-island layer, or T <= 16 alone
constexpr int SNP_ROW = SNP_TC * SNP_SLOTS + 2;  // LDS row of one k: [c][slot], +2 elements: consecutive k rows are 32 T + 1 banks apart

template <int TC>                                // TC = compile-time tap count (9, 16, 1) or 0: read it from the table
__device__ __forceinline__ void sn_pack_tile(const int64_t* __restrict__ L, int which, int tile, float inv, int call,
                                             h16raw* __restrict__ fwd_arena, h16raw* __restrict__ bwd_arena,
                                             int64_t fwd_call_stride, int64_t bwd_call_stride, h16raw* __restrict__ lds) {
    const float* __restrict__ W = reinterpret_cast<const float*>(L[SN_W]);
    const int kind = (int)L[SN_KIND];
    const int K = (int)L[SN_K], C = (int)L[SN_C], T = TC ? TC : (int)L[SN_T], Cp = (int)L[SN_CPAD];
    const bool transposed = kind & 1, ws = kind & 8;
    // kind bit 64 (bf16 build): the FORWARD pack in IEEE fp16 (the fp16 island, tcvom_conv_desc.in_f16), the data-gradient pack in the
    // build's type as ever: LDS slot t holds the build-type value, slot T + t the fp16 one
    const bool f16f = (kind & 64) && !TCVOM_BUILD_F16;
    const int TT = T;                            // slots per (k, c) of the forward pack  (all divisions below are by T or constants:
                                                 // with a runtime T the integer divisions WERE the kernel, ~250 VALU ops / element)
    const int TL = f16f ? 2 * T : T;             // ... of the LDS image
    const int rp = SNP_TC * TL + 2;              // row pitch of this layer (<= SNP_ROW)
    const int nct = (Cp + SNP_TC - 1) / SNP_TC;
    const int k0 = (tile / nct) * SNP_TK, c0 = (tile % nct) * SNP_TC;
    const int nk = min(SNP_TK, K - k0), nc = min(SNP_TC, C - c0);        // nc <= 0: a tile of padding channels only
    const float4* wstat = reinterpret_cast<const float4*>(L[SN_WS_STATS]);
    const int tid = threadIdx.x;
    // ---- read: rows are k (runs of nc * T floats) or, for ConvTranspose weights [c][k][t], c (runs of nk * T floats);
    // LDS element (kl, cl, slot) at kl * rp + cl * TT + slot: consecutive source elements land on consecutive addresses
    const int nrow = transposed ? nc : nk, run = (transposed ? nk : nc) * T;
    const int64_t row_stride = transposed ? (int64_t)K * T : (int64_t)C * T;
    const float* __restrict__ Wt = W + (transposed ? ((int64_t)c0 * K + k0) * T : ((int64_t)k0 * C + c0) * T);
    // SNP_RB rows per pass: their loads are independent and in flight together (the kernel is latency-bound otherwise)
    for (int r0 = 0; r0 < nrow; r0 += SNP_RB) {
        for (int e = tid; e < run; e += 256) {
            const int q = e / T, t = e - q * T;
            float w[SNP_RB];
#pragma unroll
            for (int u = 0; u < SNP_RB; ++u) w[u] = r0 + u < nrow ? Wt[(r0 + u) * row_stride + e] : 0.f;
#pragma unroll
            for (int u = 0; u < SNP_RB; ++u) {
                if (r0 + u >= nrow) break;
                const int r = r0 + u;
                const int kl = transposed ? q : r, cl = transposed ? r : q;
                float val = w[u] * inv;
                if (ws) { const float4 st = wstat[k0 + kl]; val = (w[u] - st.x) * st.y; }
                const h16raw hi = f2h(val);
                h16raw* dst = lds + kl * rp + cl * TL + t;
                dst[0] = hi;
                if (f16f) dst[T] = f2h_ieee(val);
            }
        }
    }
    __syncthreads();
    // ---- forward pack: [K][TT][Cp], channels c0 .. c0 + 63 of every (k, slot) row of the tile; padding channels are zero.
    // A thread gathers 8 consecutive channels from LDS and writes them as ONE 16-byte store (2-byte stores, 128 bytes per wave
    // instruction, made the stores the kernel: 317 us for the three calls of a 1080p window against ~80 us of traffic).
    h16raw* __restrict__ fdst = fwd_arena + call * fwd_call_stride + L[SN_FWD_OFF];
    const int ncp = min(SNP_TC, Cp - c0);
    const bool frag = kind & 32;
    for (int e = tid; e < nk * TT * (SNP_TC / 8); e += 256) {
        const int cl = (e % (SNP_TC / 8)) * 8, row = e / (SNP_TC / 8);
        int kl, slot;
        kl = row / T; slot = row - kl * T;
        if (cl < ncp) {                                   // (Cp % 8 == 0: the 8 channels are inside the padded row together)
            const h16raw* src = lds + kl * rp + cl * TL + slot + (f16f ? T : 0);
            h16raw v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = cl + i < nc ? src[i * TL] : (h16raw)0;
            const int64_t di = frag ? sn_frag_index(k0 + kl, slot, c0 + cl, TT, Cp) : ((int64_t)(k0 + kl) * TT + slot) * Cp + c0 + cl;
            uint4 q;
            q.x = v[0] | ((unsigned)v[1] << 16); q.y = v[2] | ((unsigned)v[3] << 16);
            q.z = v[4] | ((unsigned)v[5] << 16); q.w = v[6] | ((unsigned)v[7] << 16);
            *reinterpret_cast<uint4*>(fdst + di) = q;     // (fragment-major: the 8 channels col & 7 of one row are contiguous too)
        }
    }
    // ---- data-gradient pack: [C][T][K], output channels k0 .. k0 + 31 of every (c, t) row
    if (which == 3 && nc > 0) {
        h16raw* __restrict__ bdst = bwd_arena + call * bwd_call_stride + L[SN_BWD_OFF];
        if ((K & 7) == 0 && (nk & 7) == 0) {
            for (int e = tid; e < nc * T * (SNP_TK / 8); e += 256) {
                const int kl = (e % (SNP_TK / 8)) * 8, row = e / (SNP_TK / 8), t = row % T, cl = row / T;
                if (kl < nk) {
                    const h16raw* src = lds + kl * rp + cl * TL + t;
                    h16raw v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = src[i * rp];
                    const int64_t di = frag ? sn_frag_index(c0 + cl, t, k0 + kl, T, K) : ((int64_t)(c0 + cl) * T + t) * K + k0 + kl;
                    uint4 q;
                    q.x = v[0] | ((unsigned)v[1] << 16); q.y = v[2] | ((unsigned)v[3] << 16);
                    q.z = v[4] | ((unsigned)v[5] << 16); q.w = v[6] | ((unsigned)v[7] << 16);
                    *reinterpret_cast<uint4*>(bdst + di) = q;
                }
            }
        } else {
            for (int e = tid; e < nc * T * SNP_TK; e += 256) {
                const int kl = e % SNP_TK, row = e / SNP_TK, t = row % T, cl = row / T;
                if (kl < nk) {
                    const int64_t di = frag ? sn_frag_index(c0 + cl, t, k0 + kl, T, K) : ((int64_t)(c0 + cl) * T + t) * K + k0 + kl;
                    bdst[di] = lds[kl * rp + cl * TL + t];
                }
            }
        }
    }
}

// pack: one thread per OUTPUT element.  which = 0 fwd pack [K][T][Cpad], 1 bwd pack [C][T][K]; 2 / 3: tiled (above)
__global__ __launch_bounds__(256) void sn_pack_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work,
                                                      const float* __restrict__ sigma, int L_total, int call,
                                                      h16raw* __restrict__ fwd_arena, h16raw* __restrict__ bwd_arena,
                                                      int64_t fwd_call_stride, int64_t bwd_call_stride)
{
    __shared__ h16raw tile_lds[SNP_TK * SNP_ROW];
    // `call` < 0: the work entry carries its own call in bits 8.. of `which` (all power-iteration calls of a window packed in ONE
    // launch: the fp32 weights are read once per tile from HBM and the later calls hit the cache)
    const int layer = work[blockIdx.x * 3], wraw = work[blockIdx.x * 3 + 1], which = wraw & 0xff;
    if (call < 0) call = wraw >> 8;
    const int64_t base = (int64_t)work[blockIdx.x * 3 + 2] * 256;
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const float* W = reinterpret_cast<const float*>(L[SN_W]);
    const int kind = (int)L[SN_KIND];
    const int K = (int)L[SN_K], C = (int)L[SN_C], T = (int)L[SN_T], Cp = (int)L[SN_CPAD];
    const float inv = (kind & 2) ? 1.f : 1.f / sigma[(int64_t)call * L_total + layer];
    if (which >= 2) {
        const int tile = work[blockIdx.x * 3 + 2];
        if (T == 9) sn_pack_tile<9>(L, which, tile, inv, call, fwd_arena, bwd_arena, fwd_call_stride, bwd_call_stride, tile_lds);
        else if (T == 16) sn_pack_tile<16>(L, which, tile, inv, call, fwd_arena, bwd_arena, fwd_call_stride, bwd_call_stride, tile_lds);
        else if (T == 1) sn_pack_tile<1>(L, which, tile, inv, call, fwd_arena, bwd_arena, fwd_call_stride, bwd_call_stride, tile_lds);
        else sn_pack_tile<0>(L, which, tile, inv, call, fwd_arena, bwd_arena, fwd_call_stride, bwd_call_stride, tile_lds);
        return;
    }
    const int64_t idx = base + threadIdx.x;
    int k, c, t;
    if (which == 0) {
        if (idx >= (int64_t)K * T * Cp) return;
        c = (int)(idx % Cp);
        t = (int)((idx / Cp) % T);
        k = (int)(idx / ((int64_t)Cp * T));
    } else {
        if (idx >= (int64_t)C * T * K) return;
        k = (int)(idx % K);
        t = (int)((idx / K) % T);
        c = (int)(idx / ((int64_t)K * T));
    }
    float val = 0.f;
    if (kind & 16) {                               // 7x7 stem, space-to-depth taps (forward pack only)
        const int so = stem_src(t, c, C);
        if (so >= 0) {
            const float4 ws = reinterpret_cast<const float4*>(L[SN_WS_STATS])[k];
            val = (W[(int64_t)k * C * 49 + so] - ws.x) * ws.y;
        }
    } else if (c < C) {
        const int64_t src = (kind & 1) ? ((int64_t)c * K + k) * T + t : ((int64_t)k * C + c) * T + t;
        val = W[src] * inv;
        if (kind & 8) {
            const float4 ws = reinterpret_cast<const float4*>(L[SN_WS_STATS])[k];
            val = (W[src] - ws.x) * ws.y;
        }
    }
    int64_t di = idx;
    if (kind & 32) di = which == 0 ? sn_frag_index(k, t, c, T, Cp) : sn_frag_index(c, t, k, T, K);
    if (which == 0) fwd_arena[call * fwd_call_stride + L[SN_FWD_OFF] + di] = ((kind & 64) && !TCVOM_BUILD_F16) ? f2h_ieee(val) : f2h(val);
    else bwd_arena[call * bwd_call_stride + L[SN_BWD_OFF] + di] = f2h(val);
}

// ------------------------------------------------------------------------------ backward
// weight-order element e -> (k, c, t) in 32-bit arithmetic, the tap count a compile-time constant for the usual shapes
__device__ __forceinline__ void sn_split(unsigned e, int kind, int K, int C, int T, int& k, int& c, int& t) {
    unsigned ab;
    if (T == 9) { ab = e / 9u; t = (int)(e - ab * 9u); }
    else if (T == 16) { ab = e >> 4; t = (int)(e & 15u); }
    else if (T == 1) { ab = e; t = 0; }
    else { ab = e / (unsigned)T; t = (int)(e - ab * (unsigned)T); }
    if (kind & 1) { c = (int)(ab / (unsigned)K); k = (int)(ab - (unsigned)c * (unsigned)K); }
    else { k = (int)(ab / (unsigned)C); c = (int)(ab - (unsigned)k * (unsigned)C); }
}

// inner[call][layer] = sum_e dW~_call[e] * W_bar[e].  SN_INNER_BLOCK elements per block: every block ends with ONE atomic on
// its layer's scalar, and same-address atomics serialise at ~0.14 us each (halo weight gradient): with 1024 elements
// per block a 512x512x3x3 layer queued 2304 of them per call and the kernel took 350 us.
constexpr int SN_INNER_BLOCK = 8192;
__global__ __launch_bounds__(256) void sn_bwd_inner_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work,
                                                           const float* __restrict__ dw_arena, int64_t dw_call_stride,
                                                           float* __restrict__ inner, int L_total)
{
    __shared__ float red[4];
    const int layer = work[blockIdx.x * 3], call = work[blockIdx.x * 3 + 1];
    const int64_t base = (int64_t)work[blockIdx.x * 3 + 2] * SN_INNER_BLOCK;
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const float* W = reinterpret_cast<const float*>(L[SN_W]);
    const int kind = (int)L[SN_KIND];
    const int K = (int)L[SN_K], C = (int)L[SN_C], T = (int)L[SN_T], Cp = (int)L[SN_CPAD];
    const int64_t numel = L[SN_NUMEL];
    const float* dw = dw_arena + call * dw_call_stride + L[SN_DW_OFF];
    float a = 0.f;
    if ((kind & 17) == 0) {
        // Conv2d layout [K][C][T]: a thread takes whole (k, c) pairs -- the pairs whose first element lies in this block's range --
        // so that the packed gradient [k][t][c] is read along c (coalesced; the element-order loop below gathers 9 rows per wave
        // instruction: 314 us per step against 258 us in this order) and W_bar as T consecutive floats per thread.  (One block
        // for all calls of a layer, W_bar read once: 303 us -- a third of the blocks, each with three gather streams.)
        const unsigned pend = (unsigned)((min(base + SN_INNER_BLOCK, numel) + T - 1) / T);
        for (unsigned pr = (unsigned)((base + T - 1) / T) + threadIdx.x; pr < pend; pr += 256) {
            const unsigned k = pr / (unsigned)C, c = pr - k * (unsigned)C;
            const float* wp = W + (int64_t)pr * T;
            const float* dp = dw + (int64_t)k * T * Cp + c;
            if (T == 9) {
#pragma unroll
                for (int t = 0; t < 9; ++t) a += dp[t * Cp] * wp[t];
            } else {
                for (int t = 0; t < T; ++t) a += dp[t * Cp] * wp[t];
            }
        }
    } else {
#pragma unroll 8
        for (int q = 0; q < SN_INNER_BLOCK / 256; ++q) {
            const int64_t e = base + q * 256 + threadIdx.x;
            if (e < numel) {
                int k, c, t;
                sn_split((unsigned)e, kind, K, C, T, k, c, t);
                a += dw[((int64_t)k * T + t) * Cp + c] * W[e];
            }
        }
    }
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) atomicAdd(inner + (int64_t)call * L_total + layer, a);
}

constexpr int SN_APPLY_TMAX = 25;
// work rows of sn_bwd_apply_kernel for one layer (the host builds its list from this)
extern "C" int tcvom_sn_apply_blocks(int32_t kind, int32_t K, int32_t C, int32_t T, int64_t numel) {
    if ((kind & 17) == 0 && T <= SN_APPLY_TMAX) return (int)(((int64_t)K * C + 255) / 256);
    return (int)((numel + 255) / 256);
}
// grad[e] = sum_calls dW~[e]/sigma - inner/sigma^2 * u[row] v[col]
// (element order: the pair-major order of sn_bwd_inner_kernel makes the gradient STORES 36-byte strided here -- 213 -> 313 us)
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work,
                                                           const float* __restrict__ dw_arena, int64_t dw_call_stride,
                                                           const float* __restrict__ inner, SnScratch sc,
                                                           const int* __restrict__ ncalls, float* __restrict__ grad_arena, float out_scale,
                                                           const float* __restrict__ dots, const int* __restrict__ dot_layers)
{
    const int layer = work[blockIdx.x * 2];
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const int64_t numel = L[SN_NUMEL];
    const int kind = (int)L[SN_KIND];
    const int K = (int)L[SN_K], C = (int)L[SN_C], T = (int)L[SN_T], Cp = (int)L[SN_CPAD];
    const int wd = (int)L[SN_WD];
    if ((kind & 17) == 0 && T <= SN_APPLY_TMAX) {
        // Conv2d weights [K][C][T] (tcvom_sn_apply_blocks: a block = 256 (k, c) pairs, all T taps): the packed gradients [k][t][c]
        // are read along c (256 consecutive floats per tap and call), the results cross an LDS buffer in element order and leave as
        // 256-float runs.  (One thread per element gathered 7 c x 9 t per wave instruction from 9 rows: 216 us per 1080p step
        // against ~80 us of traffic.)
        __shared__ float buf[256 * SN_APPLY_TMAX];
        const int64_t p0 = (int64_t)work[blockIdx.x * 2 + 1] * 256, npair = (int64_t)K * C;
        const int64_t pr = p0 + threadIdx.x;
        const int nc = ncalls[layer];
        const bool from_dot = dot_layers != nullptr && dot_layers[layer] != 0;
        if (pr < npair) {
            const int k = (int)(pr / C), c = (int)(pr - (int64_t)k * C);
            // per call: 1 / sigma and the rank-one coefficient inner / sigma^2 * u[k] (hoisted: two divisions per element and call
            // were a third of the kernel)
            constexpr int MAXC = 4;
            float isg[MAXC], cf[MAXC];
            const float* vrow[MAXC];
            const float* drow[MAXC];
#pragma unroll
            for (int call = 0; call < MAXC; ++call) {
                isg[call] = 1.f; cf[call] = 0.f; vrow[call] = nullptr; drow[call] = nullptr;
                if (call < nc) {
                    drow[call] = dw_arena + call * dw_call_stride + L[SN_DW_OFF] + (int64_t)k * T * Cp + c;
                    if (!(kind & 2)) {
                        const float sg = sc.sigma[(int64_t)call * sc.L + layer];
                        const float in_ = from_dot ? sg * dots[(int64_t)call * sc.L + layer] : inner[(int64_t)call * sc.L + layer];
                        isg[call] = 1.f / sg;
                        cf[call] = in_ * isg[call] * isg[call] * sc.uhist[(int64_t)call * sc.sum_h + L[SN_S_OFF] + k];
                        vrow[call] = sc.vhist + (int64_t)call * sc.sum_wd + L[SN_T_OFF] + (int64_t)c * T;
                    }
                }
            }
            auto taps = [&](auto tc_) {
                // TC taps at a time, all calls: the loads of a group are issued together (the kernel is bound by the bytes in
                // flight: a loop over single taps left 3 loads per thread outstanding -- 240 us against ~80 us of traffic)
                constexpr int TC = decltype(tc_)::value;
                for (int t0 = 0; t0 + TC <= T; t0 += TC) {
                    float d[MAXC][TC], v[MAXC][TC];
#pragma unroll
                    for (int call = 0; call < MAXC; ++call)
#pragma unroll
                        for (int i = 0; i < TC; ++i) {
                            d[call][i] = call < nc ? drow[call][(int64_t)(t0 + i) * Cp] : 0.f;
                            v[call][i] = (call < nc && vrow[call]) ? vrow[call][t0 + i] : 0.f;
                        }
#pragma unroll
                    for (int i = 0; i < TC; ++i) {
                        float g = 0.f;
#pragma unroll
                        for (int call = 0; call < MAXC; ++call) g += d[call][i] * isg[call] - cf[call] * v[call][i];
                        buf[threadIdx.x * T + t0 + i] = g * out_scale;
                    }
                }
            };
            if (nc <= MAXC) {
                if (T == 9) taps(std::integral_constant<int, 9>{});
                else taps(std::integral_constant<int, 1>{});
            } else {
                for (int t = 0; t < T; ++t) {                            // (more than 4 calls per window: the plain form)
                    float g = 0.f;
                    for (int call = 0; call < nc; ++call) {
                        const float d = dw_arena[call * dw_call_stride + L[SN_DW_OFF] + ((int64_t)k * T + t) * Cp + c];
                        if (kind & 2) { g += d; continue; }
                        const float sg = sc.sigma[(int64_t)call * sc.L + layer];
                        const float in_ = from_dot ? sg * dots[(int64_t)call * sc.L + layer] : inner[(int64_t)call * sc.L + layer];
                        g += d / sg - in_ / (sg * sg) * sc.uhist[(int64_t)call * sc.sum_h + L[SN_S_OFF] + k] *
                                          sc.vhist[(int64_t)call * sc.sum_wd + L[SN_T_OFF] + c * T + t];
                    }
                    buf[threadIdx.x * T + t] = g * out_scale;
                }
            }
        }
        __syncthreads();
        const int64_t e0 = p0 * T, e1 = min(numel, e0 + (int64_t)256 * T);
        float* gout = grad_arena + L[SN_GRAD_OFF];
        for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) gout[e] = buf[e - e0];
        return;
    }
    const int64_t e = (int64_t)work[blockIdx.x * 2 + 1] * 256 + threadIdx.x;
    if (e >= numel) return;
    int k, c, t, row, col;
    if (kind & 16) {                               // stem: T is the packed tap count (16), the parameter is [K][C][7][7]
        row = (int)((unsigned)e / (unsigned)wd); col = (int)((unsigned)e - (unsigned)row * (unsigned)wd);
        k = row; c = t = 0;
    } else {
        sn_split((unsigned)e, kind, K, C, T, k, c, t);
        // the SpectralNorm matrix is weight.view(shape[0], -1): rows are k ([K][C][T]) or, for ConvTranspose, c ([C][K][T])
        if (kind & 1) { row = c; col = k * T + t; } else { row = k; col = c * T + t; }
    }
    int64_t pidx = ((int64_t)k * T + t) * Cp + c;
    if (kind & 16) {                               // [K][C][7][7] element -> its slot in the [K][16][64] packed gradient
        const int v = col % 7, u = (col / 7) % 7, ch = col / 49;
        pidx = (int64_t)row * 16 * 64 + stem_dst(ch, u, v);
    }
    float g = 0.f;
    const int nc = ncalls[layer];
    const bool from_dot = dot_layers != nullptr && dot_layers[layer] != 0;
    for (int call = 0; call < nc; ++call) {
        const float d = dw_arena[call * dw_call_stride + L[SN_DW_OFF] + pidx];
        if (kind & 2) {
            g += d;
        } else {
            const float sg = sc.sigma[(int64_t)call * sc.L + layer];
            const float uu = sc.uhist[(int64_t)call * sc.sum_h + L[SN_S_OFF] + row];
            const float vv = sc.vhist[(int64_t)call * sc.sum_wd + L[SN_T_OFF] + col];
            // <dW~, W_bar>: summed by sn_bwd_inner_kernel, or sigma <dy, y> delivered by the layer's BatchNorm backward (tcvom_sn_dot)
            const float in_ = from_dot ? sg * dots[(int64_t)call * sc.L + layer] : inner[(int64_t)call * sc.L + layer];
            g += d / sg - in_ / (sg * sg) * uu * vv;
        }
    }
    grad_arena[L[SN_GRAD_OFF] + e] = g * out_scale;           // (1 / loss scale of the fp16 build; ws_backward is linear in it)
}

// ------------------------------------------------------------------------------ weight standardisation
// one block per (layer, output channel): mean, unbiased variance over the wd = C*R*S elements of the row
__global__ __launch_bounds__(256) void ws_stats_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work) {
    __shared__ float red[4];
    const int layer = work[blockIdx.x * 2], row = work[blockIdx.x * 2 + 1];
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const int wd = (int)L[SN_WD];
    const float* w = reinterpret_cast<const float*>(L[SN_W]) + (int64_t)row * wd;
    float a = 0.f;
    for (int i = threadIdx.x; i < wd; i += 256) a += w[i];
    const float mean = block_sum_256(a, red) / (float)wd;
    __syncthreads();
    float b = 0.f;
    for (int i = threadIdx.x; i < wd; i += 256) { const float d = w[i] - mean; b += d * d; }
    const float var = block_sum_256(b, red) / (float)(wd - 1);
    if (threadIdx.x == 0) {
        const float sd = sqrtf(var + 1e-12f);
        reinterpret_cast<float4*>(L[SN_WS_STATS])[row] = make_float4(mean, 1.f / (sd + 1e-5f), sd, 0.f);
    }
}

// grad (w.r.t. the standardised weight, already in the parameter's layout) -> grad w.r.t. the raw weight, in place:
//   w^ = (w - mu) / s,  s = sd + 1e-5,  sd = sqrt(var_unbiased + 1e-12)
//   dw_i = (g_i - mean(g)) / s  -  (w_i - mu) * <g, w - mu> / (s^2 * sd * (n - 1))
__global__ __launch_bounds__(256) void ws_backward_kernel(const int64_t* __restrict__ tab, const int* __restrict__ work,
                                                          float* __restrict__ grad_arena) {
    __shared__ float red[4];
    const int layer = work[blockIdx.x * 2], row = work[blockIdx.x * 2 + 1];
    const int64_t* L = tab + (int64_t)layer * SN_WORDS;
    const int wd = (int)L[SN_WD];
    const float* w = reinterpret_cast<const float*>(L[SN_W]) + (int64_t)row * wd;
    float* g = grad_arena + L[SN_GRAD_OFF] + (int64_t)row * wd;
    const float4 ws = reinterpret_cast<const float4*>(L[SN_WS_STATS])[row];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < wd; i += 256) { a += g[i]; b += g[i] * (w[i] - ws.x); }
    const float gsum = block_sum_256(a, red);
    __syncthreads();
    const float gdot = block_sum_256(b, red);
    const float gmean = gsum / (float)wd;
    const float k2 = gdot * ws.y * ws.y / (ws.z * (float)(wd - 1));
    for (int i = threadIdx.x; i < wd; i += 256) g[i] = (g[i] - gmean) * ws.y - (w[i] - ws.x) * k2;
}

extern "C" int tcvom_ws_stats(const int64_t* table, const int32_t* work_rows, int32_t n_rows, void* stream) {
    TCVOM_CHECK_ARG(table && work_rows && n_rows > 0, "ws_stats: bad args");
    hipLaunchKernelGGL(ws_stats_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, table, work_rows);
    TCVOM_LAUNCH_CHECK("ws_stats");
    return TCVOM_OK;
}

extern "C" int tcvom_ws_backward(const int64_t* table, const int32_t* work_rows, int32_t n_rows, float* grad_arena, void* stream) {
    TCVOM_CHECK_ARG(table && work_rows && n_rows > 0 && grad_arena, "ws_backward: bad args");
    hipLaunchKernelGGL(ws_backward_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, table, work_rows, grad_arena);
    TCVOM_LAUNCH_CHECK("ws_backward");
    return TCVOM_OK;
}

static SnScratch mk_scratch(const tcvom_sn_scratch* s) {
    SnScratch sc;
    sc.tvec = s->tvec; sc.svec = s->svec; sc.sigma = s->sigma; sc.uhist = s->uhist; sc.vhist = s->vhist;
    sc.sum_h = s->sum_h; sc.sum_wd = s->sum_wd; sc.L = s->num_layers;
    return sc;
}

extern "C" int tcvom_sn_power_iteration(const int64_t* table, const tcvom_sn_scratch* s,
                                        const int32_t* work_wtu, int32_t n_wtu, const int32_t* work_wv, int32_t n_wv,
                                        const int32_t* sn_layers, int32_t n_sn, int32_t call, int32_t training,
                                        void* stream) {
    TCVOM_CHECK_ARG(table && s && work_wtu && work_wv && sn_layers, "sn_power_iteration: null pointer");
    if (n_sn <= 0) return TCVOM_OK;
    hipStream_t st = (hipStream_t)stream;
    SnScratch sc = mk_scratch(s);
    // (tvec is zero here: allocated so, and the finalize kernel of every training call leaves the slices it read zero)
    if (training) hipLaunchKernelGGL(sn_wt_u_kernel, dim3(n_wtu), dim3(256), 0, st, table, work_wtu, sc.tvec);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(n_wv), dim3(256), 0, st, table, work_wv, sc.tvec, sc.svec, training ? 0 : 1);
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(n_sn), dim3(256), 0, st, table, sn_layers, sc, call, training ? 0 : 1);
    TCVOM_LAUNCH_CHECK("sn_power_iteration");
    return TCVOM_OK;
}

extern "C" int tcvom_sn_pack(const int64_t* table, const tcvom_sn_scratch* s, const int32_t* work_pack, int32_t n_pack,
                             int32_t call, void* fwd_arena, void* bwd_arena, int64_t fwd_call_stride,
                             int64_t bwd_call_stride, void* stream) {
    TCVOM_CHECK_ARG(table && s && work_pack && fwd_arena && n_pack > 0, "sn_pack: bad args");
    hipLaunchKernelGGL(sn_pack_kernel, dim3(n_pack), dim3(256), 0, (hipStream_t)stream, table, work_pack, s->sigma,
                       s->num_layers, call, (h16raw*)fwd_arena, (h16raw*)bwd_arena, fwd_call_stride, bwd_call_stride);
    TCVOM_LAUNCH_CHECK("sn_pack");
    return TCVOM_OK;
}

extern "C" int tcvom_sn_backward(const int64_t* table, const tcvom_sn_scratch* s,
                                 const int32_t* work_inner, int32_t n_inner, const int32_t* work_apply, int32_t n_apply,
                                 const int32_t* ncalls, const float* dw_arena, int64_t dw_call_stride,
                                 float* inner, int32_t max_calls, float* grad_arena, float out_scale,
                                 const float* dots, const int32_t* dot_layers, void* stream) {
    TCVOM_CHECK_ARG(table && s && work_apply && ncalls && dw_arena && inner && grad_arena, "sn_backward: null pointer");
    TCVOM_CHECK_ARG((dots == nullptr) == (dot_layers == nullptr), "sn_backward: dots and dot_layers come together");
    hipStream_t st = (hipStream_t)stream;
    SnScratch sc = mk_scratch(s);
    if (hipMemsetAsync(inner, 0, sizeof(float) * (size_t)max_calls * sc.L, st) != hipSuccess)
        return tcvom_fail(TCVOM_ERR_LAUNCH, "sn_backward: memset failed");
    if (n_inner > 0)
        hipLaunchKernelGGL(sn_bwd_inner_kernel, dim3(n_inner), dim3(256), 0, st, table, work_inner, dw_arena,
                           dw_call_stride, inner, sc.L);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(n_apply), dim3(256), 0, st, table, work_apply, dw_arena, dw_call_stride,
                       inner, sc, ncalls, grad_arena, out_scale, dots, dot_layers);
    TCVOM_LAUNCH_CHECK("sn_backward");
    return TCVOM_OK;
}
