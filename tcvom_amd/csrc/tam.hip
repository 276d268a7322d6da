// Temporal Attention Module core — fused masked local-window cross-frame attention.
// Replaces FeatureAggregationModule._attention (models/VMN/VMN_model.py:24-61): the python loop
// over the batch, F.unfold(k) [C, w*w, N] (819 MB per direction at 1080p), torch.nonzero, gather,
// mul-sum, softmax, mul-sum and the two scatters, for BOTH directions, plus `v + xb + xf` (:68).
//
//   for every os8 pixel u with mask[u] != 0:
//       logit[j] = <q_u, k_{u+d_j}> / sqrt(C)   over the w*w zero-padded neighbours (OOB: logit 0)
//       p        = softmax_j(logit)
//       out_u    = v_u + sum_j p_j kb_{u+d_j} + sum_j p'_j kf_{u+d_j}      (values are the keys)
//   known pixels: out_u = v_u, logits 0.
//
// window 7 (the configuration of every entry script): LDS-tiled kernels, see tam_tiled_kernel below -- tiles without unknown
// pixels exit at once, key halos of both directions + queries staged in LDS, one wave per unknown pixel with one lane per
// neighbour for the logits (v_dot2_f32_bf16) and one lane per channel pair for the aggregation.
// Other windows (1, 3, 5): one wave (64 lanes) per unknown pixel taken from a compacted work list; a lane owns channel pairs
// {lane + 64*i}; the w*w neighbour key slices stay in registers between the logit and the aggregation pass.
// HBM-bound: algorithmic traffic = read q, v, kb, kf + write out + 2 logit maps.
#include <cstdlib>
#include "common.h"

template <int WIN, int NP>
__global__ __launch_bounds__(256) void tam_fwd_kernel(
    const unsigned* __restrict__ q, const unsigned* __restrict__ kb, const unsigned* __restrict__ kf,
    const unsigned* __restrict__ v, const unsigned char* __restrict__ mask,
    unsigned* __restrict__ out, float* __restrict__ attb, float* __restrict__ attf, const int* __restrict__ worklist,
    int B, int H, int W, int C, float inv_sqrt_c)
{
    constexpr int W2 = WIN * WIN, R = WIN / 2;
    const int lane = threadIdx.x & 63;
    const int64_t N = (int64_t)H * W;
    const int CP = C / 2;                       // channel pairs per pixel
    // one wave per UNKNOWN pixel, taken from the compacted work list (known pixels: out = v and zero logits, done by the
    // launcher with a copy and a memset).  The kernel keeps 49 key slices + logits + probabilities in registers
    // (256 VGPRs, one wave per SIMD), so it must not be launched over the >95 % known pixels of a typical trimap.
    const int nwork = worklist[0];
    for (int wi = blockIdx.x * 4 + (threadIdx.x >> 6); wi < nwork; wi += gridDim.x * 4) {
    const int64_t pix = worklist[1 + wi];
    const int b = (int)(pix / N);
    const int64_t u = pix % N;
    const int y = (int)(u / W), x = (int)(u % W);
    const bool unknown = true;

    float o[NP][2];
    unsigned qq[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        const unsigned vv = cp < CP ? v[pix * CP + cp] : 0u;
        o[i][0] = hlo(vv);
        o[i][1] = hhi(vv);
        qq[i] = (cp < CP && unknown) ? q[pix * CP + cp] : 0u;
    }
#pragma unroll 1
    for (int dir = 0; dir < 2; ++dir) {
        const unsigned* __restrict__ k = dir == 0 ? kb : kf;
        float* __restrict__ att = dir == 0 ? attb : attf;
        if (!unknown) continue;                 // the logit maps are zero-filled by the launcher (one memset instead of
                                                // 2 x 49 scattered 4-byte stores per known pixel)
        unsigned kk[W2][NP];
        float lg[W2];
#pragma unroll
        for (int j = 0; j < W2; ++j) {
            const int yy = y + j / WIN - R, xx = x + j % WIN - R;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int cp = lane + 64 * i;
                const unsigned w = (ok && cp < CP) ? k[(((int64_t)b * H + yy) * W + xx) * CP + cp] : 0u;
                kk[j][i] = w;
                part += hlo(qq[i]) * hlo(w) + hhi(qq[i]) * hhi(w);
            }
            lg[j] = wave_sum(part) * inv_sqrt_c;
        }
        float mx = lg[0];
#pragma unroll
        for (int j = 1; j < W2; ++j) mx = fmaxf(mx, lg[j]);
        float den = 0.f;
        float pj[W2];
#pragma unroll
        for (int j = 0; j < W2; ++j) { pj[j] = __expf(lg[j] - mx); den += pj[j]; }
        const float rden = 1.f / den;
#pragma unroll
        for (int j = 0; j < W2; ++j) {
            const float p = pj[j] * rden;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                o[i][0] += p * hlo(kk[j][i]);
                o[i][1] += p * hhi(kk[j][i]);
            }
        }
        // every lane holds all logits: lane j stores logit j (strided by N in memory)
#pragma unroll
        for (int j = 0; j < W2; ++j)
            if (lane == (j & 63)) att[((int64_t)b * W2 + j) * N + u] = lg[j];
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        if (cp < CP) out[pix * CP + cp] = pack2h(o[i][0], o[i][1]);
    }
    }
}

// worklist[0] = number of unknown pixels, worklist[1..] = their flat indices (any order); worklist[0] zeroed by the launcher
__global__ void tam_compact_kernel(const unsigned char* __restrict__ mask, int* __restrict__ worklist, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i] != 0) worklist[1 + atomicAdd(worklist, 1)] = (int)i;
}

// Backward pass A (per query pixel): recompute p; dp_j = <dout, k_j>; ds = p*(dp - sum p dp) + datt;
// dq = sum_j ds_j k_j / sqrt(C).  Stores p and ds ([B][2][W2][N] fp32) for pass B.
template <int WIN, int NP>
__global__ __launch_bounds__(256) void tam_bwd_query_kernel(
    const unsigned* __restrict__ q, const unsigned* __restrict__ kb, const unsigned* __restrict__ kf,
    const unsigned char* __restrict__ mask, const unsigned* __restrict__ dout,
    const float* __restrict__ dattb, const float* __restrict__ dattf,
    unsigned* __restrict__ dq, float* __restrict__ pbuf, float* __restrict__ dsbuf, const int* __restrict__ worklist,
    int B, int H, int W, int C, float inv_sqrt_c)
{
    constexpr int W2 = WIN * WIN, R = WIN / 2;
    const int lane = threadIdx.x & 63;
    const int64_t N = (int64_t)H * W;
    const int CP = C / 2;
    const int nwork = worklist[0];                 // unknown pixels only (dq of the others is zero-filled by the launcher)
    for (int wi = blockIdx.x * 4 + (threadIdx.x >> 6); wi < nwork; wi += gridDim.x * 4) {
    const int64_t pix = worklist[1 + wi];
    const int b = (int)(pix / N);
    const int64_t u = pix % N;
    const int y = (int)(u / W), x = (int)(u % W);
    const bool unknown = true;
    float dqa[NP][2];
    unsigned qq[NP], go[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        dqa[i][0] = dqa[i][1] = 0.f;
        qq[i] = (cp < CP && unknown) ? q[pix * CP + cp] : 0u;
        go[i] = (cp < CP && unknown) ? dout[pix * CP + cp] : 0u;
    }
#pragma unroll 1
    for (int dir = 0; dir < 2; ++dir) {
        const unsigned* __restrict__ k = dir == 0 ? kb : kf;
        const float* __restrict__ datt = dir == 0 ? dattb : dattf;
        float* __restrict__ pb = pbuf + ((int64_t)(b * 2 + dir) * W2) * N;
        float* __restrict__ db = dsbuf + ((int64_t)(b * 2 + dir) * W2) * N;
        if (!unknown) continue;                 // pass B reads p / ds of unknown query pixels only
        unsigned kk[W2][NP];
        float lg[W2], dp[W2];
#pragma unroll
        for (int j = 0; j < W2; ++j) {
            const int yy = y + j / WIN - R, xx = x + j % WIN - R;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            float part = 0.f, part2 = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int cp = lane + 64 * i;
                const unsigned w = (ok && cp < CP) ? k[(((int64_t)b * H + yy) * W + xx) * CP + cp] : 0u;
                kk[j][i] = w;
                part += hlo(qq[i]) * hlo(w) + hhi(qq[i]) * hhi(w);
                part2 += hlo(go[i]) * hlo(w) + hhi(go[i]) * hhi(w);
            }
            lg[j] = wave_sum(part) * inv_sqrt_c;
            dp[j] = wave_sum(part2);
        }
        float mx = lg[0];
#pragma unroll
        for (int j = 1; j < W2; ++j) mx = fmaxf(mx, lg[j]);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < W2; ++j) { lg[j] = __expf(lg[j] - mx); den += lg[j]; }
        const float rden = 1.f / den;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < W2; ++j) { lg[j] *= rden; dot += lg[j] * dp[j]; }
#pragma unroll
        for (int j = 0; j < W2; ++j) {
            const float da = datt ? datt[((int64_t)b * W2 + j) * N + u] : 0.f;   // uniform address: broadcast load
            const float ds = lg[j] * (dp[j] - dot) + da;
            const float dss = ds * inv_sqrt_c;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                dqa[i][0] += dss * hlo(kk[j][i]);
                dqa[i][1] += dss * hhi(kk[j][i]);
            }
            if (lane == (j & 63)) { pb[(int64_t)j * N + u] = lg[j]; db[(int64_t)j * N + u] = dss; }
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        if (cp < CP) dq[pix * CP + cp] = pack2h(dqa[i][0], dqa[i][1]);
    }
    }
}

// (Round 4, measured and dropped: pass B on 8 x 8 key-pixel tiles with the dout / q halos staged in LDS once for both directions --
//  all-unknown 195 -> 147 us, but 96 -> 112 us on the bench window (a 5-pixel band), with the hits walked four at a time as well: the
//  [b][dir][j][u] layout of p / ds makes lane j's fetch 49 cache lines per wave either way; a tile-major layout written by pass A
//  would be the next step.)
// Backward pass B (per key pixel v, one direction per blockIdx.y): gather form of the scatter
//   dk_v = sum_j [u = v - d_j in image, unknown]  p_j(u) * dout_u + ds_j(u)/sqrt(C) * q_u
template <int WIN, int NP>
__global__ __launch_bounds__(256) void tam_bwd_key_kernel(
    const unsigned* __restrict__ q, const unsigned char* __restrict__ mask, const unsigned* __restrict__ dout,
    const float* __restrict__ pbuf, const float* __restrict__ dsbuf,
    unsigned* __restrict__ dkb, unsigned* __restrict__ dkf, int B, int H, int W, int C)
{
    constexpr int W2 = WIN * WIN, R = WIN / 2;
    const int lane = threadIdx.x & 63;
    const int dir = blockIdx.y;
    const int64_t N = (int64_t)H * W;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= B * N) return;
    const int b = (int)(pix / N);
    const int64_t vv = pix % N;
    const int y = (int)(vv / W), x = (int)(vv % W);
    const int CP = C / 2;
    const float* __restrict__ pb = pbuf + ((int64_t)(b * 2 + dir) * W2) * N;
    const float* __restrict__ db = dsbuf + ((int64_t)(b * 2 + dir) * W2) * N;
    float acc[NP][2];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i][0] = acc[i][1] = 0.f;
    // which of the W2 neighbours are unknown pixels: lane j tests neighbour j, one ballot (a serial loop of W2 dependent mask loads
    // per wave -- all of them misses for the 97 % of the key pixels that have no unknown neighbour -- was most of the kernel: 195 us)
    bool hit = false;
    if (lane < W2) {
        const int uy = y - (lane / WIN - R), ux = x - (lane % WIN - R);
        if (uy >= 0 && uy < H && ux >= 0 && ux < W) hit = mask[(int64_t)b * N + (int64_t)uy * W + ux] != 0;
    }
    unsigned long long bits = __ballot(hit);
    while (bits) {
        const int j = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const int uy = y - (j / WIN - R), ux = x - (j % WIN - R);
        const int64_t u = (int64_t)uy * W + ux;
        const float p = pb[(int64_t)j * N + u], ds = db[(int64_t)j * N + u];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int cp = lane + 64 * i;
            if (cp < CP) {
                const unsigned g = dout[((int64_t)b * N + u) * CP + cp];
                const unsigned qv = q[((int64_t)b * N + u) * CP + cp];
                acc[i][0] += p * hlo(g) + ds * hlo(qv);
                acc[i][1] += p * hhi(g) + ds * hhi(qv);
            }
        }
    }
    unsigned* __restrict__ dk = dir == 0 ? dkb : dkf;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        if (cp < CP) dk[pix * CP + cp] = pack2h(acc[i][0], acc[i][1]);
    }
}

// ------------------------------------------------------------------------------------------ LDS-tiled kernels (C <= 128)
// A workgroup owns an 8 x 8 pixel tile; tiles without unknown pixels exit at once.  The (8+6) x (8+6) key halos of BOTH
// directions and the query tile are staged in LDS (zero rows outside the image).  One wave per unknown pixel:
//   phase A  lane j (< 49) owns neighbour j: logit_j = <q, k_j> with v_dot2_f32_bf16 over 16-byte chunks; lane j starts at
//            chunk j so that the 16 lanes of an LDS read group hit 16 different banks although all key rows start on bank 0;
//            the softmax is then TWO wave reductions (the one-wave-per-pixel kernel above needs 49, one per neighbour)
//   phase B  lane = channel pair: out = v + sum_j p_j k_j, p_j broadcast with v_readlane, key rows read conflict-free
typedef __attribute__((ext_vector_type(2))) act_t tam_h2;
__device__ __forceinline__ float tam_dot8(const uint4 a, const uint4 b, float acc) {
    acc = dot2_h16(__builtin_bit_cast(tam_h2, a.x), __builtin_bit_cast(tam_h2, b.x), acc, false);
    acc = dot2_h16(__builtin_bit_cast(tam_h2, a.y), __builtin_bit_cast(tam_h2, b.y), acc, false);
    acc = dot2_h16(__builtin_bit_cast(tam_h2, a.z), __builtin_bit_cast(tam_h2, b.z), acc, false);
    return dot2_h16(__builtin_bit_cast(tam_h2, a.w), __builtin_bit_cast(tam_h2, b.w), acc, false);
}
__device__ __forceinline__ float tam_lane(float v, int j) {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), j));
}

// MODE 0: forward (out, logits).  MODE 1: backward pass A (dq, p and ds / sqrt(C) for pass B); `g` = dout, datt* may be NULL.
// TH x TW = pixel tile, C8MAX = 16-byte chunks per pixel the LDS arrays are sized for (16: C <= 128, tiles 8x8;
// 32: C <= 256 as in the FBA / DIM bases, tiles 4x8 forward and 4x4 backward so that everything stays below 160 KiB).
// TAM_NW waves per workgroup: a tile's unknown pixels are spread over them one pixel per wave, and the LDS footprint allows ONE
// workgroup per CU -- with 4 waves (one per SIMD, nothing to hide the LDS latency of the dependent chains) a fully unknown 8 x 8
// tile took 16 pixels x 7.7 us per wave, and the densest tile sets the launch time (157 us on a 3 %-unknown window).
constexpr int TAM_NW = 16;
template <int WIN, int MODE, int TH, int TW, int C8MAX>
__global__ __launch_bounds__(TAM_NW * 64) void tam_tiled_kernel(
    const uint4* __restrict__ q, const uint4* __restrict__ kb, const uint4* __restrict__ kf, const unsigned* __restrict__ v,
    const uint4* __restrict__ g, const unsigned char* __restrict__ mask, unsigned* __restrict__ out,
    float* __restrict__ att0, float* __restrict__ att1, const float* __restrict__ datt0, const float* __restrict__ datt1,
    int H, int W, int C, float inv_sqrt_c, int cnt_hi)
{
    constexpr int W2 = WIN * WIN, R = WIN / 2, HH = TH + 2 * R, HW = TW + 2 * R, NT = TH * TW, NP = C8MAX / 16;
    __shared__ uint4 halo[2 * HH * HW * C8MAX];        // both directions
    __shared__ uint4 qt[NT * C8MAX];
    __shared__ uint4 gt[MODE == 1 ? NT * C8MAX : 1];
    __shared__ int list[NT];
    __shared__ int cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C8 = C >> 3, CP = C >> 1;
    const int b = blockIdx.z, ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const int64_t N = (int64_t)H * W;
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (tid < NT) {
        const int y = ty0 + tid / TW, x = tx0 + tid % TW;
        if (y < H && x < W && mask[b * N + (int64_t)y * W + x] != 0) list[atomicAdd(&cnt, 1)] = tid;
    }
    __syncthreads();
    const int nu = cnt;
    if (nu == 0 || nu >= cnt_hi) return;                    // (denser tiles: tam_mfma_kernel)
    for (int idx = tid; idx < 2 * HH * HW * C8; idx += TAM_NW * 64) {
        const int d = idx / (HH * HW * C8), rem = idx - d * (HH * HW * C8);
        const int r = rem / C8, c = rem - r * C8;
        const int y = ty0 + r / HW - R, x = tx0 + r % HW - R;
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if (y >= 0 && y < H && x >= 0 && x < W) val = (d == 0 ? kb : kf)[(b * N + (int64_t)y * W + x) * C8 + c];
        halo[(d * HH * HW + r) * C8 + c] = val;
    }
    for (int idx = tid; idx < NT * C8; idx += TAM_NW * 64) {
        const int p = idx / C8, c = idx - p * C8;
        const int y = ty0 + p / TW, x = tx0 + p % TW;
        const bool in = y < H && x < W;
        qt[idx] = in ? q[(b * N + (int64_t)y * W + x) * C8 + c] : make_uint4(0u, 0u, 0u, 0u);
        if (MODE == 1) gt[idx] = in ? g[(b * N + (int64_t)y * W + x) * C8 + c] : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int j = lane < W2 ? lane : W2 - 1;            // idle lanes shadow the last neighbour
    for (int li = wave; li < nu; li += TAM_NW) {
        const int t = list[li];
        const int py = t / TW, px = t % TW;
        const int64_t u = (int64_t)(ty0 + py) * W + tx0 + px;
        const int64_t pix = b * N + u;
        float o[NP][2];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            o[i][0] = o[i][1] = 0.f;
            const int cp = lane + 64 * i;
            if (MODE == 0 && cp < CP) { const unsigned vv = v[pix * CP + cp]; o[i][0] = hlo(vv); o[i][1] = hhi(vv); }
        }
#pragma unroll 1
        for (int dir = 0; dir < 2; ++dir) {
            const uint4* hl = halo + dir * HH * HW * C8;
            const int rj = ((py + j / WIN) * HW + px + j % WIN) * C8;
            float lg = 0.f, dp = 0.f;
            for (int sidx = 0; sidx < C8; ++sidx) {
                int c = sidx + lane;
                c = c >= C8 ? c % C8 : c;
                const uint4 kv = hl[rj + c];
                lg = tam_dot8(qt[t * C8 + c], kv, lg);
                if (MODE == 1) dp = tam_dot8(gt[t * C8 + c], kv, dp);
            }
            lg *= inv_sqrt_c;
            const float mx = wave_max(lane < W2 ? lg : -3.0e38f);
            const float e = lane < W2 ? __expf(lg - mx) : 0.f;
            const float p = e / wave_sum(e);
            float wgt;                                  // weight of neighbour j in phase B
            if (MODE == 0) {
                wgt = p;
                if (lane < W2) (dir == 0 ? att0 : att1)[((int64_t)b * W2 + lane) * N + u] = lg;
            } else {
                const float dot = wave_sum(lane < W2 ? p * dp : 0.f);
                const float* datt = dir == 0 ? datt0 : datt1;
                const float da = (datt && lane < W2) ? datt[((int64_t)b * W2 + lane) * N + u] : 0.f;
                wgt = (p * (dp - dot) + da) * inv_sqrt_c;
                if (lane < W2) {                        // p and ds / sqrt(C) for pass B: [b][dir][j][u]
                    att0[(((int64_t)b * 2 + dir) * W2 + lane) * N + u] = p;
                    att1[(((int64_t)b * 2 + dir) * W2 + lane) * N + u] = wgt;
                }
            }
            const unsigned* hp = reinterpret_cast<const unsigned*>(hl) + (py * HW + px) * CP;
#pragma unroll
            for (int jj = 0; jj < W2; ++jj) {
                const float pj = tam_lane(wgt, jj);
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int cp = lane + 64 * i;
                    const unsigned kv = hp[((jj / WIN) * HW + jj % WIN) * CP + (cp < CP ? cp : 0)];
                    o[i][0] += pj * hlo(kv);
                    o[i][1] += pj * hhi(kv);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int cp = lane + 64 * i;
            if (cp < CP) out[pix * CP + cp] = pack2h(o[i][0], o[i][1]);
        }
    }
}

// ------------------------------------------------------------------------------------------ MFMA tile kernel (window 7, C = 128)
// The same 8 x 8 pixel tile with its 14 x 14 key halos in LDS, but the two contractions of a tile are DENSE products on the matrix
// cores instead of one wave per pixel on the vector ALUs:
//     S^T [224 keys][64 queries] = K_halo [224][128] * Q_tile^T        (196 halo keys padded to 7 blocks of 32; 56 MFMA 32x32x16)
//     O^T [128 ch][64 queries]  += K_halo^T [128][224] * P [224][64]   (P = softmax of the 49 in-window entries of a column, 0 elsewhere)
// 4x the useful flops (a query needs 49 of the 196 columns), at ~30x the rate.  Wave = (query block of 32, direction).  A lane of
// the S^T accumulators holds ONE query (lane & 31) and 16 key rows per block, so the softmax is a reduction over its own registers
// plus ONE exchange with the partner lane (lane ^ 32).  P feeds the second product straight from those registers as the B operand:
// the reduction index of a k-step is taken in the accumulator's own row order (rows 16 t + 4 h + r and 16 t + 8 + 4 h + r of a block
// for lane half h), and the A operand -- K^T, channels x keys -- is read with that row order through the transposing LDS read
// (ds_read_b64_tr_b16: 4 key rows per read), so no lane exchange and no LDS round trip of P.
// LDS rows are 256 bytes (128 channels); the 16-byte chunk c of row r sits at c ^ 4 (r & 3) ^ ((r >> 2) & 3): the 32 rows of a
// ds_read_b128 fragment read (GEMM 1) fall on 16 distinct bank groups, and the 4 rows of a transposing read on distinct 64-byte groups.
// MODE 0: forward (out = v + both directions, logits).  MODE 1: backward pass A (dq, p and ds / sqrt(C) for pass B): a third product
// dP^T = K_halo * dO^T of the first kind, the softmax backward in registers, dq^T = K_halo^T * dS.
__device__ __forceinline__ int tam_swz(int c, int r) { return c ^ ((r & 3) << 2) ^ ((r >> 2) & 3); }
template <int MODE>
__global__ __launch_bounds__(256) void tam_mfma_kernel(
    const uint4* __restrict__ q, const uint4* __restrict__ kb, const uint4* __restrict__ kf, const unsigned* __restrict__ v,
    const uint4* __restrict__ g, const unsigned char* __restrict__ mask, unsigned* __restrict__ out,
    float* __restrict__ att0, float* __restrict__ att1, const float* __restrict__ datt0, const float* __restrict__ datt1,
    int H, int W, float inv_sqrt_c, int cnt_lo, int self_init)
{
    // self_init != 0 (every active tile runs here, cnt_lo == 1): the kernel writes EVERY pixel of its tile -- known pixels get out = v and
    // zero logits (forward) / dq = 0 (backward) -- so the launcher needs no memsets of the logit maps / dq and no copy of v
    // (three ~5 us fill / copy launches per call: a fifth of the all-unknown forward)
    constexpr int WIN = 7, W2 = 49, R = 3, TH = 8, TW = 8, HW = 14, NKEY = 196, NK = 224, C8 = 16, NT = 64;
    extern __shared__ __attribute__((aligned(16))) uint4 tam_lds[];
    uint4* halo = tam_lds;                                  // [2][224][16]
    uint4* qt = halo + 2 * NK * C8;                         // [64][16]
    uint4* gt = qt + NT * C8;                               // [64][16] (MODE 1)
    // (the query / gradient tiles are dead after the first products: the 2 x 16 KB exchange area of the epilogue lies over them; the
    //  forward kernel has no gradient tile, its second half is extra)
    float* comb = reinterpret_cast<float*>(qt);             // [2 query blocks][4][16][64] fp32
    __shared__ unsigned char unk[NT];
    __shared__ int cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const int64_t N = (int64_t)H * W;
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (tid < NT) {
        const int y = ty0 + tid / TW, x = tx0 + tid % TW;
        const bool u = y < H && x < W && mask[b * N + (int64_t)y * W + x] != 0;
        unk[tid] = u ? 1 : 0;
        if (u) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (cnt < cnt_lo) {                                     // (sparser tiles: the one-wave-per-pixel kernel, see the launchers)
        if (self_init && cnt == 0) {
            // a tile of known pixels: out = v (forward) or dq = 0 (backward), zero logits
            for (int idx = tid; idx < NT * C8; idx += 256) {
                const int p = idx >> 4, c = idx & 15;
                const int y = ty0 + p / TW, x = tx0 + p % TW;
                if (y < H && x < W) {
                    const int64_t o = (b * N + (int64_t)y * W + x) * C8 + c;
                    reinterpret_cast<uint4*>(out)[o] = MODE == 0 ? reinterpret_cast<const uint4*>(v)[o] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
            if (MODE == 0) {
                for (int idx = tid; idx < 2 * W2 * NT; idx += 256) {
                    const int p = idx & 63, jd = idx >> 6;              // jd = dir * 49 + j
                    const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
                    if (y < H && x < W) (jd < W2 ? att0 : att1)[((int64_t)b * W2 + (jd < W2 ? jd : jd - W2)) * N + (int64_t)y * W + x] = 0.f;
                }
            }
        }
        return;
    }
    {
        // thread -> (chunk c = tid & 15, halo row r = (tid >> 4) + 16 i): 14 rows of 16 threads per pass and direction, all loads of a
        // thread issued before its stores
        const int c = tid & 15;
        uint4 val[2][14];
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const int hy = r / HW, hx = r - hy * HW;
            const int y = ty0 + hy - R, x = tx0 + hx - R;
            const bool ok = r < NKEY && y >= 0 && y < H && x >= 0 && x < W;
            const int64_t off = (b * N + (int64_t)y * W + x) * C8 + c;
            val[0][i] = ok ? kb[off] : make_uint4(0u, 0u, 0u, 0u);
            val[1][i] = ok ? kf[off] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int r = (tid >> 4) + 16 * i;
            halo[r * C8 + tam_swz(c, r)] = val[0][i];
            halo[(NK + r) * C8 + tam_swz(c, r)] = val[1][i];
        }
    }
    for (int idx = tid; idx < NT * C8; idx += 256) {
        const int p = idx >> 4, c = idx & 15;
        const int y = ty0 + p / TW, x = tx0 + p % TW;
        const bool in = y < H && x < W;
        qt[p * C8 + tam_swz(c, p)] = in ? q[(b * N + (int64_t)y * W + x) * C8 + c] : make_uint4(0u, 0u, 0u, 0u);
        if (MODE == 1) gt[p * C8 + tam_swz(c, p)] = in ? g[(b * N + (int64_t)y * W + x) * C8 + c] : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int qb = wave & 1, dir = wave >> 1;
    const int n = lane & 31, h = lane >> 5;
    const int qi = qb * 32 + n, qy = qi >> 3, qx = qi & 7;
    const bool q_unk = unk[qi] != 0;
    const bool q_in = ty0 + qy < H && tx0 + qx < W;
    const int64_t u = (int64_t)(ty0 + qy) * W + tx0 + qx;       // (valid when q_in)
    const uint4* hl = halo + dir * NK * C8;
    // ---- GEMM 1 (and the dP product of the backward): accumulators [7 key blocks][16 rows 8 g + 4 h + r]
    f32x16_t acc[7], accd[MODE == 1 ? 7 : 1];
#pragma unroll
    for (int k7 = 0; k7 < 7; ++k7)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[k7][r] = 0.f; if (MODE == 1) accd[k7][r] = 0.f; }
    {
        h16x8_t fq[8], fg[MODE == 1 ? 8 : 1];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            fq[ks] = *reinterpret_cast<const h16x8_t*>(qt + qi * C8 + tam_swz(ks * 2 + h, qi));
            if (MODE == 1) fg[ks] = *reinterpret_cast<const h16x8_t*>(gt + qi * C8 + tam_swz(ks * 2 + h, qi));
        }
#pragma unroll
        for (int k7 = 0; k7 < 7; ++k7) {
            const int row = k7 * 32 + n;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const h16x8_t fk = *reinterpret_cast<const h16x8_t*>(hl + row * C8 + tam_swz(ks * 2 + h, row));
                acc[k7] = mfma16(fk, fq[ks], acc[k7], 0, 0, 0);
                if (MODE == 1) accd[k7] = mfma16(fk, fg[ks], accd[k7], 0, 0, 0);
            }
        }
    }
    // ---- softmax over the 49 in-window keys of this lane's query (this lane + its partner lane ^ 32 hold them)
    float mx = -3.0e38f;
#pragma unroll
    for (int k7 = 0; k7 < 7; ++k7)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const int row0 = k7 * 32 + 8 * (r16 >> 2) + (r16 & 3);         // (a constant after unrolling; + 4 for the upper lane half)
            const int ky = h ? (row0 + 4) / HW : row0 / HW, kx = h ? (row0 + 4) % HW : row0 % HW;
            const bool rel = (unsigned)(ky - qy) < (unsigned)WIN && (unsigned)(kx - qx) < (unsigned)WIN;
            const float l = rel ? acc[k7][r16] * inv_sqrt_c : -3.0e38f;
            acc[k7][r16] = l;
            mx = fmaxf(mx, l);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float den = 0.f, dot = 0.f;
    float* attd = dir == 0 ? att0 : att1;
#pragma unroll
    for (int k7 = 0; k7 < 7; ++k7)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const float l = acc[k7][r16];
            const bool rel = l > -1.0e38f;
            if (MODE == 0 && rel && (q_unk || (self_init && q_in))) {
                const int row0 = k7 * 32 + 8 * (r16 >> 2) + (r16 & 3);
                const int ky = h ? (row0 + 4) / HW : row0 / HW, kx = h ? (row0 + 4) % HW : row0 % HW;
                attd[((int64_t)b * W2 + (ky - qy) * WIN + (kx - qx)) * N + u] = q_unk ? l : 0.f;
            }
            const float e = rel ? __expf(l - mx) : 0.f;
            acc[k7][r16] = e;
            den += e;
        }
    den += __shfl_xor(den, 32, 64);
    const float rden = 1.f / den;
    if (MODE == 1) {
#pragma unroll
        for (int k7 = 0; k7 < 7; ++k7)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) dot += acc[k7][r16] * rden * accd[k7][r16];
        dot += __shfl_xor(dot, 32, 64);
    }
    // ---- GEMM 2: O^T / dq^T [4 channel blocks][query]; B fragments from the registers, A = K^T through transposing reads
    f32x16_t acc2[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;
    typedef __attribute__((address_space(3))) const void* tam_lp;
    const unsigned hl_addr = (unsigned)(uintptr_t)(tam_lp)hl;
    const int trow = (lane & 15) >> 2, cig = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const float* dattd = dir == 0 ? datt0 : datt1;
    // backward: the incoming logit gradients of this query's 49 neighbours, all loads in flight at once (inside the product loop they
    // were one exposed L2 round trip per k-step: 14 per wave)
    float dal[MODE == 1 ? 7 : 1][16];
    auto load_datt = [&](int k7) {
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const int row0 = k7 * 32 + 8 * (r16 >> 2) + (r16 & 3);
            const int ky = h ? (row0 + 4) / HW : row0 / HW, kx = h ? (row0 + 4) % HW : row0 % HW;
            const bool rel = (unsigned)(ky - qy) < (unsigned)WIN && (unsigned)(kx - qx) < (unsigned)WIN;
            dal[k7][r16] = (rel && q_unk && dattd) ? dattd[((int64_t)b * W2 + (ky - qy) * WIN + (kx - qx)) * N + u] : 0.f;
        }
    };
    if (MODE == 1) load_datt(0);                            // one key block ahead of its use (16 registers in flight)
#pragma unroll
    for (int k7 = 0; k7 < 7; ++k7) {
        if (MODE == 1) {
            if (k7 + 1 < 7) load_datt(k7 + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float w8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r16 = (2 * t + (i >> 2)) * 4 + (i & 3);
                float p = acc[k7][r16] * rden;
                if (MODE == 1) {
                    const int row0 = k7 * 32 + 8 * (r16 >> 2) + (r16 & 3);
                    const int ky = h ? (row0 + 4) / HW : row0 / HW, kx = h ? (row0 + 4) % HW : row0 % HW;
                    const bool rel = (unsigned)(ky - qy) < (unsigned)WIN && (unsigned)(kx - qx) < (unsigned)WIN;
                    float ds = 0.f;
                    if (rel && q_unk) {
                        ds = (p * (accd[k7][r16] - dot) + dal[k7][r16]) * inv_sqrt_c;
                        if (att1) {
                            const int64_t o2 = (((int64_t)b * 2 + dir) * W2 + (ky - qy) * WIN + (kx - qx)) * N + u;
                            att0[o2] = p;                  // p and ds / sqrt(C) for the vector pass B: [b][dir][j][u] fp32
                            att1[o2] = ds;
                        } else {
                            // pass B on the matrix cores (tam_mfma_key_kernel): one 32-bit word (p, ds / sqrt(C)) in the 16-bit storage
                            // type per (query, neighbour), QUERY-major [b][dir][u][49] -- a key tile gathers the 49-word rows of its
                            // 14 x 14 query halo, a lane's gather stays inside one 196-byte row
                            reinterpret_cast<unsigned*>(att0)[(((int64_t)b * 2 + dir) * N + u) * W2 + (ky - qy) * WIN + (kx - qx)] = pack2h(p, ds);
                        }
                    }
                    p = ds;
                }
                w8[i] = p;
            }
            uint4 pk = make_uint4(pack2h(w8[0], w8[1]), pack2h(w8[2], w8[3]), pack2h(w8[4], w8[5]), pack2h(w8[6], w8[7]));
            const h16x8_t fp = __builtin_bit_cast(h16x8_t, pk);
            const int rlo = k7 * 32 + 16 * t + 4 * h + trow, rhi = rlo + 8;
            TrFrag fa[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const unsigned alo = hl_addr + rlo * 256 + ((((cb ^ (rlo & 3)) << 2) | (cig ^ ((rlo >> 2) & 3))) << 4) + (lane & 1) * 8;
                const unsigned ahi = hl_addr + rhi * 256 + ((((cb ^ (rhi & 3)) << 2) | (cig ^ ((rhi >> 2) & 3))) << 4) + (lane & 1) * 8;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[cb].lo) : "v"(alo));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[cb].hi) : "v"(ahi));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                tr_fence(fa[cb]);
                acc2[cb] = mfma16(tr_value(fa[cb]), fp, acc2[cb], 0, 0, 0);
            }
        }
    }
    // ---- the two directions of a query block meet in LDS; the direction-0 wave adds v (forward) and stores the unknown queries
    __syncthreads();
    float* cm = comb + qb * (4 * 16 * 64);
    if (dir == 1) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) cm[(cb * 16 + r) * 64 + lane] = acc2[cb][r];
    }
    __syncthreads();
    if (dir == 0 && !q_unk && self_init && q_in) {
        // a known pixel inside an active tile: out = v / dq = 0 (lane half h takes channels 64 h .. 64 h + 63)
        const int64_t o = (b * N + u) * C8 + 8 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            reinterpret_cast<uint4*>(out)[o + c] = MODE == 0 ? reinterpret_cast<const uint4*>(v)[o + c] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (dir == 0 && q_unk) {
        const int64_t pix = b * N + u;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = cb * 32 + 8 * gq + 4 * h;             // 4 consecutive channels
                float o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = acc2[cb][gq * 4 + r] + cm[(cb * 16 + gq * 4 + r) * 64 + lane];
                if (MODE == 0) {
                    const uint2 vv = *reinterpret_cast<const uint2*>(v + pix * 64 + (ch >> 1));
                    o4[0] += hlo(vv.x); o4[1] += hhi(vv.x); o4[2] += hlo(vv.y); o4[3] += hhi(vv.y);
                }
                *reinterpret_cast<uint2*>(out + pix * 64 + (ch >> 1)) = make_uint2(pack2h(o4[0], o4[1]), pack2h(o4[2], o4[3]));
            }
    }
}

// ------------------------------------------------------------------------------------------ backward pass B on the matrix cores
// (window 7, C = 128, after tam_mfma_kernel<1> wrote (p, ds / sqrt(C)) query-major).  A workgroup owns an 8 x 8 tile of KEY pixels; the
// queries that attend to them lie in the 14 x 14 halo around it.  Per direction
//     dK^T [128 ch][64 keys] = dO_halo^T [128][224 queries] * P [224][64] + Q_halo^T [128][224] * dS [224][64]
// (P[u][v] = p_j(u) with j the offset v - u when it lies in the window and u is an unknown pixel, 0 otherwise): the mirror image of the
// forward's second product -- there the halo holds keys and P comes out of the accumulators, here the halo holds dO / Q and the B
// fragments are gathered from pass A's rows.  Wave = (key block of 32, direction); A fragments through the transposing LDS read with
// the same row order as the forward (rows 16 s + 4 h + r and 16 s + 8 + 4 h + r of k-step s for lane half h).  k-steps whose 16 halo
// queries are all known are skipped (a band-shaped unknown region leaves most of them out); a tile without an unknown query in its
// halo exits at once (the launcher zero-fills dk).  Replaces the one-wave-per-key-pixel gather kernel: 195 us all-unknown at 136 x 240.
__global__ __launch_bounds__(256) void tam_mfma_key_kernel(
    const uint4* __restrict__ q, const uint4* __restrict__ g, const unsigned char* __restrict__ mask, const unsigned* __restrict__ pd,
    unsigned* __restrict__ dkb, unsigned* __restrict__ dkf, int H, int W)
{
    constexpr int R = 3, TW = 8, HW = 14, NQ = 196, NK = 224, C8 = 16, W2 = 49;
    extern __shared__ __attribute__((aligned(16))) uint4 tam_lds[];
    uint4* gh = tam_lds;                                    // dO halo [224][16], rows of known / outside queries are zero
    uint4* qh = gh + NK * C8;                               // Q halo
    __shared__ __attribute__((aligned(16))) unsigned char um[NK];
    __shared__ int cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, ty0 = blockIdx.y * 8, tx0 = blockIdx.x * 8;
    const int64_t N = (int64_t)H * W;
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (tid < NK) {
        const int hy = (tid * 4682) >> 16, hx = tid - hy * HW;          // tid / 14 for tid < 224
        const int y = ty0 + hy - R, x = tx0 + hx - R;
        const bool u = tid < NQ && y >= 0 && y < H && x >= 0 && x < W && mask[b * N + (int64_t)y * W + x] != 0;
        um[tid] = u ? 1 : 0;
        if (u) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (cnt == 0) {                                         // no query attends to this tile's keys: dk = 0
        for (int idx = tid; idx < 2 * 64 * C8; idx += 256) {
            const int d = idx >> 10, p = (idx >> 4) & 63, c = idx & 15;
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            if (y < H && x < W) reinterpret_cast<uint4*>(d == 0 ? dkb : dkf)[(b * N + (int64_t)y * W + x) * C8 + c] = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    const int kb = wave & 1, dir = wave >> 1;
    const int n = lane & 31, h = lane >> 5;
    const int kq = kb * 32 + n, ky = kq >> 3, kx = kq & 7;
    const unsigned* pdd = pd + ((int64_t)b * 2 + dir) * N * W2;
    const unsigned* um32 = reinterpret_cast<const unsigned*>(um);
    // ---- the (p, ds) words of all 14 k-steps: element i of step s <-> halo query row 16 s + 8 (i >> 2) + 4 h + (i & 3).  Issued before
    // the halos are staged: one exposed memory latency per tile instead of one per k-step
    unsigned w[14][8];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const unsigned mlo = um32[s * 4 + h], mhi = um32[s * 4 + 2 + h];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = s * 16 + 8 * (i >> 2) + 4 * h + (i & 3);
            const int uy = (row * 4682) >> 16, ux = row - uy * HW;
            const int dy = ky + R - uy, dx = kx + R - ux;
            const bool unk = (((i >> 2) ? mhi : mlo) >> (8 * (i & 3))) & 1u;
            const bool ok = unk && (unsigned)(dy + R) < 7u && (unsigned)(dx + R) < 7u;
            const int64_t u = (int64_t)(ty0 - R + uy) * W + (tx0 - R + ux);
            w[s][i] = ok ? pdd[u * W2 + (dy + R) * 7 + (dx + R)] : 0u;
        }
    }
    {
        const int c = tid & 15;
        uint4 vg[14], vq[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const int hy = (r * 4682) >> 16, hx = r - hy * HW;
            const bool ok = um[r] != 0;
            const int64_t off = (b * N + (int64_t)(ty0 + hy - R) * W + (tx0 + hx - R)) * C8 + c;
            vg[i] = ok ? g[off] : make_uint4(0u, 0u, 0u, 0u);
            vq[i] = ok ? q[off] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int r = (tid >> 4) + 16 * i;
            gh[r * C8 + tam_swz(c, r)] = vg[i];
            qh[r * C8 + tam_swz(c, r)] = vq[i];
        }
    }
    __syncthreads();
    // which of the 14 k-steps hold an unknown query (wave-uniform)
    unsigned long long ksm;
    {
        bool any = false;
        if (lane < 14) {
            const uint4 m4 = *reinterpret_cast<const uint4*>(um + lane * 16);
            any = (m4.x | m4.y | m4.z | m4.w) != 0u;
        }
        ksm = __ballot(any);
    }
    f32x16_t acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    typedef __attribute__((address_space(3))) const void* tam_lp;
    const unsigned gh_addr = (unsigned)(uintptr_t)(tam_lp)gh, qh_addr = (unsigned)(uintptr_t)(tam_lp)qh;
    const int trow = (lane & 15) >> 2, cig = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        if (!((ksm >> s) & 1ull)) continue;
        uint4 pp, dd;
        pp.x = (w[s][0] & 0xffffu) | (w[s][1] << 16); pp.y = (w[s][2] & 0xffffu) | (w[s][3] << 16);
        pp.z = (w[s][4] & 0xffffu) | (w[s][5] << 16); pp.w = (w[s][6] & 0xffffu) | (w[s][7] << 16);
        dd.x = (w[s][0] >> 16) | (w[s][1] & 0xffff0000u); dd.y = (w[s][2] >> 16) | (w[s][3] & 0xffff0000u);
        dd.z = (w[s][4] >> 16) | (w[s][5] & 0xffff0000u); dd.w = (w[s][6] >> 16) | (w[s][7] & 0xffff0000u);
        const h16x8_t fp = __builtin_bit_cast(h16x8_t, pp), fd = __builtin_bit_cast(h16x8_t, dd);
        // ---- A fragments: dO^T and Q^T, channels x queries, through the transposing reads
        const int rlo = s * 16 + 4 * h + trow, rhi = rlo + 8;
        TrFrag fg[4], fq[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const unsigned olo = rlo * 256 + ((((cb ^ (rlo & 3)) << 2) | (cig ^ ((rlo >> 2) & 3))) << 4) + (lane & 1) * 8;
            const unsigned ohi = rhi * 256 + ((((cb ^ (rhi & 3)) << 2) | (cig ^ ((rhi >> 2) & 3))) << 4) + (lane & 1) * 8;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fg[cb].lo) : "v"(gh_addr + olo));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fg[cb].hi) : "v"(gh_addr + ohi));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fq[cb].lo) : "v"(qh_addr + olo));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fq[cb].hi) : "v"(qh_addr + ohi));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { tr_fence(fg[cb]); tr_fence(fq[cb]); }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = mfma16(tr_value(fg[cb]), fp, acc[cb], 0, 0, 0);      // (four independent chains)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = mfma16(tr_value(fq[cb]), fd, acc[cb], 0, 0, 0);
    }
    const int y = ty0 + ky, x = tx0 + kx;
    if (y < H && x < W) {
        unsigned* dk = (dir == 0 ? dkb : dkf) + (b * N + (int64_t)y * W + x) * 64;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = cb * 32 + 8 * gq + 4 * h;
                *reinterpret_cast<uint2*>(dk + (ch >> 1)) = make_uint2(pack2h(acc[cb][gq * 4 + 0], acc[cb][gq * 4 + 1]),
                                                                       pack2h(acc[cb][gq * 4 + 2], acc[cb][gq * 4 + 3]));
            }
    }
}

#define TAM_DISPATCH(KERNEL, GRID, ...)                                                                        \
    do {                                                                                                       \
        const int np = C <= 128 ? 1 : 2;                                                                       \
        if (window == 7 && np == 1) hipLaunchKernelGGL((KERNEL<7, 1>), GRID, dim3(256), 0, st, __VA_ARGS__);   \
        else if (window == 7) hipLaunchKernelGGL((KERNEL<7, 2>), GRID, dim3(256), 0, st, __VA_ARGS__);         \
        else if (window == 5 && np == 1) hipLaunchKernelGGL((KERNEL<5, 1>), GRID, dim3(256), 0, st, __VA_ARGS__); \
        else if (window == 5) hipLaunchKernelGGL((KERNEL<5, 2>), GRID, dim3(256), 0, st, __VA_ARGS__);         \
        else if (window == 3 && np == 1) hipLaunchKernelGGL((KERNEL<3, 1>), GRID, dim3(256), 0, st, __VA_ARGS__); \
        else if (window == 3) hipLaunchKernelGGL((KERNEL<3, 2>), GRID, dim3(256), 0, st, __VA_ARGS__);         \
        else if (np == 1) hipLaunchKernelGGL((KERNEL<1, 1>), GRID, dim3(256), 0, st, __VA_ARGS__);             \
        else hipLaunchKernelGGL((KERNEL<1, 2>), GRID, dim3(256), 0, st, __VA_ARGS__);                          \
    } while (0)

static int tam_check(int B, int H, int W, int C, int window) {
    if (B <= 0 || H <= 0 || W <= 0) return tcvom_fail(TCVOM_ERR_ARG, "tam: bad shape");
    if (C % 2 != 0 || C > 256) return tcvom_fail(TCVOM_ERR_ARG, "tam: C=%d must be even and <= 256", C);
    if (!(window == 1 || window == 3 || window == 5 || window == 7)) return tcvom_fail(TCVOM_ERR_ARG, "tam: window=%d", window);
    return TCVOM_OK;
}

// the heavy per-unknown-pixel kernels run on a fixed grid and walk the work list
static int tam_heavy_grid(int64_t pixels) {
    int64_t b = (pixels + 3) / 4;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int tcvom_tam_fwd(const void* q, const void* kb, const void* kf, const void* v, const uint8_t* mask,
                             void* out, float* attb, float* attf, int32_t* worklist, int32_t B, int32_t H, int32_t W,
                             int32_t C, int32_t window, void* stream) {
    TCVOM_CHECK_ARG(q && kb && kf && v && mask && out && attb && attf && worklist, "tam_fwd: null pointer");
    if (int e = tam_check(B, H, W, C, window)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * W;
    TCVOM_CHECK_ARG(n < (1ll << 31), "tam_fwd: too many pixels");
    const float isc = 1.0f / sqrtf((float)C);
    const size_t att_bytes = sizeof(float) * (size_t)n * window * window;
    static const int dense0 = getenv("TCVOM_TAM_DENSE") ? atoi(getenv("TCVOM_TAM_DENSE")) : 1;
    // window 7, C = 128 with every active tile on the MFMA kernel: that kernel writes every pixel itself (known ones: out = v, zero logits)
    const bool self_init = window == 7 && C == 128 && dense0 <= 1;
    if (!self_init &&
        (hipMemsetAsync(attb, 0, att_bytes, st) != hipSuccess || hipMemsetAsync(attf, 0, att_bytes, st) != hipSuccess ||
         hipMemsetAsync(worklist, 0, sizeof(int32_t), st) != hipSuccess ||
         hipMemcpyAsync(out, v, sizeof(h16raw) * (size_t)n * C, hipMemcpyDeviceToDevice, st) != hipSuccess))
        return tcvom_fail(TCVOM_ERR_LAUNCH, "tam_fwd: memset / copy failed");
    // window 7, C = 128: tiles with at least TCVOM_TAM_DENSE unknown pixels go to the MFMA tile kernel (its cost does not depend on the
    // count), sparser ones to the one-wave-per-pixel tile kernel (cost ~ count / 16 waves); each launch skips the other's tiles.
    // Default 1 = every active tile on the MFMA kernel: on the bench window (a 5-pixel band, 8 .. 40 unknown pixels per active tile)
    // forward 58 -> 38 us, all-unknown 104 -> 47 us; the split launches (16 / 24 / 32) measured slower than either kernel alone.
    static const int dense = getenv("TCVOM_TAM_DENSE") ? atoi(getenv("TCVOM_TAM_DENSE")) : 1;      // 65: never the MFMA kernel
    int tile_hi = 1 << 30;
    if (window == 7 && C == 128 && dense <= 64) {
        auto kern = tam_mfma_kernel<0>;
        constexpr size_t lds = (size_t)(2 * 224 * 16 + 2 * 64 * 16) * 16;
        static bool attr = false;
        if (!attr) {
            const hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "tam: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(ea)); }
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(cdiv(W, 8), cdiv(H, 8), B), dim3(256), lds, st, (const uint4*)q, (const uint4*)kb, (const uint4*)kf,
                           (const unsigned*)v, (const uint4*)nullptr, mask, (unsigned*)out, attb, attf, (const float*)nullptr,
                           (const float*)nullptr, H, W, isc, dense < 1 ? 1 : dense, self_init ? 1 : 0);
        tile_hi = dense;
        if (dense <= 1) { TCVOM_LAUNCH_CHECK("tam_fwd"); return TCVOM_OK; }
    }
    if (window == 7 && C % 8 == 0) {
#define TAM_TILED_FWD(TH, TW, CM)                                                                                     \
        hipLaunchKernelGGL((tam_tiled_kernel<7, 0, TH, TW, CM>), dim3(cdiv(W, TW), cdiv(H, TH), B), dim3(TAM_NW * 64), 0, st,   \
                           (const uint4*)q, (const uint4*)kb, (const uint4*)kf, (const unsigned*)v, (const uint4*)nullptr, \
                           mask, (unsigned*)out, attb, attf, (const float*)nullptr, (const float*)nullptr, H, W, C, isc, tile_hi)
        if (C <= 128) TAM_TILED_FWD(8, 8, 16); else TAM_TILED_FWD(4, 8, 32);
#undef TAM_TILED_FWD
        TCVOM_LAUNCH_CHECK("tam_fwd");
        return TCVOM_OK;
    }
    hipLaunchKernelGGL(tam_compact_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, mask, worklist, n);
    const dim3 grid(tam_heavy_grid(n));
    TAM_DISPATCH(tam_fwd_kernel, grid, (const unsigned*)q, (const unsigned*)kb, (const unsigned*)kf, (const unsigned*)v,
                 mask, (unsigned*)out, attb, attf, (const int*)worklist, B, H, W, C, isc);
    TCVOM_LAUNCH_CHECK("tam_fwd");
    return TCVOM_OK;
}

extern "C" int tcvom_tam_bwd(const void* q, const void* kb, const void* kf, const uint8_t* mask, const void* dout,
                             const float* dattb, const float* dattf, void* dq, void* dkb, void* dkf,
                             float* pbuf, float* dsbuf, const int32_t* worklist, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t window, void* stream) {
    TCVOM_CHECK_ARG(q && kb && kf && mask && dout && dq && dkb && dkf && pbuf && dsbuf && worklist, "tam_bwd: null pointer");
    if (int e = tam_check(B, H, W, C, window)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * W;
    const dim3 grid(tam_heavy_grid(n));
    const dim3 grid2(cdiv(n, 4), 2);
    const float isc = 1.0f / sqrtf((float)C);
    static const int dense = getenv("TCVOM_TAM_DENSE") ? atoi(getenv("TCVOM_TAM_DENSE")) : 1;      // (pass A: 90 -> 96 us on the band window, 118 -> 90 us all-unknown)
    static const bool key_valu0 = getenv("TCVOM_TAM_KEY_VALU") != nullptr;
    const bool self_init = window == 7 && C == 128 && dense <= 1;   // (the MFMA kernels write every pixel themselves)
    if (!self_init && hipMemsetAsync(dq, 0, sizeof(h16raw) * (size_t)n * C, st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "tam_bwd: memset failed");
    int tile_hi = 1 << 30;
    if (window == 7 && C == 128 && dense <= 64) {
        auto kern = tam_mfma_kernel<1>;
        constexpr size_t lds = (size_t)(2 * 224 * 16 + 2 * 64 * 16) * 16;
        static bool attr = false;
        if (!attr) {
            const hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "tam: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(ea)); }
            attr = true;
        }
        // every active tile on the MFMA kernel (dense <= 1): pass B runs on the matrix cores too, fed by query-major (p, ds) words
        static const bool key_valu = getenv("TCVOM_TAM_KEY_VALU") != nullptr;           // A/B switch: the one-wave-per-key gather kernel
        const bool key_mfma = dense <= 1 && !key_valu;
        hipLaunchKernelGGL(kern, dim3(cdiv(W, 8), cdiv(H, 8), B), dim3(256), lds, st, (const uint4*)q, (const uint4*)kb, (const uint4*)kf,
                           (const unsigned*)nullptr, (const uint4*)dout, mask, (unsigned*)dq, pbuf, key_mfma ? (float*)nullptr : dsbuf,
                           dattb, dattf, H, W, isc, dense < 1 ? 1 : dense, self_init ? 1 : 0);
        tile_hi = dense;
        (void)key_valu0;
        if (key_mfma) {
            auto kkern = tam_mfma_key_kernel;
            constexpr size_t klds = (size_t)(2 * 224 * 16) * 16;
            static bool kattr = false;
            if (!kattr) {
                const hipError_t ea = hipFuncSetAttribute((const void*)kkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)klds);
                if (ea != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "tam: hipFuncSetAttribute(%d bytes of LDS): %s", (int)klds, hipGetErrorString(ea)); }
                kattr = true;
            }
            // (no memset of dk: a key tile without an unknown query in its halo writes its own zeros)
            hipLaunchKernelGGL(kkern, dim3(cdiv(W, 8), cdiv(H, 8), B), dim3(256), klds, st, (const uint4*)q, (const uint4*)dout, mask,
                               (const unsigned*)pbuf, (unsigned*)dkb, (unsigned*)dkf, H, W);
            TCVOM_LAUNCH_CHECK("tam_bwd");
            return TCVOM_OK;
        }
    }
    if (window == 7 && C % 8 == 0 && tile_hi <= 1) {
        // (every tile went to the MFMA kernel)
    } else if (window == 7 && C % 8 == 0) {
#define TAM_TILED_BWD(TH, TW, CM)                                                                                     \
        hipLaunchKernelGGL((tam_tiled_kernel<7, 1, TH, TW, CM>), dim3(cdiv(W, TW), cdiv(H, TH), B), dim3(TAM_NW * 64), 0, st,   \
                           (const uint4*)q, (const uint4*)kb, (const uint4*)kf, (const unsigned*)nullptr, (const uint4*)dout, \
                           mask, (unsigned*)dq, pbuf, dsbuf, dattb, dattf, H, W, C, isc, tile_hi)
        if (C <= 128) TAM_TILED_BWD(8, 8, 16); else TAM_TILED_BWD(4, 4, 32);
#undef TAM_TILED_BWD
    } else {
        TAM_DISPATCH(tam_bwd_query_kernel, grid, (const unsigned*)q, (const unsigned*)kb, (const unsigned*)kf, mask,
                     (const unsigned*)dout, dattb, dattf, (unsigned*)dq, pbuf, dsbuf, (const int*)worklist, B, H, W, C, isc);
    }
    TAM_DISPATCH(tam_bwd_key_kernel, grid2, (const unsigned*)q, mask, (const unsigned*)dout, pbuf, dsbuf,
                 (unsigned*)dkb, (unsigned*)dkf, B, H, W, C);
    TCVOM_LAUNCH_CHECK("tam_bwd");
    return TCVOM_OK;
}
