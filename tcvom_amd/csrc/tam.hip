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
    int H, int W, int C, float inv_sqrt_c)
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
    if (nu == 0) return;
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
    if (hipMemsetAsync(attb, 0, att_bytes, st) != hipSuccess || hipMemsetAsync(attf, 0, att_bytes, st) != hipSuccess ||
        hipMemsetAsync(worklist, 0, sizeof(int32_t), st) != hipSuccess ||
        hipMemcpyAsync(out, v, sizeof(h16raw) * (size_t)n * C, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return tcvom_fail(TCVOM_ERR_LAUNCH, "tam_fwd: memset / copy failed");
    if (window == 7 && C % 8 == 0) {
#define TAM_TILED_FWD(TH, TW, CM)                                                                                     \
        hipLaunchKernelGGL((tam_tiled_kernel<7, 0, TH, TW, CM>), dim3(cdiv(W, TW), cdiv(H, TH), B), dim3(TAM_NW * 64), 0, st,   \
                           (const uint4*)q, (const uint4*)kb, (const uint4*)kf, (const unsigned*)v, (const uint4*)nullptr, \
                           mask, (unsigned*)out, attb, attf, (const float*)nullptr, (const float*)nullptr, H, W, C, isc)
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
    if (hipMemsetAsync(dq, 0, sizeof(h16raw) * (size_t)n * C, st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "tam_bwd: memset failed");
    if (window == 7 && C % 8 == 0) {
#define TAM_TILED_BWD(TH, TW, CM)                                                                                     \
        hipLaunchKernelGGL((tam_tiled_kernel<7, 1, TH, TW, CM>), dim3(cdiv(W, TW), cdiv(H, TH), B), dim3(TAM_NW * 64), 0, st,   \
                           (const uint4*)q, (const uint4*)kb, (const uint4*)kf, (const unsigned*)nullptr, (const uint4*)dout, \
                           mask, (unsigned*)dq, pbuf, dsbuf, dattb, dattf, H, W, C, isc)
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
