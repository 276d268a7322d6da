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
// One wave (64 lanes) per pixel; a lane owns channel pairs {lane + 64*i}; the w*w neighbour key
// slices stay in registers between the logit and the aggregation pass, so each key element is
// fetched once per (pixel, direction) — from L2, since neighbouring pixels share 42/49 of them.
// HBM-bound: algorithmic traffic = read q, v, kb, kf + write out + 2 logit maps.
#include "common.h"

template <int WIN, int NP>
__global__ __launch_bounds__(256) void tam_fwd_kernel(
    const unsigned* __restrict__ q, const unsigned* __restrict__ kb, const unsigned* __restrict__ kf,
    const unsigned* __restrict__ v, const unsigned char* __restrict__ mask,
    unsigned* __restrict__ out, float* __restrict__ attb, float* __restrict__ attf,
    int B, int H, int W, int C, float inv_sqrt_c)
{
    constexpr int W2 = WIN * WIN, R = WIN / 2;
    const int lane = threadIdx.x & 63;
    const int64_t N = (int64_t)H * W;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= B * N) return;
    const int b = (int)(pix / N);
    const int64_t u = pix % N;
    const int y = (int)(u / W), x = (int)(u % W);
    const int CP = C / 2;                       // channel pairs per pixel
    const bool unknown = mask[pix] != 0;

    float o[NP][2];
    unsigned qq[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        const unsigned vv = cp < CP ? v[pix * CP + cp] : 0u;
        o[i][0] = bflo(vv);
        o[i][1] = bfhi(vv);
        qq[i] = (cp < CP && unknown) ? q[pix * CP + cp] : 0u;
    }
#pragma unroll 1
    for (int dir = 0; dir < 2; ++dir) {
        const unsigned* __restrict__ k = dir == 0 ? kb : kf;
        float* __restrict__ att = dir == 0 ? attb : attf;
        if (!unknown) {
            for (int j = lane; j < W2; j += 64) att[((int64_t)b * W2 + j) * N + u] = 0.f;
            continue;
        }
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
                part += bflo(qq[i]) * bflo(w) + bfhi(qq[i]) * bfhi(w);
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
                o[i][0] += p * bflo(kk[j][i]);
                o[i][1] += p * bfhi(kk[j][i]);
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
        if (cp < CP) out[pix * CP + cp] = pack2bf(o[i][0], o[i][1]);
    }
}

// Backward pass A (per query pixel): recompute p; dp_j = <dout, k_j>; ds = p*(dp - sum p dp) + datt;
// dq = sum_j ds_j k_j / sqrt(C).  Stores p and ds ([B][2][W2][N] fp32) for pass B.
template <int WIN, int NP>
__global__ __launch_bounds__(256) void tam_bwd_query_kernel(
    const unsigned* __restrict__ q, const unsigned* __restrict__ kb, const unsigned* __restrict__ kf,
    const unsigned char* __restrict__ mask, const unsigned* __restrict__ dout,
    const float* __restrict__ dattb, const float* __restrict__ dattf,
    unsigned* __restrict__ dq, float* __restrict__ pbuf, float* __restrict__ dsbuf,
    int B, int H, int W, int C, float inv_sqrt_c)
{
    constexpr int W2 = WIN * WIN, R = WIN / 2;
    const int lane = threadIdx.x & 63;
    const int64_t N = (int64_t)H * W;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= B * N) return;
    const int b = (int)(pix / N);
    const int64_t u = pix % N;
    const int y = (int)(u / W), x = (int)(u % W);
    const int CP = C / 2;
    const bool unknown = mask[pix] != 0;
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
        if (!unknown) {
            for (int j = lane; j < W2; j += 64) { pb[(int64_t)j * N + u] = 0.f; db[(int64_t)j * N + u] = 0.f; }
            continue;
        }
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
                part += bflo(qq[i]) * bflo(w) + bfhi(qq[i]) * bfhi(w);
                part2 += bflo(go[i]) * bflo(w) + bfhi(go[i]) * bfhi(w);
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
                dqa[i][0] += dss * bflo(kk[j][i]);
                dqa[i][1] += dss * bfhi(kk[j][i]);
            }
            if (lane == (j & 63)) { pb[(int64_t)j * N + u] = lg[j]; db[(int64_t)j * N + u] = dss; }
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        if (cp < CP) dq[pix * CP + cp] = pack2bf(dqa[i][0], dqa[i][1]);
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
#pragma unroll 7
    for (int j = 0; j < W2; ++j) {
        const int uy = y - (j / WIN - R), ux = x - (j % WIN - R);
        if (uy < 0 || uy >= H || ux < 0 || ux >= W) continue;
        const int64_t u = (int64_t)uy * W + ux;
        if (mask[(int64_t)b * N + u] == 0) continue;
        const float p = pb[(int64_t)j * N + u], ds = db[(int64_t)j * N + u];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int cp = lane + 64 * i;
            if (cp < CP) {
                const unsigned g = dout[((int64_t)b * N + u) * CP + cp];
                const unsigned qv = q[((int64_t)b * N + u) * CP + cp];
                acc[i][0] += p * bflo(g) + ds * bflo(qv);
                acc[i][1] += p * bfhi(g) + ds * bfhi(qv);
            }
        }
    }
    unsigned* __restrict__ dk = dir == 0 ? dkb : dkf;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int cp = lane + 64 * i;
        if (cp < CP) dk[pix * CP + cp] = pack2bf(acc[i][0], acc[i][1]);
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

extern "C" int tcvom_tam_fwd(const void* q, const void* kb, const void* kf, const void* v, const uint8_t* mask,
                             void* out, float* attb, float* attf, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t window, void* stream) {
    TCVOM_CHECK_ARG(q && kb && kf && v && mask && out && attb && attf, "tam_fwd: null pointer");
    if (int e = tam_check(B, H, W, C, window)) return e;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv((int64_t)B * H * W, 4));
    const float isc = 1.0f / sqrtf((float)C);
    TAM_DISPATCH(tam_fwd_kernel, grid, (const unsigned*)q, (const unsigned*)kb, (const unsigned*)kf, (const unsigned*)v,
                 mask, (unsigned*)out, attb, attf, B, H, W, C, isc);
    TCVOM_LAUNCH_CHECK("tam_fwd");
    return TCVOM_OK;
}

extern "C" int tcvom_tam_bwd(const void* q, const void* kb, const void* kf, const uint8_t* mask, const void* dout,
                             const float* dattb, const float* dattf, void* dq, void* dkb, void* dkf,
                             float* pbuf, float* dsbuf, int32_t B, int32_t H, int32_t W, int32_t C, int32_t window,
                             void* stream) {
    TCVOM_CHECK_ARG(q && kb && kf && mask && dout && dq && dkb && dkf && pbuf && dsbuf, "tam_bwd: null pointer");
    if (int e = tam_check(B, H, W, C, window)) return e;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv((int64_t)B * H * W, 4));
    const dim3 grid2(cdiv((int64_t)B * H * W, 4), 2);
    const float isc = 1.0f / sqrtf((float)C);
    TAM_DISPATCH(tam_bwd_query_kernel, grid, (const unsigned*)q, (const unsigned*)kb, (const unsigned*)kf, mask,
                 (const unsigned*)dout, dattb, dattf, (unsigned*)dq, pbuf, dsbuf, B, H, W, C, isc);
    TAM_DISPATCH(tam_bwd_key_kernel, grid2, (const unsigned*)q, mask, (const unsigned*)dout, pbuf, dsbuf,
                 (unsigned*)dkb, (unsigned*)dkf, B, H, W, C);
    TCVOM_LAUNCH_CHECK("tam_bwd");
    return TCVOM_OK;
}
