// Small HBM-bound layout / resampling kernels around the conv stack (all NHWC bf16, 16-byte
// accesses): AvgPool2d(2) of the encoder down-sample path (resnet_enc.py:108-114), nearest x2
// up-sampling of the decoder identity path (resnet_dec.py:112-118), ReflectionPad2d(1) of the
// guidance head (res_gca_enc.py:20-33), bias gradients, bf16 matrix transposes, and the final
// 32->1 3x3 conv + (tanh+1)/2 of the decoder (resnet_dec.py:80, VMN_GCA.py:45-47).
#include "common.h"

static int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}
#define GRID_STRIDE(v, n) \
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (n); v += (int64_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- avgpool 2x2 s2 (fwd) / its backward
// DUAL = 1 (fp16 island of the bf16 build, tcvom_avgpool2_f16): x is IEEE fp16; the result goes to y in the build's type and to y16 in IEEE fp16
template <int DUAL = 0>
__global__ void avgpool2_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, uint4* __restrict__ y16 = nullptr) {
    const int OH = H / 2, OW = W / 2;
    const int64_t n = (int64_t)N * OH * OW * C8;
    GRID_STRIDE(v, n) {
        const int c = (int)(v % C8);
        int64_t t = v / C8;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int nn = (int)(t / OH);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const uint4 q = x[(((int64_t)nn * H + 2 * oh + dy) * W + 2 * ow + dx) * C8 + c];
                if constexpr (DUAL) unpack8_ieee(q, f); else unpack8(q, f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += f[k];
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= 0.25f;
        y[v] = pack8(acc);
        if constexpr (DUAL) y16[v] = pack8_ieee(acc);
    }
}
// dx[n,h,w] = scale * dy[n,h/2,w/2]      (avgpool backward: scale .25; nearest-upsample forward: scale 1)
__global__ void upsample2_kernel(const uint4* __restrict__ y, uint4* __restrict__ x, int N, int H, int W, int C8, float scale) {
    const int OH = H / 2, OW = W / 2;
    const int64_t n = (int64_t)N * H * W * C8;
    GRID_STRIDE(v, n) {
        const int c = (int)(v % C8);
        int64_t t = v / C8;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int nn = (int)(t / H);
        uint4 q = y[(((int64_t)nn * OH + h / 2) * OW + w / 2) * C8 + c];
        if (scale != 1.f) {
            float f[8];
            unpack8(q, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] *= scale;
            q = pack8(f);
        }
        x[v] = q;
    }
}
// y[n,oh,ow] = scale * sum_{2x2} x[...]   (nearest-upsample backward: scale 1)
__global__ void sumpool2_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, float scale) {
    const int OH = H / 2, OW = W / 2;
    const int64_t n = (int64_t)N * OH * OW * C8;
    GRID_STRIDE(v, n) {
        const int c = (int)(v % C8);
        int64_t t = v / C8;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int nn = (int)(t / OH);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                unpack8(x[(((int64_t)nn * H + 2 * oh + dy) * W + 2 * ow + dx) * C8 + c], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += f[k];
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= scale;
        y[v] = pack8(acc);
    }
}

// ---------------------------------------------------------------- reflection pad 1 (fwd) and its backward
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ void reflect_pad1_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8) {
    const int PH = H + 2, PW = W + 2;
    const int64_t n = (int64_t)N * PH * PW * C8;
    GRID_STRIDE(v, n) {
        const int c = (int)(v % C8);
        int64_t t = v / C8;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int nn = (int)(t / PH);
        y[v] = x[(((int64_t)nn * H + reflect_idx(ph - 1, H)) * W + reflect_idx(pw - 1, W)) * C8 + c];
    }
}
__global__ void reflect_pad1_bwd_kernel(const uint4* __restrict__ dy, uint4* __restrict__ dx, int N, int H, int W, int C8) {
    const int PH = H + 2, PW = W + 2;
    const int64_t n = (int64_t)N * H * W * C8;
    GRID_STRIDE(v, n) {
        const int c = (int)(v % C8);
        int64_t t = v / C8;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int nn = (int)(t / H);
        // padded rows that map to h: h+1 always; 0 if h==1; PH-1 if h==H-2
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + 1;
        if (h == 1) hs[nh++] = 0;
        if (h == H - 2) hs[nh++] = PH - 1;
        ws[nw++] = w + 1;
        if (w == 1) ws[nw++] = 0;
        if (w == W - 2) ws[nw++] = PW - 1;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
                unpack8(dy[(((int64_t)nn * PH + hs[a]) * PW + ws[b]) * C8 + c], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += f[k];
            }
        dx[v] = pack8(acc);
    }
}

// ---------------------------------------------------------------- z = a + b (+ c)
__global__ void add_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c,
                           uint4* __restrict__ z, int64_t nvec) {
    GRID_STRIDE(v, nvec) {
        float fa[8], fb[8], fc[8];
        unpack8(a[v], fa);
        unpack8(b[v], fb);
        if (c) unpack8(c[v], fc);
#pragma unroll
        for (int k = 0; k < 8; ++k) fa[k] += fb[k] + (c ? fc[k] : 0.f);
        z[v] = pack8(fa);
    }
}

// ---------------------------------------------------------------- bias gradient: db[k] = sum_p dy[p][k]
__global__ __launch_bounds__(256) void colsum_kernel(const h16raw* __restrict__ dy, float* __restrict__ out,
                                                     int64_t P, int K, int ld, int rows_per_block) {
    // columns are processed in chunks of Kc = min(K, 256); within a chunk 256/Kc row slices run concurrently
    const int Kc = K < 256 ? K : 256;
    const int RS = 256 / Kc;
    const int cl = threadIdx.x % Kc, rsub = threadIdx.x / Kc;
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    __shared__ float red[256];
    for (int cb = 0; cb < K; cb += Kc) {
        float a = 0.f;
        if (rsub < RS && cb + cl < K)
            for (int64_t p = pbeg + rsub; p < pend; p += RS) a += h2f(dy[p * ld + cb + cl]);
        __syncthreads();
        red[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < Kc && cb + threadIdx.x < K) {
            float s = 0.f;
            for (int r = 0; r < RS; ++r) s += red[r * Kc + threadIdx.x];
            atomicAdd(out + cb + threadIdx.x, s);
        }
    }
}

// The same with 16-byte loads: a thread owns ONE channel octet (8 running sums) and walks the rows with stride 256 / K8 -- the
// scalar form above issues one 2-byte load per element (28 us for 32640 x 128: 0.3 TB/s).  K % 8 == 0, K <= 2048, ld % 8 == 0.
__global__ __launch_bounds__(256) void colsum_v8_kernel(const uint4* __restrict__ dy, float* __restrict__ out, int64_t P, int K8,
                                                        int64_t ld8, int rows_per_block) {
    __shared__ float red[256 * 8];
    const int RS = 256 / K8;                                   // row slices running concurrently
    const int o = threadIdx.x % K8, rs = threadIdx.x / K8;
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rs < RS) {
        int64_t p = pbeg + rs;
        for (; p + 3 * RS < pend; p += 4 * RS) {               // four loads in flight
            const uint4 q0 = dy[p * ld8 + o], q1 = dy[(p + RS) * ld8 + o], q2 = dy[(p + 2 * RS) * ld8 + o], q3 = dy[(p + 3 * RS) * ld8 + o];
            float f[8];
            unpack8(q0, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += f[k];
            unpack8(q1, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += f[k];
            unpack8(q2, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += f[k];
            unpack8(q3, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += f[k];
        }
        for (; p < pend; p += RS) {
            float f[8];
            unpack8(dy[p * ld8 + o], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += f[k];
        }
    }
    // red[slice][K]; then thread c adds the slices of channel c and issues ONE atomic: the lanes of an atomic instruction
    // cover contiguous addresses (whole cache lines per instruction -- 8 instructions of 16 scattered lanes per block, the
    // first form of this kernel, ran 3x slower than the scalar kernel: the atomics are executed per cache line)
    const int K = K8 * 8;
    if (rs < RS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[rs * K + o * 8 + k] = a[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) {
        float sum = 0.f;
        for (int r = 0; r < RS; ++r) sum += red[r * K + c];
        atomicAdd(out + c, sum);
    }
}

// ---------------------------------------------------------------- bf16 matrix transpose  in[R][ldi] -> out[Cc][ldo]
__global__ __launch_bounds__(256) void transpose_kernel(const h16raw* __restrict__ in, h16raw* __restrict__ out,
                                                        int R, int Cc, int64_t ldi, int64_t ldo,
                                                        int64_t in_bstride, int64_t out_bstride) {
    __shared__ h16raw tile[64][66];
    in += blockIdx.z * in_bstride;
    out += blockIdx.z * out_bstride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int rr = r0 + r, cc = c0 + tx;
        tile[r][tx] = (rr < R && cc < Cc) ? in[(int64_t)rr * ldi + cc] : (h16raw)0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int cc = c0 + c, rr = r0 + tx;
        if (cc < Cc && rr < ldo) out[(int64_t)cc * ldo + rr] = rr < R ? tile[tx][c] : (h16raw)0;
    }
}

// 16-byte variant (ldi, ldo multiples of 8, 16-byte aligned bases): 64x64 tiles, uint4 global loads and stores, the
// transposition happens in the LDS read (8 two-byte reads per output chunk).  2.5 -> ~5 TB/s on the 134 MB attention
// matrices of the GCA backward.
__global__ __launch_bounds__(256) void transpose_v8_kernel(const h16raw* __restrict__ in, h16raw* __restrict__ out,
                                                           int R, int Cc, int64_t ldi, int64_t ldo,
                                                           int64_t in_bstride, int64_t out_bstride) {
    __shared__ h16raw tile[64][72];
    in += blockIdx.z * in_bstride;
    out += blockIdx.z * out_bstride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t8 = threadIdx.x & 7, tr = threadIdx.x >> 3;           // 8 chunks x 32 rows per pass
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = tr + 32 * i, rr = r0 + r, cc = c0 + t8 * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (rr < R && cc + 8 <= Cc) v = *reinterpret_cast<const uint4*>(in + (int64_t)rr * ldi + cc);
        else if (rr < R && cc < Cc) {
            h16raw tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) tmp[k] = cc + k < Cc ? in[(int64_t)rr * ldi + cc + k] : (h16raw)0;
            v = *reinterpret_cast<uint4*>(tmp);
        }
        *reinterpret_cast<uint4*>(&tile[r][t8 * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tr + 32 * i, cc = c0 + c, rr = r0 + t8 * 8;      // output row cc, output columns rr .. rr+7
        if (cc < Cc && rr < ldo) {
            h16raw tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) tmp[k] = tile[t8 * 8 + k][c];    // rows >= R were loaded as zeros
            *reinterpret_cast<uint4*>(out + (int64_t)cc * ldo + rr) = *reinterpret_cast<uint4*>(tmp);
        }
    }
}

// ---------------------------------------------------------------- final conv C->1 (KS x KS, pad KS/2, bias) + output map
// MODE 0: alpha = (tanh(pre) + 1) / 2   (GCA decoder head, resnet_dec.py:139-141; KS = 3)
// MODE 1: alpha = clamp(pre, 0, 1)      (DIM alpha_pred, models/DIM/vggnet.py:76,123; KS = 5)
// MODE 2: alpha = pre                   (IndexNet pred[0][0] 32 -> 1, models/Index/net.py:16-22; KS = 5; BN + ReLU6 follow)
// x NHWC bf16 [N,H,W,C]; w fp32 [KS*KS][C] (tap-major); alpha fp32 [N,H,W]
template <int MODE>
__device__ __forceinline__ float head_out(float pre) {
    return MODE == 0 ? 0.5f * (tanhf(pre) + 1.f) : (MODE == 1 ? fminf(fmaxf(pre, 0.f), 1.f) : pre);
}
template <int MODE>
__device__ __forceinline__ float head_dpre(float dalpha, float alpha) {
    if (MODE == 0) {
        const float th = 2.f * alpha - 1.f;
        return dalpha * 0.5f * (1.f - th * th);
    }
    if (MODE == 2) return dalpha;
    return (alpha > 0.f && alpha < 1.f) ? dalpha : 0.f;
}

template <int KS, int MODE>
__global__ __launch_bounds__(256) void head_conv_fwd_kernel(const h16raw* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ alpha,
                                                            int N, int H, int W, int C) {
    constexpr int T = KS * KS, R = KS / 2;
    extern __shared__ float ws[];
    for (int i = threadIdx.x; i < T * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int64_t n = (int64_t)N * H * W;
    const int C8 = C / 8;
    GRID_STRIDE(p, n) {
        const int xw = (int)(p % W);
        const int yh = (int)((p / W) % H);
        const int nn = (int)(p / ((int64_t)W * H));
        float acc = bias[0];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int ih = yh + t / KS - R, iw = xw + t % KS - R;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            const uint4* src = reinterpret_cast<const uint4*>(x + (((int64_t)nn * H + ih) * W + iw) * C);
            for (int c8 = 0; c8 < C8; ++c8) {
                float f[8];
                unpack8(src[c8], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += f[k] * ws[t * C + c8 * 8 + k];
            }
        }
        alpha[p] = head_out<MODE>(acc);
    }
}
// (Measured and dropped, round 3: the channel contraction on v_dot2 with hi + residual 16-bit weight pairs in LDS -- 167 us with one
//  thread per pixel, 102 us with one thread per (pixel, 8 channels) and lane shuffles, against 96 us for this form: the kernel is
//  bound by the 9-fold re-read of the input through the vector memory path, not by its conversions; an LDS-tiled form would be next.)
// dpre = d(loss) / d(pre-activation) of the head, once per pixel (the data- and the weight-gradient kernels read it 9 / 25 times)
template <int MODE>
__global__ __launch_bounds__(256) void head_dpre_kernel(const float* __restrict__ dalpha, const float* __restrict__ alpha,
                                                        float* __restrict__ dpre, int64_t n) {
    GRID_STRIDE(p, n) dpre[p] = head_dpre<MODE>(dalpha[p], alpha[p]);
}
// dx[p][c] = sum_t dpre[p - off_t] w[t][c]
template <int KS>
__global__ __launch_bounds__(256) void head_conv_bwd_data_kernel(const float* __restrict__ dpre, const float* __restrict__ w,
                                                                 h16raw* __restrict__ dx, int N, int H, int W, int C) {
    constexpr int T = KS * KS, R = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float ws[];
    for (int i = threadIdx.x; i < T * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int C8 = C / 8;
    const int64_t n = (int64_t)N * H * W * C8;
    GRID_STRIDE(v, n) {
        const int c8 = (int)(v % C8);
        const int64_t p = v / C8;
        const int xw = (int)(p % W);
        const int yh = (int)((p / W) % H);
        const int nn = (int)(p / ((int64_t)W * H));
        // output pixel q = p - (t offset):  x[p] contributed to pre[q] with tap t where p = q + off_t.  The T scalars first (they
        // are independent loads), then the multiply-adds.
        float dp[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int qh = yh - (t / KS - R), qw = xw - (t % KS - R);
            const bool ok = qh >= 0 && qh < H && qw >= 0 && qw < W;
            dp[t] = ok ? dpre[((int64_t)nn * H + qh) * W + qw] : 0.f;
        }
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 w0 = *reinterpret_cast<const float4*>(ws + t * C + c8 * 8), w1 = *reinterpret_cast<const float4*>(ws + t * C + c8 * 8 + 4);
            acc[0] += dp[t] * w0.x; acc[1] += dp[t] * w0.y; acc[2] += dp[t] * w0.z; acc[3] += dp[t] * w0.w;
            acc[4] += dp[t] * w1.x; acc[5] += dp[t] * w1.y; acc[6] += dp[t] * w1.z; acc[7] += dp[t] * w1.w;
        }
        *reinterpret_cast<uint4*>(dx + p * C + c8 * 8) = pack8(acc);
    }
}
// dw[t][c] = sum_p dpre[p] x[p + off_t][c];  db = sum_p dpre[p].
// Input-stationary: x is read ONCE (16 bytes = 8 channels per thread and pixel) and multiplied with the KS*KS
// neighbouring dpre values (fp32 scalars, L1/L2-resident); KS*KS*8 accumulators per thread, pixel lanes combined
// through LDS, one atomicAdd per (tap, channel) and block, spread over `replicas` copies of dw that the caller sums
// (2048 blocks onto ONE copy measured ~500 us of pure atomic serialisation).  A tap-stationary version that re-read x
// once per tap took 505 us (3x3, 64 channels) and 4.0 ms (5x5, 32 channels) at 1088x1920 against ~60 us for one pass.
template <int KS>
__global__ __launch_bounds__(256) void head_conv_bwd_weight_kernel(const float* __restrict__ dpre, const h16raw* __restrict__ x,
                                                                   float* __restrict__ dw, float* __restrict__ db,
                                                                   int N, int H, int W, int C, int rows_per_block, int replicas) {
    constexpr int T = KS * KS, R = KS / 2;
    __shared__ float red[256 * T];
    const int C8 = C / 8;
    const int64_t P = (int64_t)N * H * W;
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    const int c8 = threadIdx.x % C8, pl = threadIdx.x / C8, PL = 256 / C8;
    float acc[T][8];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    int xw = (int)((pbeg + pl) % W), yh = (int)(((pbeg + pl) / W) % H);
    for (int64_t q = pbeg + pl; q < pend; q += PL, xw += PL) {
        while (xw >= W) { xw -= W; yh = yh + 1 == H ? 0 : yh + 1; }
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + q * C + c8 * 8), f);
        if (c8 == 0) bsum += dpre[q];
        // x[q] is tap t of the output pixel o = q - off_t.  The T scalars first (independent loads; one bounds test per pixel for
        // the interior instead of four per tap), then the multiply-adds.
        float dp[T];
        if (yh >= R && yh < H - R && xw >= R && xw < W - R) {
#pragma unroll
            for (int t = 0; t < T; ++t) dp[t] = dpre[q - (int64_t)(t / KS - R) * W - (t % KS - R)];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int oh = yh - (t / KS - R), ow = xw - (t % KS - R);
                dp[t] = (oh < 0 || oh >= H || ow < 0 || ow >= W) ? 0.f : dpre[q - (int64_t)(t / KS - R) * W - (t % KS - R)];
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[t][k] += dp[t] * f[k];
    }
    // combine the PL pixel lanes: for every k, red[t][thread]; then thread i < T*C8 sums (t = i / C8, c8' = i % C8)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) red[t * 256 + threadIdx.x] = acc[t][k];
        __syncthreads();
        for (int i = threadIdx.x; i < T * C8; i += 256) {
            const int t = i / C8, cc = i % C8;
            float sacc = 0.f;
            for (int l = 0; l < PL; ++l) sacc += red[t * 256 + l * C8 + cc];
            atomicAdd(dw + (int64_t)(blockIdx.x % replicas) * T * C + t * C + cc * 8 + k, sacc);
        }
    }
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sacc = 0.f;
        for (int l = 0; l < PL; ++l) sacc += red[l * C8];
        atomicAdd(db + blockIdx.x % replicas, sacc);
    }
}

// ---------------------------------------------------------------- C ABI
extern "C" int tcvom_avgpool2(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && y && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "avgpool2: bad args");
    hipLaunchKernelGGL(avgpool2_kernel<0>, dim3(grid_for((int64_t)N * H * W * C / 32)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, (uint4*)y, N, H, W, C / 8, (uint4*)nullptr);
    TCVOM_LAUNCH_CHECK("avgpool2");
    return TCVOM_OK;
}
// the same inside the fp16 island of the bf16 build (gca_net.py: the AvgPool2d of encoder layer2's downsample branch): x16 IEEE fp16 in,
// y (the build's type: what the 1x1 conv's weight gradient reads) and y16 (IEEE fp16: what its forward reads) out
extern "C" int tcvom_avgpool2_f16(const void* x16, void* y, void* y16, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x16 && y && y16 && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "avgpool2_f16: bad args");
    hipLaunchKernelGGL(avgpool2_kernel<1>, dim3(grid_for((int64_t)N * H * W * C / 32)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x16, (uint4*)y, N, H, W, C / 8, (uint4*)y16);
    TCVOM_LAUNCH_CHECK("avgpool2_f16");
    return TCVOM_OK;
}
extern "C" int tcvom_upsample2(const void* y, void* x, int32_t N, int32_t H, int32_t W, int32_t C, float scale, void* stream) {
    TCVOM_CHECK_ARG(x && y && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "upsample2: bad args");
    hipLaunchKernelGGL(upsample2_kernel, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)y, (uint4*)x, N, H, W, C / 8, scale);
    TCVOM_LAUNCH_CHECK("upsample2");
    return TCVOM_OK;
}
extern "C" int tcvom_sumpool2(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, float scale, void* stream) {
    TCVOM_CHECK_ARG(x && y && C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "sumpool2: bad args");
    hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_for((int64_t)N * H * W * C / 32)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, (uint4*)y, N, H, W, C / 8, scale);
    TCVOM_LAUNCH_CHECK("sumpool2");
    return TCVOM_OK;
}
extern "C" int tcvom_reflect_pad1(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(x && y && C % 8 == 0 && H >= 2 && W >= 2, "reflect_pad1: bad args");
    hipLaunchKernelGGL(reflect_pad1_kernel, dim3(grid_for((int64_t)N * (H + 2) * (W + 2) * C / 8)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4*)x, (uint4*)y, N, H, W, C / 8);
    TCVOM_LAUNCH_CHECK("reflect_pad1");
    return TCVOM_OK;
}
extern "C" int tcvom_reflect_pad1_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    TCVOM_CHECK_ARG(dx && dy && C % 8 == 0 && H >= 3 && W >= 3, "reflect_pad1_bwd: bad args");
    hipLaunchKernelGGL(reflect_pad1_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4*)dy, (uint4*)dx, N, H, W, C / 8);
    TCVOM_LAUNCH_CHECK("reflect_pad1_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_add(const void* a, const void* b, const void* c, void* z, int64_t numel, void* stream) {
    TCVOM_CHECK_ARG(a && b && z && numel % 8 == 0, "add: bad args");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(numel / 8)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a,
                       (const uint4*)b, (const uint4*)c, (uint4*)z, numel / 8);
    TCVOM_LAUNCH_CHECK("add");
    return TCVOM_OK;
}
extern "C" int tcvom_colsum(const void* dy, float* out, int64_t P, int32_t K, int32_t ld, void* stream) {
    TCVOM_CHECK_ARG(dy && out && K >= 1 && (K >= 256 || 256 % K == 0), "colsum: K=%d must divide 256 or be >= 256", K);
    int64_t blocks = (P + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    const int rpb = (int)((P + blocks - 1) / blocks);
    if (hipMemsetAsync(out, 0, sizeof(float) * K, (hipStream_t)stream) != hipSuccess)
        return tcvom_fail(TCVOM_ERR_LAUNCH, "colsum: memset failed");
    if (K % 8 == 0 && K <= 2048 && ld % 8 == 0 && ((uintptr_t)dy & 15) == 0) {
        int64_t b8 = (P + 127) / 128;                          // >= 128 rows per block: the atomics stay a small share
        if (b8 > 512) b8 = 512;
        if (b8 < 1) b8 = 1;
        const int rpb8 = (int)((P + b8 - 1) / b8);
        hipLaunchKernelGGL(colsum_v8_kernel, dim3((int)b8), dim3(256), 0, (hipStream_t)stream, (const uint4*)dy, out, P, K / 8, (int64_t)(ld / 8), rpb8);
        TCVOM_LAUNCH_CHECK("colsum");
        return TCVOM_OK;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const h16raw*)dy, out, P, K, ld, rpb);
    TCVOM_LAUNCH_CHECK("colsum");
    return TCVOM_OK;
}
extern "C" int tcvom_transpose_bf16(const void* in, void* out, int32_t R, int32_t Cc, int64_t ldi, int64_t ldo,
                                    int32_t batch, int64_t in_bstride, int64_t out_bstride, void* stream) {
    TCVOM_CHECK_ARG(in && out && R > 0 && Cc > 0 && ldi >= Cc && ldo >= R, "transpose_bf16: bad args");
    dim3 grid(cdiv(Cc, 64), cdiv(ldo, 64), batch > 1 ? batch : 1);
    const bool v8 = ldi % 8 == 0 && ldo % 8 == 0 && in_bstride % 8 == 0 && out_bstride % 8 == 0 &&
                    ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    if (v8)
        hipLaunchKernelGGL(transpose_v8_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const h16raw*)in, (h16raw*)out, R, Cc,
                           ldi, ldo, in_bstride, out_bstride);
    else
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const h16raw*)in, (h16raw*)out, R, Cc,
                           ldi, ldo, in_bstride, out_bstride);
    TCVOM_LAUNCH_CHECK("transpose_bf16");
    return TCVOM_OK;
}
extern "C" int tcvom_head_conv_fwd(const void* x, const float* w, const float* bias, float* alpha, int32_t N, int32_t H,
                                   int32_t W, int32_t C, int32_t ksize, int32_t mode, void* stream) {
    TCVOM_CHECK_ARG(x && w && bias && alpha && C % 8 == 0 && C <= 256, "head_conv_fwd: bad args");
    TCVOM_CHECK_ARG((ksize == 3 && mode == 0) || (ksize == 5 && (mode == 1 || mode == 2)), "head_conv_fwd: ksize=%d mode=%d not instantiated", ksize, mode);
    const dim3 g(grid_for((int64_t)N * H * W));
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3)
        hipLaunchKernelGGL((head_conv_fwd_kernel<3, 0>), g, dim3(256), 9 * C * sizeof(float), st, (const h16raw*)x, w, bias, alpha, N, H, W, C);
    else if (mode == 1)
        hipLaunchKernelGGL((head_conv_fwd_kernel<5, 1>), g, dim3(256), 25 * C * sizeof(float), st, (const h16raw*)x, w, bias, alpha, N, H, W, C);
    else
        hipLaunchKernelGGL((head_conv_fwd_kernel<5, 2>), g, dim3(256), 25 * C * sizeof(float), st, (const h16raw*)x, w, bias, alpha, N, H, W, C);
    TCVOM_LAUNCH_CHECK("head_conv_fwd");
    return TCVOM_OK;
}
extern "C" int tcvom_head_conv_bwd(const float* dalpha, const float* alpha, const void* x, const float* w, void* dx,
                                   float* dpre, float* dw, float* db, int32_t N, int32_t H, int32_t W, int32_t C,
                                   int32_t ksize, int32_t mode, int32_t replicas, void* stream) {
    TCVOM_CHECK_ARG(dalpha && alpha && x && w && dx && dpre && dw && db && C % 8 == 0 && C <= 256 && 256 % C == 0, "head_conv_bwd: bad args");
    TCVOM_CHECK_ARG(replicas >= 1 && replicas <= 64, "head_conv_bwd: replicas=%d outside 1..64", replicas);
    TCVOM_CHECK_ARG((ksize == 3 && mode == 0) || (ksize == 5 && (mode == 1 || mode == 2)), "head_conv_bwd: ksize=%d mode=%d not instantiated", ksize, mode);
    hipStream_t st = (hipStream_t)stream;
    const int T = ksize * ksize;
    const int64_t P = (int64_t)N * H * W;
    {
        const dim3 gp(grid_for(P));
        if (mode == 0) hipLaunchKernelGGL(head_dpre_kernel<0>, gp, dim3(256), 0, st, dalpha, alpha, dpre, P);
        else if (mode == 1) hipLaunchKernelGGL(head_dpre_kernel<1>, gp, dim3(256), 0, st, dalpha, alpha, dpre, P);
        else hipLaunchKernelGGL(head_dpre_kernel<2>, gp, dim3(256), 0, st, dalpha, alpha, dpre, P);
    }
    const dim3 g(grid_for(P * C / 8));
    if (ksize == 3)
        hipLaunchKernelGGL((head_conv_bwd_data_kernel<3>), g, dim3(256), T * C * sizeof(float), st, dpre, w, (h16raw*)dx, N, H, W, C);
    else
        hipLaunchKernelGGL((head_conv_bwd_data_kernel<5>), g, dim3(256), T * C * sizeof(float), st, dpre, w, (h16raw*)dx, N, H, W, C);
    if (hipMemsetAsync(dw, 0, sizeof(float) * T * C * replicas, st) != hipSuccess || hipMemsetAsync(db, 0, sizeof(float) * replicas, st) != hipSuccess)
        return tcvom_fail(TCVOM_ERR_LAUNCH, "head_conv_bwd: memset failed");
    // every block ends with T*C atomicAdds; `replicas` copies of dw (summed by the caller) keep them from serialising
    int64_t blocks = (P + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    const int rpb = (int)((P + blocks - 1) / blocks);
    if (ksize == 3)
        hipLaunchKernelGGL((head_conv_bwd_weight_kernel<3>), dim3((int)blocks), dim3(256), 0, st, dpre, (const h16raw*)x, dw, db, N, H, W, C, rpb, replicas);
    else
        hipLaunchKernelGGL((head_conv_bwd_weight_kernel<5>), dim3((int)blocks), dim3(256), 0, st, dpre, (const h16raw*)x, dw, db, N, H, W, C, rpb, replicas);
    TCVOM_LAUNCH_CHECK("head_conv_bwd");
    return TCVOM_OK;
}
