// BatchNorm (train-mode batch statistics / eval-mode running statistics) around the igemm
// convolutions, fused with activation and residual adds.  Replaces nn.BatchNorm2d + ReLU /
// LeakyReLU(0.2) + `out += identity` of the reference BasicBlocks
// (models/GCA/encoders/resnet_enc.py:33-49, decoders/resnet_dec.py:43-59) and the
// conv->ReLU->BN shortcut order of res_gca_enc.py:47-55.  All HBM-bound streaming kernels:
// 16-byte (8 x bf16) accesses per lane, fp32 math, statistics combined in fp64.
#include <cstdlib>
#include "common.h"

// ---------------------------------------------------------------- statistics finalize
// stage 1 (only when there are many groups): [G][2][C] fp32 -> [BN_SLICES][2][C] fp64, slice s sums groups
// g = s, s + BN_SLICES, ...  One block = 32 channels x 8 sub-slices of one slice; grid (C/32, BN_SLICES).
#define BN_SLICES 64
#define FIN_SL 32           // sub-slices (of 32 channels each) per block of the finalize kernels: 1024 threads, 4 load chains each
__global__ __launch_bounds__(256) void bn_partial_reduce_kernel(const float* __restrict__ partial, int G, int C,
                                                                double* __restrict__ out) {
    __shared__ double s1[8][32], s2[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int slice = blockIdx.y;
    partial += (int64_t)blockIdx.z * G * 2 * C;                     // frame (BatchNorm statistics group) of a batched call
    out += (int64_t)blockIdx.z * BN_SLICES * 2 * C;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 4 independent chains so that the (L2-latency-bound) loads of consecutive groups overlap
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        const int64_t st = (int64_t)8 * BN_SLICES * 2 * C;
        int g = slice + sl * BN_SLICES;
        for (; g + 24 * BN_SLICES < G; g += 32 * BN_SLICES) {
            const float* q = partial + (int64_t)g * 2 * C + c;
            const float x0 = q[0], y0 = q[C], x1 = q[st], y1 = q[st + C], x2 = q[2 * st], y2 = q[2 * st + C], x3 = q[3 * st], y3 = q[3 * st + C];
            a += (double)x0; b += (double)y0; a1 += (double)x1; b1 += (double)y1;
            a2 += (double)x2; b2 += (double)y2; a3 += (double)x3; b3 += (double)y3;
        }
        for (; g < G; g += 8 * BN_SLICES) {
            a += (double)partial[(int64_t)g * 2 * C + c];
            b += (double)partial[(int64_t)g * 2 * C + C + c];
        }
        a += a1 + a2 + a3;
        b += b1 + b2 + b3;
    }
    s1[sl][cl] = a;
    s2[sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
        out[(int64_t)slice * 2 * C + c] = a;
        out[(int64_t)slice * 2 * C + C + c] = b;
    }
}


// ---------------------------------------------------------------- SyncBatchNorm: in-kernel peer exchange
// train_ddp.py:271-273 converts every BatchNorm to SyncBatchNorm: the per-channel (sum, sum of squares) -- and in backward
// (sum dy, sum dy * xhat) -- are added over the ranks before they are used.  Instead of a collective per BatchNorm call (the
// reference: all_gather / all_reduce of <= 4 KB, ~370 per step), every rank owns a MAILBOX (uncached device memory, mapped into
// the peers through hipIpc): ring slot (seq % ring), one region per sender.  The finalize kernel PUSHES its local sums into the
// region [slot][my rank] of every peer over xGMI and then PULLS the `world` regions of its own mailbox.  A value travels as two
// 8-byte granules {32 data bits, 32-bit tag = seq}: each granule is one naturally aligned store, so a reader that sees the tag
// sees the data -- no fence, no flag, no ordering between granules (the LL protocol of the collective libraries).  Every rank adds
// the `world` contributions in rank order: bit-identical statistics on all ranks.  A slot is rewritten `ring` exchanges later;
// a rank can be at most one exchange ahead of the slowest reader of its pushes (it needs that reader's push to finish its own
// exchange), so ring >= 2 suffices on one stream.
struct BnSync {
    const unsigned long long* const* peers;   // device table [world]: base of rank r's mailbox as mapped in this process; NULL = no exchange
    int world, rank;
    unsigned int seq;                          // tag of this exchange (never 0: fresh mailboxes are zero)
    long long slot_off;                        // granule offset of the ring slot: (seq % ring) * world * cap2
    long long cap2;                            // granules per (slot, sender) = 2 * capacity in doubles
    long long timeout;                         // wall_clock64() ticks (100 MHz) a pull may spin before it gives up
    int* status;                               // set to seq when a pull timed out (host-visible: pinned memory)
    long long* wait;                           // device int64[2] or NULL: max / sum of the pull spin times (diagnostic)
};

// Threads (sl, cl) of a FIN_SL x 32 block; on entry the sl == 0 threads hold the LOCAL sums (a, b) of channel c of `frame`;
// on return they hold the sums over all ranks.  Every thread of the block must call it (barriers inside).
__device__ __forceinline__ void bn_sync_exchange(const BnSync& sy, double (*s1)[32], double (*s2)[32], int sl, int cl, int c, int C,
                                                 int frame, double& a, double& b) {
    if (sl == 0) { s1[0][cl] = a; s2[0][cl] = b; }
    __syncthreads();
    const bool active = sl < sy.world && c < C;
    const long long ia = ((long long)(frame * 2) * C + c) * 2, ib = ((long long)(frame * 2 + 1) * C + c) * 2;
    double pa = 0.0, pb = 0.0;
    if (active) {
        const unsigned long long ua = (unsigned long long)__double_as_longlong(s1[0][cl]), ub = (unsigned long long)__double_as_longlong(s2[0][cl]);
        const unsigned long long tag = (unsigned long long)sy.seq << 32;
        unsigned long long* dst = const_cast<unsigned long long*>(sy.peers[sl]) + sy.slot_off + (long long)sy.rank * sy.cap2;
        __hip_atomic_store(dst + ia, tag | (ua & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + ia + 1, tag | (ua >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + ib, tag | (ub & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + ib + 1, tag | (ub >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();                                               // s1[0] / s2[0] have been read: the pulls may overwrite them
    if (active) {
        unsigned long long* src = const_cast<unsigned long long*>(sy.peers[sy.rank]) + sy.slot_off + (long long)sl * sy.cap2;
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long g0 = __hip_atomic_load(src + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long g1 = __hip_atomic_load(src + ia + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long g2 = __hip_atomic_load(src + ib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long g3 = __hip_atomic_load(src + ib + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(g0 >> 32) == sy.seq && (unsigned)(g1 >> 32) == sy.seq && (unsigned)(g2 >> 32) == sy.seq && (unsigned)(g3 >> 32) == sy.seq) {
                pa = __longlong_as_double((long long)((g0 & 0xffffffffull) | (g1 << 32)));
                pb = __longlong_as_double((long long)((g2 & 0xffffffffull) | (g3 << 32)));
                break;
            }
            if (wall_clock64() - t0 > sy.timeout) {                // a peer never arrived: report, do not hang the device
                if (sy.status) __hip_atomic_store(sy.status, (int)sy.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // ... and POISON this rank's sums: statistics from partial sums would differ between the ranks silently; NaN
                // scale / shift / gradient coefficients turn the loss NaN at once (the host raises MailboxTimeout on every rank
                // before the optimizer step: GradientAverager exchanges the status word, tcvom_amd/ddp.py)
                pa = pb = __longlong_as_double(0x7ff8000000000000ll);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        // how long this rank waited for its neighbour's push: one sample per (exchange, workgroup), from the thread that pulls the
        // NEXT rank's region (a peer's push; in a one-rank mailbox its own)
        if (sy.wait && cl == 0 && frame == 0 && sl == (sy.rank + 1) % sy.world) {
            const long long dt = wall_clock64() - t0;
            atomicMax((unsigned long long*)sy.wait, (unsigned long long)dt);
            atomicAdd((unsigned long long*)sy.wait + 1, (unsigned long long)dt);
        }
    }
    if (sl < FIN_SL) { s1[sl][cl] = pa; s2[sl][cl] = pb; }
    __syncthreads();
    if (sl == 0) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < sy.world; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
    }
}

// stage 2 / single stage.  PT = float (raw partials [G][2][C]) or double (stage-1 output).
template <typename PT>
__global__ __launch_bounds__(FIN_SL * 32) void bn_finalize_kernel(
    const PT* __restrict__ partial, int G, int C, double count, double unbias_count,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float eps, float* __restrict__ scale_shift, float* __restrict__ saved, int64_t slot_stride, const BnSync sy)
{
    __shared__ double s1[FIN_SL][32], s2[FIN_SL][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    partial += (int64_t)blockIdx.y * G * 2 * C;                     // frame of a batched call
    scale_shift += blockIdx.y * slot_stride;
    saved += blockIdx.y * slot_stride;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 4 independent chains so that the (L2-latency-bound) loads of consecutive groups overlap
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        int g = sl;
        for (; g + 3 * FIN_SL < G; g += 4 * FIN_SL) {
            const PT* q = partial + (int64_t)g * 2 * C + c;
            const PT x0 = q[0], y0 = q[C], x1 = q[2 * FIN_SL * C], y1 = q[(2 * FIN_SL + 1) * C], x2 = q[4 * FIN_SL * C], y2 = q[(4 * FIN_SL + 1) * C],
                     x3 = q[6 * FIN_SL * C], y3 = q[(6 * FIN_SL + 1) * C];
            a += (double)x0; b += (double)y0; a1 += (double)x1; b1 += (double)y1;
            a2 += (double)x2; b2 += (double)y2; a3 += (double)x3; b3 += (double)y3;
        }
        for (; g < G; g += FIN_SL) {
            a += (double)partial[(int64_t)g * 2 * C + c];
            b += (double)partial[(int64_t)g * 2 * C + C + c];
        }
        a += a1 + a2 + a3;
        b += b1 + b2 + b3;
    }
    s1[sl][cl] = a;
    s2[sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < FIN_SL; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
    }
    if (sy.peers) bn_sync_exchange(sy, s1, s2, sl, cl, c, C, blockIdx.y, a, b);      // SyncBatchNorm: sums over the ranks (block-uniform branch)
    if (sl == 0 && c < C) {
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale_shift[c] = sc;
        scale_shift[C + c] = beta[c] - (float)mean * sc;
        saved[c] = (float)mean;
        saved[C + c] = invstd;
        if (running_mean) {
            const double unb = unbias_count > 1.0 ? var * unbias_count / (unbias_count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale_shift, float* __restrict__ saved)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - rm[c] * sc;
    saved[c] = rm[c];
    saved[C + c] = invstd;
}

// conv output `y` is 16-bit in the build's storage type (YF32 = 0) or IEEE fp16 (2: the fp16 island of the bf16 build -- 11 significant
// bits instead of bf16's 8 at the same bytes; in the fp16 build the same thing as 0)
// (unpack8_ieee: common.h)
static inline int bn_y_mode(int y_fp32) {
#ifdef TCVOM_F16
    return 0;
#else
    return y_fp32 == 2 ? 2 : 0;
#endif
}
template <int YF32>
__device__ __forceinline__ void load_y8(const void* __restrict__ y, int64_t v, float* f) {
    if (YF32 == 1) {
        const float4* p = reinterpret_cast<const float4*>(y) + 2 * v;
        const float4 a = p[0], b = p[1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        unpack8(reinterpret_cast<const uint4*>(y)[v], f);
    }
}

// ---------------------------------------------------------------- apply: z = act(y*s + b + res1) + res2
__device__ __forceinline__ float act_fwd(float x, int act) {
    return act == 1 ? fmaxf(x, 0.f) : (act == 2 ? (x > 0.f ? x : 0.2f * x) : (act == 3 ? (x > 0.f ? x : 0.01f * x) : x));
}
__device__ __forceinline__ float act_grad(float pre, int act) {
    return act == 1 ? (pre > 0.f ? 1.f : 0.f) : (act == 2 ? (pre > 0.f ? 1.f : 0.2f) : (act == 3 ? (pre > 0.f ? 1.f : 0.01f) : 1.f));
}

// negative-side slope of the activation (1 = none): act(x) = x > 0 ? x : slope * x, act'(x) = x > 0 ? 1 : slope -- branch-free
// act 4 = ReLU6 (IndexNet / MobileNetV2 blocks): slope 0 below, capped at 6 above (gradient 0 at and beyond both ends, like
// torch's hardtanh backward)
__device__ __forceinline__ float act_slope(int act) { return (act == 1 || act == 4) ? 0.f : (act == 2 ? 0.2f : (act == 3 ? 0.01f : 1.f)); }
__device__ __forceinline__ float act_cap(int act) { return act == 4 ? 6.f : __builtin_inff(); }
// 8 consecutive per-channel coefficients as two 16-byte loads (the vectors are 32-byte aligned: C and the slot strides are
// multiples of 8 floats); all loads of a thread's coefficient set are issued before the first one is waited for
__device__ __forceinline__ void load_coef8(const float* __restrict__ p, float* f) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// ---------------------------------------------------------------- fp16 build: saturation counters (tcvom_overflow_sink)
// process-wide: the backward kernels are launched from autograd's worker thread, not from the thread that set the sink
#include <atomic>
static std::atomic<int*> g_overflow_sink{nullptr};
extern "C" int tcvom_overflow_sink(int32_t* counters) {
    g_overflow_sink.store((int*)counters, std::memory_order_relaxed);
    return TCVOM_OK;
}
#ifdef TCVOM_F16
__device__ __forceinline__ unsigned sat8(const uint4& q) {
    // |x| == 65504 = 0x7bff in any of the 8 halves
    auto h2 = [](unsigned u) { return (unsigned)((u & 0x7fffu) == 0x7bffu) | (unsigned)(((u >> 16) & 0x7fffu) == 0x7bffu); };
    return h2(q.x) | h2(q.y) | h2(q.z) | h2(q.w);
}
#else
__device__ __forceinline__ unsigned sat8(const uint4&) { return 0u; }
#endif

template <int YF32> struct YRaw { uint4 a; };
template <> struct YRaw<1> { float4 a, b; };
template <int YF32>
__device__ __forceinline__ YRaw<YF32> load_yraw(const void* __restrict__ y, int64_t v) {
    YRaw<YF32> r;
    if constexpr (YF32 == 1) { const float4* p = reinterpret_cast<const float4*>(y) + 2 * v; r.a = p[0]; r.b = p[1]; }
    else r.a = reinterpret_cast<const uint4*>(y)[v];
    return r;
}
template <int YF32>
__device__ __forceinline__ void unpack_yraw(const YRaw<YF32>& r, float* f) {
    if constexpr (YF32 == 1) { f[0] = r.a.x; f[1] = r.a.y; f[2] = r.a.z; f[3] = r.a.w; f[4] = r.b.x; f[5] = r.b.y; f[6] = r.b.z; f[7] = r.b.w; }
    else if constexpr (YF32 == 2) unpack8_ieee(r.a, f);
    else unpack8(r.a, f);
}

// A block owns a contiguous pixel range; a thread owns ONE channel octet for the whole range (its 16 scale/shift
// values live in registers) and walks the pixels with stride 256/C8: every access is a 16-byte load/store and
// consecutive lanes cover consecutive 16-byte chunks of a pixel row.
// DUAL = 1 (the fp16 island of the bf16 build, tcvom_bn_apply_f16): z is written twice -- in the build's type to `z` (what the backward,
// the weight gradient and every consumer outside the island read) and as IEEE fp16 to `z16` (what the next forward conv of the island
// and the residual input of its block read); res1 is IEEE fp16 when res1_f16 is set.
template <int YF32, int DUAL = 0>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const void* __restrict__ y, const float* __restrict__ scale_shift,
    const uint4* __restrict__ res1, const uint4* __restrict__ res2, uint4* __restrict__ z,
    int64_t P, int C8, int C, int act, int rows_per_block, int64_t slot_stride, int* __restrict__ overflow,
    unsigned char* __restrict__ mask_out = nullptr, uint4* __restrict__ z16 = nullptr, int res1_f16 = 0)
{
    // mask_out (or NULL): one byte per 8-channel vector, bit k = the pre-activation value of channel k is positive.  The backward of a
    // site with a residual input (z = act(norm(y) + res1)) then takes the activation's slope from that bit instead of re-reading res1
    // in both of its passes: 1/16 of the residual tensor's bytes, twice (the bottleneck outputs of the FBA trunk are 400 MB each).
    const int oct = threadIdx.x % C8, prow = threadIdx.x / C8, RP = 256 / C8;
    if (prow >= RP) return;
    unsigned sat = 0u;
    // blockIdx.y = frame of a batched call: P pixels per frame, own (scale, shift) vector
    scale_shift += blockIdx.y * slot_stride;
    const int64_t fo = (int64_t)blockIdx.y * P * C8;
    float sc[8], sh[8];
    load_coef8(scale_shift + oct * 8, sc);
    load_coef8(scale_shift + C + oct * 8, sh);
    const float slope = act_slope(act), cap = act_cap(act);
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    // two-deep software pipeline: the loads of pixel row p + RP are in flight while row p is computed and stored
    int64_t p = pbeg + prow;
    if (p >= pend) return;
    int64_t v = fo + p * C8 + oct;
    YRaw<YF32> yr = load_yraw<YF32>(y, v);
    uint4 q1 = res1 ? res1[v] : uint4{0, 0, 0, 0}, q2 = res2 ? res2[v] : uint4{0, 0, 0, 0};
    while (true) {
        const int64_t pn = p + RP;
        const bool more = pn < pend;
        const int64_t vn = fo + (more ? pn : p) * C8 + oct;
        const YRaw<YF32> yn = load_yraw<YF32>(y, vn);
        const uint4 n1 = res1 ? res1[vn] : uint4{0, 0, 0, 0}, n2 = res2 ? res2[vn] : uint4{0, 0, 0, 0};
        float f[8], r1[8], r2[8];
        unpack_yraw<YF32>(yr, f);
        if constexpr (YF32 == 0) sat |= sat8(yr.a);
        if (DUAL && res1_f16) unpack8_ieee(q1, r1); else unpack8(q1, r1);
        unpack8(q2, r2);
        unsigned bits = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = f[k] * sc[k] + sh[k] + r1[k];
            bits |= x > 0.f ? (1u << k) : 0u;
            x = fminf(x > 0.f ? x : slope * x, cap);
            f[k] = x + r2[k];
        }
        z[v] = pack8(f);
        if constexpr (DUAL) z16[v] = pack8_ieee(f);
        if (mask_out) mask_out[v] = (unsigned char)bits;
        if (!more) break;
        p = pn; v = vn; yr = yn; q1 = n1; q2 = n2;
    }
    if (overflow && sat) atomicAdd(overflow + 1, 1);                  // a conv output at the fp16 saturation value (never in a healthy step)
}

// ---------------------------------------------------------------- backward, pass 1: per-channel sums
// block = 256 threads = RP pixel rows x C8 channel octets (C8 <= 256); partial[block][2][C]
// HAS3: a third addend of the incoming gradient (dz3, frames dz3_f0 .. dz3_f1 - 1 like dz2): the outputs of the encoder stages have three
// consumers (the next stage's conv1, its down-sampling branch, the shortcut branch) -- summed here in fp32 instead of by an element-wise
// pass (and its 16-bit rounding) in front of the two BatchNorm-backward kernels
template <int YF32, bool HAS3 = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const uint4* __restrict__ dz, const uint4* __restrict__ dz2, const void* __restrict__ y, const uint4* __restrict__ res1,
    const float* __restrict__ scale_shift, const float* __restrict__ saved,
    float* __restrict__ partial, int64_t P, int C8, int C, int act, int rows_per_block, int64_t slot_stride, int dz2_f0, int dz2_f1,
    int* __restrict__ overflow, const unsigned char* __restrict__ mask = nullptr, const uint4* __restrict__ dz3 = nullptr, int dz3_f0 = 0,
    int dz3_f1 = 0)
{
    // mask (or NULL): the activation bits bn_apply_kernel wrote (then res1 is NULL: the pre-activation sign comes from the bit)
    extern __shared__ float red[];        // [2][256][8]
    unsigned sat = 0u;
    const int tid = threadIdx.x;
    // dz2 may cover the frames dz2_f0 .. dz2_f1 - 1 only (a consumer that ran for the interior frames of a window): zero elsewhere
    if (dz2) dz2 = ((int)blockIdx.y >= dz2_f0 && (int)blockIdx.y < dz2_f1) ? dz2 - (int64_t)dz2_f0 * P * C8 : nullptr;
    if (HAS3 && dz3) dz3 = ((int)blockIdx.y >= dz3_f0 && (int)blockIdx.y < dz3_f1) ? dz3 - (int64_t)dz3_f0 * P * C8 : nullptr;
    scale_shift += blockIdx.y * slot_stride;                         // blockIdx.y = frame of a batched call
    saved += blockIdx.y * slot_stride;
    partial += (int64_t)blockIdx.y * gridDim.x * 2 * C;
    const int64_t fo = (int64_t)blockIdx.y * P * C8;
    const int oct = tid % C8;
    const int prow = tid / C8;
    const int RP = 256 / C8;
    const int c0 = oct * 8;
    float sg[8], sx[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sg[k] = 0.f; sx[k] = 0.f; }
    load_coef8(scale_shift + c0, sc);
    load_coef8(scale_shift + C + c0, sh);
    load_coef8(saved + c0, mu);
    load_coef8(saved + C + c0, is);
    const float slope = act_slope(act), cap = act_cap(act);
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    if (prow < RP && pbeg + prow < pend) {
        // two-deep software pipeline (see bn_apply_kernel)
        int64_t p = pbeg + prow;
        int64_t v = fo + p * C8 + oct;
        uint4 qg = dz[v], qh = dz2 ? dz2[v] : uint4{0, 0, 0, 0}, q3 = (HAS3 && dz3) ? dz3[v] : uint4{0, 0, 0, 0};
        YRaw<YF32> yr = load_yraw<YF32>(y, v);
        uint4 q1 = res1 ? res1[v] : uint4{0, 0, 0, 0};
        unsigned mb = mask ? mask[v] : 0u;
        while (true) {
            const int64_t pn = p + RP;
            const bool more = pn < pend;
            const int64_t vn = fo + (more ? pn : p) * C8 + oct;
            const uint4 ng = dz[vn], nh = dz2 ? dz2[vn] : uint4{0, 0, 0, 0}, n3 = (HAS3 && dz3) ? dz3[vn] : uint4{0, 0, 0, 0};
            const YRaw<YF32> yn = load_yraw<YF32>(y, vn);
            const uint4 n1 = res1 ? res1[vn] : uint4{0, 0, 0, 0};
            const unsigned nb = mask ? mask[vn] : 0u;
            float g[8], g2[8], yy[8], r1[8];
            unpack8(qg, g);
            unpack8(qh, g2);
            sat |= sat8(qg) | sat8(qh);
            if constexpr (HAS3) {
                float g3[8];
                unpack8(q3, g3);
                sat |= sat8(q3);
#pragma unroll
                for (int k = 0; k < 8; ++k) g2[k] += g3[k];
            }
            unpack_yraw<YF32>(yr, yy);
            unpack8(q1, r1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float pre = yy[k] * sc[k] + sh[k] + r1[k];
                const float fac = mask ? (((mb >> k) & 1u) ? 1.f : slope) : (pre > 0.f ? (pre < cap ? 1.f : 0.f) : slope);
                const float gg = (g[k] + g2[k]) * fac;
                sg[k] += gg;
                sx[k] += gg * (yy[k] - mu[k]) * is[k];
            }
            if (!more) break;
            p = pn; qg = ng; qh = nh; q3 = n3; yr = yn; q1 = n1; mb = nb;
        }
    }
    if (overflow && __any(sat != 0u) && (tid & 63) == 0) atomicAdd(overflow, 1);      // (never taken in a healthy step)
    float* r_g = red;                 // [256][8]
    float* r_x = red + 256 * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { r_g[tid * 8 + k] = sg[k]; r_x[tid * 8 + k] = sx[k]; }
    __syncthreads();
    if (tid < C8) {
        float ag[8], ax[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ag[k] = 0.f; ax[k] = 0.f; }
        for (int r = 0; r < RP; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ag[k] += r_g[(r * C8 + tid) * 8 + k];
                ax[k] += r_x[(r * C8 + tid) * 8 + k];
            }
        }
        float* po = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 8; ++k) { po[tid * 8 + k] = ag[k]; po[C + tid * 8 + k] = ax[k]; }
    }
}

struct SnDot { float* out; long long stride; float eps, scale; };         // eps < 0: eval-mode statistics
static const SnDot kNoDot = {nullptr, 0, 0.f, 0.f};
static SnDot make_dot(const tcvom_sn_dot* d) {
    if (!d || !d->out) return kNoDot;
    SnDot o = {d->out, (long long)d->frame_stride, d->training ? d->eps : -1.f, d->scale};
    return o;
}

template <typename PT>
__global__ __launch_bounds__(FIN_SL * 32) void bn_bwd_finalize_kernel(
    const PT* __restrict__ partial, int G, int C, double count,
    const float* __restrict__ gamma, const float* __restrict__ saved,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef, int accumulate, int64_t slot_stride, const BnSync sy,
    const SnDot sd)
{
    __shared__ double s1[FIN_SL][32], s2[FIN_SL][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    partial += (int64_t)blockIdx.y * G * 2 * C;                     // frame of a batched call
    saved += blockIdx.y * slot_stride;
    coef += (int64_t)blockIdx.y * 3 * C;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 4 independent chains so that the (L2-latency-bound) loads of consecutive groups overlap
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        int g = sl;
        for (; g + 3 * FIN_SL < G; g += 4 * FIN_SL) {
            const PT* q = partial + (int64_t)g * 2 * C + c;
            const PT x0 = q[0], y0 = q[C], x1 = q[2 * FIN_SL * C], y1 = q[(2 * FIN_SL + 1) * C], x2 = q[4 * FIN_SL * C], y2 = q[(4 * FIN_SL + 1) * C],
                     x3 = q[6 * FIN_SL * C], y3 = q[(6 * FIN_SL + 1) * C];
            a += (double)x0; b += (double)y0; a1 += (double)x1; b1 += (double)y1;
            a2 += (double)x2; b2 += (double)y2; a3 += (double)x3; b3 += (double)y3;
        }
        for (; g < G; g += FIN_SL) {
            a += (double)partial[(int64_t)g * 2 * C + c];
            b += (double)partial[(int64_t)g * 2 * C + C + c];
        }
        a += a1 + a2 + a3;
        b += b1 + b2 + b3;
    }
    s1[sl][cl] = a;
    s2[sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < FIN_SL; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
        // accumulate: the S calls of one BatchNorm (frames on concurrent streams) add into one gradient buffer
        // (SyncBatchNorm: gamma / beta gradients stay the LOCAL sums -- torch semantics; the gradient all-reduce averages them)
        if (dbeta) { if (accumulate) atomicAdd(dbeta + c, (float)a); else dbeta[c] = (float)a; }
        if (dgamma) { if (accumulate) atomicAdd(dgamma + c, (float)b); else dgamma[c] = (float)b; }
    }
    if (sy.peers) bn_sync_exchange(sy, s1, s2, sl, cl, c, C, blockIdx.y, a, b);      // the dx coefficients use the sums over the ranks
    if (sl == 0 && c < C) {
        // dy = gi * g - c1 - xhat * c2   (bn_bwd_apply); the same form serves GroupNorm (gn_bwd_finalize_kernel)
        const float gi = gamma[c] * saved[C + c];
        coef[c] = (float)(a / count) * gi;
        coef[C + c] = (float)(b / count) * gi;
        coef[2 * C + c] = gi;
    }
    if (sd.out && threadIdx.x < 64) {
        // <dy, y> over the frame = d(loss)/d(alpha) of y = conv(x, alpha W~) at alpha = 1 = <dW~, W~>: SpectralNorm's backward needs
        // <dW~, W_bar> = sigma <dW~, W~> (models/GCA/ops.py:25-45 through autograd) and gets it here from the two BatchNorm sums
        // instead of a pass over the weight gradient.  Per channel, with dy = gi (g - mean g - xhat mean(g xhat)):
        //   training: sum dy y = gi * sum(g xhat) * invstd * eps     (BatchNorm is scale-invariant up to eps)
        //   eval    : sum dy y = gi * (mean * sum g + sum(g xhat) / invstd)
        float d = 0.f;
        if (sl == 0 && c < C) {
            const double mean = saved[c], is = saved[C + c], gi = (double)gamma[c] * is;
            d = (float)((sd.eps >= 0.f ? gi * b * is * (double)sd.eps : gi * (mean * a + b / is)) * (double)sd.scale);
        }
        d = wave_sum(d);
        if (threadIdx.x == 0 && d != 0.f) atomicAdd(sd.out + blockIdx.y * sd.stride, d);
    }
}

template <int YF32, bool HAS3 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const uint4* __restrict__ dz, const uint4* __restrict__ dz2, const void* __restrict__ y, const uint4* __restrict__ res1,
    const float* __restrict__ scale_shift, const float* __restrict__ saved, const float* __restrict__ coef,
    uint4* __restrict__ dy, uint4* __restrict__ dres1, int64_t P, int C8, int C, int act, int training, int in_relu,
    int rows_per_block, int64_t slot_stride, int dz2_f0, int dz2_f1, const unsigned char* __restrict__ mask = nullptr,
    const uint4* __restrict__ dz3 = nullptr, int dz3_f0 = 0, int dz3_f1 = 0, uint4* __restrict__ dsum = nullptr)
{
    // dsum (HAS3 instantiations, or NULL): the incoming gradient itself, dz + dz2 + dz3 in the 16-bit type -- the gradient of a residual
    // that is added AFTER the activation (res2 of the decoder blocks), otherwise an element-wise pass over the same addends
    const int oct = threadIdx.x % C8, prow = threadIdx.x / C8, RP = 256 / C8;
    if (prow >= RP) return;
    if (dz2) dz2 = ((int)blockIdx.y >= dz2_f0 && (int)blockIdx.y < dz2_f1) ? dz2 - (int64_t)dz2_f0 * P * C8 : nullptr;
    if (HAS3 && dz3) dz3 = ((int)blockIdx.y >= dz3_f0 && (int)blockIdx.y < dz3_f1) ? dz3 - (int64_t)dz3_f0 * P * C8 : nullptr;
    scale_shift += blockIdx.y * slot_stride;                         // blockIdx.y = frame of a batched call
    saved += blockIdx.y * slot_stride;
    coef += (int64_t)blockIdx.y * 3 * C;
    const int64_t fo = (int64_t)blockIdx.y * P * C8;
    float sc[8], sh[8], mu[8], is[8], c1[8], c2[8], gi[8];
    load_coef8(scale_shift + oct * 8, sc);
    load_coef8(scale_shift + C + oct * 8, sh);
    load_coef8(saved + oct * 8, mu);
    load_coef8(saved + C + oct * 8, is);
    load_coef8(coef + oct * 8, c1);
    load_coef8(coef + C + oct * 8, c2);
    load_coef8(coef + 2 * C + oct * 8, gi);
    if (!training) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { c1[k] = 0.f; c2[k] = 0.f; }
    }
    const float slope = act_slope(act), cap = act_cap(act);
    const int64_t pbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t pend = pbeg + rows_per_block < P ? pbeg + rows_per_block : P;
    int64_t p = pbeg + prow;
    if (p >= pend) return;
    // two-deep software pipeline (see bn_apply_kernel)
    int64_t v = fo + p * C8 + oct;
    uint4 qg = dz[v], qh = dz2 ? dz2[v] : uint4{0, 0, 0, 0}, q3 = (HAS3 && dz3) ? dz3[v] : uint4{0, 0, 0, 0};
    YRaw<YF32> yr = load_yraw<YF32>(y, v);
    uint4 q1 = res1 ? res1[v] : uint4{0, 0, 0, 0};
    unsigned mb = mask ? mask[v] : 0u;
    while (true) {
        const int64_t pn = p + RP;
        const bool more = pn < pend;
        const int64_t vn = fo + (more ? pn : p) * C8 + oct;
        const uint4 ng = dz[vn], nh = dz2 ? dz2[vn] : uint4{0, 0, 0, 0}, n3 = (HAS3 && dz3) ? dz3[vn] : uint4{0, 0, 0, 0};
        const YRaw<YF32> yn = load_yraw<YF32>(y, vn);
        const uint4 n1 = res1 ? res1[vn] : uint4{0, 0, 0, 0};
        const unsigned nb = mask ? mask[vn] : 0u;
        float g[8], g2[8], yy[8], r1[8], o[8];
        unpack8(qg, g);
        unpack8(qh, g2);
        if constexpr (HAS3) {
            float g3[8];
            unpack8(q3, g3);
#pragma unroll
            for (int k = 0; k < 8; ++k) g2[k] += g3[k];
            if (dsum) {
                float t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = g[k] + g2[k];
                dsum[v] = pack8(t);
            }
        }
        unpack_yraw<YF32>(yr, yy);
        unpack8(q1, r1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float pre = yy[k] * sc[k] + sh[k] + r1[k];
            const float fac = mask ? (((mb >> k) & 1u) ? 1.f : slope) : (pre > 0.f ? (pre < cap ? 1.f : 0.f) : slope);
            const float gg = (g[k] + g2[k]) * fac;
            g[k] = gg;
            const float xh = (yy[k] - mu[k]) * is[k];
            o[k] = gi[k] * gg - c1[k] - xh * c2[k];
            if (in_relu && yy[k] <= 0.f) o[k] = 0.f;
        }
        dy[v] = pack8(o);
        if (dres1) dres1[v] = pack8(g);
        if (!more) break;
        p = pn; v = vn; qg = ng; qh = nh; q3 = n3; yr = yn; q1 = n1; mb = nb;
    }
}

// pixel rows per block for the streaming BN kernels: ~8 loop iterations per thread (fewer iterations / more blocks
// measured SLOWER: every thread first loads its 16..56 per-channel coefficients; MORE iterations for the wide layers -- 32 / 64 for
// C >= 512, round 5 -- measured slower too: FBA 47.3 -> 47.7 / 48.1 ms, GCA 23.67 -> 23.80), at most 8192 blocks
static int bn_rows_per_block(int64_t pixels, int C) {
    const int rp = 256 / (C / 8);
    int64_t rows = (int64_t)rp * 8;
    if ((pixels + rows - 1) / rows > 8192) rows = ((pixels + 8191) / 8192 + rp - 1) / rp * rp;
    return (int)rows;
}

static int stream_grid(int64_t n, int per_block) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

static const BnSync kNoSync = {nullptr, 1, 0, 0u, 0, 0, 0, nullptr, nullptr};

// host view of tcvom_bn_sync -> kernel argument
static int make_sync(const tcvom_bn_sync* s, int32_t C, int32_t nframes, BnSync* out, const char* who) {
    if (!s) { *out = kNoSync; return TCVOM_OK; }
    TCVOM_CHECK_ARG(s->peers && s->world >= 1 && s->world <= FIN_SL && s->rank >= 0 && s->rank < s->world && s->seq != 0 && s->ring >= 2,
                    "%s: bad tcvom_bn_sync (world %d, rank %d, seq %u, ring %d; at most %d ranks)", who, s->world, s->rank, s->seq, s->ring, FIN_SL);
    TCVOM_CHECK_ARG((int64_t)nframes * 2 * C <= s->capacity, "%s: %d frames x 2 x %d channels exceed the mailbox capacity of %lld doubles",
                    who, nframes, C, (long long)s->capacity);
    out->peers = (const unsigned long long* const*)s->peers;
    out->world = s->world;
    out->rank = s->rank;
    out->seq = s->seq;
    out->cap2 = 2 * s->capacity;
    out->slot_off = (long long)(s->seq % (uint32_t)s->ring) * s->world * out->cap2;
    out->timeout = s->timeout_ticks > 0 ? s->timeout_ticks : 3000000000ll;      // default 30 s
    out->status = s->status;
    out->wait = (long long*)s->wait_ticks;
    return TCVOM_OK;
}

static int bn_finalize_impl(const float* partial, int32_t groups, int32_t C, int64_t count, int64_t unbias_count,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, float* scale_shift, float* saved, double* scratch,
                            int32_t nframes, int64_t slot_stride, const BnSync& sy, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const double ub = (double)(unbias_count > 0 ? unbias_count : count);
    if (groups > 4 * BN_SLICES && scratch) {
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(cdiv(C, 32), BN_SLICES, nframes), dim3(256), 0, st, partial, groups, C, scratch);
        // the second stage reads nframes blocks of BN_SLICES double partials
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, (const double*)scratch, BN_SLICES, C,
                           (double)count, ub, gamma, beta, running_mean, running_var, momentum, eps, scale_shift, saved, slot_stride, sy);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, partial, groups, C,
                           (double)count, ub, gamma, beta, running_mean, running_var, momentum, eps, scale_shift, saved, slot_stride, sy);
    }
    TCVOM_LAUNCH_CHECK("bn_finalize");
    return TCVOM_OK;
}

extern "C" int tcvom_bn_finalize(const float* partial, int32_t groups, int32_t C, int64_t count, int64_t unbias_count,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* scale_shift, float* saved, double* scratch,
                                 int32_t nframes, int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(partial && gamma && beta && scale_shift && saved && groups > 0 && C > 0 && count > 0, "bn_finalize: bad args");
    TCVOM_CHECK_ARG(nframes >= 1 && (nframes == 1 || (!running_mean && !running_var)), "bn_finalize: running statistics of a batched call are updated by tcvom_bn_ema_multi");
    return bn_finalize_impl(partial, groups, C, count, unbias_count, gamma, beta, running_mean, running_var, momentum, eps, scale_shift,
                            saved, scratch, nframes, slot_stride, kNoSync, stream);
}

// SyncBatchNorm forward: the local partial sums are combined, exchanged with the peers INSIDE the finalize kernel (bn_sync_exchange)
// and finalised with the global count: the same launches as tcvom_bn_finalize, no collective call, no host involvement.
extern "C" int tcvom_bn_finalize_sync(const float* partial, int32_t groups, int32_t C, int64_t count, int64_t unbias_count,
                                      const float* gamma, const float* beta, float eps, float* scale_shift, float* saved,
                                      double* scratch, int32_t nframes, int64_t slot_stride, const tcvom_bn_sync* sync, void* stream) {
    TCVOM_CHECK_ARG(partial && gamma && beta && scale_shift && saved && groups > 0 && C > 0 && count > 0 && nframes >= 1 && sync,
                    "bn_finalize_sync: bad args");
    BnSync sy;
    const int rc = make_sync(sync, C, nframes, &sy, "bn_finalize_sync");
    if (rc != TCVOM_OK) return rc;
    return bn_finalize_impl(partial, groups, C, count, unbias_count, gamma, beta, nullptr, nullptr, 0.f, eps, scale_shift, saved, scratch,
                            nframes, slot_stride, sy, stream);
}
/* doubles of scratch tcvom_bn_finalize needs for C channels */
extern "C" int tcvom_bn_finalize_scratch_doubles(int32_t C) { return BN_SLICES * 2 * C; }

// Deferred running-statistics update (momentum EMA with the unbiased variance), from the (mean, invstd) a train-mode
// call saved.  Frames of a window run on concurrent streams, so the EMA of the S calls of one BatchNorm is applied
// afterwards, in call order, instead of inside bn_finalize.
__global__ void bn_ema_kernel(const float* __restrict__ saved, float* __restrict__ rm, float* __restrict__ rv, int C,
                              float momentum, float eps, double unbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = saved[c], invstd = saved[C + c];
    double var = 1.0 / ((double)invstd * (double)invstd) - (double)eps;
    if (var < 0.0) var = 0.0;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(var * unbias);
}

extern "C" int tcvom_bn_ema_update(const float* saved, float* running_mean, float* running_var, int32_t C, float momentum,
                                   float eps, int64_t unbias_count, void* stream) {
    TCVOM_CHECK_ARG(saved && running_mean && running_var && C > 0 && unbias_count > 0, "bn_ema_update: bad args");
    const double ub = unbias_count > 1 ? (double)unbias_count / (double)(unbias_count - 1) : 1.0;
    hipLaunchKernelGGL(bn_ema_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, saved, running_mean, running_var,
                       C, momentum, eps, ub);
    TCVOM_LAUNCH_CHECK("bn_ema_update");
    return TCVOM_OK;
}

// All deferred EMAs of a window in ONE launch: block b owns BatchNorm b and applies the EMA of its train-mode calls
// in call order (bit s of mask[b] = call slot s was a train-mode call).  table row: running_mean ptr, running_var
// ptr, address of slot 0's (mean, invstd), C, floats between slots, momentum/eps as two packed fp32.
struct BnEmaArgs { uint32_t mask[256]; float unbias[256]; };
__global__ __launch_bounds__(128) void bn_ema_multi_kernel(const int64_t* __restrict__ table, const BnEmaArgs args) {
    const int b = blockIdx.x;
    const uint32_t mask = args.mask[b];
    if (!mask) return;
    const int64_t* row = table + (int64_t)b * 6;
    float* rm = reinterpret_cast<float*>(row[0]);
    float* rv = reinterpret_cast<float*>(row[1]);
    const float* saved0 = reinterpret_cast<const float*>(row[2]);
    const int C = (int)row[3];
    const int64_t stride = row[4];
    const float momentum = __int_as_float((int)(row[5] & 0xffffffffll)), eps = __int_as_float((int)(row[5] >> 32));
    const double unbias = (double)args.unbias[b];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float m = rm[c], v = rv[c];
        for (int s = 0; s < 32; ++s) {
            if (!((mask >> s) & 1u)) continue;
            const float* sv = saved0 + s * stride;
            const float mean = sv[c], invstd = sv[C + c];
            double var = 1.0 / ((double)invstd * (double)invstd) - (double)eps;
            if (var < 0.0) var = 0.0;
            m = (1.f - momentum) * m + momentum * mean;
            v = (1.f - momentum) * v + momentum * (float)(var * unbias);
        }
        rm[c] = m;
        rv[c] = v;
    }
}

extern "C" int tcvom_bn_ema_multi(const int64_t* table, int32_t nbn, const uint32_t* masks, const float* unbias, void* stream) {
    TCVOM_CHECK_ARG(table && masks && unbias && nbn > 0, "bn_ema_multi: bad args");
    for (int b0 = 0; b0 < nbn; b0 += 256) {
        BnEmaArgs a;
        const int n = nbn - b0 < 256 ? nbn - b0 : 256;
        for (int i = 0; i < 256; ++i) {
            a.mask[i] = i < n ? masks[b0 + i] : 0u;
            a.unbias[i] = i < n ? unbias[b0 + i] : 1.f;
        }
        hipLaunchKernelGGL(bn_ema_multi_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, table + (int64_t)b0 * 6, a);
    }
    TCVOM_LAUNCH_CHECK("bn_ema_multi");
    return TCVOM_OK;
}

extern "C" int tcvom_bn_eval_coeffs(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* scale_shift, float* saved, void* stream) {
    TCVOM_CHECK_ARG(gamma && beta && running_mean && running_var && scale_shift && saved && C > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale_shift, saved);
    TCVOM_LAUNCH_CHECK("bn_eval_coeffs");
    return TCVOM_OK;
}

static int bn_apply_impl(const void* y, const float* scale_shift, const void* res1, const void* res2, void* z, uint8_t* mask,
                         int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes, int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(y && scale_shift && z && pixels > 0 && C > 0 && C % 8 == 0 && nframes >= 1, "bn_apply: bad args (C=%d)", C);
    TCVOM_CHECK_ARG(C <= 2048, "bn_apply: C=%d (multiples of 8 up to 2048)", C);
    const int rpb = bn_rows_per_block(pixels, C);
    const dim3 grid(cdiv(pixels, rpb), nframes);
    TCVOM_CHECK_ARG(y_fp32 == 0 || y_fp32 == 2, "bn_apply: y_fp32 = %d (0: the build's 16-bit type, 2: IEEE fp16)", y_fp32);
    switch (bn_y_mode(y_fp32)) {
    case 2: hipLaunchKernelGGL(bn_apply_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream,
                           y, scale_shift, (const uint4*)res1, (const uint4*)res2, (uint4*)z, pixels, C / 8, C, act, rpb, slot_stride,
                           g_overflow_sink.load(std::memory_order_relaxed), mask); break;
    default: hipLaunchKernelGGL(bn_apply_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream,
                           y, scale_shift, (const uint4*)res1, (const uint4*)res2, (uint4*)z, pixels, C / 8, C, act, rpb, slot_stride,
                           g_overflow_sink.load(std::memory_order_relaxed), mask); break;
    }
    TCVOM_LAUNCH_CHECK("bn_apply");
    return TCVOM_OK;
}
extern "C" int tcvom_bn_apply(const void* y, const float* scale_shift, const void* res1, const void* res2, void* z,
                              int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes, int64_t slot_stride,
                              void* stream) {
    return bn_apply_impl(y, scale_shift, res1, res2, z, nullptr, pixels, C, act, y_fp32, nframes, slot_stride, stream);
}
// ... and one byte per 8-channel vector with the signs of the pre-activation values (bit k = channel k positive), for
// tcvom_bn_bwd_reduce_mask / tcvom_bn_bwd_apply_mask: the backward of a residual site without re-reading res1
extern "C" int tcvom_bn_apply_mask(const void* y, const float* scale_shift, const void* res1, const void* res2, void* z, uint8_t* mask,
                                   int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes, int64_t slot_stride,
                                   void* stream) {
    TCVOM_CHECK_ARG(mask && act != 4, "bn_apply_mask: null mask / a capped activation (one bit cannot hold ReLU6's two thresholds)");
    return bn_apply_impl(y, scale_shift, res1, res2, z, mask, pixels, C, act, y_fp32, nframes, slot_stride, stream);
}

// The apply pass of the fp16 island (bf16 build: encoder stem, layer1, layer2 of the GCA network): y is IEEE fp16 (y_fp32 must be 2), z is
// stored in the build's type AND as IEEE fp16 (z16); res1 is IEEE fp16 when res1_f16 != 0; mask may be NULL.
extern "C" int tcvom_bn_apply_f16(const void* y, const float* scale_shift, const void* res1, int32_t res1_f16, const void* res2, void* z,
                                  void* z16, uint8_t* mask, int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes,
                                  int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(y && scale_shift && z && z16 && pixels > 0 && C > 0 && C % 8 == 0 && C <= 2048 && nframes >= 1, "bn_apply_f16: bad args (C=%d)", C);
    TCVOM_CHECK_ARG(y_fp32 == 2, "bn_apply_f16: the conv output must be IEEE fp16 (y_fp32 = 2), got %d", y_fp32);
    TCVOM_CHECK_ARG(!(mask && act == 4), "bn_apply_f16: a capped activation has no one-bit mask");
    const int rpb = bn_rows_per_block(pixels, C);
    const dim3 grid(cdiv(pixels, rpb), nframes);
    hipLaunchKernelGGL((bn_apply_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream,
                       y, scale_shift, (const uint4*)res1, (const uint4*)res2, (uint4*)z, pixels, C / 8, C, act, rpb, slot_stride,
                       g_overflow_sink.load(std::memory_order_relaxed), mask, (uint4*)z16, res1_f16);
    TCVOM_LAUNCH_CHECK("bn_apply_f16");
    return TCVOM_OK;
}

// Partial-sum groups (= blocks per frame) of the backward reduction.  A batched call of >= 3 frames stays at <= 4 * BN_SLICES groups
// per frame -- >= 768 blocks in all, enough to stream at full rate -- so that its finalize is ONE launch (more groups go through
// bn_partial_reduce first: 28 extra dependent 5 us launches per 1080p step for the os1 / os2 / os4 layers).
extern "C" int tcvom_bn_bwd_groups_n(int64_t pixels, int32_t C, int32_t nframes) {
    const int rows = 256 / (C / 8);
    int64_t per = (int64_t)rows * 8;                // >= 8 loop iterations per thread
    int64_t g = (pixels + per - 1) / per;
    if (g > 2048) g = 2048;
    if (nframes >= 3 && g > 4 * BN_SLICES) g = 4 * BN_SLICES;
    if (g < 1) g = 1;
    return (int)g;
}
extern "C" int tcvom_bn_bwd_groups(int64_t pixels, int32_t C) { return tcvom_bn_bwd_groups_n(pixels, C, 1); }

static int bn_bwd_reduce_impl(const void* dz, const void* dz2, const void* y, const void* res1, const uint8_t* mask, const float* scale_shift,
                              const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act,
                              int32_t y_fp32, int32_t nframes, int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1,
                              void* stream, const void* dz3 = nullptr, int32_t dz3_f0 = 0, int32_t dz3_f1 = 0) {
    TCVOM_CHECK_ARG(dz2_f0 >= 0 && dz2_f0 <= dz2_f1 && dz2_f1 <= nframes, "bn_bwd_reduce: dz2 frames %d..%d of %d", dz2_f0, dz2_f1, nframes);
    TCVOM_CHECK_ARG(!dz3 || (dz3_f0 >= 0 && dz3_f0 <= dz3_f1 && dz3_f1 <= nframes), "bn_bwd_reduce: dz3 frames %d..%d of %d", dz3_f0, dz3_f1, nframes);
    TCVOM_CHECK_ARG(dz && y && scale_shift && saved && partial && pixels > 0 && C >= 8 && C % 8 == 0 && C <= 2048 && nframes >= 1,
                    "bn_bwd_reduce: bad args (C=%d)", C);
    const int groups = tcvom_bn_bwd_groups_n(pixels, C, nframes);
    const int rpb = (int)((pixels + groups - 1) / groups);
    const dim3 grid(groups, nframes);
    TCVOM_CHECK_ARG(y_fp32 == 0 || y_fp32 == 2, "bn_bwd_reduce: y_fp32 = %d (0: the build's 16-bit type, 2: IEEE fp16)", y_fp32);
#define BN_BWD_REDUCE_LAUNCH(YM, H3)                                                                                              \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<YM, H3>), grid, dim3(256), 2 * 256 * 8 * sizeof(float), (hipStream_t)stream,          \
                       (const uint4*)dz, (const uint4*)dz2, y, (const uint4*)res1, scale_shift, saved, partial, pixels, C / 8, C, act, rpb, \
                       slot_stride, dz2_f0, dz2_f1, g_overflow_sink.load(std::memory_order_relaxed), mask, (const uint4*)dz3, dz3_f0, dz3_f1)
    if (bn_y_mode(y_fp32) == 2) { if (dz3) BN_BWD_REDUCE_LAUNCH(2, true); else BN_BWD_REDUCE_LAUNCH(2, false); }
    else { if (dz3) BN_BWD_REDUCE_LAUNCH(0, true); else BN_BWD_REDUCE_LAUNCH(0, false); }
#undef BN_BWD_REDUCE_LAUNCH
    TCVOM_LAUNCH_CHECK("bn_bwd_reduce");
    return TCVOM_OK;
}
extern "C" int tcvom_bn_bwd_reduce_ranged(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                                          const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act,
                                          int32_t y_fp32, int32_t nframes, int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1,
                                          void* stream) {
    return bn_bwd_reduce_impl(dz, dz2, y, res1, nullptr, scale_shift, saved, partial, pixels, C, act, y_fp32, nframes, slot_stride, dz2_f0,
                              dz2_f1, stream);
}
// the residual site's activation signs from tcvom_bn_apply_mask's bytes instead of from (y, res1)
extern "C" int tcvom_bn_bwd_reduce_mask(const void* dz, const void* dz2, const void* y, const uint8_t* mask, const float* scale_shift,
                                        const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act,
                                        int32_t y_fp32, int32_t nframes, int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1,
                                        void* stream) {
    TCVOM_CHECK_ARG(mask && act != 4, "bn_bwd_reduce_mask: null mask / capped activation");
    return bn_bwd_reduce_impl(dz, dz2, y, nullptr, mask, scale_shift, saved, partial, pixels, C, act, y_fp32, nframes, slot_stride, dz2_f0,
                              dz2_f1, stream);
}
// three addends: dz (all frames) + dz2 / dz3 (each for its frame range; NULL = absent); res1 XOR mask as in the two entries above
extern "C" int tcvom_bn_bwd_reduce3(const void* dz, const void* dz2, int32_t dz2_f0, int32_t dz2_f1, const void* dz3, int32_t dz3_f0,
                                    int32_t dz3_f1, const void* y, const void* res1, const uint8_t* mask, const float* scale_shift,
                                    const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act, int32_t y_fp32,
                                    int32_t nframes, int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(!(res1 && mask) && !(mask && act == 4), "bn_bwd_reduce3: res1 and mask exclude each other; no mask with a capped activation");
    return bn_bwd_reduce_impl(dz, dz2, y, res1, mask, scale_shift, saved, partial, pixels, C, act, y_fp32, nframes, slot_stride, dz2_f0,
                              dz2_f1, stream, dz3, dz3_f0, dz3_f1);
}
extern "C" int tcvom_bn_bwd_reduce(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                                   const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act,
                                   int32_t y_fp32, int32_t nframes, int64_t slot_stride, void* stream) {
    return tcvom_bn_bwd_reduce_ranged(dz, dz2, y, res1, scale_shift, saved, partial, pixels, C, act, y_fp32, nframes, slot_stride, 0,
                                      nframes, stream);
}

static int bn_bwd_finalize_impl(const float* partial, int32_t groups, int32_t C, int64_t count,
                                const float* gamma, const float* saved, float* dgamma, float* dbeta,
                                float* coef, double* scratch, int32_t accumulate, int32_t nframes, int64_t slot_stride,
                                const BnSync& sy, const SnDot& sd, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (groups > 4 * BN_SLICES && scratch) {
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(cdiv(C, 32), BN_SLICES, nframes), dim3(256), 0, st, partial, groups, C, scratch);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, (const double*)scratch, BN_SLICES,
                           C, (double)count, gamma, saved, dgamma, dbeta, coef, accumulate, slot_stride, sy, sd);
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<float>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, partial, groups, C,
                           (double)count, gamma, saved, dgamma, dbeta, coef, accumulate, slot_stride, sy, sd);
    }
    TCVOM_LAUNCH_CHECK("bn_bwd_finalize");
    return TCVOM_OK;
}

extern "C" int tcvom_bn_bwd_finalize(const float* partial, int32_t groups, int32_t C, int64_t count,
                                     const float* gamma, const float* saved, float* dgamma, float* dbeta,
                                     float* coef, double* scratch, int32_t accumulate, int32_t nframes, int64_t slot_stride,
                                     const tcvom_sn_dot* dot, void* stream) {
    TCVOM_CHECK_ARG(partial && gamma && saved && coef && groups > 0 && C > 0 && count > 0 && nframes >= 1, "bn_bwd_finalize: bad args");
    TCVOM_CHECK_ARG(nframes == 1 || accumulate, "bn_bwd_finalize: the frames of a batched call must ACCUMULATE dgamma/dbeta");
    return bn_bwd_finalize_impl(partial, groups, C, count, gamma, saved, dgamma, dbeta, coef, scratch, accumulate, nframes, slot_stride,
                                kNoSync, make_dot(dot), stream);
}

// SyncBatchNorm backward: (sum dy, sum dy * xhat) exchanged inside the finalize kernel; dgamma / dbeta from the LOCAL sums.
extern "C" int tcvom_bn_bwd_finalize_sync(const float* partial, int32_t groups, int32_t C, int64_t count,
                                          const float* gamma, const float* saved, float* dgamma, float* dbeta,
                                          float* coef, double* scratch, int32_t accumulate, int32_t nframes, int64_t slot_stride,
                                          const tcvom_bn_sync* sync, const tcvom_sn_dot* dot, void* stream) {
    TCVOM_CHECK_ARG(partial && gamma && saved && coef && groups > 0 && C > 0 && count > 0 && nframes >= 1 && sync, "bn_bwd_finalize_sync: bad args");
    TCVOM_CHECK_ARG(nframes == 1 || accumulate, "bn_bwd_finalize_sync: the frames of a batched call must ACCUMULATE dgamma/dbeta");
    BnSync sy;
    const int rc = make_sync(sync, C, nframes, &sy, "bn_bwd_finalize_sync");
    if (rc != TCVOM_OK) return rc;
    return bn_bwd_finalize_impl(partial, groups, C, count, gamma, saved, dgamma, dbeta, coef, scratch, accumulate, nframes, slot_stride, sy,
                                make_dot(dot), stream);
}

// ---------------------------------------------------------------- GroupNorm (FBA base: models/FBA/layers_WS.py:26-27)
// The conv epilogue's per-channel (sum, sum of squares) partials of ONE sample are combined over the channels of
// a group; the per-channel (scale, shift) / (mean, invstd) vectors then drive the same apply kernels as BatchNorm.
// One block per sample ("frame" of a batched call).
// One block per (group, sample): thread t sums channel (t % cpg) of the group over the partial slices t / cpg, ...
// (cpg = channels per group: 2 .. 64, a power of two here), then the block combines the group.
template <typename PT>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const PT* __restrict__ partial, int G, int C, int ngroups, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale_shift, float* __restrict__ saved, int64_t slot_stride) {
    __shared__ double ra[256], rb[256];
    const int grp = blockIdx.x, frame = blockIdx.y;
    const int cpg = C / ngroups;
    partial += (int64_t)frame * G * 2 * C;
    scale_shift += frame * slot_stride;
    saved += frame * slot_stride;
    const int lanes = 256 / cpg > 0 ? 256 / cpg : 1;       // slice lanes per channel (cpg <= 256)
    double a = 0.0, b = 0.0;
    for (int cc = threadIdx.x % cpg; cc < cpg; cc += 256) {
        const int c = grp * cpg + cc;
        for (int g = threadIdx.x / cpg; g < G; g += lanes) {
            a += (double)partial[(int64_t)g * 2 * C + c];
            b += (double)partial[(int64_t)g * 2 * C + C + c];
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
        __syncthreads();
    }
    const double m = count * cpg, mean = ra[0] / m;
    double var = rb[0] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float gm = (float)mean, gr = (float)(1.0 / sqrt(var + (double)eps));
    for (int cc = threadIdx.x; cc < cpg; cc += 256) {
        const int c = grp * cpg + cc;
        const float sc = gamma[c] * gr;
        scale_shift[c] = sc;
        scale_shift[C + c] = beta[c] - gm * sc;
        saved[c] = gm;
        saved[C + c] = gr;
    }
}

// partial: per-channel (sum g, sum g * xhat) of bn_bwd_reduce.  dx = rstd * (gamma * g - G1/m - xhat * G2/m) with the
// group sums G1 = sum_c gamma_c S1_c, G2 = sum_c gamma_c S2_c, m = pixels * channels per group.  One block per (group, sample).
template <typename PT>
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const PT* __restrict__ partial, int G, int C, int ngroups, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ saved,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, int64_t slot_stride) {
    __shared__ double ra[256], rb[256];
    __shared__ double cs[256], cq[256];                    // per-channel totals of the group (cpg <= 256)
    const int grp = blockIdx.x, frame = blockIdx.y;
    const int cpg = C / ngroups;
    partial += (int64_t)frame * G * 2 * C;
    saved += frame * slot_stride;
    coef += (int64_t)frame * 3 * C;
    const int lanes = 256 / cpg > 0 ? 256 / cpg : 1;
    const int cc0 = threadIdx.x % cpg, l0 = threadIdx.x / cpg;
    double a = 0.0, b = 0.0;
    if (l0 < lanes) {
        const int c = grp * cpg + cc0;
        for (int g = l0; g < G; g += lanes) {
            a += (double)partial[(int64_t)g * 2 * C + c];
            b += (double)partial[(int64_t)g * 2 * C + C + c];
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < cpg) {                          // channel totals: sum over the slice lanes
        double sa = 0.0, sb = 0.0;
        for (int l = 0; l < lanes; ++l) { sa += ra[l * cpg + threadIdx.x]; sb += rb[l * cpg + threadIdx.x]; }
        cs[threadIdx.x] = sa;
        cq[threadIdx.x] = sb;
        const int c = grp * cpg + threadIdx.x;
        if (dbeta) atomicAdd(dbeta + c, (float)sa);
        if (dgamma) atomicAdd(dgamma + c, (float)sb);
    }
    __syncthreads();
    double g1 = 0.0, g2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
        const double gmm = (double)gamma[grp * cpg + j];
        g1 += gmm * cs[j];
        g2 += gmm * cq[j];
    }
    const double m = count * cpg;
    if ((int)threadIdx.x < cpg) {
        const int c = grp * cpg + threadIdx.x;
        const float rstd = saved[C + c];
        coef[c] = rstd * (float)(g1 / m);
        coef[C + c] = rstd * (float)(g2 / m);
        coef[2 * C + c] = gamma[c] * rstd;
    }
}

extern "C" int tcvom_gn_finalize(const float* partial, int32_t groups, int32_t C, int64_t count, int32_t num_groups,
                                 const float* gamma, const float* beta, float eps, float* scale_shift, float* saved,
                                 double* scratch, int32_t nframes, int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(partial && gamma && beta && scale_shift && saved && groups > 0 && count > 0 && nframes >= 1, "gn_finalize: bad args");
    TCVOM_CHECK_ARG(C > 0 && num_groups > 0 && C % num_groups == 0 && C / num_groups <= 256 && 256 % (C / num_groups) == 0, "gn_finalize: C=%d groups=%d", C, num_groups);
    hipStream_t st = (hipStream_t)stream;
    if (groups > 4 * BN_SLICES && scratch) {
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(cdiv(C, 32), BN_SLICES, nframes), dim3(256), 0, st, partial, groups, C, scratch);
        hipLaunchKernelGGL(gn_finalize_kernel<double>, dim3(num_groups, nframes), dim3(256), 0, st, (const double*)scratch, BN_SLICES, C, num_groups,
                           (double)count, gamma, beta, eps, scale_shift, saved, slot_stride);
    } else {
        hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(num_groups, nframes), dim3(256), 0, st, partial, groups, C, num_groups, (double)count,
                           gamma, beta, eps, scale_shift, saved, slot_stride);
    }
    TCVOM_LAUNCH_CHECK("gn_finalize");
    return TCVOM_OK;
}

extern "C" int tcvom_gn_bwd_finalize(const float* partial, int32_t groups, int32_t C, int64_t count, int32_t num_groups,
                                     const float* gamma, const float* saved, float* dgamma, float* dbeta, float* coef,
                                     double* scratch, int32_t nframes, int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(partial && gamma && saved && coef && groups > 0 && count > 0 && nframes >= 1, "gn_bwd_finalize: bad args");
    TCVOM_CHECK_ARG(C > 0 && num_groups > 0 && C % num_groups == 0 && C / num_groups <= 256 && 256 % (C / num_groups) == 0, "gn_bwd_finalize: C=%d groups=%d", C, num_groups);
    hipStream_t st = (hipStream_t)stream;
    if (groups > 4 * BN_SLICES && scratch) {
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(cdiv(C, 32), BN_SLICES, nframes), dim3(256), 0, st, partial, groups, C, scratch);
        hipLaunchKernelGGL(gn_bwd_finalize_kernel<double>, dim3(num_groups, nframes), dim3(256), 0, st, (const double*)scratch, BN_SLICES, C, num_groups,
                           (double)count, gamma, saved, dgamma, dbeta, coef, slot_stride);
    } else {
        hipLaunchKernelGGL(gn_bwd_finalize_kernel<float>, dim3(num_groups, nframes), dim3(256), 0, st, partial, groups, C, num_groups, (double)count,
                           gamma, saved, dgamma, dbeta, coef, slot_stride);
    }
    TCVOM_LAUNCH_CHECK("gn_bwd_finalize");
    return TCVOM_OK;
}

// ---------------------------------------------------------------- cross-rank (SyncBatchNorm) statistics
// The reference converts every BatchNorm to SyncBatchNorm before DDP (train_ddp.py:213): the per-channel sums are
// added over the ranks before mean / variance (forward) and before the dx coefficients (backward).  The host does
// the all-reduce (RCCL) on a [2][C] fp64 vector between these calls.
template <typename PT>
__global__ __launch_bounds__(FIN_SL * 32) void bn_sums_kernel(const PT* __restrict__ partial, int G, int C, double* __restrict__ sums) {
    __shared__ double s1[FIN_SL][32], s2[FIN_SL][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    partial += (int64_t)blockIdx.y * G * 2 * C;                     // frame of a batched call: sums[frame][2][C]
    sums += (int64_t)blockIdx.y * 2 * C;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 4 independent chains so that the (L2-latency-bound) loads of consecutive groups overlap
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        int g = sl;
        for (; g + 3 * FIN_SL < G; g += 4 * FIN_SL) {
            const PT* q = partial + (int64_t)g * 2 * C + c;
            const PT x0 = q[0], y0 = q[C], x1 = q[2 * FIN_SL * C], y1 = q[(2 * FIN_SL + 1) * C], x2 = q[4 * FIN_SL * C], y2 = q[(4 * FIN_SL + 1) * C],
                     x3 = q[6 * FIN_SL * C], y3 = q[(6 * FIN_SL + 1) * C];
            a += (double)x0; b += (double)y0; a1 += (double)x1; b1 += (double)y1;
            a2 += (double)x2; b2 += (double)y2; a3 += (double)x3; b3 += (double)y3;
        }
        for (; g < G; g += FIN_SL) {
            a += (double)partial[(int64_t)g * 2 * C + c];
            b += (double)partial[(int64_t)g * 2 * C + C + c];
        }
        a += a1 + a2 + a3;
        b += b1 + b2 + b3;
    }
    s1[sl][cl] = a;
    s2[sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < FIN_SL; ++k) { a += s1[k][cl]; b += s2[k][cl]; }
        sums[c] = a;
        sums[C + c] = b;
    }
}

__global__ void bn_local_grads_kernel(const double* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                      int accumulate, int nframes) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;                                        // the frames of a batched call are calls of ONE BatchNorm
    for (int f = 0; f < nframes; ++f) { a += sums[(int64_t)f * 2 * C + c]; b += sums[(int64_t)f * 2 * C + C + c]; }
    if (dbeta) { if (accumulate) atomicAdd(dbeta + c, (float)a); else dbeta[c] = (float)a; }
    if (dgamma) { if (accumulate) atomicAdd(dgamma + c, (float)b); else dgamma[c] = (float)b; }
}

extern "C" int tcvom_bn_reduce_sums(const float* partial, int32_t groups, int32_t C, double* sums, double* scratch,
                                    int32_t nframes, void* stream) {
    TCVOM_CHECK_ARG(partial && sums && groups > 0 && C > 0 && nframes >= 1, "bn_reduce_sums: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (groups > 4 * BN_SLICES && scratch) {
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(cdiv(C, 32), BN_SLICES, nframes), dim3(256), 0, st, partial, groups, C, scratch);
        hipLaunchKernelGGL(bn_sums_kernel<double>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, (const double*)scratch, BN_SLICES, C, sums);
    } else {
        hipLaunchKernelGGL(bn_sums_kernel<float>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, partial, groups, C, sums);
    }
    TCVOM_LAUNCH_CHECK("bn_reduce_sums");
    return TCVOM_OK;
}

extern "C" int tcvom_bn_finalize_sums(const double* sums, int32_t C, int64_t count, int64_t unbias_count, const float* gamma,
                                      const float* beta, float eps, float* scale_shift, float* saved, int32_t nframes,
                                      int64_t slot_stride, void* stream) {
    TCVOM_CHECK_ARG(sums && gamma && beta && scale_shift && saved && C > 0 && count > 0 && nframes >= 1, "bn_finalize_sums: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, (hipStream_t)stream, sums, 1, C, (double)count,
                       (double)(unbias_count > 0 ? unbias_count : count), gamma, beta, (float*)nullptr, (float*)nullptr, 0.f, eps,
                       scale_shift, saved, slot_stride, kNoSync);
    TCVOM_LAUNCH_CHECK("bn_finalize_sums");
    return TCVOM_OK;
}

extern "C" int tcvom_bn_bwd_finalize_sums(const double* sums_all, const double* sums_local, int32_t C, int64_t count,
                                          const float* gamma, const float* saved, float* dgamma, float* dbeta, float* coef,
                                          int32_t accumulate, int32_t nframes, int64_t slot_stride, const tcvom_sn_dot* dot, void* stream) {
    TCVOM_CHECK_ARG(sums_all && sums_local && gamma && saved && coef && C > 0 && count > 0 && nframes >= 1, "bn_bwd_finalize_sums: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3(cdiv(C, 32), nframes), dim3(FIN_SL * 32), 0, st, sums_all, 1, C, (double)count, gamma,
                       saved, (float*)nullptr, (float*)nullptr, coef, 0, slot_stride, kNoSync, make_dot(dot));
    // gamma / beta gradients stay LOCAL sums (torch SyncBatchNorm semantics); the gradient all-reduce averages them
    hipLaunchKernelGGL(bn_local_grads_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, sums_local, C, dgamma, dbeta, accumulate, nframes);
    TCVOM_LAUNCH_CHECK("bn_bwd_finalize_sums");
    return TCVOM_OK;
}

static int bn_bwd_apply_impl(const void* dz, const void* dz2, const void* y, const void* res1, const uint8_t* mask, const float* scale_shift,
                             const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                             int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                             int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream, const void* dz3 = nullptr,
                             int32_t dz3_f0 = 0, int32_t dz3_f1 = 0, void* dsum = nullptr) {
    TCVOM_CHECK_ARG(dz2_f0 >= 0 && dz2_f0 <= dz2_f1 && dz2_f1 <= nframes, "bn_bwd_apply: dz2 frames %d..%d of %d", dz2_f0, dz2_f1, nframes);
    TCVOM_CHECK_ARG(!dz3 || (dz3_f0 >= 0 && dz3_f0 <= dz3_f1 && dz3_f1 <= nframes), "bn_bwd_apply: dz3 frames %d..%d of %d", dz3_f0, dz3_f1, nframes);
    TCVOM_CHECK_ARG(dz && y && scale_shift && saved && coef && dy && pixels > 0 && C % 8 == 0 && nframes >= 1, "bn_bwd_apply: bad args");
    TCVOM_CHECK_ARG(C <= 2048, "bn_bwd_apply: C=%d (multiples of 8 up to 2048)", C);
    const int rpb = bn_rows_per_block(pixels, C);
    const dim3 grid(cdiv(pixels, rpb), nframes);
    TCVOM_CHECK_ARG(y_fp32 == 0 || y_fp32 == 2, "bn_bwd_apply: y_fp32 = %d (0: the build's 16-bit type, 2: IEEE fp16)", y_fp32);
#define BN_BWD_APPLY_LAUNCH(YM, H3)                                                                                               \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<YM, H3>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)dz, (const uint4*)dz2, y, \
                       (const uint4*)res1, scale_shift, saved, coef, (uint4*)dy, (uint4*)dres1, pixels, C / 8, C, act, training,    \
                       in_relu, rpb, slot_stride, dz2_f0, dz2_f1, mask, (const uint4*)dz3, dz3_f0, dz3_f1, (uint4*)dsum)
    if (bn_y_mode(y_fp32) == 2) { if (dz3 || dsum) BN_BWD_APPLY_LAUNCH(2, true); else BN_BWD_APPLY_LAUNCH(2, false); }
    else { if (dz3 || dsum) BN_BWD_APPLY_LAUNCH(0, true); else BN_BWD_APPLY_LAUNCH(0, false); }
#undef BN_BWD_APPLY_LAUNCH
    TCVOM_LAUNCH_CHECK("bn_bwd_apply");
    return TCVOM_OK;
}
extern "C" int tcvom_bn_bwd_apply_ranged(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                                         const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                                         int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                                         int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream) {
    return bn_bwd_apply_impl(dz, dz2, y, res1, nullptr, scale_shift, saved, coef, dy, dres1, pixels, C, act, training, in_relu, y_fp32,
                             nframes, slot_stride, dz2_f0, dz2_f1, stream);
}
extern "C" int tcvom_bn_bwd_apply_mask(const void* dz, const void* dz2, const void* y, const uint8_t* mask, const float* scale_shift,
                                       const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                                       int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                                       int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream) {
    TCVOM_CHECK_ARG(mask && act != 4, "bn_bwd_apply_mask: null mask / capped activation");
    return bn_bwd_apply_impl(dz, dz2, y, nullptr, mask, scale_shift, saved, coef, dy, dres1, pixels, C, act, training, in_relu, y_fp32,
                             nframes, slot_stride, dz2_f0, dz2_f1, stream);
}
extern "C" int tcvom_bn_bwd_apply3(const void* dz, const void* dz2, int32_t dz2_f0, int32_t dz2_f1, const void* dz3, int32_t dz3_f0,
                                   int32_t dz3_f1, const void* y, const void* res1, const uint8_t* mask, const float* scale_shift,
                                   const float* saved, const float* coef, void* dy, void* dres1, void* dsum, int64_t pixels, int32_t C,
                                   int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes, int64_t slot_stride,
                                   void* stream) {
    TCVOM_CHECK_ARG(!(res1 && mask) && !(mask && act == 4), "bn_bwd_apply3: res1 and mask exclude each other; no mask with a capped activation");
    TCVOM_CHECK_ARG(!dsum || ((!dz2 || (dz2_f0 == 0 && dz2_f1 == nframes)) && (!dz3 || (dz3_f0 == 0 && dz3_f1 == nframes)) && dsum != dz),
                    "bn_bwd_apply3: dsum with whole-tensor addends only");
    return bn_bwd_apply_impl(dz, dz2, y, res1, mask, scale_shift, saved, coef, dy, dres1, pixels, C, act, training, in_relu, y_fp32,
                             nframes, slot_stride, dz2_f0, dz2_f1, stream, dz3, dz3_f0, dz3_f1, dsum);
}
extern "C" int tcvom_bn_bwd_apply(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                                  const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                                  int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                                  int64_t slot_stride, void* stream) {
    return tcvom_bn_bwd_apply_ranged(dz, dz2, y, res1, scale_shift, saved, coef, dy, dres1, pixels, C, act, training, in_relu, y_fp32,
                                     nframes, slot_stride, 0, nframes, stream);
}
