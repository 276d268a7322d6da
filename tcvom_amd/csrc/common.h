// Shared device/host helpers for libtcvom_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tcvom_hip.h"

// ---- the 16-bit storage type of activations and packed weights ("h16" below)
// One library per type, same ABI: libtcvom_hip.so stores bf16 (8 exponent / 7 mantissa bits), libtcvom_hip_f16.so (built with
// -DTCVOM_F16) stores IEEE fp16 (5 / 10).  Same MFMA rate (v_mfma_f32_32x32x16_{bf16,f16}), same bytes; fp32 accumulation,
// statistics, softmax and losses in both.  fp16's three extra mantissa bits lower the storage-rounding floor of the alpha matte
// ~35x (tests/test_bf16_noise_floor.py); its narrow exponent is why the fp16 build runs the backward under a loss scale
// (tcvom_amd/ops.py: LOSS_SCALE).  tcvom_act_dtype() tells the caller which one it loaded.
typedef unsigned short h16raw;  // bit pattern of one stored element
#ifdef TCVOM_F16
typedef _Float16 act_t;
#define mfma16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define dot2_h16 __builtin_amdgcn_fdot2
#else
typedef __bf16 act_t;
#define mfma16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define dot2_h16 __builtin_amdgcn_fdot2_f32_bf16
#endif
#ifdef TCVOM_F16
#define TCVOM_BUILD_F16 1
#else
#define TCVOM_BUILD_F16 0
#endif
typedef __attribute__((ext_vector_type(8))) act_t h16x8_t;
typedef __attribute__((ext_vector_type(2))) act_t h16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- error reporting across the C ABI (no exceptions cross the boundary) ----
extern thread_local char g_tcvom_err[512];
int tcvom_fail(int code, const char* fmt, ...);
#define TCVOM_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return tcvom_fail(TCVOM_ERR_ARG, __VA_ARGS__); } while (0)
#define TCVOM_LAUNCH_CHECK(name) \
    do { hipError_t e_ = hipGetLastError(); \
         if (e_ != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// ---- h16 <-> f32 ----
#ifdef TCVOM_F16
__device__ __forceinline__ float h2f(h16raw h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float hlo(unsigned u) { return (float)__builtin_bit_cast(h16x2_t, u).x; }
__device__ __forceinline__ float hhi(unsigned u) { return (float)__builtin_bit_cast(h16x2_t, u).y; }
// RNE with saturation to the largest finite value (65504): an overflow must not turn into inf / NaN downstream
__device__ __forceinline__ float h16_clamp(float a) { return __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); }      // one v_med3_f32
__device__ __forceinline__ unsigned pack2h(float a, float b) {
    h16x2_t v = {(act_t)h16_clamp(a), (act_t)h16_clamp(b)};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ h16raw f2h(float a) {
    act_t v = (act_t)h16_clamp(a);
    return __builtin_bit_cast(h16raw, v);
}
#else
__device__ __forceinline__ float h2f(h16raw h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float hlo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2h(float a, float b) {
    h16x2_t v = {(act_t)a, (act_t)b};           // RNE; lowers to v_cvt_pk_bf16_f32 on gfx950
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ h16raw f2h(float a) {
    act_t v = (act_t)a;
    return __builtin_bit_cast(h16raw, v);
}
#endif
// two values as IEEE fp16 (round to nearest even, saturated at the largest finite value), whatever the build's storage type: the conv
// outputs of the high-precision layers of the bf16 build (tcvom_conv_desc.out_fp32 = 2)
__device__ __forceinline__ unsigned pack2_ieee(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t_;
    const f16x2_t_ v = {(_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ h16raw f2h_ieee(float a) {
    const _Float16 v = (_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    return __builtin_bit_cast(h16raw, v);
}
__device__ __forceinline__ void unpack8_ieee(const uint4& q, float* f) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t_;
    const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2_t_ h = __builtin_bit_cast(f16x2_t_, u[i]);
        f[2 * i] = (float)h.x;
        f[2 * i + 1] = (float)h.y;
    }
}
__device__ __forceinline__ uint4 pack8_ieee(const float* f) {
    uint4 q;
    q.x = pack2_ieee(f[0], f[1]); q.y = pack2_ieee(f[2], f[3]);
    q.z = pack2_ieee(f[4], f[5]); q.w = pack2_ieee(f[6], f[7]);
    return q;
}
// ---- the "fp16 island" of the bf16 build (tcvom_conv_desc.in_f16, gca_net.py: F16_ISLAND): the forward convs of the encoder stem,
// layer1 and layer2 take IEEE fp16 activations and IEEE fp16 packed weights whatever the build stores -- same MFMA rate, same bytes,
// three more significant bits in every stored value of the layers that inject >= 99 % of the storage noise of the alpha matte.
// XF = 1 selects v_mfma_f32_32x32x16_f16 on the raw operand bits; XF = 0 (and every XF in the fp16 build) the build's own instruction.
template <int XF>
__device__ __forceinline__ f32x16_t mfma16x(h16x8_t a, h16x8_t b, f32x16_t c) {
#ifndef TCVOM_F16
    if constexpr (XF) {
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t_;
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t_, a), __builtin_bit_cast(f16x8_t_, b), c, 0, 0, 0);
    } else
#endif
    return mfma16(a, b, c, 0, 0, 0);
}
// ... and the 16-bit store of such a kernel: IEEE fp16 in XF mode, the build's type otherwise
template <int XF>
__device__ __forceinline__ unsigned pack2x(float a, float b) {
    if constexpr (XF) return pack2_ieee(a, b); else return pack2h(a, b);
}
__device__ __forceinline__ void unpack8(const uint4& q, float* f) {
    f[0] = hlo(q.x); f[1] = hhi(q.x); f[2] = hlo(q.y); f[3] = hhi(q.y);
    f[4] = hlo(q.z); f[5] = hhi(q.z); f[6] = hlo(q.w); f[7] = hhi(q.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 q;
    q.x = pack2h(f[0], f[1]); q.y = pack2h(f[2], f[3]);
    q.z = pack2h(f[4], f[5]); q.w = pack2h(f[6], f[7]);
    return q;
}

// ---- wave / block reductions (wave = 64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over a 256-thread block; result valid in every thread.  `red` = 4+ floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- 8 sums over the 32 lanes of each half wave, as a HALVING butterfly: the statistics epilogues of the conv / GEMM kernels hold, per
// lane, the (sum, sum of squares) of 4 consecutive channels over the lane's own pixels: t[0..3] = sums, t[4..7] = sums of squares.
// Stage 1 (lane ^ 1) leaves every lane with 4 of the 8 values summed over its pair, stage 2 (lane ^ 2) with 2 values summed over its
// quad -- index 4 b0 + 2 b1 + {0, 1} for lane bits b0, b1 -- which two row rotations (by 4 and by 8 lanes) and one 16-lane swap then
// sum over the 8 quads: 12 cross-lane operations instead of the 40 of a full butterfly on all 8 values (5 dependent DPP stages x 8
// values were 8 us of a 256 x 256 tile's epilogue).  Result: EVERY lane of the half wave holds the two totals of its (b0, b1):
//   b0 == 0: out[0], out[1] = sums of channels 2 b1, 2 b1 + 1;     b0 == 1: sums of SQUARES of channels 2 b1, 2 b1 + 1.
#define H16_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (x)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ void reduce8_pairs(const float (&t)[8], int lane, float& o0, float& o1) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float send = b0 ? t[i] : t[4 + i], keep = b0 ? t[4 + i] : t[i];
        k[i] = keep + H16_DPP(send, 0xB1);                  // quad_perm [1,0,3,2]: lane ^ 1
    }
    float m[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = b1 ? k[i] : k[2 + i], keep = b1 ? k[2 + i] : k[i];
        m[i] = keep + H16_DPP(send, 0x4E);                  // quad_perm [2,3,0,1]: lane ^ 2
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        m[i] += H16_DPP(m[i], 0x124);                       // row_ror:4  -> two quads of the row
        m[i] += H16_DPP(m[i], 0x128);                       // row_ror:8  -> the four quads of the 16-lane row
        m[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, m[i]), 0x401F));   // lane ^ 16
    }
    o0 = m[0];
    o1 = m[1];
}
// ... and the store: stats row `sp` ([2][K] floats: sums, then sums of squares), `mrow` = first of the lane's 4 channels
__device__ __forceinline__ void reduce8_store(const float (&t)[8], int lane, float* sp, int K, int mrow, bool valid) {
    float o0, o1;
    reduce8_pairs(t, lane, o0, o1);
    if ((lane & 28) == 0 && valid)                          // the four lanes of quad 0 of each half wave (lanes 0..3, 32..35)
        *reinterpret_cast<float2*>(sp + ((lane & 1) ? K : 0) + mrow + (lane & 2)) = make_float2(o0, o1);
}

// ---- gfx950 LDS transposing reads (ds_read_b64_tr_b16) issued as inline asm
// The compiler treats the ds_read_tr builtin as a possible reader of what an in-flight global_load_lds writes and puts
// s_waitcnt vmcnt(0) in front of the first one: the next stage's DMA then never overlaps this stage's MFMAs.  The reduction
// loop therefore issues the transposing reads as inline asm (invisible to the waitcnt pass) and counts lgkmcnt itself:
// tr_issue (2 x ds_read_b64_tr_b16), one s_waitcnt lgkmcnt(0), tr_fence per fragment (an empty asm the MFMA depends on, so
// that it cannot be scheduled above the wait).
typedef __attribute__((ext_vector_type(2))) unsigned int tr_u32x2_t;
struct TrFrag { tr_u32x2_t lo, hi; };
__device__ __forceinline__ void tr_issue(TrFrag& f, const h16raw* p_lo, const h16raw* p_hi) {
    typedef __attribute__((address_space(3))) const void* lp_t;
    const unsigned a_lo = (unsigned)(uintptr_t)(lp_t)p_lo, a_hi = (unsigned)(uintptr_t)(lp_t)p_hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a_lo));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a_hi));
}
__device__ __forceinline__ void tr_fence(TrFrag& f) { asm volatile("" : "+v"(f.lo), "+v"(f.hi)); }
__device__ __forceinline__ h16x8_t tr_value(const TrFrag& f) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const u32x4_t v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3);
    return __builtin_bit_cast(h16x8_t, v);
}
// the same with the k-step / row displacement as an immediate: lo at addr + OFF, hi (k rows + 4) at addr + OFF + HI bytes
template <int OFF, int HI>
__device__ __forceinline__ void tr_issue_imm(TrFrag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(addr), "n"(OFF + HI));
}

