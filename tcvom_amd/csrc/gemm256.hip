// Dense NT GEMM on 256x256 tiles with two staggered wave groups, for the attention GEMMs of GuidedCxtAtten
// (models/GCA/ops.py:177,204 and their gradients: 8160 x 8160 x 576, 8160 x 2048 x 8160 at 1080p):
//
//     out[n][m] = epilogue( sum_k B[n][k] * A[m][k] )          A: [M][K] (k contiguous), B: [N][K], out: [N][ldo]
//
// The single-phase loop of igemm_nt (all 8 waves: wait -> barrier -> DMA burst -> ds_reads -> MFMAs) leaves the matrix
// pipe idle while the texture path is saturated by the DMA burst and vice versa (traced: 53 % MFMA issue at 256x256).
// Here the K-tile (64 deep) is processed in 4 phases -- the 4 quadrants (64 x 32) of a wave's 128 x 64 output tile --
// each a LOAD section (ds_reads of the operand sub-tiles the quadrant newly needs) and an MFMA section (8 MFMAs
// 32x32x16), separated by workgroup barriers.  Waves 4..7 execute ONE EXTRA barrier up front, so at any time one
// group (one wave per SIMD) is in its MFMA section while the other (its SIMD neighbour) is in a LOAD section:
//
//     barrier interval     8t    8t+1  8t+2  8t+3  8t+4  8t+5  8t+6  8t+7  8t+8
//     group 0 (waves 0-3)  L1    M1*   L2    M2*   L3    M3    L4    M4w   L1'          * issues the LDS-DMA of K-tile t+1
//     group 1 (waves 4-7)  M4    L1    M1*   L2    M2*   L3    M3    L4w   M4           w ends with s_waitcnt vmcnt(0)
//
// LDS: two K-tile buffers (2 x (256 + 256) rows x 128 B = 128 KiB), rows unpadded, 16-byte chunk c of row r stored in slot
// c ^ ((r >> 1) & 7) (applied on the DMA source side, as in igemm_nt).  Hazards: buffer (t+1) & 1 held K-tile t-1, whose
// last reads (group 1, L4) are retired by the lgkmcnt(0) at the head of its M4 in interval 8t -- the DMA of tile t+1
// is first issued in interval 8t+1; it must have landed before interval 8t+8, where group 0 starts reading it: every
// wave drains vmcnt before the barrier that ends interval 8t+7.
#include <cstdlib>
#include <type_traits>
#include "common.h"

struct Gemm256Args {
    const h16raw* A;
    const h16raw* B;
    void* out;
    const float* bias;
    const float* mscale;
    const float* mdiag;
    const h16raw* zero_page;
    int M, N, K, ldo, act, out_fp32, batch;
    long long a_bstride, b_bstride, out_bstride, vec_bstride;
    // fused softmax backward (GuidedCxtAtten, tcvom_gca_dp_softmax_bwd): out = bf16( P * (acc - delta[n]) * mscale[m] ), zeros
    // in the padding columns M <= m < ldo.  P has the layout of `out`, delta is [batch][N].
    const h16raw* P;
    const float* delta;
    // optional transposed copies written by the same epilogue: Tt[b][m][n] = out[b][n][m], Pt[b][m][n] = P[b][n][m] ([ldt][ldt] per
    // batch entry): the operands of the dV = P^T dO and M' = T^T G GEMMs, which otherwise cost two N x N transpose passes
    h16raw* Tt;
    h16raw* Pt;
    int ldt;
    // paired launch (gemm_nt256_pair): grid z in [batch, 2 batch) multiplies the SAME A with B2 into out2
    const h16raw* B2;
    void* out2;
    long long b2_bstride;
    // EPI 3 (tcvom_gca_scores_softmax): out = bf16( exp(S - tile row max) ), stats[b][n][tile_m][2] = (tile row max, sum of exps)
    float* stats;
    // EPI 1 (16-bit output, 256-row tiles): per-channel (sum, sum of squares) of the tile's outputs for the BatchNorm / GroupNorm that
    // follows a 1 x 1 conv -- the layout igemm_nt<256,256> writes: cstats[group][2][M], group = cstats_goff + z * cstats_bstride +
    // 4 * (pixel tile) + wave column, one group per wave column (64 pixels), reduced by bn_finalize / gn_finalize
    float* cstats;
    int cstats_goff;
    long long cstats_bstride;
    // K-split tail (fp32 output, linear epilogue only): flat_nx > 0 -> 1-D grid of split_r * split_s + (tiles - split_r) workgroups
    // over the nx x ny x nz tiles in x-fastest order; the LAST split_r tiles are each computed by split_s workgroups (launched
    // first), one per 1/split_s of the reduction, and added into the zeroed output with fp32 atomics.  576 tiles (the paired
    // dq / dk product at 1080p) are 2.25 rounds of 256 workgroups: 512 whole tiles + 64 x 4 quarter-length ones instead of 3 rounds.
    int flat_nx, flat_ny, split_r, split_s;
    // k-major B operand (template BT = 1): B[k][n] with `ldb` elements per k row and `krows` valid rows (zeros
    // beyond) instead of B[n][k] -- the N x N matrices P and T of GuidedCxtAtten's backward are read as they lie in memory by the
    // products that contract their ROW index (dV = P^T dO, M' = T^T G): no transposed copies P^T / T^T (2 x 400 MB written by
    // the softmax-backward epilogue and read back per 3-frame launch at 1080p)
    int ldb, krows;
    // k-major A operand (template AT = 1, 256-row tiles): A[k][m] with `lda` elements per k row and `krows_a` valid rows -- V of O = P V
    // and dO of dV = P^T dO as they lie in memory, no transposed copies V^T / dO^T
    int lda, krows_a;
};

#ifndef G256_STAGED_EPI
#define G256_STAGED_EPI 1     // 1: the plain epilogue of the 256-row tiles goes through per-wave LDS regions (whole-row stores)
#endif
#ifndef G256_PHASES
#define G256_PHASES 2         // barrier phases per K-tile (2 or 4), see the main loop
#endif
// (Round 6, the softmax-backward product <2, 4>: life-cycle stamps put a tile at 1.3-2.3 us of prologue, 60-62 us of main loop, 6-9 us
// until the P tile has landed, 2.6 us of arithmetic, 1.8 us of stores.  Starting the first round of workgroups in 8 phases 0.5 .. 2 us
// apart -- so that an eighth of the CUs fetches P at a time instead of all 256 in one burst -- left the fetch at 6-9 us and the launch
// at 940-966 us: the fetch time is not a burst effect.  Dropped.)
// (An L2 prefetch of K-tile t+2 -- one 4-byte LDS-DMA per lane and K-tile into a dump area, i.e. one cache line per tile row --
// measured 8 % SLOWER: 772 -> 833 us; the one-K-tile prefetch distance is not what separates the 3-frame launch, 259 us per
// round of workgroups, from the one-frame launch whose operands stay in the 256 MB Infinity Cache, 224 us.)
// (Issuing the LDS-DMA instructions in the LOAD sections of the four-phase loop, as the guide's 8-phase template does, instead of
// behind MFMA pairs measured 6-8 % SLOWER here: O = P V 792 -> 852 us per 3-frame launch.)
#ifdef G256_LIFE
// life cycle of two workgroups (first and a late round) in 100 MHz ticks: entry, first K-tile landed, main loop done, epilogue phases,
// stores issued, stores retired -- tools/g256_life.py
__device__ unsigned long long g256_life[64];
extern "C" int tcvom_life256_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g256_life), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#define G_LIFE(i) if (life_on) g256_life[life_slot * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define G_LIFE(i)
#endif
#ifdef G256_TRACE
__device__ unsigned long long g256_trace[512];
extern "C" int tcvom_trace256_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g256_trace), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : -1;
}
#endif

// EPI: 0 = fp32 output, 1 = bf16 output, 2 = fused softmax backward, 3 = softmax numerators + per-tile row statistics (one
// instantiation each: a single kernel with all epilogues spilled registers in the main loop)
template <int EPI, int MF, int BT = 0, int AT = 0>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const Gemm256Args g) {
    static_assert(AT == 0 || MF == 4, "the k-major A operand is built for the 256-row tiles");
    // MF = 32-row A fragments per wave: 4 -> 256 x 256 tiles; 3 -> 192 (A rows) x 256 tiles for M = 576 (the d(query) / d(key)
    // GEMMs of GuidedCxtAtten: 3 x 192 instead of 3 x 256 with a quarter of the MFMAs multiplying padding).  The quadrant
    // (m1, *) then holds one fragment: phases 3 and 4 issue 4 MFMAs instead of 8.
    constexpr int TM = MF * 64, TN = 256, HM = TM / 2, A_IT = TM / 64;
    constexpr int SLOT = (TM + TN) * 64;                 // bf16 elements per K-tile buffer
    __shared__ __attribute__((aligned(16))) h16raw lds[2 * SLOT];
    // EPI 2 / 3: the tile's 256 column coefficients c[m] (and d[m]) wait in LDS from the prologue on -- read per (a, q) step of the
    // epilogue from global memory they were 16 dependent L2 round trips per workgroup behind the main loop (the scores product has
    // only 9 K-tiles: its epilogue was longer than its main loop)
    __shared__ __attribute__((aligned(16))) float coef[(EPI == 2 || EPI == 3) ? 512 : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // wm = wave group (0: rows 0..127 of the tile, 1: rows 128..255)
#ifdef G256_LIFE
    const int life_slot = (blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0) ? 0 : (blockIdx.x == 8 && blockIdx.y == gridDim.y / 2 && blockIdx.z == gridDim.z - 1) ? 1 : -1;
    const bool life_on = life_slot >= 0 && tid == 0;
    G_LIFE(0)
#endif

    // XCD-aware tile order over the pixel (N) tiles
    int bx, by = blockIdx.y, bzz = blockIdx.z, kpart = -1;
    {
        int nwg = gridDim.x, fx = blockIdx.x;
        if (g.flat_nx > 0) {                           // flat grid with a K-split tail (see Gemm256Args)
            const int nsplit = g.split_r * g.split_s, f = blockIdx.x;
            int tile;
            if (f < nsplit) {
                kpart = f / g.split_r;
                tile = (int)gridDim.x - nsplit + (f - kpart * g.split_r);          // whole tiles: 0 .. gridDim.x - nsplit - 1
            } else tile = f - nsplit;
            nwg = g.flat_nx;
            fx = tile % nwg;
            const int yz = tile / nwg;
            by = yz % g.flat_ny;
            bzz = yz / g.flat_ny;
        }
        const int q = nwg >> 3, r = nwg & 7, xcd = fx & 7, idx = fx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const bool second = bzz >= g.batch;
    const int bz = second ? bzz - g.batch : bzz;
    constexpr bool bt = BT == 1;                          // the B operand is k-major
    const h16raw* A = g.A + bz * g.a_bstride;
    const h16raw* B = second ? g.B2 + bz * g.b2_bstride : g.B + bz * g.b_bstride;
    int ntile = g.K >> 6;
    int krows = g.krows, krows_a = g.krows_a;
    if (kpart >= 0) {                                  // this workgroup's share of the reduction
        const int t0 = ntile * kpart / g.split_s, t1 = ntile * (kpart + 1) / g.split_s;
        if (AT != 0) { A += (int64_t)t0 * 64 * g.lda; krows_a -= t0 * 64; } else A += t0 * 64;
        if (BT != 0 && bt) { B += (int64_t)t0 * 64 * g.ldb; krows -= t0 * 64; } else B += t0 * 64;
        ntile = t1 - t0;
    }
    void* const gout = second ? g.out2 : g.out;
    const float* bias = g.bias ? g.bias + bz * g.vec_bstride : nullptr;
    const float* mscale = g.mscale ? g.mscale + bz * g.vec_bstride : nullptr;
    const float* mdiag = g.mdiag ? g.mdiag + bz * g.vec_bstride : nullptr;
    const int n0 = bx * TN, m0 = by * TM;
    const int K = g.K;

    // DMA pieces (8 rows x 128 bytes per wave-instruction; lane -> row +lane/8, 16-byte chunk kc): B piece (it, wave) covers tile
    // rows (it*8 + wave)*8 .. +7; the A rows are split by wave group -- group wm fetches the HM rows its own waves multiply
    // (piece s of wave w: rows wm*HM + (s*4 + (w & 3))*8 ..), so a group's A pieces are needed one barrier interval later than B
    const int kc8 = (((lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7)) << 3);
    const int a_piece = wm * (HM / 8) + (wave & 3);                    // + s*4
    int64_t a_off[4], b_off[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ra = m0 + (a_piece + it * 4) * 8 + (lane >> 3), rb = n0 + (it * 8 + wave) * 8 + (lane >> 3);
        a_off[it] = (it < A_IT && ra < g.M) ? (int64_t)ra * K + kc8 : -1;
        b_off[it] = rb < g.N ? (int64_t)rb * K + kc8 : -1;
        if (AT != 0) {
            // k-major A: group wm's half tile [64 k][128 m] in LDS, 256-byte rows; piece (it, wave & 3) = k rows 4 (it*4 + (wave & 3))
            // + (lane >> 4); the lane at chunk position lane & 15 fetches chunk c = position ^ 4 (row & 3) (see G_READ_A)
            const int kr = (it * 4 + (wave & 3)) * 4 + (lane >> 4), col = m0 + wm * HM + (((lane & 15) ^ ((kr & 3) << 2)) << 3);
            a_off[it] = (col < g.M && col + 8 <= g.lda) ? (int64_t)kr * g.lda + col : -1;
        }
        if (BT != 0 && bt) {
            // k-major B tile [64 k][256 n] in LDS, 512-byte rows: piece (it, wave) = k rows (it*8 + wave)*2 + (lane >> 5); the lane
            // at chunk position lane & 31 of its row fetches the 16-byte chunk (8 n) c = position ^ 4 (row & 3) (see G_READ_B)
            const int kr = (it * 8 + wave) * 2 + (lane >> 5), col = n0 + (((lane & 31) ^ ((kr & 3) << 2)) << 3);
            b_off[it] = col + 8 <= g.ldb ? (int64_t)kr * g.ldb + col : -1;
        }
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define G_ISSUE_A(t, it)                                                                                     \
    {                                                                                                        \
        const h16raw* src_ = a_off[it] >= 0 ? A + a_off[it] + (t) * 64 : g.zero_page;                       \
        if (AT != 0)                                                                                         \
            src_ = (a_off[it] >= 0 && (t) * 64 + ((it) * 4 + (wave & 3)) * 4 + (lane >> 4) < krows_a) ? A + a_off[it] + (int64_t)(t) * 64 * g.lda : g.zero_page; \
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(lds + ((t) & 1) * SLOT + (a_piece + (it) * 4) * 512), 16, 0, 0); \
    }
#define G_ISSUE_B(t, it)                                                                                     \
    {                                                                                                        \
        const h16raw* src_ = b_off[it] >= 0 ? B + b_off[it] + (t) * 64 : g.zero_page;                       \
        if (BT != 0 && bt)                                                                                   \
            src_ = (b_off[it] >= 0 && (t) * 64 + ((it) * 8 + wave) * 2 + (lane >> 5) < krows) ? B + b_off[it] + (int64_t)(t) * 64 * g.ldb : g.zero_page; \
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(lds + ((t) & 1) * SLOT + TM * 64 + ((it) * 8 + wave) * 512), 16, 0, 0); \
    }

    f32x16_t acc[MF][2];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int a_row = wm * HM + (lane & 31), b_row = wn * 64 + (lane & 31);
    const int a_swz = (a_row >> 1) & 7, b_swz = (b_row >> 1) & 7;     // the same for rows +32, +64, +96
    const int khalf = lane >> 5;
    h16x8_t fa[2][4], fb[2][4];                       // A sub-tile (2 row-fragments x 4 k16), both B sub-tiles (1 x 4 each)

    // k-major A: fragment (mh, a) = columns mh*64 + a*32 + (lane & 31) of the group's [64 k][128 m] half tile, through the transposing
    // LDS read like the k-major B (rows 256 bytes apart: chunk c of row k at position c ^ 4 (k & 3), i.e. the fragment index
    // mh*2 + a -- bits 2, 3 of the chunk -- XOR (k & 3))
    const unsigned tra_b0 = (unsigned)(uintptr_t)(lptr_t)(lds) + wm * (HM * 64 * 2) + (((lane >> 5) << 3) + ((lane & 15) >> 2)) * 256
                          + ((((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) << 4) + (lane & 1) * 8;
    const int tra_s2 = (lane & 15) >> 2;
    TrFrag fat[2][4];
#define G_READ_A(buf, mh)                                                                                    \
    if (AT != 0) {                                                                                           \
        _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) {                                                   \
            const unsigned ad_ = tra_b0 + (unsigned)(((((mh) * 2 + a_) ^ tra_s2) << 6)) + (buf) * (SLOT * 2); \
            tr_issue_imm<0, 1024>(fat[a_][0], ad_); tr_issue_imm<4096, 1024>(fat[a_][1], ad_);               \
            tr_issue_imm<8192, 1024>(fat[a_][2], ad_); tr_issue_imm<12288, 1024>(fat[a_][3], ad_);           \
        }                                                                                                    \
    } else {                                                                                                 \
        _Pragma("unroll") for (int a_ = 0; a_ < ((mh) * 2 + 1 < MF ? 2 : 1); ++a_)                           \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                              \
                fa[a_][kk_] = *reinterpret_cast<const h16x8_t*>(lds + (buf) * SLOT + (a_row + (mh) * 64 + a_ * 32) * 64 + (((kk_ * 2 + khalf) ^ a_swz) << 3)); \
    }
#define G_FIX_A()                                                                                            \
    if (AT != 0) {                                                                                           \
        _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                                     \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) { tr_fence(fat[a_][kk_]); fa[a_][kk_] = tr_value(fat[a_][kk_]); } \
    }
    // k-major B: the MFMA fragment (lane: n = lane & 31, 8 consecutive k from 8 (lane >> 5)) comes out of the [k][n] tile through the
    // transposing LDS read -- a 16-lane group addresses a 4 (k) x 16 (n) block, lane L the 4 n at row L >> 2 / column 4 (L & 3), and
    // receives column L, rows 0..3 -- two reads (k rows +0..3, +4..7) per fragment.  Bank layout: the 4 rows of a group are 512 B
    // apart, so chunk c of row k sits at position c ^ 4 (k & 3): the two groups of a half wave then cover 8 distinct 32-byte
    // columns of the 256-byte bank line.
    const unsigned tr_b0 = (unsigned)(uintptr_t)(lptr_t)(lds + TM * 64) + (((lane >> 5) << 3) + ((lane & 15) >> 2)) * 512 + (lane & 1) * 8;
    const int tr_c0 = wn * 8 + ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1), tr_sw = ((lane & 15) >> 2) << 2;
    const unsigned tr_addr[2] = {tr_b0 + (unsigned)(((tr_c0) ^ tr_sw) << 4), tr_b0 + (unsigned)(((tr_c0 + 4) ^ tr_sw) << 4)};
    TrFrag ft[2][4];
#define G_READ_B(buf, nh)                                                                                    \
    if (BT != 0 && bt) {                                                                                     \
        const unsigned ad_ = tr_addr[nh] + (buf) * (SLOT * 2);                                               \
        tr_issue_imm<0, 2048>(ft[nh][0], ad_); tr_issue_imm<8192, 2048>(ft[nh][1], ad_);                     \
        tr_issue_imm<16384, 2048>(ft[nh][2], ad_); tr_issue_imm<24576, 2048>(ft[nh][3], ad_);                \
    } else {                                                                                                 \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                                  \
            fb[nh][kk_] = *reinterpret_cast<const h16x8_t*>(lds + (buf) * SLOT + TM * 64 + (b_row + (nh) * 32) * 64 + (((kk_ * 2 + khalf) ^ b_swz) << 3)); \
    }
    // after the s_waitcnt that retires the transposing reads (they are inline asm: invisible to the compiler's wait insertion)
#define G_FIX_B()                                                                                            \
    if (BT != 0 && bt) {                                                                                     \
        _Pragma("unroll") for (int nh_ = 0; nh_ < 2; ++nh_)                                                  \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) { tr_fence(ft[nh_][kk_]); fb[nh_][kk_] = tr_value(ft[nh_][kk_]); } \
    }
    // 8 MFMAs of one quadrant; `DMA` = 0/1/2: interleave the A / B DMA instructions of K-tile tn behind MFMA pairs
#define G_MFMA(mh, nh, DMA, tn)                                                                              \
    {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) {                                                \
            acc[(mh) * 2 + 0][nh] = mfma16(fa[0][kk_], fb[nh][kk_], acc[(mh) * 2 + 0][nh], 0, 0, 0); \
            if constexpr ((mh) * 2 + 1 < MF)                                                                 \
                acc[(mh) * 2 + 1][nh] = mfma16(fa[1][kk_], fb[nh][kk_], acc[(mh) * 2 + 1][nh], 0, 0, 0); \
            if ((DMA) == 1 && kk_ < A_IT && (tn) < ntile) G_ISSUE_A(tn, kk_)                                 \
            if ((DMA) == 2 && (tn) < ntile) G_ISSUE_B(tn, kk_)                                               \
        }                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    }
#define G_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef G256_TRACE                                      // tools/g256_trace.py: (arrive, leave) cycle stamps of every barrier
    const bool trace_on = blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == 4);
    int trace_i = 0;
    // slots 254 / 255 of a group's block: shader-clock cycles and 100 MHz reference ticks of the whole main loop (their ratio is
    // the clock the CU actually ran at)
    const unsigned long long trace_c0 = __builtin_readcyclecounter(), trace_r0 = __builtin_amdgcn_s_memrealtime();
#define G_BAR()                                                                                              \
    {                                                                                                        \
        if (trace_on && trace_i < 250) g256_trace[(wave >> 2) * 256 + trace_i++] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_s_barrier();                                                                        \
        if (trace_on && trace_i < 250) g256_trace[(wave >> 2) * 256 + trace_i++] = __builtin_readcyclecounter(); \
    }
#else
#define G_BAR() __builtin_amdgcn_s_barrier()
#endif

    // prologue: K-tile 0 into buffer 0 (all waves), then the stagger
#pragma unroll
    for (int it = 0; it < 4; ++it) { if (it < A_IT) G_ISSUE_A(0, it) G_ISSUE_B(0, it) }
    if constexpr (EPI == 2 || EPI == 3) {
        // waves 0..3: c[m0 .. m0 + 255], waves 4..7: d[...] (zeros beyond M / without a diagonal term); 4 bytes per lane
        const int ci = (wave & 3) * 64 + lane;
        const float* vec = wave < 4 ? mscale : mdiag;
        const void* src_ = (vec && m0 + ci < g.M) ? (const void*)(vec + m0 + ci) : (const void*)g.zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(coef + wave * 64), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G_LIFE(1)
    G_BAR();
    if (wm == 1) G_BAR();                              // group 1 runs one barrier interval behind group 0

#if G256_PHASES == 2
    // Two phases per K-tile: A = quadrants (m0, n0), (m0, n1) (16 MFMAs, loads A-m0 and all of B), B = (m1, n1), (m1, n0) (loads
    // A-m1 only): half the barriers of the four-phase loop and 24 instead of 28 ds_reads per K-tile.
    //     barrier interval     4t    4t+1   4t+2   4t+3   4t+4
    //     group 0              LA    MA*    LB     MBw    LA'            * issues the LDS-DMA of K-tile t+1 (B pieces first)
    //     group 1              MB    LA     MA*    LBv    MBw            v vmcnt(A_IT): its B pieces have landed;  w vmcnt(0)
    // Buffer (t+1) & 1 held K-tile t-1, whose last reads (group 1, LB) retire at the head of its MB in interval 4t; the first DMA
    // into it is issued in 4t+1.  Group 0 reads tile t+1 from 4t+4 on (A rows of group 0, all B rows): group 0's pieces are
    // drained at the end of 4t+3, group 1's B pieces too (they were issued first: vmcnt(A_IT) leaves only its own A pieces in
    // flight); group 1's A rows are read by group 1 alone, from 4t+5 on, and are drained at the end of its MB (4t+4).
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        G_READ_A(buf, 0) G_READ_B(buf, 0) G_READ_B(buf, 1)
        G_BAR();
        G_WAIT_LDS();
        G_FIX_A()
        G_FIX_B()
        G_MFMA(0, 0, 2, t + 1)
        G_MFMA(0, 1, 1, t + 1)
        G_BAR();
        G_READ_A(buf, 1)
        if (wm == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT) : "memory");
        G_BAR();
        G_WAIT_LDS();
        G_FIX_A()
        G_MFMA(1, 1, 0, 0)
        G_MFMA(1, 0, 0, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G_BAR();
    }
#else
    static_assert(BT == 0 && AT == 0, "the four-phase loop has no k-major operand path");
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        // phase 1: quadrant (m0, n0)
        G_READ_A(buf, 0) G_READ_B(buf, 0)
        G_BAR();
        G_WAIT_LDS();
        G_MFMA(0, 0, 1, t + 1)
        G_BAR();
        // phase 2: quadrant (m0, n1)
        G_READ_B(buf, 1)
        G_BAR();
        G_WAIT_LDS();
        G_MFMA(0, 1, 2, t + 1)
        G_BAR();
        // phase 3: quadrant (m1, n1)
        G_READ_A(buf, 1)
        G_BAR();
        G_WAIT_LDS();
        G_MFMA(1, 1, 0, 0)
        G_BAR();
        // phase 4: quadrant (m1, n0); the DMA of K-tile t+1 must have landed before anybody enters phase 1 of t+1:
        // group 0 drains at the end of its M4, group 1 (one interval behind) at the end of its L4
        G_READ_B(buf, 0)
        if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G_BAR();
        G_WAIT_LDS();
        G_MFMA(1, 0, 0, 0)
        if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G_BAR();
    }
#endif
    if (wm == 0) G_BAR();                              // balance group 1's extra barrier
#ifdef G256_TRACE
    if (trace_on) {
        g256_trace[(wave >> 2) * 256 + 254] = __builtin_readcyclecounter() - trace_c0;
        g256_trace[(wave >> 2) * 256 + 255] = __builtin_amdgcn_s_memrealtime() - trace_r0;
    }
#endif
#undef G_MFMA
#undef G_READ_A
#undef G_READ_B
#undef G_FIX_B
#undef G_FIX_A
#undef G_ISSUE_A
#undef G_ISSUE_B
#undef G_BAR
#undef G_WAIT_LDS

    // ------------------------------------------------------------------ epilogue
    // A lane holds 4 consecutive m of ONE row n, its 32 neighbours 32 different rows.
    if constexpr (EPI == 3) coef[tid] *= 1.4426950408889634f;          // c[m], d[m] in base-2 units (512 threads, 512 floats)
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0): nothing of the main loop is in flight
    __builtin_amdgcn_s_barrier();
    G_LIFE(2)
    int pglob[2];
    bool pvalid[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        pglob[b] = n0 + wn * 64 + b * 32 + (lane & 31);
        pvalid[b] = pglob[b] < g.N;
    }
    char* lb = reinterpret_cast<char*>(lds);
    const int64_t obase = (int64_t)bz * g.out_bstride;
    if constexpr (EPI == 2) {
        // ---- fused softmax backward: T = P * (acc - delta[n]) * c[m] in bf16; tile = 256 rows x 512 bytes, IN PLACE over the
        // P tile: 16-byte chunk c of row r at position c ^ (r & 15)
        const h16raw* Pb = g.P + obase;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = (it * 8 + wave) * 2 + (lane >> 5), cp = lane & 31, c = cp ^ (r & 15);
            const bool ok = n0 + r < g.N && m0 + c * 8 < g.ldo;
            const h16raw* src = ok ? Pb + (int64_t)(n0 + r) * g.ldo + m0 + c * 8 : g.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lb + (it * 8 + wave) * 1024), 16, 0, 0);
        }
        float dl[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) dl[b] = pvalid[b] ? g.delta[(int64_t)bz * g.N + pglob[b]] : 0.f;
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        G_LIFE(3)
        // Transposed copy of the LDS tile ([n][m], 512-byte rows, chunk-swizzled) into dst[m][n] with the gfx950 transposing LDS
        // read: the 16 lanes of a group address a 4 (n) x 16 (m) block -- lane L the 4 contiguous m of row L >> 2 -- and lane L
        // receives column m = L, rows 0..3 (tools/probes/tr_probe2.hip); two reads give 8 consecutive n of one m = one 16-byte
        // store.  Wave w owns the tile columns m = 32 w .. 32 w + 31; the four groups of a wave take four consecutive 8-n
        // blocks, so one store instruction writes 64 contiguous bytes of 16 rows (the register form this replaces -- a lane
        // holds 4 m of ONE n -- stored 4 bytes per lane after a pair exchange: 8x the store instructions).
        auto store_transposed = [&](h16raw* dst) {
            const int L = lane & 15, gq = lane >> 4;
            h16raw* db = dst + (int64_t)bz * g.ldt * g.ldt;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const int mcol = wave * 32 + mh * 16, mread = mcol + 4 * (L & 3), mg = m0 + mcol + L;
#pragma unroll
                for (int nb4 = 0; nb4 < 256; nb4 += 128) {
                    TrFrag f[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r0 = nb4 + u * 32 + gq * 8 + (L >> 2), r1 = r0 + 4;
                        tr_issue(f[u], reinterpret_cast<const h16raw*>(lb + r0 * 512 + (((mread >> 3) ^ (r0 & 15)) << 4) + (mread & 4) * 2),
                                 reinterpret_cast<const h16raw*>(lb + r1 * 512 + (((mread >> 3) ^ (r1 & 15)) << 4) + (mread & 4) * 2));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        tr_fence(f[u]);
                        const int ng = n0 + nb4 + u * 32 + gq * 8;
                        if (mg < g.ldt && ng + 8 <= g.ldt)
                            *reinterpret_cast<uint4*>(db + (int64_t)mg * g.ldt + ng) = __builtin_bit_cast(uint4, tr_value(f[u]));
                    }
                }
            }
        };
        if (g.Tt) {
            store_transposed(g.Pt);                    // P^T from the P tile, before T overwrites it in place
            __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int a = 0; a < MF; ++a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ml = wm * HM + a * 32 + 8 * q + 4 * (lane >> 5);
                const float4 c4 = *reinterpret_cast<const float4*>(coef + ml);           // (zeros in the columns M <= m)
                const float sc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int r = wn * 64 + b * 32 + (lane & 31);
                    uint2* cell = reinterpret_cast<uint2*>(lb + r * 512 + (((ml >> 3) ^ (r & 15)) << 4) + (ml & 4) * 2);
                    const uint2 pp = *cell;
                    uint2 o;                           // (padding columns M <= m < ldo: sc = 0 -> zeros)
                    o.x = pack2h(hlo(pp.x) * (acc[a][b][q * 4 + 0] - dl[b]) * sc[0], hhi(pp.x) * (acc[a][b][q * 4 + 1] - dl[b]) * sc[1]);
                    o.y = pack2h(hlo(pp.y) * (acc[a][b][q * 4 + 2] - dl[b]) * sc[2], hhi(pp.y) * (acc[a][b][q * 4 + 3] - dl[b]) * sc[3]);
                    *cell = o;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        G_LIFE(4)
        h16raw* Tb = reinterpret_cast<h16raw*>(gout) + obase;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = (it * 8 + wave) * 2 + (lane >> 5), cp = lane & 31, c = cp ^ (r & 15);
            const uint4 v = *reinterpret_cast<const uint4*>(lb + r * 512 + cp * 16);
            if (n0 + r < g.N && m0 + c * 8 < g.ldo) *reinterpret_cast<uint4*>(Tb + (int64_t)(n0 + r) * g.ldo + m0 + c * 8) = v;
        }
        if (g.Tt) store_transposed(g.Tt);              // T^T from the finished T tile
        G_LIFE(5)
#ifdef G256_LIFE
        __builtin_amdgcn_s_waitcnt(0x0070);
        G_LIFE(6)
#endif
        return;
    } else {
    // ---- plain epilogue (bias / scale / diagonal / activation)
    constexpr bool F32 = EPI == 0;
    if constexpr (MF == 4 && G256_STAGED_EPI) {
        // Each wave turns its own 64 (n) x 128 (m) tile through a private LDS region, 32 rows at a time, so that the global stores
        // are whole rows: a lane's MFMA results are 4 consecutive m of ONE row n and its 32 neighbours sit in 32 different rows --
        // stored straight from the registers that is 32 segments of 32 bytes (fp32) / 8 bytes (bf16) per instruction, and the
        // GEMMs with a short reduction (the 1x1 convs of the FBA bottlenecks: 4 .. 8 K-tiles; S = G G^T: 9) spend more time
        // there than in their main loop.  No barrier: a wave only reads what it wrote (LDS operations of a wave execute in order).
        // Row r of the region holds 512 (fp32) / 256 (bf16) bytes; its 16-byte chunk c sits at position c ^ (r & 31) / c ^ (r & 15)
        // (conflict-free ds_write_b128 / ds_write_b64 for the MFMA layout, plain rows for the ds_read_b128 that feed the stores).
        char* wl = lb + wave * (F32 ? 16384 : 8192);
        const int nl = lane & 31, h = lane >> 5;
        const bool al16 = F32 || ((g.ldo & 7) == 0 && (g.out_bstride & 7) == 0 && ((uintptr_t)gout & 15) == 0);
        const float slope = g.act == 1 ? 0.f : g.act == 3 ? 0.01f : 1.f;          // activation as max(x, slope * x)
        // (the diagonal term is decided once per tile, not per element: `if (mdiag && ..)` / `if (act == ..)` inside the 128-value
        // loops were scalar branches)
        if constexpr (EPI == 3) {
            // ---- S' = acc * c[m] - d[m] [m == n] in place, then the softmax numerators of THIS tile: a row n of the tile has its
            // 256 columns m in 2 lane halves x 2 wave groups; max and sum go lane -> partner lane (xor 32) -> partner wave (LDS)
            float* red = reinterpret_cast<float*>(lb + 96 * 1024);       // 2 x [2 wm][4 wn][2 b][32] floats, beyond the staging regions
            const float ninf = -__builtin_inff();
            float tmax[2] = {ninf, ninf};
            // The scores live in base-2 units from here on (the coefficients in LDS were multiplied by log2 e at the head of the
            // epilogue): x' = log2e (acc c[m] - d[m] [m == n]), numerator = exp2(x' - max x') -- one v_exp_f32 per value, no multiply.
            // Per-element work the tile does not need is decided ONCE per tile: the diagonal term exists in the tiles on the
            // diagonal only (m0 == n0: 1 tile in 32 at 1080p), padding columns M <= m in the last tile column only.
            auto pass1 = [&](auto diag_, auto pad_) {
                constexpr bool DIAG = decltype(diag_)::value, PAD = decltype(pad_)::value;
#pragma unroll
                for (int a = 0; a < MF; ++a) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ml = wm * HM + a * 32 + 8 * q + 4 * h, mrow = m0 + ml;
                        const bool mv = mrow < g.M;
                        const float4 sc4 = *reinterpret_cast<const float4*>(coef + ml);
                        float4 dg4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (DIAG) dg4 = *reinterpret_cast<const float4*>(coef + 256 + ml);
                        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float x = acc[a][b][q * 4 + r] * sc[r];
                                if constexpr (DIAG) x -= (mrow + r) == pglob[b] ? dg[r] : 0.f;
                                if constexpr (PAD) x = mv ? x : ninf;         // padding columns M <= m: no part of the row
                                acc[a][b][q * 4 + r] = x;
                                tmax[b] = fmaxf(tmax[b], x);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);        // keep the 16 coefficient reads from being hoisted together (spills)
                    }
                }
            };
            const bool diag_tile = m0 == n0 && mdiag != nullptr, pad_tile = m0 + TM > g.M;
            if (pad_tile) { if (diag_tile) pass1(std::true_type{}, std::true_type{}); else pass1(std::false_type{}, std::true_type{}); }
            else if (diag_tile) pass1(std::true_type{}, std::false_type{});
            else pass1(std::false_type{}, std::false_type{});
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tmax[b] = fmaxf(tmax[b], __shfl_xor(tmax[b], 32, 64));
                if (h == 0) red[((wm * 4 + wn) * 2 + b) * 32 + nl] = tmax[b];
            }
            __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
            G_LIFE(3)
            float tsum[2] = {0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tmax[b] = fmaxf(tmax[b], red[(((wm ^ 1) * 4 + wn) * 2 + b) * 32 + nl]);
                // (a row whose every column of this tile is padding cannot occur: the last tile column holds M - (tiles - 1) 256 >= 1
                //  valid columns, so tmax is finite and exp2(-inf - tmax) = 0 for the padding columns)
#pragma unroll
                for (int a = 0; a < MF; ++a)
#pragma unroll
                    for (int r16 = 0; r16 < 16; ++r16) {
                        const float e = __builtin_amdgcn_exp2f(acc[a][b][r16] - tmax[b]);
                        acc[a][b][r16] = e;
                        tsum[b] += e;
                    }
                tsum[b] += __shfl_xor(tsum[b], 32, 64);
                if (h == 0) red[512 + ((wm * 4 + wn) * 2 + b) * 32 + nl] = tsum[b];
            }
            __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
            G_LIFE(4)
            if (wm == 0 && h == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (pvalid[b]) {
                        float* st = g.stats + (((int64_t)bz * g.N + pglob[b]) * gridDim.y + by) * 2;
                        st[0] = tmax[b] * 0.6931471805599453f;         // back to natural units (softmax_rescale: exp(max - row max))
                        st[1] = tsum[b] + red[512 + ((1 * 4 + wn) * 2 + b) * 32 + nl];
                    }
            }
        }
        if constexpr (EPI == 1) {
            if (g.cstats) {
                // ---- normalisation statistics of this conv's outputs (the launcher allows it without a scale / diagonal term only), summed over the wave's 64 pixels in fp32 before the 16-bit rounding, as igemm_nt does
                float* sp0 = g.cstats + ((int64_t)g.cstats_goff + bz * g.cstats_bstride + (int64_t)bx * 4 + wn) * 2 * g.M;
#pragma unroll
                for (int a = 0; a < MF; ++a) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mrow = m0 + wm * HM + a * 32 + 8 * q + 4 * h;
                        float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (bias && mrow < g.M) bs4 = *reinterpret_cast<const float4*>(bias + mrow);
                        const float bsv[4] = {bs4.x, bs4.y, bs4.z, bs4.w};
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float x = acc[a][b][q * 4 + r] + bsv[r];
                                x = fmaxf(x, x * slope);
                                const float xs = pvalid[b] ? x : 0.f;
                                t8[r] += xs;
                                t8[4 + r] = fmaf(xs, xs, t8[4 + r]);
                            }
                        }
                        reduce8_store(t8, lane, sp0, g.M, mrow, mrow < g.M);
                    }
                }
            }
        }
        auto emit = [&](auto diag_) {
        constexpr bool DIAG = decltype(diag_)::value;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int a = 0; a < MF; ++a) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mrow = m0 + wm * HM + a * 32 + 8 * q + 4 * h;
                    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), dg4 = bs;
                    if constexpr (EPI != 3) {
                        if (mrow < g.M) {                     // M % 4 == 0: a lane's 4 rows are valid together
                            if (bias) bs = *reinterpret_cast<const float4*>(bias + mrow);
                            if (mscale) sc4 = *reinterpret_cast<const float4*>(mscale + mrow);
                            if (DIAG) dg4 = *reinterpret_cast<const float4*>(mdiag + mrow);
                        }
                    }
                    const float bsv[4] = {bs.x, bs.y, bs.z, bs.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (EPI == 3) {
                            v[r] = acc[a][b][q * 4 + r];          // the softmax numerator computed above
                        } else {
                            float x = acc[a][b][q * 4 + r] * sc[r] + bsv[r];
                            if (DIAG) x -= (mrow + r) == pglob[b] ? dg[r] : 0.f;
                            v[r] = fmaxf(x, x * slope);
                        }
                    }
                    if constexpr (F32)
                        *reinterpret_cast<float4*>(wl + nl * 512 + (((a * 8 + 2 * q + h) ^ nl) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        *reinterpret_cast<uint2*>(wl + nl * 256 + (((a * 4 + q) ^ (nl & 15)) << 4) + h * 8) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
                }
            }
            const int nbase = n0 + wn * 64 + b * 32, mbase = m0 + wm * HM;
            if constexpr (F32) {
                float* o = reinterpret_cast<float*>(gout) + obase;
                if (kpart >= 0) {
                    // K-split over every tile (gemm_tt256: weight gradients): the partial products are ADDED.  Atomics execute per
                    // cache line: 64 CONSECUTIVE floats of one row per instruction (2 lines); a lane adding its float4 as four
                    // scalar atomics spreads every instruction over 8 lines (measured: 0.23 TB/s of atomic traffic, 4x slower)
#pragma unroll 4
                    for (int i = 0; i < 64; ++i) {
                        const int row = i >> 1, e = (i & 1) * 64 + lane;
                        const float v = *reinterpret_cast<const float*>(wl + row * 512 + (((e >> 2) ^ row) << 4) + (e & 3) * 4);
                        const int n = nbase + row, m = mbase + e;
                        if (n < g.N && m < g.M) __hip_atomic_fetch_add(o + (int64_t)n * g.ldo + m, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = i * 2 + h, c = nl ^ row;                     // (row < 32: row & 31 == row)
                    const float4 t = *reinterpret_cast<const float4*>(wl + row * 512 + nl * 16);
                    const int n = nbase + row, m = mbase + c * 4;
                    if (kpart < 0 && n < g.N && m < g.M) *reinterpret_cast<float4*>(o + (int64_t)n * g.ldo + m) = t;
                }
            } else {
                h16raw* o = reinterpret_cast<h16raw*>(gout) + obase;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = i * 4 + (lane >> 4), pc = lane & 15, c = pc ^ (row & 15);
                    const uint4 t = *reinterpret_cast<const uint4*>(wl + row * 256 + pc * 16);
                    const int n = nbase + row, m = mbase + c * 8;
                    if (n < g.N && m < g.M) {
                        h16raw* dst = o + (int64_t)n * g.ldo + m;
                        if (al16 && m + 8 <= g.M) *reinterpret_cast<uint4*>(dst) = t;
                        else {
                            *reinterpret_cast<uint2*>(dst) = make_uint2(t.x, t.y);
                            if (m + 8 <= g.M) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(t.z, t.w);
                        }
                    }
                }
            }
        }
        };
        if constexpr (EPI == 3) emit(std::false_type{});
        else if (mdiag) emit(std::true_type{}); else emit(std::false_type{});
        G_LIFE(5)
#ifdef G256_LIFE
        __builtin_amdgcn_s_waitcnt(0x0070);
        G_LIFE(6)
#endif
    } else {
    if constexpr (F32 && MF == 3) {
        if (kpart >= 0) {
            // K-split tail: this workgroup's partial products are ADDED to the zeroed output.  Atomics execute per cache line, so
            // each wave turns its 64 (n) x 96 (m) block through a private LDS region, 32 rows at a time, and issues them along the
            // rows (64 consecutive floats per instruction = 2 lines; straight from the MFMA layout an instruction touches 32 rows).
            float* wl = reinterpret_cast<float*>(lb + wave * 12800);              // [32 rows][100 floats] per wave
            const int nl = lane & 31, h = lane >> 5;
            float* o = reinterpret_cast<float*>(gout) + obase;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int a = 0; a < MF; ++a) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ml = a * 32 + 8 * q + 4 * h, mrow = m0 + wm * HM + ml;
                        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
                        if (mscale && mrow < g.M) sc4 = *reinterpret_cast<const float4*>(mscale + mrow);
                        *reinterpret_cast<float4*>(wl + nl * 100 + ml) = make_float4(acc[a][b][q * 4 + 0] * sc4.x, acc[a][b][q * 4 + 1] * sc4.y,
                                                                                      acc[a][b][q * 4 + 2] * sc4.z, acc[a][b][q * 4 + 3] * sc4.w);
                    }
                }
                const int nbase = n0 + wn * 64 + b * 32, mbase = m0 + wm * HM;
#pragma unroll 4
                for (int i = 0; i < 48; ++i) {
                    const int e = i * 64 + lane, row = e / 96, col = e - row * 96;
                    const float v = wl[row * 100 + col];
                    const int n = nbase + row, m = mbase + col;
                    if (n < g.N && m < g.M) __hip_atomic_fetch_add(o + (int64_t)n * g.ldo + m, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
    }
    // stored straight from the registers (the 192-row tiles)
    const float slope3 = g.act == 1 ? 0.f : g.act == 3 ? 0.01f : 1.f;
    int64_t out_off[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) out_off[b] = (int64_t)(pvalid[b] ? pglob[b] : 0) * g.ldo + obase;
#pragma unroll
    for (int a = 0; a < MF; ++a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mrow = m0 + wm * HM + a * 32 + 8 * q + 4 * (lane >> 5);
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), dg4 = bs;
            if (mrow < g.M) {                         // M % 4 == 0: a lane's 4 rows are valid together
                if (bias) bs = *reinterpret_cast<const float4*>(bias + mrow);
                if (mscale) sc4 = *reinterpret_cast<const float4*>(mscale + mrow);
                if (mdiag) dg4 = *reinterpret_cast<const float4*>(mdiag + mrow);
            }
            const float bsv[4] = {bs.x, bs.y, bs.z, bs.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[a][b][q * 4 + r] * sc[r] + bsv[r];
                    if (mdiag && (mrow + r) == pglob[b]) x -= dg[r];
                    v[r] = fmaxf(x, x * slope3);
                }
                if (pvalid[b] && mrow < g.M) {
                    if (F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(gout) + out_off[b] + mrow) = make_float4(v[0], v[1], v[2], v[3]);
                    else *reinterpret_cast<uint2*>(reinterpret_cast<h16raw*>(gout) + out_off[b] + mrow) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
                }
            }
        }
    }
    }
    }
}

// K-split tail of a launch on 192-row tiles with fp32 output and a linear epilogue: when the tiles beyond the last whole round of
// 256 workgroups are whole tile rows of the LAST output slice and would leave most of the chip idle for a full tile time, they go
// out first as split_s workgroups each (fp32 atomics into the zeroed tail); rewrites `grid` to the flat form.  < 0: error.
static int g256_split_tail(Gemm256Args& g, dim3& grid, bool eligible, void* last_out, int nb, void* stream) {
    static const bool nosplit = getenv("TCVOM_NO_KSPLIT") != nullptr;              // A/B switch
    g.flat_nx = g.flat_ny = g.split_r = g.split_s = 0;
    if (nosplit || !eligible || !g.out_fp32) return 0;
    const int nx = (int)grid.x, ny = (int)grid.y, tiles = nx * ny * (int)grid.z, rem = tiles % 256;
    const int ntile = g.K / 64;
    if (!(tiles > 256 && rem > 0 && rem <= 128 && rem % nx == 0 && rem / nx <= ny)) return 0;
    const int sp = 256 / rem < 8 ? 256 / rem : 8;                                  // 2 .. 8 parts, each >= 8 K-tiles
    if (sp < 2 || ntile / sp < 8) return 0;
    // the tail covers the last rem / nx tile rows of the LAST output slice: the whole slice is zeroed (one linear memset; its other
    // tiles store over it -- a strided 2-D memset of the tail columns alone took 14 us per launch)
    float* base = (float*)last_out + (long long)(nb - 1) * g.out_bstride;
    if (hipMemsetAsync(base, 0, sizeof(float) * (size_t)g.N * (size_t)g.ldo, (hipStream_t)stream) != hipSuccess)
        return tcvom_fail(TCVOM_ERR_LAUNCH, "gemm_nt256: memset of the K-split tail failed");
    g.flat_nx = nx; g.flat_ny = ny; g.split_r = rem; g.split_s = sp;
    grid = dim3((unsigned)(tiles - rem + rem * sp), 1, 1);
    return 0;
}

// does the dense descriptor `d` run on this kernel?  (plain row-major operands, enough 256x256 tiles to fill the chip)
int gemm_nt256_takes(const tcvom_conv_desc* d) {
    if (d->ntaps != 1 || d->tap_w[0] < 0) return 0;
    if (d->in_f16 && !TCVOM_BUILD_F16) return 0;          // IEEE fp16 operands in the bf16 build: igemm_nt
    if (d->in_step != 1 || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0 || d->wt != 1) return 0;
    if (d->C % 64 != 0 || d->K % 4 != 0 || d->ldo % 4 != 0) return 0;
    const long long P = (long long)d->N * d->PH * d->PW;
    if ((long long)d->N * d->H * d->W != P || (long long)d->N * d->OH * d->OW != P) return 0;
    const int nb = d->batch > 1 ? d->batch : 1;
    const long long tiles = ((P + 255) / 256) * ((d->K + 255) / 256) * nb;
    if (d->K < 256 || P < 256 || tiles < 192) return 0;
    return 1;
}

// ... and writes the normalisation statistics of its outputs itself?  (16-bit output on the 256-row tiles: the 1 x 1 convs in front of
// a GroupNorm / BatchNorm -- the bottleneck reduce / expand layers of the FBA trunk, models/FBA/resnet_GN_WS.py:50-137 -- ran on
// igemm_nt<256,256> for their statistics epilogue alone: 370 .. 500 TFLOP/s there against 590 .. 790 here)
int gemm_nt256_takes_stats(const tcvom_conv_desc* d) {
    static const bool off = getenv("TCVOM_NO_G256_STATS") != nullptr;              // A/B switch
    if (off || !gemm_nt256_takes(d) || d->out_fp32 || d->K % 4 != 0) return 0;
    return (long long)cdiv(d->K, 192) * 192 < (long long)cdiv(d->K, 256) * 256 ? 0 : 1;          // (not the 192-row tiles)
}
int gemm_nt256_stats_groups(const tcvom_conv_desc* d) {
    return cdiv((long long)d->N * d->PH * d->PW, 256) * 4;
}

// 1: launched; 0: not a shape for this kernel.  Called from conv_igemm_launch for dense descriptors (ntaps == 1).
// in2 / out2 non-null: paired launch, the second product in2 x w -> out2 rides in grid z (same descriptor).
// cstats non-null: per-channel statistics of the outputs (the caller checked gemm_nt256_takes_stats and passes no bias / scale / diagonal).
int gemm_nt256_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                          const tcvom_conv_desc* d, const h16raw* zero_page, void* stream, const void* in2, void* out2,
                          long long in2_bstride, float* cstats) {
    if (!gemm_nt256_takes(d)) return 0;
    if (cstats && (!gemm_nt256_takes_stats(d) || mscale || mdiag || in2)) return 0;
    const long long P = (long long)d->N * d->PH * d->PW;
    const int nb = d->batch > 1 ? d->batch : 1;
    Gemm256Args g;
    g.A = (const h16raw*)w;
    g.B = (const h16raw*)in;
    g.out = out;
    g.bias = bias;
    g.mscale = mscale;
    g.mdiag = mdiag;
    g.zero_page = zero_page;
    g.M = d->K;
    g.N = (int)P;
    g.K = d->C;
    g.ldo = d->ldo;
    g.act = d->act;
    g.out_fp32 = d->out_fp32;
    g.batch = nb;
    g.a_bstride = nb > 1 ? d->w_bstride : 0;
    g.b_bstride = nb > 1 ? d->in_bstride : 0;
    g.out_bstride = nb > 1 ? d->out_bstride : 0;
    g.vec_bstride = nb > 1 ? d->vec_bstride : 0;
    g.P = nullptr;
    g.delta = nullptr;
    g.Tt = nullptr; g.Pt = nullptr; g.ldt = 0;
    g.B2 = (const h16raw*)in2; g.out2 = out2; g.b2_bstride = nb > 1 ? in2_bstride : 0;
    g.stats = nullptr;
    g.cstats = cstats; g.cstats_goff = d->stats_group_offset; g.cstats_bstride = d->stats_bstride;
    g.flat_nx = g.flat_ny = g.split_r = g.split_s = 0;
    g.ldb = 0; g.krows = 0; g.lda = 0; g.krows_a = 0;
    // 192-row A tiles where they leave less padding than 256-row ones (M = 576: 3 x 192)
    static const bool no192 = getenv("TCVOM_NO_M192") != nullptr;                  // A/B switch
    const bool m192 = !no192 && (long long)cdiv(d->K, 192) * 192 < (long long)cdiv(d->K, 256) * 256;
    dim3 grid((unsigned)((P + 255) / 256), (unsigned)cdiv(d->K, m192 ? 192 : 256), (unsigned)(in2 ? 2 * nb : nb));
    {
        const int rc = g256_split_tail(g, grid, m192 && !bias && !mdiag && d->act == 0, in2 ? out2 : out, nb, stream);
        if (rc < 0) return rc;
    }
    if (m192) {
        if (g.out_fp32) hipLaunchKernelGGL((gemm_nt256_kernel<0, 3>), grid, dim3(512), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_nt256_kernel<1, 3>), grid, dim3(512), 0, (hipStream_t)stream, g);
    } else {
        if (g.out_fp32) hipLaunchKernelGGL((gemm_nt256_kernel<0, 4>), grid, dim3(512), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_nt256_kernel<1, 4>), grid, dim3(512), 0, (hipStream_t)stream, g);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "gemm_nt256: %s", hipGetErrorString(e));
    return 1;
}

// ---- weight gradient of a dense (1 x 1, stride 1) conv as ONE TT GEMM on the k-major operand path (both operands read as they lie
// in memory -- pixels are the reduction index and the outer memory index -- through the transposing LDS reads of gca_dv):
//     dw[k][c] += sum_p dy[p][k] * x[p][c]          dy: [P][ldy], x: [P][C], dw: [K][C] fp32 (added to: the caller's arena is zeroed)
// The output is a handful of 256 x 256 tiles (K, C <= 2048) over a reduction of 10^4 .. 10^5 pixels, so EVERY tile is split over the
// reduction (split_s workgroups per tile, fp32 atomics along the rows).  Replaces igemm_tt for these shapes (measured 480 .. 550
// TFLOP/s on the bottleneck 1 x 1 convs of the FBA base, models/FBA/resnet_GN_WS.py:50-137, against 1.25 PFLOP/s of this loop on dV).
// 1: launched; 0: not a shape for this kernel.
extern const h16raw* tcvom_zero_page(void);
int gemm_tt256_takes(const tcvom_conv_desc* d) {                                   // (shape only: bench labels)
    static const bool off = getenv("TCVOM_NO_TT256") != nullptr;                   // A/B switch
    if (off) return 0;
    if (d->ntaps != 1 || d->tap_w[0] != 0 || d->tap_dh[0] != 0 || d->tap_dw[0] != 0 || d->wt != 1) return 0;
    if (d->in_step != 1 || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return 0;
    const long long P = (long long)d->N * d->PH * d->PW;
    if ((long long)d->N * d->H * d->W != P || (long long)d->N * d->OH * d->OW != P) return 0;
    constexpr int tt_min_tiles = 8;
    // (every workgroup ends with 256 KB of atomics: worth it from 8 tiles on -- K C >= 512 K weights; the 4-tile shapes measured
    //  slower than igemm_tt)
    return d->K >= 256 && d->C >= 256 && d->K % 8 == 0 && d->C % 8 == 0 && P >= 4096 && cdiv(d->K, 256) * cdiv(d->C, 256) >= tt_min_tiles;
}
int gemm_tt256_try_launch(const void* dy, const void* x, float* dw, const tcvom_conv_desc* d, int ldy, void* stream, int nb,
                          long long dy_stride, long long x_stride, long long dw_stride) {
    // nb > 1: nb problems `*_stride` ELEMENTS apart (the frames of a frame-batched layer) in one launch -- the tiles of all problems
    // share the chip, so every tile is split into fewer parts (fewer atomics per flop)
    if (!gemm_tt256_takes(d)) return 0;
    const long long P = (long long)d->N * d->PH * d->PW;
    if (ldy % 8 != 0 || ldy < d->K) return 0;
    if (P < 4096 || P * (long long)(ldy > d->C ? ldy : d->C) >= (1ll << 31)) return 0;
    if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) & 15) != 0) return 0;
    Gemm256Args g;
    g.A = (const h16raw*)x;          // k-major A: row k = pixel p, columns m = input channels c
    g.B = (const h16raw*)dy;         // k-major B: row k = pixel p, columns n = output channels k
    g.out = dw;
    g.bias = nullptr; g.mscale = nullptr; g.mdiag = nullptr;
    g.zero_page = tcvom_zero_page();
    if (!g.zero_page) return tcvom_fail(TCVOM_ERR_LAUNCH, "gemm_tt256: could not allocate the zero page");
    g.M = d->C; g.N = d->K; g.K = (int)((P + 63) / 64 * 64); g.ldo = d->C; g.act = 0; g.out_fp32 = 1; g.batch = nb;
    g.a_bstride = x_stride; g.b_bstride = dy_stride; g.out_bstride = dw_stride; g.vec_bstride = 0;
    g.P = nullptr; g.delta = nullptr; g.Tt = nullptr; g.Pt = nullptr; g.ldt = 0;
    g.B2 = nullptr; g.out2 = nullptr; g.b2_bstride = 0;
    g.stats = nullptr;
    g.cstats = nullptr; g.cstats_goff = 0; g.cstats_bstride = 0;
    g.ldb = ldy; g.krows = (int)P;
    g.lda = d->C; g.krows_a = (int)P;
    const int nx = cdiv(d->K, 256), ny = cdiv(d->C, 256), tiles = nx * ny * nb, ntile = g.K / 64;
    // one workgroup per CU (128 KB of LDS): ~256 workgroups per problem (the frames of a layer are launched back to back), every
    // part at least 8 K-tiles long
    // parts per tile: whole rounds of 256 workgroups (one per CU), each round as long as its share of the reduction plus the
    // atomic epilogue (~20 K-tiles' worth): minimise rounds x (K-tiles per part + 20)
    int sp = 1;
    {
        long long best = -1;
        for (int c = 1; c <= 64 && ntile / c >= 8; ++c) {
            const long long rounds = ((long long)tiles * c + 255) / 256, cost = rounds * ((ntile + c - 1) / c + 20);
            if (best < 0 || cost < best) { best = cost; sp = c; }
        }
    }
    g.flat_nx = nx; g.flat_ny = ny; g.split_r = tiles; g.split_s = sp;
    const dim3 grid((unsigned)(tiles * sp), 1, 1);
    hipLaunchKernelGGL((gemm_nt256_kernel<0, 4, 1, 1>), grid, dim3(512), 0, (hipStream_t)stream, g);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "gemm_tt256: %s", hipGetErrorString(e));
    return 1;
}

// ---- GuidedCxtAtten backward, fused: T[b][i][j] = P[b][i][j] * (sum_v dO[b][i][v] V[b][j][v] - delta[b][i]) * c[b][j] (bf16,
// zero in the padding columns N <= j < ld), delta[b][i] = sum_j P dP = <dO[b][i], O[b][i]>.  Replaces the fp32 dP GEMM
// (800 MB written and read back per 3-frame launch at 1080p) + tcvom_row_softmax_bwd of models/GCA/ops.py:190's backward.
extern "C" int tcvom_gca_dp_softmax_bwd(const void* dO, const void* V, const void* P, const float* delta, const float* cvec, void* T,
                                        void* Tt, void* Pt, int32_t N, int32_t DV, int64_t ld, int32_t batch, void* stream) {
    TCVOM_CHECK_ARG(dO && V && P && delta && cvec && T, "gca_dp_softmax_bwd: null pointer");
    TCVOM_CHECK_ARG(N > 0 && N % 4 == 0 && DV % 64 == 0 && ld >= N && ld % 4 == 0 && batch >= 1, "gca_dp_softmax_bwd: N=%d DV=%d ld=%lld", N, DV, (long long)ld);
    TCVOM_CHECK_ARG(((uintptr_t)cvec % 16) == 0 && ((uintptr_t)P % 8) == 0 && ((uintptr_t)T % 8) == 0, "gca_dp_softmax_bwd: alignment");
    Gemm256Args g;
    g.A = (const h16raw*)V;          // rows m = keys j
    g.B = (const h16raw*)dO;         // columns n = queries i
    g.out = T;
    g.bias = nullptr;
    g.mscale = cvec;
    g.mdiag = nullptr;
    g.zero_page = tcvom_zero_page();
    TCVOM_CHECK_ARG(g.zero_page != nullptr, "gca_dp_softmax_bwd: could not allocate the zero page");
    g.M = N; g.N = N; g.K = DV; g.ldo = (int)ld; g.act = 0; g.out_fp32 = 0; g.batch = batch;
    g.a_bstride = (long long)N * DV;
    g.b_bstride = (long long)N * DV;
    g.out_bstride = (long long)N * ld;
    g.vec_bstride = N;
    g.P = (const h16raw*)P;
    g.delta = delta;
    TCVOM_CHECK_ARG((Tt == nullptr) == (Pt == nullptr) && (!Tt || ld % 256 == 0), "gca_dp_softmax_bwd: the transposed copies come together and need ld %% 256 == 0");
    g.Tt = (h16raw*)Tt; g.Pt = (h16raw*)Pt; g.ldt = (int)ld;
    g.B2 = nullptr; g.out2 = nullptr; g.b2_bstride = 0;
    g.stats = nullptr;
    g.cstats = nullptr; g.cstats_goff = 0; g.cstats_bstride = 0;
    g.flat_nx = g.flat_ny = g.split_r = g.split_s = 0;
    g.ldb = 0; g.krows = 0; g.lda = 0; g.krows_a = 0;
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((ld + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL((gemm_nt256_kernel<2, 4>), grid, dim3(512), 0, (hipStream_t)stream, g);
    TCVOM_LAUNCH_CHECK("gca_dp_softmax_bwd");
    return TCVOM_OK;
}


// ---- GuidedCxtAtten forward, scores + softmax without the fp32 score matrix (models/GCA/ops.py:177-190):
//   P[b][i][j] = softmax_j( c[b][j] <G[b][i], G[b][j]> - d[b][j] [i == j] ),   zeros in the padding columns N <= j < ld.
// Pass 1 (gemm_nt256 EPI 3): every 256 x 256 tile writes exp(S' - its own row maxima) as 16-bit numbers and (row max, row sum)
// of the tile into stats[b][i][tile_j][2]; pass 2 rescales every row by exp(tile max - row max) / row sum, in place.  The fp32
// N x N matrix (802 MB written and read back per 3-frame launch at 1080p) is never stored: 401 MB written, read and rewritten.
__global__ __launch_bounds__(256) void softmax_rescale_kernel(uint4* __restrict__ P, const float* __restrict__ stats, int N, int ld8, int tmt,
                                                              int flush_tiny) {
    __shared__ float f[64];
    const int64_t row = blockIdx.x;
    const float* st = stats + row * tmt * 2;
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        const float mx = t < tmt ? st[2 * t] : -__builtin_inff(), sm = t < tmt ? st[2 * t + 1] : 0.f;
        const float M = wave_max(mx);
        const float e = t < tmt ? __expf(mx - M) : 0.f;
        const float Lsum = wave_sum(sm * e);
        f[t] = e / Lsum;
    }
    __syncthreads();
    uint4* p = P + row * ld8;
    const int n8 = N >> 3;                               // N % 8 == 0
    for (int i = threadIdx.x; i < ld8; i += 256) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);            // padding columns N <= j < ld
        if (i < n8) {
            float x[8];
            unpack8(p[i], x);
            const float s = f[i >> 5];                   // 32 octets per 256-column tile
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] *= s;
#ifndef TCVOM_F16
            // bf16 keeps probabilities down to 1e-38 that carry nothing (fp16 rounds everything below 2^-25 to zero, and its parity is the
            // tighter one): as exact zeros they cost the matrix cores no switching power in O = P V, dP, dV and -- through T = P (..) --
            // dq / dk, which run 5-8 % faster on the fp16 build's mostly-zero matrices at the same instruction stream (sustained clock)
            if (flush_tiny) {
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = x[k] < 2.98023224e-08f ? 0.f : x[k];
            }
#endif
            v = pack8(x);
        }
        p[i] = v;
    }
}

extern "C" int tcvom_gca_scores_softmax_ok(int32_t N, int32_t D, int64_t ld, int32_t batch) {
    return (N >= 256 && N % 8 == 0 && D % 64 == 0 && ld >= N && ld % 256 == 0 && ld / 256 <= 64 && batch >= 1) ? 1 : 0;
}
// pass 1 alone (then tcvom_gca_softmax_rescale): P = exp(S' - tile row max), stats[b][i][tile][2] = (tile row max, tile row sum)
extern "C" int tcvom_gca_scores_exp(const void* G, const float* cvec, const float* dvec, void* P, float* stats, int32_t N, int32_t D,
                                    int64_t ld, int32_t batch, void* stream) {
    TCVOM_CHECK_ARG(G && cvec && P && stats, "gca_scores_exp: null pointer");
    TCVOM_CHECK_ARG(tcvom_gca_scores_softmax_ok(N, D, ld, batch), "gca_scores_exp: N=%d D=%d ld=%lld (N %% 8, D %% 64, ld %% 256 == 0, ld <= 16384)",
                    N, D, (long long)ld);
    TCVOM_CHECK_ARG(((uintptr_t)cvec % 16) == 0 && (!dvec || ((uintptr_t)dvec % 16) == 0) && ((uintptr_t)P % 16) == 0, "gca_scores_exp: alignment");
    Gemm256Args g;
    g.A = (const h16raw*)G;          // rows m = keys j
    g.B = (const h16raw*)G;          // columns n = queries i
    g.out = P;
    g.bias = nullptr;
    g.mscale = cvec;
    g.mdiag = dvec;
    g.zero_page = tcvom_zero_page();
    TCVOM_CHECK_ARG(g.zero_page != nullptr, "gca_scores_exp: could not allocate the zero page");
    g.M = N; g.N = N; g.K = D; g.ldo = (int)ld; g.act = 0; g.out_fp32 = 0; g.batch = batch;
    g.a_bstride = (long long)N * D;
    g.b_bstride = (long long)N * D;
    g.out_bstride = (long long)N * ld;
    g.vec_bstride = N;
    g.P = nullptr; g.delta = nullptr; g.Tt = nullptr; g.Pt = nullptr; g.ldt = 0;
    g.B2 = nullptr; g.out2 = nullptr; g.b2_bstride = 0;
    g.stats = stats;
    g.cstats = nullptr; g.cstats_goff = 0; g.cstats_bstride = 0;
    g.flat_nx = g.flat_ny = g.split_r = g.split_s = 0;
    g.ldb = 0; g.krows = 0; g.lda = 0; g.krows_a = 0;
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)(ld / 256), (unsigned)batch);
    hipLaunchKernelGGL((gemm_nt256_kernel<3, 4>), grid, dim3(512), 0, (hipStream_t)stream, g);
    TCVOM_LAUNCH_CHECK("gca_scores_exp");
    return TCVOM_OK;
}
// pass 2: every row scaled by exp(tile max - row max) / row sum, in place; zeros in the padding columns N <= j < ld
extern "C" int tcvom_gca_softmax_rescale(void* P, const float* stats, int32_t N, int64_t ld, int32_t batch, void* stream) {
    TCVOM_CHECK_ARG(P && stats && N > 0 && N % 8 == 0 && ld >= N && ld % 256 == 0 && ld / 256 <= 64 && batch >= 1 && ((uintptr_t)P % 16) == 0,
                    "gca_softmax_rescale: N=%d ld=%lld", N, (long long)ld);
    static const int flush = getenv("TCVOM_NO_P_FLUSH") ? 0 : 1;                  // A/B switch (bf16 build)
    hipLaunchKernelGGL(softmax_rescale_kernel, dim3((unsigned)((int64_t)batch * N)), dim3(256), 0, (hipStream_t)stream, (uint4*)P, stats, N,
                       (int)(ld / 8), (int)(ld / 256), flush);
    TCVOM_LAUNCH_CHECK("gca_softmax_rescale");
    return TCVOM_OK;
}
extern "C" int tcvom_gca_scores_softmax(const void* G, const float* cvec, const float* dvec, void* P, float* stats, int32_t N, int32_t D,
                                        int64_t ld, int32_t batch, void* stream) {
    const int rc = tcvom_gca_scores_exp(G, cvec, dvec, P, stats, N, D, ld, batch, stream);
    return rc != TCVOM_OK ? rc : tcvom_gca_softmax_rescale(P, stats, N, ld, batch, stream);
}


// ---- GuidedCxtAtten backward, the two products that contract the ROW index of an N x N matrix, reading it as it lies in memory
// (k-major B operand through transposing LDS reads) instead of a transposed copy:
//   dV[b][j][v] = sum_{i < N} P[b][i][j] dO[b][i][v]            (tcvom_gca_dv:    A = dO [N][DV] k-major, B = P [N][ld] k-major)
//   dWq[b][i][d] = sum_j T[b][i][j] G[b][j][d],  M'[b][j][d] = sum_i T[b][i][j] G[b][i][d]
//                                                               (tcvom_gca_dq_dk: A = Gt [D][ld]; one launch, T read both ways)
// A = dO / V [N][DV] k-major (rows beyond N count as zeros), shared set-up of tcvom_gca_dv / tcvom_gca_pv
static void g256_gca_common(Gemm256Args& g, const void* Amat, int32_t N, int32_t DV, int64_t ld, int32_t batch) {
    g.A = (const h16raw*)Amat;
    g.bias = nullptr; g.mscale = nullptr; g.mdiag = nullptr;
    g.zero_page = tcvom_zero_page();
    g.M = DV; g.N = N; g.K = (int)ld; g.ldo = DV; g.act = 0; g.out_fp32 = 1; g.batch = batch;
    g.a_bstride = (long long)N * DV;
    g.b_bstride = (long long)N * ld;
    g.out_bstride = (long long)N * DV;
    g.vec_bstride = 0;
    g.P = nullptr; g.delta = nullptr; g.Tt = nullptr; g.Pt = nullptr; g.ldt = 0;
    g.B2 = nullptr; g.out2 = nullptr; g.b2_bstride = 0;
    g.stats = nullptr;
    g.cstats = nullptr; g.cstats_goff = 0; g.cstats_bstride = 0;
    g.flat_nx = g.flat_ny = g.split_r = g.split_s = 0;
    g.ldb = 0; g.krows = 0;
    g.lda = DV; g.krows_a = N;
}
extern "C" int tcvom_gca_dv(const void* P, const void* dO, float* dV, int32_t N, int32_t DV, int64_t ld, int32_t batch, void* stream) {
    TCVOM_CHECK_ARG(P && dO && dV, "gca_dv: null pointer");
    TCVOM_CHECK_ARG(N >= 256 && DV >= 256 && DV % 8 == 0 && ld >= N && ld % 256 == 0 && batch >= 1, "gca_dv: N=%d DV=%d ld=%lld", N, DV, (long long)ld);
    TCVOM_CHECK_ARG(((uintptr_t)P % 16) == 0 && ((uintptr_t)dO % 16) == 0 && ((uintptr_t)dV % 16) == 0, "gca_dv: alignment");
    Gemm256Args g;
    g256_gca_common(g, dO, N, DV, ld, batch);        // A: rows k = queries i, columns m = value channels v
    TCVOM_CHECK_ARG(g.zero_page != nullptr, "gca_dv: could not allocate the zero page");
    g.B = (const h16raw*)P;          // k-major: row k = query i, columns n = keys j
    g.out = dV;
    g.ldb = (int)ld; g.krows = N;
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((DV + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL((gemm_nt256_kernel<0, 4, 1, 1>), grid, dim3(512), 0, (hipStream_t)stream, g);
    TCVOM_LAUNCH_CHECK("gca_dv");
    return TCVOM_OK;
}
// O[b][i][v] = sum_{j < N} P[b][i][j] V[b][j][v]      (P: [batch][N][ld] row-major = the NT B operand; V: [batch][N][DV] k-major A)
extern "C" int tcvom_gca_pv(const void* P, const void* V, float* O, int32_t N, int32_t DV, int64_t ld, int32_t batch, void* stream) {
    TCVOM_CHECK_ARG(P && V && O, "gca_pv: null pointer");
    TCVOM_CHECK_ARG(N >= 256 && DV >= 256 && DV % 8 == 0 && ld >= N && ld % 64 == 0 && batch >= 1, "gca_pv: N=%d DV=%d ld=%lld", N, DV, (long long)ld);
    TCVOM_CHECK_ARG(((uintptr_t)P % 16) == 0 && ((uintptr_t)V % 16) == 0 && ((uintptr_t)O % 16) == 0, "gca_pv: alignment");
    Gemm256Args g;
    g256_gca_common(g, V, N, DV, ld, batch);         // A: rows k = keys j, columns m = value channels v
    TCVOM_CHECK_ARG(g.zero_page != nullptr, "gca_pv: could not allocate the zero page");
    g.B = (const h16raw*)P;          // rows n = queries i, k = keys j contiguous (row stride K = ld)
    g.out = O;
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((DV + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL((gemm_nt256_kernel<0, 4, 0, 1>), grid, dim3(512), 0, (hipStream_t)stream, g);
    TCVOM_LAUNCH_CHECK("gca_pv");
    return TCVOM_OK;
}

extern "C" int tcvom_gca_dq_dk(const void* T, const void* Gt, float* dWq, float* Mp, int32_t N, int32_t D, int64_t ld, int32_t batch,
                               void* stream) {
    TCVOM_CHECK_ARG(T && Gt && dWq && Mp, "gca_dq_dk: null pointer");
    TCVOM_CHECK_ARG(N >= 256 && D >= 64 && D % 4 == 0 && ld >= N && ld % 256 == 0 && batch >= 1, "gca_dq_dk: N=%d D=%d ld=%lld", N, D, (long long)ld);
    TCVOM_CHECK_ARG(((uintptr_t)T % 16) == 0 && ((uintptr_t)Gt % 16) == 0 && ((uintptr_t)dWq % 16) == 0 && ((uintptr_t)Mp % 16) == 0, "gca_dq_dk: alignment");
    Gemm256Args g;
    g.A = (const h16raw*)Gt;         // rows m = channels d, k contiguous
    g.bias = nullptr; g.mscale = nullptr; g.mdiag = nullptr;
    g.zero_page = tcvom_zero_page();
    TCVOM_CHECK_ARG(g.zero_page != nullptr, "gca_dq_dk: could not allocate the zero page");
    g.M = D; g.N = N; g.K = (int)ld; g.ldo = D; g.act = 0; g.out_fp32 = 1; g.batch = batch;
    g.a_bstride = (long long)D * ld;
    g.b_bstride = (long long)N * ld;
    g.out_bstride = (long long)N * D;
    g.vec_bstride = 0;
    g.P = nullptr; g.delta = nullptr; g.Tt = nullptr; g.Pt = nullptr; g.ldt = 0;
    g.B2 = nullptr; g.out2 = nullptr; g.b2_bstride = 0;
    g.stats = nullptr;
    g.cstats = nullptr; g.cstats_goff = 0; g.cstats_bstride = 0;
    g.lda = 0; g.krows_a = 0;
    g.B = (const h16raw*)T;
    // Two launches, each with its own K-split tail (3 frames at 1080p: 288 tiles = one round of 256 + 32 tiles x 8 eighth-length
    // workgroups, 1.125 rounds each -- what the paired launch of tcvom_gemm_pair reaches with 576 tiles in one grid)
    for (int second = 0; second < 2; ++second) {
        g.out = second ? Mp : dWq;
        g.ldb = second ? (int)ld : 0;            // second product: T k-major -- row k = query i, columns n = keys j
        g.krows = second ? N : 0;
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)cdiv(D, 192), (unsigned)batch);
        const int rc = g256_split_tail(g, grid, true, g.out, batch, stream);
        if (rc < 0) return rc;
        if (second) hipLaunchKernelGGL((gemm_nt256_kernel<0, 3, 1>), grid, dim3(512), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((gemm_nt256_kernel<0, 3, 0>), grid, dim3(512), 0, (hipStream_t)stream, g);
    }
    TCVOM_LAUNCH_CHECK("gca_dq_dk");
    return TCVOM_OK;
}
