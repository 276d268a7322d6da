// Weight-stationary streaming 1x1 convolution for the reduction-poor pointwise layers: the expand convs of the FBA bottlenecks and
// the data gradients of their reduce convs (models/FBA/resnet_GN_WS.py:112-137: 64 -> 256 .. 512 -> 2048 at os4 / os8), the
// downsample / projection 1x1 convs of the GCA trunk (models/GCA/encoders/resnet_enc.py:76-84, decoders/resnet_dec.py:61-72) and
// their data gradients: stride 1, one tap, C in {32 .. 512} input channels, 16-bit output.
//
// Why: with C <= 512 the reduction is 1 .. 8 K-tiles of the 256 x 256 GEMM (gemm256.hip) or 1 .. 16 k-steps of the implicit GEMM
// (igemm.hip); both spend a tile's life in prologue and epilogue -- measured 1.5 - 2.4 TB/s of algorithmic traffic for layers whose
// roofline is the HBM rate (1024 <- 256 at os8: 250 MB in 150 us; 128 <- 64 at os8: 37 MB in 24 us).  Here
//   * a persistent workgroup (4 waves, two workgroups per CU) keeps the weights of its 32 .. 128 output channels in REGISTERS
//     (C / 16 A fragments per wave) for its whole run of pixel tiles;
//   * the input streams through a double-buffered LDS-DMA ring, 32 KiB per tile (TP = 16384 / C pixels), counted vmcnt: the DMA of
//     tile t + 1 and the stores of tile t - 1 are in flight under the MFMAs of tile t;
//   * results leave as whole 64-byte row segments: every wave turns its 32 pixel x 32 channel fragment through a private 2 KiB LDS
//     region (a lane's MFMA results are 4 channels of ONE pixel -- stored directly that is 64 eight-byte segments per instruction);
//   * the per-channel (sum, sum of squares) for the BatchNorm / GroupNorm that follows are running sums in registers, reduced once
//     per workgroup (one statistics group per workgroup and pixel group, as wsconv<64>).
// The workgroups of the K / 128 output slices of one pixel run are neighbours in the XCD-aware order: they read the same input
// tiles through one L2.
//
//   MFMA 32x32x16: A = weights [32 out-channels][16 k] (registers), B = pixels [32 pixels][16 k] (LDS tile).
// LDS tile: pixel-major rows of 2 C bytes; 16-byte chunk c of pixel p sits at slot c ^ f(p) (applied on the DMA source side, the
// LDS write stays lane-linear): f = p & 15 for >= 256-byte pixels, (p >> 1) & 7 for 128-byte, (p >> 2) & 3 for 64-byte ones, so
// that the 16 lanes of a ds_read_b128 group (16 consecutive pixels, one chunk) cover 16 distinct slots of the 256-byte bank line.
#include <cstdlib>
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned int pw_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int pw_u32x2_t;
typedef __attribute__((address_space(3))) void* pw_lptr_t;
#ifndef PW_ABL
#define PW_ABL 0              // study builds (wrong results): 1 = no output stores, 2 = no statistics, 4 = no DMA after the first tile, 8 = no MFMAs
#endif
#ifndef PW_DMA_AT_HEAD
#define PW_DMA_AT_HEAD 0      // study builds: 1 = the next tile's DMA instructions in one burst behind the barrier
#endif

struct PwArgs {
    const h16raw* in;
    const h16raw* wgt;
    h16raw* out;
    const float* bias;
    float* stats;
    int P;                          // pixels of one frame
    int K, ldo, act;
    int nslices, chunks;            // output-channel slices (32 MF channels each), pixel runs per frame
    int tiles_per_frame, tiles_per_wg;
    long long w_bstride;            // elements between the frames' weight copies (0: shared)
    int stats_group_offset;
    long long stats_bstride;
};

template <int C, int MF>
struct PwCfg {
    static constexpr int PS = 4 / MF;                            // pixel groups: wave = (mf, ps)
    static constexpr int CU = C / 8, PIXB = C * 2, NCC = C / 16;
    static constexpr int TP0 = C <= 64 ? 256 : 16384 / C;
    static constexpr int TP = TP0 < 128 * PS ? TP0 : 128 * PS;   // pixels per tile: <= 32 KiB and <= 4 fragments per wave
    static constexpr int TB = TP * PIXB;                         // bytes per tile buffer
    static constexpr int NI = TP / 32 / PS;                      // 32-pixel fragments per wave and tile
    static constexpr int NDMA = TB / 1024 / 4;                   // DMA wave-instructions per wave and tile
    static constexpr int NST = NI * 2;                           // store instructions per wave and tile
    static constexpr int STAGE = 2048;                           // bytes of a wave's output staging region
    static constexpr int LDS = 2 * TB + 4 * STAGE;
    static_assert(MF == 1 || MF == 2 || MF == 4, "1, 2 or 4 channel blocks per workgroup");
    static_assert(NI >= 1 && TP % (32 * PS) == 0 && TB % 4096 == 0, "tile does not split over the waves");
    static_assert(NST <= 32, "counted vmcnt");
    static constexpr int swz(int p) { return CU >= 16 ? (p & 15) : CU == 8 ? ((p >> 1) & 7) : ((p >> 2) & 3); }
};

template <int C, int MF>
__global__ __launch_bounds__(256, 2) void pwconv_kernel(const PwArgs a) {
    typedef PwCfg<C, MF> G;
    constexpr int PS = G::PS, NI = G::NI, NCC = G::NCC, TP = G::TP, TB = G::TB, CU = G::CU, PIXB = G::PIXB;
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [2][TB] input tiles, [4][STAGE] output staging
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mf = wave / PS, ps = wave % PS;
    const int half = lane >> 5, col = lane & 31;

    // unit = ((frame * chunks + chunk) * nslices + slice), XCD-aware: consecutive units (the slices of one pixel run first) share an XCD
    int unit;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int slice = unit % a.nslices;
    const int fc = unit / a.nslices, chunk = fc % a.chunks, frame = fc / a.chunks;
    const int t_begin = chunk * a.tiles_per_wg;
    const int t_end = min(a.tiles_per_frame, t_begin + a.tiles_per_wg);
    const int kbase = slice * (32 * MF) + mf * 32;               // first output channel of this wave

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<h16raw*>(a.in) + (long long)frame * a.P * C, 0, (unsigned)((long long)a.P * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + (long long)frame * a.P * a.ldo, 0, (unsigned)((long long)a.P * a.ldo * 2), 0x00020000);

    // ---- DMA of one tile: instruction j of this wave covers the 64 16-byte units (j * 4 + wave) * 64 .. of the tile buffer; unit u =
    // (pixel u / CU, slot u % CU) holds chunk slot ^ f(pixel).  Out-of-frame pixels use an out-of-range offset: zeros.
    // p0 < 0: nothing to fetch (the slot of a tile past the run: zeros into the idle buffer, no branch in the MFMA stream)
    auto issue_piece = [&](int p0, int buf, int j) {
        const int u = (j * 4 + wave) * 64 + lane, p = u / CU, s = u % CU;
        const int c8 = s ^ G::swz(p);
        const unsigned off = (p0 >= 0 && p0 + p < a.P) ? (unsigned)(((p0 + p) * C + c8 * 8) * 2) : 0xffffffffu;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (pw_lptr_t)(lds + buf * TB + (j * 4 + wave) * 1024), 16, (int)off, 0, 0, 0);
#else
        (void)off;
#endif
    };
    if (t_begin < t_end) {
#pragma unroll
        for (int j = 0; j < G::NDMA; ++j) issue_piece(t_begin * TP, 0, j);
    }

    // ---- weights -> registers: A fragment cc = rows kbase + col, channels cc * 16 + half * 8 .. + 7
    h16x8_t wr[NCC];
    {
        const h16raw* wsrc = a.wgt + (long long)frame * a.w_bstride + (long long)(kbase + col) * C + half * 8;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) wr[cc] = *reinterpret_cast<const h16x8_t*>(wsrc + cc * 16);
    }
    f32x16_t binit;
#pragma unroll
    for (int r = 0; r < 16; ++r) binit[r] = a.bias ? a.bias[kbase + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
    const float slope = a.act == 1 ? 0.f : (a.act == 3 ? 0.01f : 1.f);
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[g][r] = s2[g][r] = 0.f;
    // the BUILTIN wait (everything in flight, the first tile's DMA included) is modelled by the compiler's waitcnt pass: the
    // weight / bias registers cost no vmcnt(0) inside the tile loop, where it would drain the next tile's DMA
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // B fragment (i, cc): pixel (ps * NI + i) * 32 + col, chunk cc * 2 + half at slot chunk ^ f(pixel); f is the same for every i
    const int bswz = G::swz(col);
    const char* brow = lds + ((ps * NI) * 32 + col) * PIXB;
    // output staging: this wave's region, row = pixel (64 bytes = its 32 channels), 16-byte slot g at g ^ ((row >> 2) & 3)
    char* stg = lds + 2 * TB + wave * G::STAGE;
    char* stw = stg + col * 64 + half * 8;
    const int wswz = (col >> 2) & 3;
    unsigned stw_addr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) stw_addr[g] = (unsigned)(uintptr_t)(pw_lptr_t)(stw + ((g ^ wswz) << 4));
    const int rq = lane >> 2, rc = lane & 3;                     // read side: pixel rq (+16), slot rc

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        // the DMA of this tile is the oldest thing in flight; the stores of the previous tile (issued after it) may stay
        if (tile == t_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NST) : "memory");
        __builtin_amdgcn_s_barrier();               // tile landed for every wave; every wave is done reading the other buffer
#if PW_DMA_AT_HEAD
        {
            const int np0h = tile + 1 < t_end ? (tile + 1) * TP : -1;
#pragma unroll
            for (int j = 0; j < G::NDMA; ++j) issue_piece(np0h, buf ^ 1, j);
        }
#endif
        // the DMA instructions of the next tile ride between the MFMAs of this one (an LDS-DMA instruction holds its wave's issue
        // for ~100 cycles: all of them at the head of the tile were NDMA x 100 cycles in which this wave fed no MFMA)
        const int np0 = tile + 1 < t_end ? (tile + 1) * TP : -1;
        f32x16_t acc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = binit;
        const char* bt = brow + buf * TB;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            const int so = ((cc * 2 + half) ^ bswz) << 4;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                constexpr int TOTAL = NCC * NI;
                const int m = cc * NI + i;                       // (compile-time after unrolling)
#if !PW_DMA_AT_HEAD
                // DMA instruction j goes out in front of MFMA j * STEP (+ 1 where there is room: the first MFMA of the tile starts at once).
                // (As a loop over j with `m == f(j)` tests the nest no longer unrolled at 64 k-steps x 16 instructions -- a C = 1024
                //  instantiation, 2 x 64 KiB tiles and one workgroup per CU, kept its weights in scratch; with that fixed it measured
                //  neutral on the FBA step, 46.21 vs 46.16 ms, and is not built.)
                constexpr int STEP = TOTAL / G::NDMA, OFF = STEP >= 2 ? 1 : 0;
                static_assert(TOTAL % G::NDMA == 0, "the DMA instructions spread evenly over the MFMAs of a tile");
                if (!(PW_ABL & 4) && m % STEP == OFF) issue_piece(np0, buf ^ 1, m / STEP);
#endif
                const h16x8_t b = *reinterpret_cast<const h16x8_t*>(bt + i * 32 * PIXB + so);
                if (!(PW_ABL & 8)) acc[i] = mfma16(wr[cc], b, acc[i], 0, 0, 0);
                else acc[i][cc & 15] += (float)b[0];
            }
        }
        // ---- epilogue: activation, statistics, 16-bit rows through the staging region
        const int pw0 = tile * TP + ps * NI * 32;   // first pixel of this wave's fragments
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool pin = pw0 + i * 32 + col < a.P;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][g * 4 + r];
                    x = fmaxf(x, x * slope);
                    x = pin ? x : 0.f;
                    v[r] = x;
                    if (!(PW_ABL & 2)) {
                        s1[g][r] += x;
                        s2[g][r] = fmaf(x, x, s2[g][r]);
                    }
                }
                // (inline asm: a compiler-visible LDS WRITE gets an s_waitcnt vmcnt(0) in front of it while an LDS-DMA is in flight --
                //  the next tile's DMA would be drained at every tile's epilogue)
                const pw_u32x2_t pk = {pack2h(v[0], v[1]), pack2h(v[2], v[3])};
                asm volatile("ds_write_b64 %0, %1" ::"v"(stw_addr[g]), "v"(pk) : "memory");
            }
            // (LDS operations of one wave execute in order: no barrier between the writes above and the reads below)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int q = rq + 16 * jj;
                const pw_u32x4_t row = *reinterpret_cast<const pw_u32x4_t*>(stg + q * 64 + ((rc ^ ((q >> 2) & 3)) << 4));
                const int px = pw0 + i * 32 + q;
                const unsigned o = px < a.P ? (unsigned)((px * a.ldo + kbase + rc * 8) * 2) : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b128(row, orsrc, (int)((PW_ABL & 1) ? 0xffffffffu : o), 0, 0);
            }
        }
    }
    // ---- statistics: one group per (workgroup, pixel group): [group][2][K]
    if (a.stats && t_begin < t_end) {
        float* sp = a.stats + ((long long)a.stats_group_offset + (long long)frame * a.stats_bstride + chunk * PS + ps) * 2 * a.K;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { t[r] = s1[g][r]; t[4 + r] = s2[g][r]; }
            reduce8_store(t, lane, sp, a.K, kbase + 8 * g + 4 * half, true);
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
struct PwPlan { bool ok; int C, mf, tp, ps; };

static PwPlan pw_plan(const tcvom_conv_desc* d, int nphase) {
    PwPlan p;
    p.ok = false;
    static const bool disabled = getenv("TCVOM_NO_PWCONV") != nullptr;          // A/B switch
    constexpr int maxc = 512;
    if (disabled || nphase != 1 || d->ntaps != 1 || d->tap_w[0] != 0 || d->wt != 1 || d->tap_dh[0] != 0 || d->tap_dw[0] != 0) return p;
    if (d->w_layout != 0 || d->out_fp32) return p;
    if (d->in_f16 && !TCVOM_BUILD_F16) return p;          // IEEE fp16 operands in the bf16 build: igemm_nt
    if (d->in_step != 1 || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return p;
    if (d->PH != d->H || d->PW != d->W || d->OH != d->H || d->OW != d->W) return p;
    const int C = d->C, K = d->K;
    if (!(C == 32 || C == 64 || C == 128 || C == 256 || C == 512) || C > maxc) return p;
    p.mf = K % 128 == 0 ? 4 : K == 64 ? 2 : K == 32 ? 1 : 0;
    if (p.mf == 0 || d->ldo % 8 != 0 || d->ldo < K) return p;
    p.ps = 4 / p.mf;
    p.tp = C <= 64 ? 256 : 16384 / C;
    if (p.tp < 32 * p.ps) return p;                       // (C = 256 with K = 32, C = 512 with K < 128: no instantiation)
    if (p.tp > 128 * p.ps) p.tp = 128 * p.ps;             // (PwCfg::TP)
    const long long P = (long long)d->N * d->H * d->W;
    constexpr int minp = 1024;
    if (P < minp || P * C >= (1ll << 30) || P * d->ldo >= (1ll << 30)) return p;        // 32-bit byte offsets inside a frame
    const int nb = d->batch > 1 ? d->batch : 1;
    if (nb > 1) {
        if (d->in_bstride != P * C || d->out_bstride != P * d->ldo || d->vec_bstride != 0) return p;
    }
    p.C = C;
    p.ok = true;
    return p;
}

// persistent workgroups: two per CU over all frames and slices of the launch
static void pw_grid(const tcvom_conv_desc* d, const PwPlan& p, int* nslices, int* chunks, int* tiles_per_frame, int* tiles_per_wg) {
    const int nb = d->batch > 1 ? d->batch : 1;
    const long long P = (long long)d->N * d->H * d->W;
    *nslices = d->K / (32 * p.mf);
    *tiles_per_frame = cdiv(P, p.tp);
    int c = 512 / (nb * *nslices);
    if (c < 1) c = 1;
    if (c > *tiles_per_frame) c = *tiles_per_frame;
    *tiles_per_wg = cdiv(*tiles_per_frame, c);
    *chunks = cdiv(*tiles_per_frame, *tiles_per_wg);
}

// statistics groups ONE frame of the launch writes (one per pixel run and pixel group), or 0 when the shape is not handled here
int pwconv_stats_groups(const tcvom_conv_desc* d, int nphase) {
    const PwPlan p = pw_plan(d, nphase);
    if (!p.ok) return 0;
    int ns, ch, tpf, tpw;
    pw_grid(d, p, &ns, &ch, &tpf, &tpw);
    return ch * p.ps;
}

const char* pwconv_variant(const tcvom_conv_desc* d, int nphase) {
    const PwPlan p = pw_plan(d, nphase);
    if (!p.ok) return nullptr;
    static const char* names[5][3] = {{"pwconv<32,1>", "pwconv<32,2>", "pwconv<32,4>"}, {"pwconv<64,1>", "pwconv<64,2>", "pwconv<64,4>"},
                                      {"pwconv<128,1>", "pwconv<128,2>", "pwconv<128,4>"}, {"pwconv<256,1>", "pwconv<256,2>", "pwconv<256,4>"},
                                      {"pwconv<512,1>", "pwconv<512,2>", "pwconv<512,4>"}};
    const int ci = p.C == 32 ? 0 : p.C == 64 ? 1 : p.C == 128 ? 2 : p.C == 256 ? 3 : 4, mi = p.mf == 1 ? 0 : p.mf == 2 ? 1 : 2;
    return names[ci][mi];
}

template <int C, int MF>
static hipError_t pw_launch(const PwArgs& a, int grid, hipStream_t st) {
    typedef PwCfg<C, MF> G;
    static_assert(2 * G::LDS <= 160 * 1024, "two workgroups per CU");
    static bool attr = false;
    hipError_t e = hipSuccess;
    if (!attr) {
        e = hipFuncSetAttribute((const void*)pwconv_kernel<C, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        attr = true;
    }
    hipLaunchKernelGGL((pwconv_kernel<C, MF>), dim3(grid), dim3(256), G::LDS, st, a);
    return e;
}

// returns 1 when the conv was launched here, 0 when the caller should use another kernel, < 0 on error
int pwconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                      float* stats, const tcvom_conv_desc* d, int nphase, void* stream) {
    const PwPlan p = pw_plan(d, nphase);
    if (!p.ok) return 0;
    // The plan took the shape, so tcvom_conv_stats_groups told the caller THIS kernel's statistics layout: a launch-time reason to decline
    // (a column scale / diagonal term, an operand that is not 16-byte aligned) must not silently hand the launch to a kernel that writes
    // another number of groups into the buffer sized for this one -- without statistics the fall-through is harmless.
    const bool decline = mscale || mdiag || (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) != 0;
    if (decline && stats)
        return tcvom_fail(TCVOM_ERR_ARG, "pwconv: statistics were sized for the pointwise kernel but the launch cannot use it (column scale / "
                                         "diagonal term or an operand that is not 16-byte aligned)");
    if (decline) return 0;
    PwArgs a;
    a.in = (const h16raw*)in;
    a.wgt = (const h16raw*)w;
    a.out = (h16raw*)out;
    a.bias = bias;
    a.stats = stats;
    a.P = d->N * d->H * d->W;
    a.K = d->K; a.ldo = d->ldo; a.act = d->act;
    const int nb = d->batch > 1 ? d->batch : 1;
    a.w_bstride = nb > 1 ? d->w_bstride : 0;
    a.stats_group_offset = d->stats_group_offset;
    a.stats_bstride = nb > 1 ? d->stats_bstride : 0;
    pw_grid(d, p, &a.nslices, &a.chunks, &a.tiles_per_frame, &a.tiles_per_wg);
    if (stats && nb > 1 && d->stats_bstride < (long long)a.chunks * p.ps)
        return tcvom_fail(TCVOM_ERR_ARG, "pwconv: stats_bstride %lld < groups per frame", (long long)d->stats_bstride);
    const int grid = nb * a.chunks * a.nslices;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define PW_CASE(C_, MF_) else if (p.C == C_ && p.mf == MF_) e = pw_launch<C_, MF_>(a, grid, st)
    if (false) {}
    PW_CASE(32, 1); PW_CASE(32, 2); PW_CASE(32, 4);
    PW_CASE(64, 1); PW_CASE(64, 2); PW_CASE(64, 4);
    PW_CASE(128, 1); PW_CASE(128, 2); PW_CASE(128, 4);
    PW_CASE(256, 2); PW_CASE(256, 4);
    PW_CASE(512, 4);
    else return 0;
#undef PW_CASE
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "pwconv: %s", hipGetErrorString(e));
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "pwconv: %s", hipGetErrorString(e2));
    return 1;
}
