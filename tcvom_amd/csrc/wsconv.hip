// Weight-stationary 3x3 convolution for the 64- and 128-channel layers of the ResNet trunk (os4 / os8 of vmn_gca:
// BasicBlock convs resnet_enc.py:33-49, resnet_dec.py:43-59, shortcut branches res_gca_enc.py:47-55, the TAM
// projections VMN_model.py:13-15) and their data gradients: stride 1, same size, C == K in {64, 128}, full 3x3 stencil.
//
// Why: these layers are 9.6 GFLOP per 1080p frame whatever their level, and the implicit-GEMM kernel (igemm.hip) moves
// every input pixel through the L2 -> LDS DMA path once PER TAP plus the weight tile once per pixel tile: 32 KB per
// 2 MFLOP, which is what bounds it (measured: 57 us for the 3 frames of an os8 layer against 13 us of MFMA time, 105 us
// for an os4 layer).  Here
//   * the WEIGHTS of the layer live in REGISTERS for the lifetime of a persistent workgroup: 8 waves x 36 A fragments
//     (32 output channels x 16 k each, 144 VGPRs) = 32 x 576 x 8 = the whole [K][9][C] matrix of a 128-channel layer;
//   * the input is DMA'd (global_load_lds) ONCE per pixel tile as a (TH + 2) x (TW + 2) halo, double buffered, and all 9
//     taps read their B fragments from it with immediate offsets;
//   so the L2 -> LDS traffic per tile falls from 576 KB to 45 KB (C = 128) and LDS reads from 1.5 KB to 1 KB per MFMA.
//
// Wave roles (8 waves): wave = (mf, ks, ps).  mf = 32-row block of output channels; ks = which 64-channel window of the
// input channels this wave reduces over (C = 128: two windows -> the two partial sums of a pixel are exchanged through
// LDS at the end of a tile); ps = pixel group (C = 64: four groups of 4 fragments each).  Every wave runs 36 k-steps
// (9 taps x 4 chunks of 16 channels) x 4 pixel fragments = 144 MFMAs per tile.
//   MFMA 32x32x16 bf16:  A = weights [32 out-channels][16 k] (registers),  B = pixels [32 pixels][16 k] (LDS halo).
// LDS halo image: pixel-major; 16-byte channel chunk c of halo pixel (hy, hx) is stored at slot c ^ f(hx) (applied on
// the DMA source side, the LDS write stays lane-linear), f = hx & 15 for 256-byte pixels, (hx >> 1) & 7 for 128-byte
// pixels: the 16 lanes of a ds_read_b128 group -- 16 pixels with distinct hx mod 16 -- then cover 16 distinct slots.
#include <cstdlib>
#include <type_traits>
#include "common.h"

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}


typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef const __attribute__((address_space(1))) void* ws_gptr_t;
typedef __attribute__((address_space(3))) void* ws_lptr_t;

struct WsArgs {
    const h16raw* in;
    const h16raw* wgt;
    void* out;
    const float* bias;
    float* stats;
    const h16raw* zero_page;
    int H, W, K, ldo, wt, act;
    unsigned in_bytes, out_bytes, stats_bytes;
    int tiles_x, tiles_y, tiles_per_frame, tiles_per_wg, wgs_per_frame;
    int spf;                      // samples per frame (every frame has its own weight copy, w_bstride elements apart)
    long long w_bstride;
    int stats_group_offset;
    long long stats_bstride;
    int wslot[9];                 // weight slot of the canonical tap t = (dh + 1) * 3 + (dw + 1)
    int w_frag;                   // weights are fragment-major (tcvom_conv_desc.w_layout = 1)
    unsigned long long* trace;    // NULL, or 64 cycle stamps of workgroup 8 / wave 0 (tcvom_conv_trace_read, env TCVOM_CONV_TRACE)
};
#define WS_STAMP(i) if (tracing) a.trace[i] = __builtin_readcyclecounter()

// compile-time geometry of one kernel configuration
// OM_: output type -- 0 the build's 16-bit storage type, 2 IEEE fp16 whatever the build stores (the conv results of the fp16 island of
// the bf16 build: 11 significant bits in front of the BatchNorm at bf16's bytes)
// XF_ = 1: IEEE fp16 operands whatever the build stores (tcvom_conv_desc.in_f16: the fp16 island of the bf16 build, common.h)
template <int C_, int MF_, int PS_, int NI_, int FH_, int FW_, int OM_ = 0, int XF_ = 0>
struct WsCfg {
    static constexpr int C = C_, MF = MF_, PS = PS_, NI = NI_, FH = FH_, FW = FW_, NT = 9, OM = OM_, XF = XF_;
    static constexpr int OB = 2;                               // bytes per output element
    static constexpr int CU = C / 8, PIXB = C * 2;
    static constexpr int TH = PS * NI * FH, TW = FW, HW = TW + 2, HH = TH + 2, HPIX = HH * HW;
    static constexpr int NDMA = (HPIX * CU + 63) / 64;          // DMA wave-instructions (1 KiB each) per halo
    static constexpr int DMA_IT = (NDMA + 3) / 4;               // ... per wave
    static constexpr int SLOTB = NDMA * 1024;                   // bytes per halo buffer
    static constexpr int DUMPB = 1024;                          // where the DMA instructions past NDMA of the last round land
    static constexpr int NCC = C / 16;                          // 16-channel chunks per tap
    static constexpr int NS = NT * NCC;                         // k-steps (A fragments held by a wave)
    static constexpr int PF = 2;                                // B fragments are requested PF k-steps ahead of their MFMAs
    // schedule of the work that rides on the MFMA stream of a tile: the halo DMA of the NEXT tile (DPS instructions per
    // k-step from step 0) and the epilogue of the PREVIOUS tile (NQ pieces spread over steps E0 .. E1 - 1)
    static constexpr int DPS = 1;                               // one DMA instruction per k-step from step 0
    static_assert(DMA_IT <= NS / 2, "the halo DMA must be issued in the first half of the k-steps");
    static constexpr int E0 = (DMA_IT + DPS - 1) / DPS;
    static constexpr int E1 = NS - NS / 8;                      // the last stores get an eighth of the tile to complete
    // C = 128 (no registers to spare): the channel sums are reduced and stored per TILE by an extra piece per channel group
    // (its 40 DPP adds fit the MFMA shadow there); C = 64: 32 running sums per lane, reduced once per workgroup
    static constexpr bool SPT = C_ == 128;                      // 288 weight registers
    static constexpr int NQG = NI + (SPT ? 1 : 0);              // pieces per channel group
    static constexpr int NQ = 4 * NQG;                          // epilogue pieces: 4 channel groups x (NI fragments (+ statistics))
    static constexpr int step_of(int q) { return E0 + q * (E1 - E0) / NQ; }
    // C = 128: 288 weight + 64 accumulator registers leave no room for a register copy of the previous tile's results: they
    // wait in LDS (16 KiB per wave, [quad q = fragment * 4 + group][lane] float4) and every fragment piece fetches its
    // float4 one k-step ahead with an inline-asm read that the hand-counted lgkmcnt waits include
    static constexpr bool OLDS = SPT;
    static constexpr int XCHB = OLDS ? 4 * 16 * 1024 : 0;
    // fragment pieces scheduled AT k-step s (their LDS reads are issued in step s - 1)
    static constexpr int frag_pieces_at(int s) {
        int n = 0;
        for (int q = 0; q < NQ; ++q) n += (step_of(q) == s && q % NQG < NI) ? 1 : 0;
        return s >= E0 && s < E1 ? n : 0;
    }
    static constexpr int er(int s) { return OLDS && s >= 0 ? frag_pieces_at(s + 1) : 0; }   // reads issued in step s
    static constexpr int nb(int s) { return s >= 0 && s < NS ? NI : 0; }                    // B reads OF k-step s
    // LDS operations issued after the B reads of k-step s (issued in step s - PF) and before the wait at the TOP of step s:
    // the epilogue fetches of steps s - PF .. s - 1 and the B reads of k-steps s + 1 .. s + PF - 1
    static constexpr int later_than_b(int s) {
        int n = 0;
        for (int u = s - PF; u < s; ++u) n += er(u) + (u > s - PF ? nb(u + PF) : 0);
        return n;
    }
    static_assert(MF * PS == 4 && FH * FW == 32 && (C == 64 || C == 128) && (OM == 0 || OM == 2), "unsupported configuration");
    static constexpr int tap_of(int s) { return s / NCC; }         // stencil position of k-step s
    static_assert(HW % 2 == 0, "halo rows must hold an even number of pixels (bank parity of 128-byte pixels)");
};

// per-wave state carried through the k-steps of a tile
template <class G>
struct WsCtx {
    const WsArgs& a;
    char* lds;
    int lane, half, fy, fx, wave, mf, ps, frame;
    f32x16_t oreg[G::NI];         // results of the PREVIOUS tile, drained while this tile's MFMAs run (register form)
    f32x4_t ereg;                 // ... LDS form: the float4 of the next fragment piece
    unsigned xaddr;               // ... LDS form: LDS address of this lane's float4 of quad 0
    float s1[G::SPT ? 1 : 4][4], s2[G::SPT ? 1 : 4][4];   // channel sums [group][channel]: of the group being drained (SPT) or over the whole tile run
    // next tile (halo DMA)
    bool has_next;
    int nn, ny0, nx0, nslot;      // sample, halo origin (tile origin - 1), buffer
    int dhy, dhx, dsl;            // halo pixel / 16-byte slot of this lane's unit in the DMA instruction being issued
    __amdgpu_buffer_rsrc_t irsrc; // the input as a raw buffer: out-of-image units use an out-of-range offset and load zeros
    // previous tile (epilogue)
    bool has_prev;
    int ptile, py0, pgx;          // tile index inside the frame, first row of this wave's fragments, this lane's column
    unsigned pbase;               // BYTE offset of (sample, row py0 + fy, column pgx, channel mf*32 + 4*half) in the output
    __amdgpu_buffer_rsrc_t orsrc, srsrc;   // output / statistics as raw buffers: out-of-range offsets drop a store without a branch
    float slope;                  // activation as max(x, slope * x): 1 none, 0 ReLU, 0.01 LeakyReLU
    __device__ WsCtx(const WsArgs& a_) : a(a_) {}
};

// ---- halo DMA instruction IT of this wave: unit u = (IT*4 + wave)*64 + lane -> halo pixel p = u / CU, LDS slot u % CU.
// Branch-free (a branch would end the scheduling region of the MFMAs it rides behind): buffer_load ... lds with an
// out-of-range offset for the units outside the image (they load zeros).  From one instruction to the next the pixel moves
// on by 256 / CU, the slot stays: (hy, hx) are carried instead of divided out again.
template <class G, int IT>
__device__ __forceinline__ void ws_dma_piece(WsCtx<G>& c) {
    constexpr int STEP = 256 / G::CU;
    static_assert(256 % G::CU == 0 && STEP <= G::HW, "one row wrap per DMA instruction at most");
    if constexpr (IT == 0) {
        int ln = c.lane;
        asm volatile("" : "+v"(ln));      // opaque: no hoisting of this arithmetic out of the tile loop (registers)
        const int u = c.wave * 64 + ln, p = u / G::CU;
        c.dsl = u % G::CU;
        c.dhy = p / G::HW;
        c.dhx = p - c.dhy * G::HW;
    } else {
        c.dhx += STEP;
        const bool wrap = c.dhx >= G::HW;
        c.dhx = wrap ? c.dhx - G::HW : c.dhx;
        c.dhy += wrap ? 1 : 0;
    }
    const int j = IT * 4 + c.wave;
    const int f = G::CU >= 16 ? (c.dhx & 15) : ((c.dhx >> 1) & 7);
    const int y = c.ny0 + c.dhy, x = c.nx0 + c.dhx;
    bool ok = (unsigned)y < (unsigned)c.a.H && (unsigned)x < (unsigned)c.a.W;
    if constexpr (((IT * 4 + 3) * 64 + 63) / G::CU >= G::HPIX) ok = ok && c.dhy < G::HH;     // padding units of the last rounds
    const unsigned off = (unsigned)((((c.nn * c.a.H + y) * c.a.W + x) * G::C + ((c.dsl ^ f) << 3)) * 2);
    // (the last round has 4 * DMA_IT - NDMA instructions too many: they land in a dump area)
    char* dst = (IT * 4 + 3 < G::NDMA || j < G::NDMA) ? c.lds + c.nslot * G::SLOTB + j * 1024 : c.lds + 2 * G::SLOTB;
#if defined(__HIP_DEVICE_COMPILE__)        // (the host pass of hipcc has no such builtin)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.irsrc, (ws_lptr_t)dst, 16, (int)(ok ? off : 0xffffffffu), 0, 0, 0);
#else
    (void)dst; (void)ok; (void)off;
#endif
}
template <class G, int IT, int N>
__device__ __forceinline__ void ws_dma_pieces(WsCtx<G>& c) {
    if constexpr (N > 0 && IT < G::DMA_IT) {
        ws_dma_piece<G, IT>(c);
        ws_dma_pieces<G, IT + 1, N - 1>(c);
    }
}

// (mov_dpp: no `old` operand -- lanes a row mask disables hold garbage afterwards, here rows 0 and 2 of the last level, unused;
// with an `old` of 0 the compiler kept v_mov 0 + v_mov_dpp + v_add instead of one v_add_f32_dpp)
#define WS_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xF, true))

// ---- epilogue piece Q of the previous tile.  Q = g * (NI + 1) + j: j < NI: activation, store and channel sums of the 4
// channels mf*32 + 8g + 4*half ..+3 of fragment j; j == NI: the BatchNorm partial statistics of group g.  VALU / VMEM only (no
// LDS operation: the k-step pipeline counts lgkmcnt by hand).
// 8 sums over the 32 pixel lanes of each half wave with DPP adds (VALU rate, no LDS): within quads, half rows, rows, then
// row 0 -> row 1 / row 2 -> row 3 (row_bcast15): lanes 16..31 and 48..63 hold the totals
__device__ __forceinline__ void ws_reduce8(float (&t)[8]) {
    // level by level: 8 independent adds per level (a dependent DPP chain pays 2 wait states per link)
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += WS_DPP(t[r], 0xB1, 0xF);       // quad_perm [1,0,3,2]
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += WS_DPP(t[r], 0x4E, 0xF);       // quad_perm [2,3,0,1]
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += WS_DPP(t[r], 0x141, 0xF);      // row_half_mirror
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += WS_DPP(t[r], 0x140, 0xF);      // row_mirror
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] += WS_DPP(t[r], 0x142, 0xA);      // row_bcast15 into rows 1 and 3
}
// end of a tile's k-steps (the only exposed VALU work per tile): activation, and zeros for the pixels of a border tile that
// lie outside the image (their sums must not count); both only where needed (wave-uniform branches)
template <class G>
__device__ __forceinline__ void ws_finish_tile(WsCtx<G>& c, f32x16_t (&acc)[G::NI], int y0, int x0) {
    const WsArgs& a = c.a;
    if (a.act != 0) {
#pragma unroll
        for (int i = 0; i < G::NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(acc[i][r], acc[i][r] * c.slope);
    }
    if (y0 + G::TH > a.H || x0 + G::TW > a.W) {
        const bool xin = x0 + c.fx < a.W;
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            const bool in = xin && y0 + (c.ps * G::NI + i) * G::FH + c.fy < a.H;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = in ? acc[i][r] : 0.f;
        }
    }
}
template <class G, int Q>
__device__ __forceinline__ void ws_epi_piece(WsCtx<G>& c, const f32x4_t vals) {
    // branch-free (predicated by out-of-range buffer offsets): a branch would end the scheduling region the MFMAs sit in
    constexpr int g = Q / G::NQG, j = Q % G::NQG, sg = G::SPT ? 0 : g;
    const WsArgs& a = c.a;
    if constexpr (j < G::NI) {
        // (activation and the zeroing of out-of-image pixels happened at the hand-off: ws_finish_tile)
        const int gy = c.py0 + j * G::FH + c.fy;
        const bool pv = gy < a.H && c.pgx < a.W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c.s1[sg][r] += vals[r];
            c.s2[sg][r] = fmaf(vals[r], vals[r], c.s2[sg][r]);
        }
        const unsigned o = pv ? c.pbase + (unsigned)(j * G::FH) * (unsigned)(a.W * a.ldo * G::OB) + (unsigned)(8 * G::OB) * g : 0xffffffffu;
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
        if constexpr (G::OM == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pack2_ieee(vals[0], vals[1]), pack2_ieee(vals[2], vals[3])}, c.orsrc, (int)o, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pack2h(vals[0], vals[1]), pack2h(vals[2], vals[3])}, c.orsrc, (int)o, 0, 0);
    } else {
        float t[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = c.s1[0][r]; t[4 + r] = c.s2[0][r]; }
        ws_reduce8(t);
        const int mrow = c.mf * 32 + 8 * g + 4 * c.half;
        const unsigned grp = (unsigned)(a.stats_group_offset + c.frame * (int)a.stats_bstride + c.ptile * G::PS + c.ps);
        const unsigned so = ((c.lane & 31) == 16 && c.has_prev) ? (grp * 2u * a.K + mrow) * 4u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{t[0], t[1], t[2], t[3]}), c.srsrc, (int)so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{t[4], t[5], t[6], t[7]}), c.srsrc, (int)so, a.K * 4, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) c.s1[0][r] = c.s2[0][r] = 0.f;
    }
}
// the BatchNorm partial statistics of this wave's tile run: one group per (workgroup, ps), written once at the end
template <class G>
__device__ __forceinline__ void ws_stats_store(WsCtx<G>& c, int group) {
    const WsArgs& a = c.a;
    if constexpr (!G::SPT) {
        if (!a.stats) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { t[r] = c.s1[g][r]; t[4 + r] = c.s2[g][r]; }
            ws_reduce8(t);
            if ((c.lane & 31) == 16) {
                const int mrow = c.mf * 32 + 8 * g + 4 * c.half;
                float* sp = a.stats + ((int64_t)a.stats_group_offset + (int64_t)c.frame * a.stats_bstride + group) * 2 * a.K + mrow;
                *reinterpret_cast<float4*>(sp) = make_float4(t[0], t[1], t[2], t[3]);
                *reinterpret_cast<float4*>(sp + a.K) = make_float4(t[4], t[5], t[6], t[7]);
            }
        }
    }
}
// the pieces scheduled for k-step S (interleaved form)
template <class G, int S, int Q>
__device__ __forceinline__ void ws_epi_pieces(WsCtx<G>& c) {
    if constexpr (Q < G::NQ) {
        if constexpr (G::step_of(Q) == S) {
            constexpr int g = Q / G::NQG, j = Q % G::NQG;
            if constexpr (j >= G::NI) ws_epi_piece<G, Q>(c, f32x4_t{0.f, 0.f, 0.f, 0.f});
            else if constexpr (G::OLDS) ws_epi_piece<G, Q>(c, c.ereg);
            else ws_epi_piece<G, Q>(c, f32x4_t{c.oreg[j][g * 4], c.oreg[j][g * 4 + 1], c.oreg[j][g * 4 + 2], c.oreg[j][g * 4 + 3]});
        }
        ws_epi_pieces<G, S, Q + 1>(c);
    }
}
// all pieces straight from the accumulators (the last tile of a workgroup: no MFMAs left to hide behind)
template <class G, int Q>
__device__ __forceinline__ void ws_epi_all(WsCtx<G>& c, const f32x16_t (&acc)[G::NI]) {
    if constexpr (Q < G::NQ) {
        constexpr int g = Q / G::NQG, j = Q % G::NQG;
        if constexpr (j >= G::NI) ws_epi_piece<G, Q>(c, f32x4_t{0.f, 0.f, 0.f, 0.f});
        else ws_epi_piece<G, Q>(c, f32x4_t{acc[j][g * 4], acc[j][g * 4 + 1], acc[j][g * 4 + 2], acc[j][g * 4 + 3]});
        ws_epi_all<G, Q + 1>(c, acc);
    }
}
// LDS form: request the float4 of the fragment piece scheduled at k-step S (quad j * 4 + g)
template <class G, int S, int Q>
__device__ __forceinline__ void ws_epi_fetch(WsCtx<G>& c) {
    if constexpr (Q < G::NQ) {
        constexpr int g = Q / G::NQG, j = Q % G::NQG;
        if constexpr (G::step_of(Q) == S && j < G::NI)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.ereg) : "v"(c.xaddr), "n"((j * 4 + g) * 1024));
        ws_epi_fetch<G, S, Q + 1>(c);
    }
}

// B fragment reads of k-step S (tap t = S / NCC, 16-channel chunk cc = S % NCC): NI x ds_read_b128 from
// (bbase[t % 3] ^ (cc << 5)) + buffer base, the row / column displacement of (fragment i, tap) as the immediate offset
template <class G, int S, int I>
__device__ __forceinline__ void ws_read(u32x4_t (&bq)[G::PF + 1][G::NI], unsigned ad) {
    constexpr int t = G::tap_of(S);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[S % (G::PF + 1)][I]) : "v"(ad), "n"(((I * G::FH + t / 3) * G::HW + t % 3) * G::PIXB));
    if constexpr (I + 1 < G::NI) ws_read<G, S, I + 1>(bq, ad);
}
template <class G, int S>
__device__ __forceinline__ void ws_read_step(u32x4_t (&bq)[G::PF + 1][G::NI], const unsigned (&bbase)[3], unsigned lb) {
    constexpr int t = G::tap_of(S), cc = S % G::NCC;
    unsigned b = bbase[t % 3];
    asm volatile("" : "+v"(b));           // opaque: the 3 * NCC address variants are recomputed (2 VALU) instead of held in registers
    ws_read<G, S, 0>(bq, (b ^ (unsigned)(cc << 5)) + lb);
}
template <class G, int S>
__device__ __forceinline__ void ws_prefetch(u32x4_t (&bq)[G::PF + 1][G::NI], const unsigned (&bbase)[3], unsigned lb) {
    if constexpr (S < G::PF && S < G::NS) {
        ws_read_step<G, S>(bq, bbase, lb);
        ws_prefetch<G, S + 1>(bq, bbase, lb);
    }
}
// k-steps S .. NS-1: reads of step S + PF, this step's share of the next halo's DMA and of the previous tile's epilogue,
// counted wait for the reads of step S, NI MFMAs; the order is pinned
template <class G, int S, int I0, int I1>
__device__ __forceinline__ void ws_read_some(u32x4_t (&bq)[G::PF + 1][G::NI], unsigned ad) {
    if constexpr (I0 < I1 && I0 < G::NI) {
        constexpr int t = G::tap_of(S);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[S % (G::PF + 1)][I0]) : "v"(ad), "n"(((I0 * G::FH + t / 3) * G::HW + t % 3) * G::PIXB));
        ws_read_some<G, S, I0 + 1, I1>(bq, ad);
    }
}
// One k-step.  An in-order wave stalls in front of every MFMA but the first of a back-to-back group until the matrix pipe is
// free, so everything else has to sit BETWEEN the MFMAs: the B reads of k-step S + PF go behind MFMAs 0 and 1, this step's
// DMA instruction or epilogue piece shares a scheduling region with the remaining MFMAs (group barriers ask for an MFMA /
// VALU alternation).  lgkmcnt is counted by hand (the reads are inline asm).
template <class G, int S>
__device__ __forceinline__ void ws_step(WsCtx<G>& c, const h16x8_t (&wr)[G::NS], f32x16_t (&acc)[G::NI], u32x4_t (&bq)[G::PF + 1][G::NI],
                                        const unsigned (&bbase)[3], unsigned lb) {
    constexpr int set = S % (G::PF + 1), NI = G::NI, SN = S + G::PF;
    constexpr bool rd = SN < G::NS;
    static_assert(NI == 4, "issue pattern written for 4 fragments per wave");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(G::later_than_b(S)) : "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(bq[set][i]));
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = mfma16x<G::XF>(wr[S], __builtin_bit_cast(h16x8_t, bq[set][0]), acc[0]);
    __builtin_amdgcn_sched_barrier(0);
    unsigned ad = 0;
    if constexpr (rd) {
        constexpr int t = G::tap_of(SN), cc = SN % G::NCC;
        unsigned b = bbase[t % 3];
        asm volatile("" : "+v"(b));       // opaque: the 3 * NCC address variants are recomputed (2 VALU) instead of held in registers
        ad = (b ^ (unsigned)(cc << 5)) + lb;
        ws_read_some<G, SN, 0, 2>(bq, ad);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[1] = mfma16x<G::XF>(wr[S], __builtin_bit_cast(h16x8_t, bq[set][1]), acc[1]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (rd) ws_read_some<G, SN, 2, 4>(bq, ad);
    if constexpr (G::er(S) > 0) ws_epi_fetch<G, S + 1, 0>(c);   // unconditional: the counts rely on it
    __builtin_amdgcn_sched_barrier(0);
#ifndef WS_ABL
#define WS_ABL 0            // kernel study builds only: 1 = no DMA pieces, 2 = no epilogue pieces, 3 = neither (wrong results)
#endif
    constexpr bool has_dma = S * G::DPS < G::DMA_IT && !(WS_ABL & 1);
    constexpr bool has_epi = S >= G::E0 && S < G::E1 && !(WS_ABL & 2);
    acc[2] = mfma16x<G::XF>(wr[S], __builtin_bit_cast(h16x8_t, bq[set][2]), acc[2]);
    acc[3] = mfma16x<G::XF>(wr[S], __builtin_bit_cast(h16x8_t, bq[set][3]), acc[3]);
    if constexpr (has_dma) ws_dma_pieces<G, S * G::DPS, G::DPS>(c);      // without a next tile: zeros into the idle buffer
    if constexpr (has_epi) {
        if constexpr (G::er(S - 1) > 0) {                       // the float4 requested in step S - 1
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(G::nb(SN) + G::er(S)) : "memory");
            asm volatile("" : "+v"(c.ereg));
        }
        ws_epi_pieces<G, S, 0>(c);                              // without a previous tile: every store is out of range
    }
    if constexpr (has_dma || has_epi) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        // MFMA 2
        __builtin_amdgcn_sched_group_barrier(0x6, 8, 0);        // up to 8 VALU / SALU
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        // MFMA 3
        __builtin_amdgcn_sched_group_barrier(0x6, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < G::NS) ws_step<G, S + 1>(c, wr, acc, bq, bbase, lb);
}

template <int C, int MF, int PS, int NI, int FH, int FW, int OM = 0, int XF = 0>
__global__ __launch_bounds__(256) void wsconv_kernel(const WsArgs a) {
    typedef WsCfg<C, MF, PS, NI, FH, FW, OM, XF> G;
    constexpr int NT = G::NT;
    constexpr int TH = G::TH, TW = G::TW, HW = G::HW, PIXB = G::PIXB, CU = G::CU, NCC = G::NCC, NS = G::NS, SLOTB = G::SLOTB;
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [2][SLOTB] halo, DUMPB, (+ XCHB bytes of results and the bias in the LDS form)

    const int tid = threadIdx.x;
    WsCtx<G> c(a);
    c.lds = lds;
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.mf = c.wave / PS;
    c.ps = c.wave % PS;
    c.half = c.lane >> 5;
    const int col = c.lane & 31;
    c.fy = FW == 32 ? 0 : (col >> 4);
    c.fx = FW == 32 ? col : (col & 15);
    const int K = a.K;
    const bool tracing = a.trace != nullptr && blockIdx.x == 8 && tid == 0;
    WS_STAMP(0);

    c.frame = blockIdx.x / a.wgs_per_frame;
    const int jw = blockIdx.x - c.frame * a.wgs_per_frame;
    const int t_begin = jw * a.tiles_per_wg;
    const int t_end = min(a.tiles_per_frame, t_begin + a.tiles_per_wg);
    if (t_begin >= t_end) return;
    const int txy = a.tiles_x * a.tiles_y;
#pragma unroll
    for (int g = 0; g < (G::SPT ? 1 : 4); ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) c.s1[g][r] = c.s2[g][r] = 0.f;
    // a conv bias is the initial value of the accumulators: row (r & 3) + 8 (r >> 2) + 4 half of this wave's 32 channels.
    // Register form: 16 registers; LDS form (no registers to spare): MF*32 floats behind the result area, read at tile start.
    f32x16_t binit;
    float* lbias = reinterpret_cast<float*>(lds + 2 * SLOTB + G::DUMPB + G::XCHB);
    if constexpr (G::OLDS) {
        if (tid < MF * 32) lbias[tid] = (a.bias && tid < K) ? a.bias[tid] : 0.f;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = c.mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * c.half;
            binit[r] = (a.bias && m < K) ? a.bias[m] : 0.f;
        }
    }
    c.slope = a.act == 1 ? 0.f : (a.act == 3 ? 0.01f : 1.f);
    c.orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    c.irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16raw*>(a.in), 0, a.in_bytes, 0x00020000);
    c.dhy = c.dhx = c.dsl = 0;
    c.srsrc = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, a.stats ? a.stats_bytes : 0, 0x00020000);
    c.pgx = 0x7fffffff;
    c.py0 = 0; c.ptile = 0; c.pbase = 0;
    c.xaddr = (unsigned)(uintptr_t)(ws_lptr_t)lds + 2 * SLOTB + G::DUMPB + c.wave * 16 * 1024 + c.lane * 16;
    c.ereg = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (!G::OLDS) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) c.oreg[i][r] = 0.f;
    }

#define WS_SET_NEXT(tile, slot_)                                                                              \
    {                                                                                                         \
        const int tl_ = (tile), r_ = tl_ % txy;                                                               \
        c.nn = c.frame * a.spf + tl_ / txy;                                                                   \
        c.ny0 = (r_ / a.tiles_x) * TH - 1;                                                                    \
        c.nx0 = (r_ % a.tiles_x) * TW - 1;                                                                    \
        c.nslot = (slot_);                                                                                    \
    }
#define WS_SET_PREV(tile)                                                                                     \
    {                                                                                                         \
        const int rr_ = (tile) % txy;                                                                         \
        c.ptile = (tile);                                                                                     \
        c.py0 = (rr_ / a.tiles_x) * TH + c.ps * NI * FH;                                                      \
        c.pgx = (rr_ % a.tiles_x) * TW + c.fx;                                                                \
        c.pbase = (unsigned)((((c.frame * a.spf + (tile) / txy) * a.H + c.py0 + c.fy) * a.W + c.pgx) * a.ldo + c.mf * 32 + 4 * c.half) * (unsigned)G::OB; \
        c.has_prev = true;                                                                                    \
    }
    WS_SET_NEXT(t_begin, 0)
    ws_dma_pieces<G, 0, G::DMA_IT>(c);
    WS_STAMP(1);

    // ---- weights -> registers: A fragment of k-step s = t*NCC + cc: rows mf*32 + col, channels cc*16 + half*8 ..+7 of tap t
    h16x8_t wr[NS];
    {
        const h16raw* wsrc = a.wgt + (int64_t)c.frame * a.w_bstride;
        const int m = c.mf * 32 + col;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ws = a.wslot[t];
#pragma unroll
            for (int cc = 0; cc < NCC; ++cc) {
                h16x8_t v = __builtin_bit_cast(h16x8_t, u32x4_t{0u, 0u, 0u, 0u});
                if (a.w_frag) {               // one contiguous 1 KiB block per fragment
                    if (ws >= 0) v = *reinterpret_cast<const h16x8_t*>(wsrc + ((((int64_t)c.mf * a.wt + ws) * NCC + cc) * 64 + c.lane) * 8);
                } else if (m < K && ws >= 0)
                    v = *reinterpret_cast<const h16x8_t*>(wsrc + ((int64_t)m * a.wt + ws) * C + cc * 16 + c.half * 8);
                wr[t * NCC + cc] = v;
            }
        }
    }

    // ---- B fragment addressing (bytes inside a halo buffer): pixel (row (ps*NI + i)*FH + fy + dhi, column fx + dwi) of the halo,
    // dhi, dwi = 0..2; chunk c8 = cc*2 + half stored at slot c8 ^ f(fx + dwi).  Everything but the lane part is an
    // immediate: addr = (bbase[dwi] ^ (cc << 5)) + ((i*FH + dhi)*HW + dwi)*PIXB
    unsigned bbase[3];
#pragma unroll
    for (int dwi = 0; dwi < 3; ++dwi) {
        const int hx = c.fx + dwi;
        const int f = CU >= 16 ? (hx & 15) : ((hx >> 1) & 7);
        bbase[dwi] = (unsigned)(((c.ps * NI * FH + c.fy) * HW + c.fx) * PIXB + ((c.half ^ f) << 4));
    }

    // the BUILTIN wait (vmcnt(0), other counters untouched) is modelled by the compiler's waitcnt pass: without it the first
    // use of a weight register inside the tile loop gets a vmcnt(0) that also drains the NEXT tile's halo DMA, every tile
    __builtin_amdgcn_s_waitcnt(0x0F70);
    WS_STAMP(2);
    int slot = 0;
    c.has_prev = false;
    for (int tile = t_begin; tile < t_end; ++tile) {
        __builtin_amdgcn_s_barrier();               // halo of `tile` complete; every wave is done with the other buffer
        WS_STAMP(4 + (tile - t_begin) * 4);
        c.has_next = tile + 1 < t_end;
        if (c.has_next) WS_SET_NEXT(tile + 1, slot ^ 1)
        else { c.ny0 = -0x100000; c.nslot = slot ^ 1; }         // every halo pixel out of the image: the zero page

        f32x16_t acc[NI];
        if constexpr (G::OLDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t bq4 = *reinterpret_cast<const f32x4_t*>(lbias + c.mf * 32 + 8 * q + 4 * c.half);
#pragma unroll
                for (int r = 0; r < 4; ++r) binit[q * 4 + r] = bq4[r];
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = binit;
        // software pipeline: the NI fragment reads of k-step s + PF are issued in front of the NI MFMAs of k-step s (one wave per
        // SIMD: the latency of a read is covered by this wave's own MFMAs).  The reads are inline asm with hand-counted
        // lgkmcnt: the compiler's own waits came out as lgkmcnt(0) every second k-step, i.e. it waited for reads issued one
        // MFMA earlier.  With one wave per SIMD nothing else hides non-MFMA work either, so the halo DMA of the next tile and the
        // epilogue of the previous one are cut into pieces that ride behind the MFMAs of this tile's k-steps (measured before:
        // MFMA loop 9.7 k cycles, DMA issue 3.6 k, epilogue 9.5 k per tile).
        u32x4_t bq[G::PF + 1][NI];
        const unsigned lb = (unsigned)(uintptr_t)(ws_lptr_t)lds + slot * SLOTB;
        ws_prefetch<G, 0>(bq, bbase, lb);
        ws_step<G, 0>(c, wr, acc, bq, bbase, lb);
        WS_STAMP(5 + (tile - t_begin) * 4);
        ws_finish_tile<G>(c, acc, ((tile % txy) / a.tiles_x) * TH, ((tile % txy) % a.tiles_x) * TW);
        if (tile + 1 == t_end) {                    // the last tile's epilogue has no MFMAs to hide behind
            WS_SET_PREV(tile)
            ws_epi_all<G, 0>(c, acc);
            break;
        }
        // hand the results to the next tile's k-steps
        if constexpr (G::OLDS) {
            f32x4_t* x4 = reinterpret_cast<f32x4_t*>(lds + 2 * SLOTB + G::DUMPB + c.wave * 16 * 1024) + c.lane;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) x4[(i * 4 + g) * 64] = f32x4_t{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) c.oreg[i] = acc[i];
        }
        WS_SET_PREV(tile)
        WS_STAMP(6 + (tile - t_begin) * 4);
        // the next tile's halo (and this tile's share of output stores) must have landed before the barrier at the top
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_STAMP(7 + (tile - t_begin) * 4);
        slot ^= 1;
    }
    ws_stats_store<G>(c, jw * PS + c.ps);
    if (tracing) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.trace[3] = __builtin_readcyclecounter(); }
#undef WS_SET_NEXT
#undef WS_SET_PREV
}

// cycle stamps of one workgroup (profiling aid, tools/ws_trace.py): enabled by the environment variable TCVOM_CONV_TRACE
static unsigned long long* ws_trace_buffer() {
    static unsigned long long* buf = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        if (getenv("TCVOM_CONV_TRACE") != nullptr && hipMalloc((void**)&buf, 64 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
        if (buf) (void)hipMemset(buf, 0, 64 * sizeof(unsigned long long));
    }
    return buf;
}
extern "C" int tcvom_conv_trace_read(uint64_t* host, int32_t n) {
    unsigned long long* buf = ws_trace_buffer();
    if (!buf || n < 1 || n > 64) return tcvom_fail(TCVOM_ERR_ARG, "conv_trace_read: tracing is off (set TCVOM_CONV_TRACE) or bad n");
    return hipMemcpy(host, buf, n * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess ? TCVOM_OK : TCVOM_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------- host side
struct WsPlan { bool ok, xf; int C, th, tw; int wslot[9]; };

static WsPlan ws_plan(const tcvom_conv_desc* d, int nphase) {
    WsPlan p;
    p.ok = false;
    p.xf = d->in_f16 != 0 && !TCVOM_BUILD_F16;
    static const bool disabled = getenv("TCVOM_NO_WSCONV") != nullptr;      // A/B switch (tools/igemm_bench.py, tests)
    if (disabled || nphase != 1) return p;
    if (d->C != d->K || (d->C != 64 && d->C != 128) || d->ldo % 4 != 0) return p;
    if (d->in_step != 1 || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return p;
    if (d->PH != d->H || d->PW != d->W || d->OH != d->H || d->OW != d->W) return p;
    const int nb = d->batch > 1 ? d->batch : 1;
    if (d->batch > 1) {
        if (d->in_bstride != (long long)d->N * d->H * d->W * d->C || d->out_bstride != (long long)d->N * d->H * d->W * d->ldo) return p;
        if (d->vec_bstride != 0) return p;
    }
    // (one FRAME must stay below 2^31 elements -- the conv engine's own limit, igemm.hip; a frame-batched launch whose frames
    //  together exceed it is split into runs of frames by wsconv_try_launch: fragment-major weights have no other kernel)
    if ((long long)d->N * d->H * d->W * d->C >= (1ll << 31) || (long long)d->N * d->H * d->W * d->ldo >= (1ll << 31)) return p;
    for (int t = 0; t < 9; ++t) p.wslot[t] = -1;
    int n = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        if (d->tap_w[t] < 0) continue;
        const int dh = d->tap_dh[t], dw = d->tap_dw[t];
        if (dh < -1 || dh > 1 || dw < -1 || dw > 1) return p;
        const int c = (dh + 1) * 3 + (dw + 1);
        if (p.wslot[c] >= 0) return p;                  // (a tap twice: not a plain 3x3 stencil)
        p.wslot[c] = d->tap_w[t];
        ++n;
    }
    if (n != 9) return p;
    if (d->out_fp32 == 1) return p;                     // (fp32 results: the implicit GEMM)
    if (d->out_fp32 < 0 || d->out_fp32 > 2) return p;
    if (p.xf != (d->out_fp32 == 2)) return p;           // IEEE fp16 results come with IEEE fp16 operands (the fp16 island), and only with them
    p.C = d->C;
    p.th = 8;
    p.tw = d->C == 64 ? 32 : 16;
    // tiny images (the 64 x 64 golden windows) would leave most of a tile masked: the implicit GEMM serves them -- unless
    // the weights are packed for this kernel
    if (d->w_layout == 0 && (d->H < p.th / 2 || d->W < p.tw / 2)) return p;
    if (d->w_layout != 0 && (d->w_layout != 1 || d->wt != 9)) return p;
    p.ok = true;
    return p;
}

static int ws_ps(const WsPlan& p) { return p.C == 64 ? 2 : 1; }     // pixel groups (waves with their own fragments) per workgroup

// persistent workgroups: one per CU over all frames of the launch, every frame the same number
static void ws_grid(const tcvom_conv_desc* d, const WsPlan& p, int* tiles_per_frame, int* tiles_per_wg, int* wgs_per_frame) {
    const int nb = d->batch > 1 ? d->batch : 1;
    *tiles_per_frame = d->N * cdiv(d->H, p.th) * cdiv(d->W, p.tw);
    int wpf = 256 / nb;
    if (wpf < 1) wpf = 1;
    if (wpf > *tiles_per_frame) wpf = *tiles_per_frame;
    *tiles_per_wg = cdiv(*tiles_per_frame, wpf);
    *wgs_per_frame = cdiv(*tiles_per_frame, *tiles_per_wg);
}

// number of statistics groups ONE frame of the launch writes (one per workgroup and pixel group), or 0 when the shape is
// not handled here
int wsconv_stats_groups(const tcvom_conv_desc* d, int nphase) {
    const WsPlan p = ws_plan(d, nphase);
    if (!p.ok) return 0;
    int tpf, tpw, wpf;
    ws_grid(d, p, &tpf, &tpw, &wpf);
    // C = 128 / doubled taps: one group per (tile, pixel group); C = 64: one per (workgroup, pixel group)
    return p.C == 128 ? tpf * ws_ps(p) : wpf * ws_ps(p);
}

// returns 1 when the conv was launched here, 0 when the caller should use another kernel, < 0 on error
int wsconv_try_launch(const void* in, const void* w, void* out, const float* bias, const float* mscale, const float* mdiag,
                      float* stats, const tcvom_conv_desc* d, int nphase, const h16raw* zero_page, void* stream) {
    if (mscale || mdiag) return 0;
    const WsPlan p = ws_plan(d, nphase);
    if (!p.ok) return 0;
    WsArgs a;
    a.in = (const h16raw*)in;
    a.wgt = (const h16raw*)w;
    a.out = out;
    a.bias = bias;
    a.stats = stats;
    a.zero_page = zero_page;
    a.H = d->H; a.W = d->W; a.K = d->K; a.ldo = d->ldo; a.wt = d->wt; a.act = d->act;
    a.tiles_x = cdiv(d->W, p.tw);
    a.tiles_y = cdiv(d->H, p.th);
    const int nb = d->batch > 1 ? d->batch : 1;
    a.spf = d->N;
    a.w_bstride = nb > 1 ? d->w_bstride : 0;
    a.stats_group_offset = d->stats_group_offset;
    a.stats_bstride = nb > 1 ? d->stats_bstride : 0;
    for (int t = 0; t < 9; ++t) a.wslot[t] = p.wslot[t];
    a.w_frag = d->w_layout == 1;
    a.trace = ws_trace_buffer();
    ws_grid(d, p, &a.tiles_per_frame, &a.tiles_per_wg, &a.wgs_per_frame);      // (from the WHOLE batch: the statistics layout the caller sized)
    const long long gpf = p.C == 128 ? (long long)a.tiles_per_frame * ws_ps(p) : (long long)a.wgs_per_frame * ws_ps(p);
    const int ob = 2;
    if (stats && nb > 1 && d->stats_bstride < gpf)
        return tcvom_fail(TCVOM_ERR_ARG, "wsconv: stats_bstride %lld < groups per frame", (long long)d->stats_bstride);
    {
        const long long sb = ((long long)a.stats_group_offset + (nb > 1 ? (long long)(nb - 1) * a.stats_bstride : 0) + gpf) * 2 * d->K * 4;
        if (sb >= (1ll << 32)) return d->w_layout == 1 ? tcvom_fail(TCVOM_ERR_ARG, "wsconv: statistics buffer beyond 4 GiB") : 0;
        a.stats_bytes = (unsigned)sb;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    // frames per launch: the buffer descriptors of the kernel address 32-bit byte ranges -- a batch whose frames together reach
    // 2^31 elements goes out as runs of frames (same tiles, same statistics groups per frame)
    const long long frame_elems = (long long)d->N * d->H * d->W * (d->C > d->ldo ? d->C : d->ldo);
    long long fmax = ((1ll << 31) - 1) / (frame_elems > 0 ? frame_elems : 1);
    static const int test_fmax = getenv("TCVOM_WS_MAX_FRAMES") ? atoi(getenv("TCVOM_WS_MAX_FRAMES")) : 0;      // (tests: force the split)
    if (test_fmax > 0 && test_fmax < fmax) fmax = test_fmax;
    if (fmax < 1) fmax = 1;
    const WsArgs base = a;
    for (int f0 = 0; f0 < nb; f0 += (int)fmax) {
        const int nf = nb - f0 < fmax ? nb - f0 : (int)fmax;
        a = base;
        a.in = base.in + (long long)f0 * d->N * d->H * d->W * d->C;
        a.out = (char*)base.out + (long long)f0 * d->N * d->H * d->W * d->ldo * ob;
        a.wgt = base.wgt + (long long)f0 * base.w_bstride;
        a.stats_group_offset = base.stats_group_offset + (int)((long long)f0 * base.stats_bstride);
        a.out_bytes = (unsigned)((long long)d->N * nf * d->H * d->W * d->ldo * ob);
        a.in_bytes = (unsigned)((long long)d->N * nf * d->H * d->W * d->C * 2);
        const dim3 grid(a.wgs_per_frame * nf);
        if (p.xf && p.C == 64) {                 // fp16 island of the bf16 build: IEEE fp16 operands and results
            auto kern = wsconv_kernel<64, 2, 2, 4, 1, 32, 2, 1>;
            constexpr size_t lds_bytes = 2 * WsCfg<64, 2, 2, 4, 1, 32>::SLOTB + 1024;
            static bool attr = false;
            if (!attr) { e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
        } else if (p.xf) {
            auto kern = wsconv_kernel<128, 4, 1, 4, 2, 16, 2, 1>;
            constexpr size_t lds_bytes = 2 * WsCfg<128, 4, 1, 4, 2, 16>::SLOTB + 1024 + WsCfg<128, 4, 1, 4, 2, 16>::XCHB + 128 * 4;
            static bool attr = false;
            if (!attr) { e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
        } else if (p.C == 64) {
            auto kern = wsconv_kernel<64, 2, 2, 4, 1, 32>;
            constexpr size_t lds_bytes = 2 * WsCfg<64, 2, 2, 4, 1, 32>::SLOTB + 1024;
            static bool attr = false;
            if (!attr) { e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
        } else {
            auto kern = wsconv_kernel<128, 4, 1, 4, 2, 16>;
            constexpr size_t lds_bytes = 2 * WsCfg<128, 4, 1, 4, 2, 16>::SLOTB + 1024 + WsCfg<128, 4, 1, 4, 2, 16>::XCHB + 128 * 4;
            static_assert(lds_bytes <= 160 * 1024, "LDS budget");
            static bool attr = false;
            if (!attr) { e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
        }
    }
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "wsconv: %s", hipGetErrorString(e));
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "wsconv: %s", hipGetErrorString(e2));
    return 1;
}
