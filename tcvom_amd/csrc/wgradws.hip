// Accumulator-stationary weight gradient of the stride-1 3x3 convolutions (the BasicBlock / shortcut / TAM layers of
// resnet_enc.py:33-49, resnet_dec.py:43-59, res_gca_enc.py:47-55 at os4 .. os32):
//     dw[k][slot(t)][c] += sum_p dy[p][k] * x[p + t][c],      t = the 9 taps, C and K multiples of 64.
//
// Why: the implicit-GEMM TT kernel (igemm.hip) gives every workgroup ONE 128 x 128 block of dw, i.e. one tap: the dy tile is
// DMA'd once per tap and x once per output-channel tile (measured 3.3x the algorithmic bytes through L2 -> LDS, 71 us for the
// three frames of an os8 layer against ~13 us of MFMA time).  Here a persistent workgroup owns a [64 k][9 taps][CWIN c] block of
// dw in REGISTERS (CWIN/16 waves x 9 accumulator tiles of 32 x 32 = 144 AGPRs each: wave = (k fragment, c fragment), one tile per
// tap; 8 waves = 2 per SIMD for CWIN = 128), walks over pixel tiles and
// DMAs, per tile, the x halo of its CWIN input channels and the dy tile of its 64 output channels ONCE; all 9 taps read their
// operands from that halo.  dw is added to global memory once per workgroup, at the end (fp32 atomics: the workgroups of one
// [k][c] block split the pixels).
//
//   MFMA 32x32x16 bf16: A = dy^T [32 k][16 pixels], B = x^T(tap) [32 c][16 pixels]; the reduction runs over pixels, both
//   operands are pixel-major in LDS and are read with ds_read_b64_tr_b16 (layout as in halo.hip: halo_wgrad_kernel).
// LDS images: pixel-major; the 64-byte channel group g of pixel p sits at group position g ^ s(p) (applied on the DMA source
// side), s = hx & 3 for 256-byte pixels and (hx >> 1) & 1 for 128-byte pixels, so that the 4 consecutive pixels a 32-lane
// group of a transposing read touches lie in distinct banks.
#include <cstdlib>
#include <type_traits>
#include "common.h"

typedef __attribute__((address_space(3))) void* wg_lptr_t;
#ifndef WG_ABL
#define WG_ABL 0          // kernel ablations for timing (1: no atomics, 2: no DMA inside the tile loop); 0 in the product
#endif

// One launch serves problems of up to WG_MAX_GEO different geometries (same channel window CWIN): the atomic flush of the
// accumulators costs a launch ~45 us whatever its length (every workgroup adds 1 - 2 blocks of 295 KB), so the layers with a
// geometry of their own (the channel-changing convs of a trunk: 3 - 9 problems per launch, 105 - 150 us each at 180 - 570 TFLOP/s)
// ride in the launch of the big groups instead.
#define WG_MAX_PROBLEMS 96                 // 3 pointer arrays + the tables below stay inside the 4 KiB kernel-argument segment
#define WG_MAX_GEO 8
struct WgGeo {
    int N, H, W, C, K, wt;                 // N, H, W: the samples the tiles walk over -- for a dilation-d conv the N * d * d sub-grids
                                           // (pixels of one residue class mod d) of H / d x W / d pixels, each a dilation-1 problem
    int dil, Hf, Wf;                       // dilation; the full image (pixel (y, x) of sub-grid (sy, sx) is (sy + y dil, sx + x dil))
    int tiles_x, tiles_y, ntiles;          // pixel tiles of one problem (N * tiles_y * tiles_x)
    int kgroups, cgroups;                  // K / 64, C / CWIN: the (k, c) blocks of dw
    unsigned dy_bytes, in_bytes;
};
struct WgArgs {
    const h16raw* dy[WG_MAX_PROBLEMS];
    const h16raw* in[WG_MAX_PROBLEMS];
    float* dw[WG_MAX_PROBLEMS];
    int qend[WG_MAX_PROBLEMS];             // end of problem i in the flat (problem, block, tile) sequence
    unsigned char geo_of[WG_MAX_PROBLEMS]; // its geometry
    WgGeo geo[WG_MAX_GEO];
    int nprob, total, per_wg;              // length of the sequence and the run of one workgroup
    int wslot[9];                          // dw slot of the canonical tap t = (dh + 1) * 3 + (dw + 1)
};

template <int CWIN_, int TW_>
struct WgCfg {
    static constexpr int NW = CWIN_ / 16;                           // waves: (2 k fragments) x (CWIN / 32 c fragments), 9 tap tiles each
    static constexpr int CWIN = CWIN_, TW = TW_, TH = 8, HW = TW + 2, HH = TH + 2;
    static constexpr int XPB = CWIN * 2, YPB = 128;                 // bytes per pixel of the x halo / the dy tile (64 k)
    static constexpr int XB = (HH * HW * XPB + 1023) / 1024 * 1024, YB = TH * TW * YPB;   // bytes of the two images (x padded to 1 KiB)
    static constexpr int XU = XB / 16, YU = YB / 16;                // 16-byte DMA units
    static constexpr int NDMA = (XU + YU + 63) / 64;                // DMA wave-instructions per tile (x units first, then dy)
    static constexpr int DMA_IT = (NDMA + NW - 1) / NW;
    static constexpr int SLOTB = NDMA * 1024;
    static constexpr int NB = 9;                              // accumulator tiles (= MFMAs per k-step) per wave
    static constexpr int NKS = TH * TW / 16;                        // k-steps (16 pixels) per tile
    static constexpr int NG = NKS * NB;                             // MFMAs per tile and wave
    static constexpr int D = 3;                                     // B fragments are requested D MFMAs ahead
    static constexpr int AISSUE = 2;                                // the A fragment of k-step s + 1 is requested before MFMA AISSUE of step s
    static_assert(XU % 64 == 0 && YU % 64 == 0, "a DMA wave-instruction covers one image");
    static_assert(AISSUE + D < NB, "the A fragment must be older than the first B fragment of its k-step");
    // LDS operations (2 per fragment) issued after the reads of B fragment g and before MFMA g
    static constexpr bool a_at(int u) { return u >= 0 && u % NB == AISSUE && u / NB + 1 < NKS; }
    static constexpr int later_than(int g) {
        int n = 0;
        for (int u = g - D; u <= g; ++u) n += (a_at(u) ? 2 : 0) + (u > g - D && u + D < NG ? 2 : 0);
        return n;
    }
};

// B fragment g = (k-step s, tile j = cf * 9 + t): pixels (row r + dh + 1, x0 + dw + 1 + q) of the halo, channels of fragment cf
template <class G, int g>
__device__ __forceinline__ void wg_read_b(TrFrag& f, const unsigned (&xb)[3]) {
    constexpr int s = g / G::NB, t = g % G::NB;
    constexpr int r = s / (G::TW / 16), xo = (s % (G::TW / 16)) * 16;
    constexpr int off = ((r + t / 3) * G::HW + xo + t % 3) * G::XPB;
    const unsigned ad = xb[t % 3];
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(ad), "n"(off));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(ad), "n"(off + 4 * G::XPB));
}
// A fragment of k-step s: pixels (row r, x0 + q) of the dy tile
template <class G, int s>
__device__ __forceinline__ void wg_read_a(TrFrag& f, unsigned ya) {
    constexpr int r = s / (G::TW / 16), xo = (s % (G::TW / 16)) * 16;
    constexpr int off = (r * G::TW + xo) * G::YPB;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(ya), "n"(off));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(ya), "n"(off + 4 * G::YPB));
}

struct WgDma {                   // per-wave state of the tile being fetched
    int H, W;                    // image (sub-grid) size of its problem
    int ym, xm;                  // tile origin (y0, x0); the x halo starts one pixel up / left
    unsigned xbase, ybase;       // byte offset of halo pixel (0, 0) channel cbase / of tile pixel (0, 0) channel kbase
    int slot;
    __amdgpu_buffer_rsrc_t xr, yr;
};
// The (lane, instruction) -> (pixel, 16-byte chunk) map of the DMA does not depend on the tile: rel[IT] = byte offset of the
// unit relative to the image's first pixel, pk[IT] = (row << 16 | column) inside the image for the bounds test (row 0x7fff:
// padding unit, never loaded).  Unit u = (IT*NW + wave)*64 + lane; units [0, XU) are the x halo, then the dy tile.
template <class G>
__device__ __forceinline__ void wg_dma_map(const WgGeo& a, int wave, int lane, unsigned (&rel)[G::DMA_IT], unsigned (&pk)[G::DMA_IT]) {
#pragma unroll
    for (int it = 0; it < G::DMA_IT; ++it) {
        const int u = (it * G::NW + wave) * 64 + lane;
        if (u < G::XU) {
            // x halo: pixel p = u / (XPB/16), LDS position sl holds chunk c = ((sl >> 2) ^ s(hx)) << 2 | (sl & 3)
            constexpr int UX = G::XPB / 16;
            const int p = u / UX, sl = u % UX;
            const int hy = p / G::HW, hx = p - hy * G::HW;
            const int sw = UX >= 16 ? (hx & 3) : ((hx >> 1) & 1);
            const int c = (((sl >> 2) ^ sw) << 2) | (sl & 3);
            rel[it] = (unsigned)(((hy * a.Wf + hx) * a.dil * a.C + c * 8) * 2);
            pk[it] = p < G::HH * G::HW ? (unsigned)(hy << 16 | hx) : 0x7fff0000u;
        } else {
            // dy tile: pixel q = v / 8, LDS position sy holds chunk ((sy >> 2) ^ ((q >> 1) & 1)) << 2 | (sy & 3)
            const int v = u - G::XU, q = v >> 3, sy = v & 7;
            const int ty = q / G::TW, tx = q - ty * G::TW;
            const int cy = (((sy >> 2) ^ ((q >> 1) & 1)) << 2) | (sy & 3);
            rel[it] = (unsigned)(((ty * a.Wf + tx) * a.dil * a.K + cy * 8) * 2);
            pk[it] = v < G::YU ? (unsigned)(ty << 16 | tx) : 0x7fff0000u;
        }
    }
}
// DMA instruction IT of this wave.  Branch-free: buffer_load ... lds with an out-of-range offset (loads zeros) outside the image.
template <class G, int IT>
__device__ __forceinline__ void wg_dma_piece(const WgArgs& a, const WgDma& d, char* lds, int wave, const unsigned (&rel)[G::DMA_IT],
                                             const unsigned (&pk)[G::DMA_IT]) {
    const int j = IT * G::NW + wave;
    const bool is_x = (IT * G::NW + G::NW - 1) * 64 + 63 < G::XU ? true : IT * G::NW * 64 >= G::XU ? false : j * 64 < G::XU;   // wave-uniform
    const int y = (is_x ? d.ym - 1 : d.ym) + (int)(pk[IT] >> 16), x = (is_x ? d.xm - 1 : d.xm) + (int)(pk[IT] & 0xffff);
    const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
    const unsigned off = (is_x ? d.xbase : d.ybase) + rel[IT];
    char* dst = (IT * G::NW + G::NW - 1 < G::NDMA || j < G::NDMA) ? lds + d.slot * G::SLOTB + j * 1024 : lds + 2 * G::SLOTB;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_x ? d.xr : d.yr, (wg_lptr_t)dst, 16, (int)(ok ? off : 0xffffffffu), 0, 0, 0);
#else
    (void)dst; (void)ok; (void)off;
#endif
}
template <class G, int IT>
__device__ __forceinline__ void wg_dma_all(const WgArgs& a, const WgDma& d, char* lds, int wave, const unsigned (&rel)[G::DMA_IT],
                                           const unsigned (&pk)[G::DMA_IT]) {
    if constexpr (IT < G::DMA_IT) {
        wg_dma_piece<G, IT>(a, d, lds, wave, rel, pk);
        wg_dma_all<G, IT + 1>(a, d, lds, wave, rel, pk);
    }
}

// MFMAs g .. NG-1 of a tile: request B fragment g + D (and, at the scheduled position, the next k-step's A fragment), wait for
// fragment g, multiply.  The order is pinned; the DMA instructions of the NEXT tile ride behind every PER-th MFMA from the start.
template <class G, int g>
__device__ __forceinline__ void wg_mfma(const WgArgs& a, const WgDma& d, char* lds, int wave, f32x16_t (&acc)[G::NB],
                                        TrFrag (&fa)[2], TrFrag (&fb)[G::D + 1], const unsigned (&xb)[3], unsigned ya,
                                        const unsigned (&rel)[G::DMA_IT], const unsigned (&pk)[G::DMA_IT]) {
    constexpr int s = g / G::NB, j = g % G::NB;
    if constexpr (g + G::D < G::NG) wg_read_b<G, g + G::D>(fb[(g + G::D) % (G::D + 1)], xb);
    if constexpr (G::a_at(g)) wg_read_a<G, s + 1>(fa[(s + 1) & 1], ya);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(G::later_than(g)) : "memory");
    tr_fence(fb[g % (G::D + 1)]);
    if constexpr (j == 0) tr_fence(fa[s & 1]);
    acc[j] = mfma16(tr_value(fa[s & 1]), tr_value(fb[g % (G::D + 1)]), acc[j], 0, 0, 0);
    constexpr int PER = G::NG / G::DMA_IT >= 6 ? 6 : 4;
#if WG_ABL != 2
    if constexpr (g % PER == PER - 1 && g / PER < G::DMA_IT) wg_dma_piece<G, g / PER>(a, d, lds, wave, rel, pk);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g + 1 < G::NG) wg_mfma<G, g + 1>(a, d, lds, wave, acc, fa, fb, xb, ya, rel, pk);
}
template <class G, int g>
__device__ __forceinline__ void wg_prefetch(TrFrag (&fb)[G::D + 1], const unsigned (&xb)[3]) {
    if constexpr (g < G::D) {
        wg_read_b<G, g>(fb[g % (G::D + 1)], xb);
        wg_prefetch<G, g + 1>(fb, xb);
    }
}

template <int CWIN, int TW>
__global__ __launch_bounds__(CWIN * 4) void wgrad_ws_kernel(const WgArgs a) {
    typedef WgCfg<CWIN, TW> G;
    extern __shared__ __attribute__((aligned(1024))) char lds[];     // [2][SLOTB] (x halo, dy tile), 1 KiB dump area
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kf = wave & 1, cf = wave >> 1;                         // this wave's 32 k rows and 32 c columns of the block
    // The launch is the flat sequence of (problem, block, tile) triples, problem i occupying [qend[i - 1], qend[i]); this workgroup
    // owns [q_begin, q_end) and flushes its accumulators whenever the sequence crosses into another (problem, block).
    const int q_begin = blockIdx.x * a.per_wg, q_end = min(a.total, q_begin + a.per_wg);
    if (q_begin >= q_end) return;

    struct Loc { int prob, gi, item, tl; };          // problem, geometry, (k, c) block of the problem, tile of the block
    int cursor = 0;                                  // problems are located front to back (q only grows)
    auto locate = [&](int q) {
        const int qq = min(q, a.total - 1);
        while (qq >= a.qend[cursor]) ++cursor;
        const int q0 = cursor ? a.qend[cursor - 1] : 0;
        Loc l;
        l.prob = cursor;
        l.gi = a.geo_of[cursor];
        const int nt = a.geo[l.gi].ntiles;
        l.item = (qq - q0) / nt;
        l.tl = (qq - q0) - l.item * nt;
        return l;
    };
    unsigned rel[G::DMA_IT], pk[G::DMA_IT];
    WgDma d;
    // (q >= q_end: an origin below the image, every unit loads zeros into the free buffer -- keeps the loop branch-free)
    auto set_tile = [&](const Loc& l, int q, int slot_) {
        const WgGeo& g = a.geo[l.gi];
        const int kg_ = l.item / g.cgroups, cg_ = l.item - kg_ * g.cgroups;
        const int txy = g.tiles_x * g.tiles_y;
        const int n_ = l.tl / txy, r_ = l.tl - n_ * txy;
        d.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16raw*>(a.in[l.prob]), 0, g.in_bytes, 0x00020000);
        d.yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16raw*>(a.dy[l.prob]), 0, g.dy_bytes, 0x00020000);
        d.H = g.H; d.W = g.W;
        d.ym = q < q_end ? (r_ / g.tiles_x) * G::TH : (1 << 20);
        d.xm = (r_ % g.tiles_x) * TW;
        const int dd_ = g.dil * g.dil, nf_ = n_ / dd_, sg_ = n_ - nf_ * dd_, sy_ = sg_ / g.dil, sx_ = sg_ - sy_ * g.dil;
        const int org_ = (nf_ * g.Hf + sy_) * g.Wf + sx_;                      /* first pixel of the sub-grid */
        d.xbase = (unsigned)(((org_ + ((d.ym - 1) * g.Wf + d.xm - 1) * g.dil) * g.C + cg_ * CWIN) * 2);
        d.ybase = (unsigned)(((org_ + (d.ym * g.Wf + d.xm) * g.dil) * g.K + kg_ * 64) * 2);
        d.slot = slot_;
    };
    Loc cur = locate(q_begin);
    int map_gi = cur.gi;                             // geometry the (lane, instruction) -> offset map of the DMA was built for
    wg_dma_map<G>(a.geo[map_gi], wave, lane, rel, pk);
    set_tile(cur, q_begin, 0);
    wg_dma_all<G, 0>(a, d, lds, wave, rel, pk);

    f32x16_t acc[G::NB];
#pragma unroll
    for (int j = 0; j < G::NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // transposing-read lane addressing (halo.hip): lane -> pixel (lane >> 5) * 8 + ((lane & 15) >> 2) (+4 for the hi half), channels
    // ((lane >> 4) & 1) * 16 + (lane & 3) * 4 ..+3 of its 32-channel fragment
    const int tr_p = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tr_c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(wg_lptr_t)lds;
    // x: halo pixel (0, q + dwi) + this wave's first c fragment; the 64-byte group g of a pixel sits at g ^ s(hx)
    unsigned xrel[3];
#pragma unroll
    for (int dwi = 0; dwi < 3; ++dwi) {
        const int hx = tr_p + dwi;
        const int sw = CWIN >= 128 ? (hx & 3) : ((hx >> 1) & 1);
        xrel[dwi] = (unsigned)(tr_p * G::XPB + ((cf ^ sw) << 6) + tr_c * 2);
    }
    const unsigned yrel = (unsigned)(G::XB + tr_p * G::YPB + ((kf ^ ((tr_p >> 1) & 1)) << 6) + tr_c * 2);

    // dw[(k * wt + slot(t)) * C + c] += acc: rows k = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column c = lane & 31
    auto flush = [&](const Loc& l) {
        const WgGeo& g = a.geo[l.gi];
        const int kg = l.item / g.cgroups, cg = l.item - kg * g.cgroups;
        float* __restrict__ dw = a.dw[l.prob];
        const int kb = kg * 64 + kf * 32 + 4 * (lane >> 5);
        const int c = cg * CWIN + cf * 32 + (lane & 31);
        const bool rows_exist = kg * 64 + kf * 32 < g.K;              // (K = 32: the odd waves' fragment lies past the weight)
#pragma unroll
        for (int j = 0; j < G::NB; ++j) {
            const int ws = a.wslot[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb + (r & 3) + 8 * (r >> 2);
#if WG_ABL == 1
                if (acc[j][r] == 1.2345f)
#endif
                if (rows_exist) atomicAdd(dw + ((int64_t)k * g.wt + ws) * g.C + c, acc[j][r]);
                acc[j][r] = 0.f;
            }
        }
    };

    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): the first tile has landed (compiler-visible)
    int slot = 0;
    Loc held = cur;                                                   // the (problem, block) the accumulators belong to
    for (int q = q_begin; q < q_end; ++q) {
        if (cur.prob != held.prob || cur.item != held.item) {         // (wave-uniform, a few times per workgroup)
            flush(held);
            held = cur;
        }
        __builtin_amdgcn_s_barrier();                                 // tile landed for every wave; the other buffer is free
        const Loc nxt = locate(q + 1);
        if (nxt.gi != map_gi) {                                       // (the next tile belongs to a problem of another geometry)
            map_gi = nxt.gi;
            wg_dma_map<G>(a.geo[map_gi], wave, lane, rel, pk);
        }
        set_tile(nxt, q + 1, slot ^ 1);
        const unsigned sb = lds0 + slot * G::SLOTB;
        unsigned xb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) xb[k] = sb + xrel[k];
        const unsigned ya = sb + yrel;
        TrFrag fa[2], fb[G::D + 1];
        wg_read_a<G, 0>(fa[0], ya);
        wg_prefetch<G, 0>(fb, xb);
        wg_mfma<G, 0>(a, d, lds, wave, acc, fa, fb, xb, ya, rel, pk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the next tile has landed (this wave's part)
        slot ^= 1;
        cur = nxt;
    }
    flush(held);
}

// ---------------------------------------------------------------------------------------------- host side
static bool wgradws_takes(const tcvom_conv_desc* d, int ldy, int* wslot, int* dil_out = nullptr) {
    static const bool disabled = getenv("TCVOM_NO_WGRADWS") != nullptr;     // A/B switch
    if (disabled) return false;
    if (d->in_step != 1 || d->out_step != 1 || d->out_off_h != 0 || d->out_off_w != 0) return false;
    if (d->H != d->OH || d->W != d->OW || d->PH != d->H || d->PW != d->W) return false;
    const int C = d->C, K = d->K;
    // C: 128-channel windows when C is a multiple of 128, else 64-channel windows (C = 64, 192, 320 ...).  K = 32: one k group whose
    // upper 32 rows (the k fragment of the odd waves) multiply the next pixel's values and are never written.
    if (ldy != K || !(K % 64 == 0 || K == 32) || C % 64 != 0) return false;
    if (d->H < 4 || d->W < 8) return false;
    int slots[9];
    for (int t = 0; t < 9; ++t) slots[t] = -1;
    int n = 0;
    // dilation: the taps are {-dil, 0, dil}^2 (ResnetDilated, models/FBA/models.py:203-217: dilation 2 and 4 with padding = dilation)
    int dil = 1;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->tap_w[t] >= 0 && abs(d->tap_dh[t]) > dil) dil = abs(d->tap_dh[t]);
    static const bool no_dil = getenv("TCVOM_WGRADWS_NO_DIL") != nullptr;     // A/B switch: dilated layers on igemm_tt
    if (dil > 1 && no_dil) return false;
    if (dil > 8 || d->H % dil != 0 || d->W % dil != 0 || d->H / dil < 4 || d->W / dil < 8) return false;
    if ((long long)d->N * dil * dil * cdiv(d->H / dil, 8) * cdiv(d->W / dil, 16) >= (1ll << 24)) return false;
    for (int t = 0; t < d->ntaps; ++t) {
        if (d->tap_w[t] < 0) continue;
        if (d->tap_dh[t] % dil != 0 || d->tap_dw[t] % dil != 0) return false;
        const int dh = d->tap_dh[t] / dil, dw_ = d->tap_dw[t] / dil;
        if (dh < -1 || dh > 1 || dw_ < -1 || dw_ > 1) return false;
        const int c = (dh + 1) * 3 + (dw_ + 1);
        if (slots[c] >= 0) return false;
        slots[c] = d->tap_w[t];
        ++n;
    }
    if (n != 9) return false;
    const long long in_b = (long long)d->N * d->H * d->W * C * 2, dy_b = (long long)d->N * d->H * d->W * K * 2;
    if (in_b >= (1ll << 31) || dy_b >= (1ll << 31)) return false;
    if (wslot) for (int t = 0; t < 9; ++t) wslot[t] = slots[t];
    if (dil_out) *dil_out = dil;
    return true;
}

// geometry of one descriptor as the kernel wants it; false: not a shape for this kernel
static bool wg_make_geo(const tcvom_conv_desc* d, int ldy, WgGeo* g, int* cwin_out, int* wslot) {
    int dil = 1;
    if (!wgradws_takes(d, ldy, wslot, &dil)) return false;
    const int C = d->C, K = d->K;
    const int cwin = C % 128 == 0 ? 128 : 64, tw = cwin == 64 ? 32 : 16;
    g->dil = dil; g->Hf = d->H; g->Wf = d->W;
    g->N = d->N * dil * dil; g->H = d->H / dil; g->W = d->W / dil; g->C = C; g->K = K; g->wt = d->wt;
    g->tiles_x = cdiv(g->W, tw);
    g->tiles_y = cdiv(g->H, 8);
    g->ntiles = g->N * g->tiles_x * g->tiles_y;
    g->kgroups = cdiv(K, 64);
    g->cgroups = C / cwin;
    g->in_bytes = (unsigned)((long long)d->N * d->H * d->W * C * 2);
    g->dy_bytes = (unsigned)((long long)d->N * d->H * d->W * K * 2);
    *cwin_out = cwin;
    return true;
}

// 1: launched, 0: not shapes for this kernel (or not ONE channel window), < 0: error.  `nprob` problems, problem i of geometry
// descs[geo_index[i]] (geo_index == NULL: all of descs[0]): the calls of one layer in a window, of several layers of one shape, and --
// since the atomic flush costs every launch the same ~45 us -- of layers of different shapes: the more problems, the fewer
// workgroups share a (problem, block) and the fewer launches pay the flush.  weights.py: WeightBank.run_deferred_wgrads.
static int wgradws_launch(const void* const* dys, const void* const* ins, float* const* dws, int nprob, const tcvom_conv_desc* descs,
                          int ngeo, const int32_t* geo_index, void* stream) {
    WgArgs a;
    if (nprob < 1 || nprob > WG_MAX_PROBLEMS || ngeo < 1 || ngeo > WG_MAX_GEO) return 0;
    int cwin = 0;
    for (int gi = 0; gi < WG_MAX_GEO; ++gi) {
        const int j = gi < ngeo ? gi : 0;
        int cw = 0, slots[9];
        if (!wg_make_geo(descs + j, descs[j].K, &a.geo[gi], &cw, slots)) return 0;
        if (gi == 0) { cwin = cw; for (int t = 0; t < 9; ++t) a.wslot[t] = slots[t]; }
        if (cw != cwin) return 0;
        for (int t = 0; t < 9; ++t) if (slots[t] != a.wslot[t]) return 0;
    }
    long long total = 0;
    for (int i = 0; i < WG_MAX_PROBLEMS; ++i) {
        const int j = i < nprob ? i : 0;
        a.dy[i] = (const h16raw*)dys[j]; a.in[i] = (const h16raw*)ins[j]; a.dw[i] = dws[j];
        const int gi = geo_index ? geo_index[j] : 0;
        if (gi < 0 || gi >= ngeo) return tcvom_fail(TCVOM_ERR_ARG, "wgrad_ws: geometry index %d of problem %d (0..%d)", gi, j, ngeo - 1);
        a.geo_of[i] = (unsigned char)gi;
        if (i < nprob) total += (long long)a.geo[gi].kgroups * a.geo[gi].cgroups * a.geo[gi].ntiles;
        if (total >= (1ll << 31)) return 0;
        a.qend[i] = (int)total;
    }
    a.nprob = nprob;
    a.total = (int)total;
    // one persistent workgroup per CU; a run shorter than 2 tiles is all pipeline fill
    int wgs = 256;
    if (total < 2ll * wgs) wgs = (int)((total + 1) / 2);
    a.per_wg = cdiv(total, wgs);
    const dim3 grid(cdiv(total, a.per_wg));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (cwin == 64) {
        auto kern = wgrad_ws_kernel<64, 32>;
        constexpr size_t lds_bytes = 2 * WgCfg<64, 32>::SLOTB + 1024;
        static_assert(lds_bytes <= 160 * 1024, "LDS budget");
        static bool attr = false;
        if (!attr) { e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    } else {
        auto kern = wgrad_ws_kernel<128, 16>;
        constexpr size_t lds_bytes = 2 * WgCfg<128, 16>::SLOTB + 1024;
        static_assert(lds_bytes <= 160 * 1024, "LDS budget");
        static bool attr = false;
        if (!attr) { e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, st, a);
    }
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "wgrad_ws: %s", hipGetErrorString(e));
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "wgrad_ws: %s", hipGetErrorString(e2));
    return 1;
}

// the problems of ONE geometry (tcvom_wgrad_igemm_batched routes eligible shapes here)
int wgradws_try_launch(const void* const* dys, const void* const* ins, float* const* dws, int nprob, const tcvom_conv_desc* d,
                       int nphase, int ldy, void* stream) {
    if (nphase != 1 || ldy != d->K) return 0;
    return wgradws_launch(dys, ins, dws, nprob, d, 1, nullptr, stream);
}

// C ABI: the weight gradients of up to WG_MAX_PROBLEMS convolutions of ONE geometry in one launch (declared in tcvom_hip.h)
extern "C" int tcvom_wgrad_ws_multi(const void* const* dy, const void* const* in, float* const* dw, int32_t nprob,
                                    const tcvom_conv_desc* d, int32_t ldy, void* stream) {
    TCVOM_CHECK_ARG(dy && in && dw && d, "wgrad_ws_multi: null pointer");
    TCVOM_CHECK_ARG(nprob >= 1 && nprob <= WG_MAX_PROBLEMS, "wgrad_ws_multi: %d problems (1..%d)", nprob, WG_MAX_PROBLEMS);
    for (int i = 0; i < nprob; ++i) TCVOM_CHECK_ARG(dy[i] && in[i] && dw[i], "wgrad_ws_multi: null pointer in problem %d", i);
    const int r = wgradws_try_launch(dy, in, dw, nprob, d, 1, ldy, stream);
    TCVOM_CHECK_ARG(r != 0, "wgrad_ws_multi: not a stride-1 3x3 convolution with C a multiple of 64 and K = 32 or a multiple of 64");
    return r < 0 ? r : TCVOM_OK;
}
extern "C" int32_t tcvom_wgrad_ws_max_problems(void) { return WG_MAX_PROBLEMS; }
extern "C" int32_t tcvom_wgrad_ws_max_geometries(void) { return WG_MAX_GEO; }
// ... of up to WG_MAX_GEO geometries that share the channel window (C a multiple of 128, or not): problem i has descs[geo_index[i]]
extern "C" int tcvom_wgrad_ws_hetero(const void* const* dy, const void* const* in, float* const* dw, int32_t nprob,
                                     const tcvom_conv_desc* descs, int32_t ngeo, const int32_t* geo_index, void* stream) {
    TCVOM_CHECK_ARG(dy && in && dw && descs && geo_index, "wgrad_ws_hetero: null pointer");
    TCVOM_CHECK_ARG(nprob >= 1 && nprob <= WG_MAX_PROBLEMS, "wgrad_ws_hetero: %d problems (1..%d)", nprob, WG_MAX_PROBLEMS);
    TCVOM_CHECK_ARG(ngeo >= 1 && ngeo <= WG_MAX_GEO, "wgrad_ws_hetero: %d geometries (1..%d)", ngeo, WG_MAX_GEO);
    for (int i = 0; i < nprob; ++i) TCVOM_CHECK_ARG(dy[i] && in[i] && dw[i], "wgrad_ws_hetero: null pointer in problem %d", i);
    const int r = wgradws_launch(dy, in, dw, nprob, descs, ngeo, geo_index, stream);
    TCVOM_CHECK_ARG(r != 0, "wgrad_ws_hetero: not stride-1 3x3 convolutions of one channel window (C a multiple of 64 / of 128, K = 32 or a multiple of 64)");
    return r < 0 ? r : TCVOM_OK;
}

// name for profiles / bench labels
const char* wgradws_variant(const tcvom_conv_desc* d, int ldy) {
    if (!wgradws_takes(d, ldy, nullptr)) return nullptr;
    return d->C % 128 == 0 ? "wgrad_ws<128>" : "wgrad_ws<64>";
}
