// Facade kernels: input preprocessing / trimap synthesis, the window losses, and Adam.
//   preprocess ........ FullModel.preprocess + make_trimap   (models/model.py:54-92)
//   masked L1 ......... single_image_loss / L1_mask / _dtSSD  (models/model.py:94-127,326-345; utils/loss_func.py:9-22)
//   attention BCE ..... the L_att block                       (models/model.py:285-323)
//   adam .............. torch.optim.Adam(weight_decay=1e-4)   (train_ddp.py:296-297,65)
// All HBM-bound streaming kernels; losses accumulate block partials with fp32 atomics into a small
// device-side accumulator so that no host synchronisation is needed (the reference syncs at :296).
#include "common.h"

#define GRID_STRIDE(v, n) \
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < (n); v += (int64_t)gridDim.x * blockDim.x)
static int sgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- preprocess, stage 1 (per pixel of every frame)
// a [F,1,H,W], fg/bg [F,3,H,W] fp32 0..255 BGR (F = B*S)
__global__ void preprocess_kernel(const float* __restrict__ a, const float* __restrict__ fg, const float* __restrict__ bg,
                                  float* __restrict__ gts, float* __restrict__ fgs, float* __restrict__ bgs,
                                  float* __restrict__ imgs, unsigned char* __restrict__ unk_raw, int64_t F, int64_t HW, float eps) {
    const float s = 1.0f / 255.0f;
    GRID_STRIDE(v, F * HW) {
        const int64_t f = v / HW, p = v % HW;
        const float g = a[v] * s;
        gts[v] = g;
        float al = g < eps ? 0.f : g;
        al = al > 1.f - eps ? 1.f : al;
        unk_raw[v] = (al > 0.f && al < 1.f) ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ff = fg[(f * 3 + (2 - c)) * HW + p] * s;      // BGR -> RGB flip
            fgs[(f * 3 + c) * HW + p] = ff;
            if (bg) {
                const float bb = bg[(f * 3 + (2 - c)) * HW + p] * s;
                bgs[(f * 3 + c) * HW + p] = bb;
                imgs[(f * 3 + c) * HW + p] = ff * g + bb * (1.f - g);
            } else {                                                  // EvalModel: `fg` is the frame itself, `a` the user trimap
                imgs[(f * 3 + c) * HW + p] = ff;
            }
        }
    }
}
// separable max filter (radius r) on uint8 planes; pass 0 = along W, pass 1 = along H
__global__ void dilate_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int64_t F, int H, int W,
                              int r, int pass) {
    GRID_STRIDE(v, F * H * W) {
        const int x = (int)(v % W), y = (int)((v / W) % H);
        unsigned char m = 0;
        if (pass == 0) {
            const int lo = max(0, x - r), hi = min(W - 1, x + r);
            for (int i = lo; i <= hi && !m; ++i) m |= in[v - x + i];
        } else {
            const int lo = max(0, y - r), hi = min(H - 1, y + r);
            for (int i = lo; i <= hi && !m; ++i) m |= in[v + (int64_t)(i - y) * W];
        }
        out[v] = m;
    }
}
// The same filter on 8-pixel words (W % 8 == 0, planes hold 0 / 1 bytes): a thread produces 8 horizontally adjacent pixels.
// pass 0: OR of the 2 r + 1 byte-shifted copies of the row, taken from the 64-bit words around the target word (funnel shifts);
// pass 1: OR of the same word position over rows y - r .. y + r (coalesced 8-byte loads).  25 byte loads per pixel become
// 5 / 25 word loads per 8 pixels: 77 -> ~10 us per pass on a 3 x 1088 x 1920 window.
__global__ __launch_bounds__(256) void dilate_words_kernel(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                           int64_t F, int H, int W8, int r, int pass) {
    GRID_STRIDE(v, F * H * W8) {
        const int xw = (int)(v % W8), y = (int)((v / W8) % H);
        unsigned long long m = 0ull;
        if (pass == 0) {
            // words xw - nw .. xw + nw cover every pixel within r of the target word's 8 pixels
            const int nw = (r + 7) >> 3;
            const unsigned long long* row = in + (v - xw);
            for (int d = -r; d <= r; ++d) {
                // pixel i of the result ORs pixel i + d of the row: bytes [8 xw + d, 8 xw + d + 8)
                const int b0 = 8 * xw + d;                            // first source byte (may be negative / beyond the row)
                const int w0 = b0 >> 3, sh = (b0 & 7) * 8;            // (arithmetic shift: floor for negative b0)
                const unsigned long long lo = (w0 >= 0 && w0 < W8) ? row[w0] : 0ull;
                const unsigned long long hi = (sh && w0 + 1 >= 0 && w0 + 1 < W8) ? row[w0 + 1] : 0ull;
                m |= sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
            }
            (void)nw;
        } else {
            const int lo = max(0, y - r), hi = min(H - 1, y + r);
            for (int i = lo; i <= hi; ++i) m |= in[v + (int64_t)(i - y) * W8];
        }
        out[v] = m;
    }
}
// stage 3: network input x8 [F,H,W,8] bf16 = {norm R,G,B, onehot bg,unk,fg, 0, 0}; trimask fp32; tris_vis fp32
__global__ void assemble_kernel(const float* __restrict__ gts, const float* __restrict__ imgs, const unsigned char* __restrict__ dil,
                                uint4* __restrict__ x8, float* __restrict__ trimask, float* __restrict__ tris_vis,
                                int64_t F, int64_t HW, float eps, int tri_channels, uint4* __restrict__ x8_f16) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
    GRID_STRIDE(v, F * HW) {
        const int64_t f = v / HW, p = v % HW;
        const float g = gts[v];
        float al = g < eps ? 0.f : g;
        al = al > 1.f - eps ? 1.f : al;
        const bool u = dil[v] != 0;
        const int cls = u ? 1 : (int)(2.f * al);
        float o[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (imgs[(f * 3 + c) * HW + p] - mean[c]) * istd[c];
        if (tri_channels == 3) {            // GCA: one-hot {bg, unknown, fg}
            o[3] = cls == 0 ? 1.f : 0.f;
            o[4] = cls == 1 ? 1.f : 0.f;
            o[5] = cls == 2 ? 1.f : 0.f;
        } else {                            // DIM / Index: one channel, 128/255 in the unknown region (models/model.py:67-69)
            o[3] = u ? 128.f / 255.f : al;
            o[4] = 0.f;
            o[5] = 0.f;
        }
        o[6] = 0.f;
        o[7] = 0.f;
        x8[v] = pack8(o);
        if (x8_f16) x8_f16[v] = pack8_ieee(o);       // the same input in IEEE fp16 (fp16 island of the bf16 build: encoder conv1 reads it)
        trimask[v] = u ? 1.f : 0.f;
        tris_vis[v] = u ? 128.f / 255.f : (tri_channels == 3 ? g : al);
    }
}

// ---------------------------------------------------------------- masked L1 (L_alpha and L_dt)
// acc[0] += sum |d| m1,  acc[1] += count(m1 > eps) ; d = (r1 - g1) - (r2 - g2), r = m ? pred : gt
// writes alphas (clamped refine) / comps when requested (single-frame form, p2 == NULL).
__global__ __launch_bounds__(256) void masked_l1_fwd_kernel(
    const float* __restrict__ p1, const float* __restrict__ g1, const float* __restrict__ m1,
    const float* __restrict__ p2, const float* __restrict__ g2, const float* __restrict__ m2,
    const float* __restrict__ fgs, const float* __restrict__ bgs, float* __restrict__ alphas, float* __restrict__ comps,
    float* __restrict__ acc, int64_t B, int64_t HW, int64_t p_stride, int64_t frame_stride, int64_t rgb_stride)
{
    // gt/mask/alphas are [B, S, 1, H, W] slices: element (b, p) at b*frame_stride + p ; predictions at
    // b*p_stride + p ; rgb at b*rgb_stride + c*HW + p
    __shared__ float red[4];
    float s = 0.f, n = 0.f;
    GRID_STRIDE(v, B * HW) {
        const int64_t b = v / HW, p = v % HW;
        const int64_t i = b * frame_stride + p;
        const int64_t ip = b * p_stride + p;
        const float m = m1[i];
        const float r1 = m != 0.f ? p1[ip] : g1[i];
        float d = r1 - g1[i];
        if (p2) {
            const float r2 = m2[i] != 0.f ? p2[ip] : g2[i];
            d -= r2 - g2[i];
        }
        s += fabsf(d) * m;
        n += m > 1.001e-5f ? 1.f : 0.f;
        if (alphas) {
            const float rc = fminf(fmaxf(r1, 0.f), 1.f);
            alphas[i] = rc;
            if (comps) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int64_t j = b * rgb_stride + c * HW + p;
                    comps[j] = fminf(fmaxf(fgs[j] * r1 + bgs[j] * (1.f - r1), 0.f), 1.f);
                }
            }
        }
    }
    s = block_sum_256(s, red);
    n = block_sum_256(n, red);
    if (threadIdx.x == 0) { atomicAdd(acc, s); atomicAdd(acc + 1, n); }
}
// The single-frame form (p2 == NULL, alphas / comps written) on float4: grid (blocks, B); no 64-bit divisions, 16-byte accesses
// (the scalar kernel above spends its time in v / HW, v % HW and 4-byte loads: 113 us for one 1088 x 1920 frame, ~107 MB).
__global__ __launch_bounds__(256) void masked_l1_fwd4_kernel(
    const float4* __restrict__ p1, const float4* __restrict__ g1, const float4* __restrict__ m1,
    const float4* __restrict__ fgs, const float4* __restrict__ bgs, float4* __restrict__ alphas, float4* __restrict__ comps,
    float* __restrict__ acc, int HW4, int64_t p_stride4, int64_t frame_stride4, int64_t rgb_stride4)
{
    __shared__ float red[4];
    const int64_t b = blockIdx.y;
    p1 += b * p_stride4; g1 += b * frame_stride4; m1 += b * frame_stride4;
    if (alphas) alphas += b * frame_stride4;
    if (comps) { comps += b * rgb_stride4; fgs += b * rgb_stride4; bgs += b * rgb_stride4; }
    float s = 0.f, n = 0.f;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < HW4; v += gridDim.x * 256) {
        const float4 m = m1[v], p = p1[v], g = g1[v];
        const float mm[4] = {m.x, m.y, m.z, m.w}, pp[4] = {p.x, p.y, p.z, p.w}, gg[4] = {g.x, g.y, g.z, g.w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = mm[k] != 0.f ? pp[k] : gg[k];
            s += fabsf(r[k] - gg[k]) * mm[k];
            n += mm[k] > 1.001e-5f ? 1.f : 0.f;
        }
        if (alphas) {
            alphas[v] = make_float4(fminf(fmaxf(r[0], 0.f), 1.f), fminf(fmaxf(r[1], 0.f), 1.f), fminf(fmaxf(r[2], 0.f), 1.f), fminf(fmaxf(r[3], 0.f), 1.f));
            if (comps) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4 f = fgs[(int64_t)c * HW4 + v], q = bgs[(int64_t)c * HW4 + v];
                    comps[(int64_t)c * HW4 + v] = make_float4(fminf(fmaxf(f.x * r[0] + q.x * (1.f - r[0]), 0.f), 1.f),
                                                              fminf(fmaxf(f.y * r[1] + q.y * (1.f - r[1]), 0.f), 1.f),
                                                              fminf(fmaxf(f.z * r[2] + q.z * (1.f - r[2]), 0.f), 1.f),
                                                              fminf(fmaxf(f.w * r[3] + q.w * (1.f - r[3]), 0.f), 1.f));
                }
            }
        }
    }
    s = block_sum_256(s, red);
    n = block_sum_256(n, red);
    if (threadIdx.x == 0) { atomicAdd(acc, s); atomicAdd(acc + 1, n); }
}
// dp1 (+)= w * sign(d) * m1 * [m1] / clamp(count) ; dp2 (+)= -w * sign(d) * m1 * [m2]      (w = gout * weight)
__global__ void masked_l1_bwd_kernel(
    const float* __restrict__ p1, const float* __restrict__ g1, const float* __restrict__ m1,
    const float* __restrict__ p2, const float* __restrict__ g2, const float* __restrict__ m2,
    const float* __restrict__ acc, const float* __restrict__ gout, float weight,
    float* __restrict__ dp1, float* __restrict__ dp2, int accumulate, int64_t B, int64_t HW, int64_t p_stride,
    int64_t frame_stride, int64_t numel_total)
{
    const float denom = fminf(fmaxf(acc[1], 1.001e-5f), (float)numel_total + 1.f);
    const float w = gout[0] * weight / denom;
    GRID_STRIDE(v, B * HW) {
        const int64_t b = v / HW, p = v % HW;
        const int64_t i = b * frame_stride + p;
        const int64_t ip = b * p_stride + p;
        const float m = m1[i];
        const float r1 = m != 0.f ? p1[ip] : g1[i];
        float d = r1 - g1[i];
        if (p2) {
            const float r2 = m2[i] != 0.f ? p2[ip] : g2[i];
            d -= r2 - g2[i];
        }
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float ga = w * sg * m;
        const float a1 = m != 0.f ? ga : 0.f;
        if (accumulate) dp1[ip] += a1; else dp1[ip] = a1;
        if (p2 && dp2) {
            const float a2 = m2[i] != 0.f ? -ga : 0.f;
            if (accumulate) dp2[ip] += a2; else dp2[ip] = a2;
        }
    }
}

// ---------------------------------------------------------------- 8x8 average pooling of the ground truth alphas
__global__ void avgpool8_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t F, int H, int W) {
    const int h = H / 8, w = W / 8;
    GRID_STRIDE(v, F * h * w) {
        const int x = (int)(v % w), y = (int)((v / w) % h);
        const int64_t f = v / ((int64_t)w * h);
        float s = 0.f;
        for (int dy = 0; dy < 8; ++dy)
            for (int dx = 0; dx < 8; ++dx) s += g[(f * H + y * 8 + dy) * W + x * 8 + dx];
        out[v] = s * (1.f / 64.f);
    }
}
// attention BCE: logits [B,W2,N]; pooled gts cg/adj [B,h,w]; mask uint8 [B,N].
//   acc[0] += sum_{j, unknown u} bce(logit, target),  acc[1] += #unknown ; dlogit (unscaled) = sigmoid(x) - t at unknown, 0 else
__global__ __launch_bounds__(256) void att_bce_kernel(const float* __restrict__ logits, const float* __restrict__ cg,
                                                      const float* __restrict__ adj, const unsigned char* __restrict__ mask,
                                                      float* __restrict__ dlogit, float* __restrict__ acc,
                                                      int B, int h, int w, int win, float thres, float smooth,
                                                      int64_t g_bstride) {
    __shared__ float red[4];
    const int W2 = win * win, R = win / 2;
    const int64_t N = (int64_t)h * w;
    float s = 0.f, n = 0.f;
    GRID_STRIDE(v, (int64_t)B * W2 * N) {
        const int64_t u = v % N;
        const int j = (int)((v / N) % W2);
        const int b = (int)(v / (N * W2));
        float dl = 0.f;
        if (mask[b * N + u]) {
            const int y = (int)(u / w) + j / win - R, x = (int)(u % w) + j % win - R;
            const float nb = (y >= 0 && y < h && x >= 0 && x < w) ? adj[b * g_bstride + (int64_t)y * w + x] : 0.f;
            const float t = fabsf(cg[b * g_bstride + u] - nb) < thres ? 1.f - smooth : 0.f;
            const float xl = logits[v];
            s += fmaxf(xl, 0.f) - xl * t + log1pf(__expf(-fabsf(xl)));
            dl = 1.f / (1.f + __expf(-xl)) - t;
            if (j == 0) n += 1.f;
        }
        if (dlogit) dlogit[v] = dl;
    }
    s = block_sum_256(s, red);
    n = block_sum_256(n, red);
    if (threadIdx.x == 0) { atomicAdd(acc, s); atomicAdd(acc + 1, n); }
}
__global__ void att_bce_bwd_kernel(const float* __restrict__ dlogit_unscaled, const float* __restrict__ acc,
                                   const float* __restrict__ gout, float weight, float* __restrict__ dlogit, int64_t n, int W2) {
    const float cnt = acc[1];
    const float w = cnt > 0.f ? gout[0] * weight / (cnt * (float)W2) : 0.f;
    GRID_STRIDE(v, n) dlogit[v] = dlogit_unscaled[v] * w;
}
// out[0] (+)= weight * acc[0] / denom   (denom_mode 0: clamp(acc[1], 1.001e-5, numel+1) ; 1: acc[1]*W2, 0 if empty)
__global__ void loss_finalize_kernel(const float* __restrict__ acc, float* __restrict__ out, float weight, int denom_mode,
                                     float numel_total, int W2, int accumulate) {
    float val;
    if (denom_mode == 0) {
        val = acc[0] / fminf(fmaxf(acc[1], 1.001e-5f), numel_total + 1.f);
    } else {
        val = acc[1] > 0.f ? acc[0] / (acc[1] * (float)W2) : 0.f;
    }
    if (accumulate) out[0] += weight * val; else out[0] = weight * val;
}

// ---------------------------------------------------------------- multi-tensor Adam (L2 weight decay folded into the gradient)
// table: per tensor 5 int64 words {param, grad, exp_avg, exp_avg_sq, numel}; work: per block {tensor, chunk}
__global__ __launch_bounds__(256) void adam_kernel(const int64_t* __restrict__ table, const int* __restrict__ work,
                                                   float lr, float beta1, float beta2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, float grad_scale, const int* __restrict__ skip) {
    if (skip && *skip != 0) return;                                   // the backward of this step overflowed: no update (block-uniform)
    const int64_t* T = table + (int64_t)work[blockIdx.x * 2] * 5;
    float* p = reinterpret_cast<float*>(T[0]);
    const float* g = reinterpret_cast<const float*>(T[1]);
    float* m = reinterpret_cast<float*>(T[2]);
    float* v = reinterpret_cast<float*>(T[3]);
    const int64_t n = T[4];
    const int64_t base = (int64_t)work[blockIdx.x * 2 + 1] * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + q * 256 + threadIdx.x;
        if (i < n) {
            const float pi = p[i];
            const float gi = g[i] * grad_scale + wd * pi;
            const float mi = beta1 * m[i] + (1.f - beta1) * gi;
            const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            p[i] = pi - (lr / bc1) * (mi / denom);
        }
    }
}

// ---------------------------------------------------------------- C ABI
static int preprocess_clips_impl(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                                 float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, void* x8_f16, float* trimask,
                                 float* tris_vis, int32_t clips, int32_t frames_per_clip, int32_t H, int32_t W,
                                 const int32_t* clip_radii, float eps, int32_t tri_channels, void* stream) {
    // bg == NULL (then bgs may be NULL too): EvalModel.preprocess (models/model.py:360-386) -- `fg` is the frame, `a` the trimap
    TCVOM_CHECK_ARG(a && fg && gts && fgs && (bgs || !bg) && imgs && unk_raw && unk_tmp && unk_dil && x8 && trimask && tris_vis && clip_radii,
                    "preprocess: null pointer");
    TCVOM_CHECK_ARG(clips > 0 && frames_per_clip > 0 && H > 0 && W > 0, "preprocess: bad shape");
    for (int c = 0; c < clips; ++c) TCVOM_CHECK_ARG(clip_radii[c] >= 0, "preprocess: negative dilation radius (clip %d)", c);
    TCVOM_CHECK_ARG(tri_channels == 3 || tri_channels == 1, "preprocess: %d trimap channels (3: one-hot, 1: DIM/Index)", tri_channels);
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W, frames = (int64_t)clips * frames_per_clip;
    hipLaunchKernelGGL(preprocess_kernel, dim3(sgrid(frames * HW)), dim3(256), 0, st, a, fg, bg, gts, fgs, bgs, imgs, unk_raw, frames, HW, eps);
    const bool words = W % 8 == 0 && ((uintptr_t)unk_raw & 7) == 0 && ((uintptr_t)unk_tmp & 7) == 0 && ((uintptr_t)unk_dil & 7) == 0;
    for (int c0 = 0; c0 < clips;) {                                   // one pair of passes per run of clips with the same radius
        int c1 = c0 + 1;
        while (c1 < clips && clip_radii[c1] == clip_radii[c0]) ++c1;
        const int r = clip_radii[c0];
        const int64_t off = (int64_t)c0 * frames_per_clip * HW, nf = (int64_t)(c1 - c0) * frames_per_clip;
        if (words) {
            const int W8 = W / 8;
            hipLaunchKernelGGL(dilate_words_kernel, dim3(sgrid(nf * H * W8)), dim3(256), 0, st, (const unsigned long long*)(unk_raw + off),
                               (unsigned long long*)(unk_tmp + off), nf, H, W8, r, 0);
            hipLaunchKernelGGL(dilate_words_kernel, dim3(sgrid(nf * H * W8)), dim3(256), 0, st, (const unsigned long long*)(unk_tmp + off),
                               (unsigned long long*)(unk_dil + off), nf, H, W8, r, 1);
        } else {
            hipLaunchKernelGGL(dilate_kernel, dim3(sgrid(nf * HW)), dim3(256), 0, st, unk_raw + off, unk_tmp + off, nf, H, W, r, 0);
            hipLaunchKernelGGL(dilate_kernel, dim3(sgrid(nf * HW)), dim3(256), 0, st, unk_tmp + off, unk_dil + off, nf, H, W, r, 1);
        }
        c0 = c1;
    }
    hipLaunchKernelGGL(assemble_kernel, dim3(sgrid(frames * HW)), dim3(256), 0, st, gts, imgs, unk_dil, (uint4*)x8, trimask, tris_vis, frames, HW, eps, tri_channels, (uint4*)x8_f16);
    TCVOM_LAUNCH_CHECK("preprocess");
    return TCVOM_OK;
}
extern "C" int tcvom_preprocess_clips(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                                      float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, float* trimask,
                                      float* tris_vis, int32_t clips, int32_t frames_per_clip, int32_t H, int32_t W,
                                      const int32_t* clip_radii, float eps, int32_t tri_channels, void* stream) {
    return preprocess_clips_impl(a, fg, bg, gts, fgs, bgs, imgs, unk_raw, unk_tmp, unk_dil, x8, nullptr, trimask, tris_vis, clips, frames_per_clip,
                                 H, W, clip_radii, eps, tri_channels, stream);
}
// ... and the network input a second time as IEEE fp16 (x8_f16, same shape; NULL = tcvom_preprocess_clips): what the first conv of the
// fp16 island of the bf16 build reads (tcvom_conv_desc.in_f16)
extern "C" int tcvom_preprocess_clips_f16(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                                          float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, void* x8_f16,
                                          float* trimask, float* tris_vis, int32_t clips, int32_t frames_per_clip, int32_t H, int32_t W,
                                          const int32_t* clip_radii, float eps, int32_t tri_channels, void* stream) {
    return preprocess_clips_impl(a, fg, bg, gts, fgs, bgs, imgs, unk_raw, unk_tmp, unk_dil, x8, x8_f16, trimask, tris_vis, clips, frames_per_clip,
                                 H, W, clip_radii, eps, tri_channels, stream);
}

extern "C" int tcvom_preprocess(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                                float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, float* trimask,
                                float* tris_vis, int64_t frames, int32_t H, int32_t W, int32_t dilate_radius, float eps,
                                int32_t tri_channels, void* stream) {
    TCVOM_CHECK_ARG(frames > 0 && frames < (1ll << 31), "preprocess: bad shape");
    return tcvom_preprocess_clips(a, fg, bg, gts, fgs, bgs, imgs, unk_raw, unk_tmp, unk_dil, x8, trimask, tris_vis, 1, (int32_t)frames, H, W,
                                  &dilate_radius, eps, tri_channels, stream);
}

extern "C" int tcvom_masked_l1_fwd(const float* p1, const float* g1, const float* m1, const float* p2, const float* g2,
                                   const float* m2, const float* fgs, const float* bgs, float* alphas, float* comps,
                                   float* acc, int64_t B, int64_t HW, int64_t p_stride, int64_t frame_stride, int64_t rgb_stride,
                                   void* stream) {
    TCVOM_CHECK_ARG(p1 && g1 && m1 && acc && B > 0 && HW > 0, "masked_l1_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, 2 * sizeof(float), st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "masked_l1_fwd: memset");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!p2 && HW % 4 == 0 && HW / 4 < (1ll << 30) && p_stride % 4 == 0 && frame_stride % 4 == 0 && rgb_stride % 4 == 0 && B <= 65535 &&
        al16(p1) && al16(g1) && al16(m1) && al16(alphas) && al16(comps) && al16(fgs) && al16(bgs) && (!comps || (fgs && bgs && alphas))) {
        const int HW4 = (int)(HW / 4);
        int blocks = (HW4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(masked_l1_fwd4_kernel, dim3(blocks, (int)B), dim3(256), 0, st, (const float4*)p1, (const float4*)g1, (const float4*)m1,
                           (const float4*)fgs, (const float4*)bgs, (float4*)alphas, (float4*)comps, acc, HW4, p_stride / 4, frame_stride / 4, rgb_stride / 4);
        TCVOM_LAUNCH_CHECK("masked_l1_fwd");
        return TCVOM_OK;
    }
    hipLaunchKernelGGL(masked_l1_fwd_kernel, dim3(sgrid(B * HW)), dim3(256), 0, st, p1, g1, m1, p2, g2, m2, fgs, bgs, alphas, comps, acc, B, HW, p_stride, frame_stride, rgb_stride);
    TCVOM_LAUNCH_CHECK("masked_l1_fwd");
    return TCVOM_OK;
}
extern "C" int tcvom_masked_l1_bwd(const float* p1, const float* g1, const float* m1, const float* p2, const float* g2,
                                   const float* m2, const float* acc, const float* gout, float weight, float* dp1,
                                   float* dp2, int32_t accumulate, int64_t B, int64_t HW, int64_t p_stride, int64_t frame_stride,
                                   void* stream) {
    TCVOM_CHECK_ARG(p1 && g1 && m1 && acc && gout && dp1 && B > 0 && HW > 0, "masked_l1_bwd: bad args");
    hipLaunchKernelGGL(masked_l1_bwd_kernel, dim3(sgrid(B * HW)), dim3(256), 0, (hipStream_t)stream, p1, g1, m1, p2, g2, m2, acc, gout, weight, dp1, dp2, accumulate, B, HW, p_stride, frame_stride, B * HW);
    TCVOM_LAUNCH_CHECK("masked_l1_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_avgpool8(const float* g, float* out, int64_t frames, int32_t H, int32_t W, void* stream) {
    TCVOM_CHECK_ARG(g && out && H % 8 == 0 && W % 8 == 0, "avgpool8: bad args");
    hipLaunchKernelGGL(avgpool8_kernel, dim3(sgrid(frames * (H / 8) * (W / 8))), dim3(256), 0, (hipStream_t)stream, g, out, frames, H, W);
    TCVOM_LAUNCH_CHECK("avgpool8");
    return TCVOM_OK;
}
extern "C" int tcvom_att_bce(const float* logits, const float* cg, const float* adj, const uint8_t* mask, float* dlogit,
                             float* acc, int32_t B, int32_t h, int32_t w, int32_t window, float thres, float smooth,
                             int64_t g_bstride, int32_t zero_acc, void* stream) {
    TCVOM_CHECK_ARG(logits && cg && adj && mask && acc, "att_bce: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (zero_acc && hipMemsetAsync(acc, 0, 2 * sizeof(float), st) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "att_bce: memset");
    // every block ends with two atomics on the same two floats: a few hundred blocks, not 4096 (same-address atomics serialise)
    int grid = sgrid((int64_t)B * window * window * h * w);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(att_bce_kernel, dim3(grid), dim3(256), 0, st, logits, cg, adj, mask, dlogit, acc, B, h, w, window, thres, smooth, g_bstride);
    TCVOM_LAUNCH_CHECK("att_bce");
    return TCVOM_OK;
}
extern "C" int tcvom_att_bce_bwd(const float* dlogit_unscaled, const float* acc, const float* gout, float weight,
                                 float* dlogit, int64_t numel, int32_t window, void* stream) {
    TCVOM_CHECK_ARG(dlogit_unscaled && acc && gout && dlogit, "att_bce_bwd: null pointer");
    hipLaunchKernelGGL(att_bce_bwd_kernel, dim3(sgrid(numel)), dim3(256), 0, (hipStream_t)stream, dlogit_unscaled, acc, gout, weight, dlogit, numel, window * window);
    TCVOM_LAUNCH_CHECK("att_bce_bwd");
    return TCVOM_OK;
}
extern "C" int tcvom_loss_finalize(const float* acc, float* out, float weight, int32_t denom_mode, float numel_total,
                                   int32_t window, int32_t accumulate, void* stream) {
    TCVOM_CHECK_ARG(acc && out, "loss_finalize: null pointer");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, out, weight, denom_mode, numel_total, window * window, accumulate);
    TCVOM_LAUNCH_CHECK("loss_finalize");
    return TCVOM_OK;
}
extern "C" int tcvom_adam_mt_guarded(const int64_t* table, const int32_t* work, int32_t nblocks, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, int64_t step, float grad_scale, const int32_t* skip_if_nonzero,
                                     void* stream) {
    TCVOM_CHECK_ARG(table && work && nblocks > 0 && step >= 1, "adam_mt: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, table, work, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale,
                       (const int*)skip_if_nonzero);
    TCVOM_LAUNCH_CHECK("adam_mt");
    return TCVOM_OK;
}
extern "C" int tcvom_adam_mt(const int64_t* table, const int32_t* work, int32_t nblocks, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    return tcvom_adam_mt_guarded(table, work, nblocks, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}
