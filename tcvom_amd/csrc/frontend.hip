// Data front-end of the video matting clips (reference dataset/VMD.py): the loader threads only decode PNGs into uint8
// frames; crop + bilinear resize + rounding (img_crop_and_resize, VMD.py:62-66), the unknown-pixel count of the crop
// search (shape_aug, VMD.py:140-152) and the bottom/right padding of the precomputed validation clips (possible_pad,
// VMD.py:187-200) run here, on the S frames of a clip per launch.  HBM-bound byte work: one pass over the crop window.
#include "common.h"

// dst[s][c][oy][ox] = floor(bilinear(src[s][ph + . ][pw + . ][chan[c]]) + 0.5), align_corners = True, bit for bit what
// F.interpolate gives the reference on the host.  ATen has two CPU kernels for this and dataset/VMD.py reaches both
// (tools/probes/interp_probe.py matched each against candidate orders):
//   form 0 (upsample_generic_Nd_kernel_impl: 1-channel alpha, or 3 channels in a multi-threaded process):
//       row = fma(w0, a, rn(w1 * b));  out = fma(h0, row_top, rn(h1 * row_bottom))
//   form 1 (cpu_upsample_linear_channels_last: 3-channel images in a process with ONE thread, i.e. every DataLoader worker):
//       wXY = rn(hX * wY);  out = fma(w11, d, fma(w10, c, fma(w00, a, rn(w01 * b))))
// with scale = (in - 1) / (out - 1) in fp32, h1 = int(scale * oy), h1l = scale * oy - h1, h0l = 1 - h1l in both.
__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int S, int Hs, int Ws,
                                                             int Cs, int4 chan, int nc, int ph, int pw, int nh, int nw, int Ho, int Wo,
                                                             float rh, float rw, int form) {
#pragma clang fp contract(off)      // no implicit fusing (rh * oy - h1 must round twice, as on the host); fmas are written out.
                                    // (HIP's __fmul_rn / __fsub_rn are header inlines compiled with contraction ON: useless here.)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)Ho * Wo;
    if (i >= per * S) return;
    const int s = (int)(i / per);
    const int r = (int)(i - s * per);
    const int oy = r / Wo, ox = r - oy * Wo;
    const float h1r = rh * (float)oy, w1r = rw * (float)ox;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < nh - 1 ? 1 : 0, w1p = w1 < nw - 1 ? 1 : 0;
    const float h1l = h1r - (float)h1, w1l = w1r - (float)w1;
    const float h0l = 1.f - h1l, w0l = 1.f - w1l;
    const uint8_t* p00 = src + (((int64_t)s * Hs + ph + h1) * Ws + pw + w1) * Cs;
    const uint8_t* p01 = p00 + w1p * Cs;
    const uint8_t* p10 = p00 + (int64_t)h1p * Ws * Cs;
    const uint8_t* p11 = p10 + w1p * Cs;
    const int ch[4] = {chan.x, chan.y, chan.z, chan.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= nc) break;
        const float a = p00[ch[c]], b = p01[ch[c]], cc = p10[ch[c]], d = p11[ch[c]];
        float v;
        if (form == 0) {
            const float top = __builtin_fmaf(w0l, a, w1l * b);
            const float bot = __builtin_fmaf(w0l, cc, w1l * d);
            v = __builtin_fmaf(h0l, top, h1l * bot);
        } else {
            const float w00 = h0l * w0l, w01 = h0l * w1l, w10 = h1l * w0l, w11 = h1l * w1l;
            v = __builtin_fmaf(w11, d, __builtin_fmaf(w10, cc, __builtin_fmaf(w00, a, w01 * b)));
        }
        v = floorf(v + 0.5f);
        dst[((int64_t)s * nc + c) * per + r] = v;
    }
}

extern "C" int tcvom_crop_resize_u8(const void* src, float* dst, int32_t S, int32_t Hs, int32_t Ws, int32_t Cs, const int32_t* chan,
                                    int32_t nc, int32_t ph, int32_t pw, int32_t nh, int32_t nw, int32_t Ho, int32_t Wo, int32_t form,
                                    void* stream) {
    TCVOM_CHECK_ARG(form == 0 || form == 1, "crop_resize_u8: form is 0 (generic) or 1 (channels-last)");
    TCVOM_CHECK_ARG(S > 0 && nc >= 1 && nc <= 4 && Cs >= 1 && Cs <= 4, "crop_resize_u8: 1..4 channels");
    TCVOM_CHECK_ARG(ph >= 0 && pw >= 0 && nh >= 1 && nw >= 1 && ph + nh <= Hs && pw + nw <= Ws, "crop_resize_u8: crop window outside the frame");
    TCVOM_CHECK_ARG(Ho >= 1 && Wo >= 1, "crop_resize_u8: empty output");
    int4 c4 = make_int4(0, 0, 0, 0);
    int* cp = &c4.x;
    for (int c = 0; c < nc; ++c) {
        TCVOM_CHECK_ARG(chan[c] >= 0 && chan[c] < Cs, "crop_resize_u8: channel index");
        cp[c] = chan[c];
    }
    // area_pixel_compute_scale<float>(in, out, align_corners = true): (in - 1) / (out - 1) in fp32, 0 when out == 1
    const float rh = Ho > 1 ? (float)(nh - 1) / (float)(Ho - 1) : 0.f;
    const float rw = Wo > 1 ? (float)(nw - 1) / (float)(Wo - 1) : 0.f;
    const int64_t n = (int64_t)S * Ho * Wo;
    hipLaunchKernelGGL(crop_resize_u8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, dst, S, Hs, Ws, Cs, c4,
                       nc, ph, pw, nh, nw, Ho, Wo, rh, rw, form);
    TCVOM_LAUNCH_CHECK("crop_resize_u8");
    return TCVOM_OK;
}

// counts[s] = #{ 0 < a < 255 } over frame s of a [S][n] float alpha (shape_aug's "does the crop contain unknown pixels" test)
__global__ __launch_bounds__(256) void count_unknown_kernel(const float* __restrict__ a, int64_t n, int32_t* __restrict__ counts) {
    const int s = blockIdx.y;
    const float* p = a + (int64_t)s * n;
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += (p[i] > 0.f && p[i] < 255.f) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(counts + s, c);
}

extern "C" int tcvom_count_unknown(const float* alpha, int32_t S, int64_t n, int32_t* counts, void* stream) {
    TCVOM_CHECK_ARG(S > 0 && n > 0, "count_unknown: empty input");
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * S, (hipStream_t)stream) != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "count_unknown: memset");
    int bx = cdiv(n, 256 * 8);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(count_unknown_kernel, dim3(bx, S), dim3(256), 0, (hipStream_t)stream, alpha, n, counts);
    TCVOM_LAUNCH_CHECK("count_unknown");
    return TCVOM_OK;
}

// dst [S][C][Ho][Wo] = src [S][C][H][W] in the top-left corner, value[c] in the bottom / right border (possible_pad)
__global__ __launch_bounds__(256) void pad_br_kernel(const float* __restrict__ src, float* __restrict__ dst, int SC, int C, int H, int W, int Ho,
                                                     int Wo, float4 value) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)Ho * Wo;
    if (i >= per * SC) return;
    const int sc = (int)(i / per);
    const int r = (int)(i - sc * per);
    const int y = r / Wo, x = r - y * Wo;
    const float v4[4] = {value.x, value.y, value.z, value.w};
    dst[i] = (y < H && x < W) ? src[((int64_t)sc * H + y) * W + x] : v4[sc % C];
}

extern "C" int tcvom_pad_bottom_right(const float* src, float* dst, int32_t S, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                      const float* value, void* stream) {
    TCVOM_CHECK_ARG(S > 0 && C >= 1 && C <= 4 && H >= 1 && W >= 1 && Ho >= H && Wo >= W, "pad_bottom_right: the target must not be smaller than the source");
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* vp = &v.x;
    for (int c = 0; c < C; ++c) vp[c] = value ? value[c] : 0.f;
    const int64_t n = (int64_t)S * C * Ho * Wo;
    hipLaunchKernelGGL(pad_br_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, S * C, C, H, W, Ho, Wo, v);
    TCVOM_LAUNCH_CHECK("pad_bottom_right");
    return TCVOM_OK;
}


// ---------------------------------------------------------------- optical flow: flow_crop_and_resize (dataset/VMD.py:68-126)
// src [F][Hs][Ws][2] fp32 (x, y displacement in pixels, NaN = invalid: the decoded flow files of a sample), crop window
// (ph, pw, nh, nw), output [F][2][Ho][Wo].  Per output pixel: corner-aligned bilinear lookup of the crop (a NaN corner makes
// the result NaN whatever its weight, as in grid_sample), NaN unless the flow is "smooth" at the lookup's top-left source pixel --
// the neighbour to the right and the neighbour below each point within 45 degrees of it (or are shorter than a pixel on
// average / zero) and differ by less than 50 pixels in length; the last column / row of the crop pass -- then the vector is
// rescaled to the new pixel size and dropped if it leaves the new frame.
#pragma clang fp contract(off)
__device__ __forceinline__ bool flow_pair_ok(float ax, float ay, float bx, float by) {
    const float dot = ax * bx + ay * by;
    const float na = sqrtf(ax * ax + ay * ay), nb = sqrtf(bx * bx + by * by);
    const float nab = na * nb;
    float c = fabsf(dot / nab);
    c = c != c ? c : fminf(fmaxf(c, 0.f), 1.0f - 1e-6f);                  // clamp keeps NaN
    bool ok = acosf(c) <= 0.78539816339744830962f;                      // pi / 4 in the reference is a double compared with a float32 angle
    ok = ok || nab == 0.f || (na + nb) < 2.f;
    return ok && fabsf(na - nb) < 50.f;                                   // comparisons with NaN are false: an invalid neighbour fails
}
__global__ __launch_bounds__(256) void flow_crop_resize_kernel(const float2* __restrict__ src, float* __restrict__ dst, int F, int Hs, int Ws,
                                                               int ph, int pw, int nh, int nw, int Ho, int Wo, float sx, float sy,
                                                               float dvx, float dvy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)Ho * Wo;
    if (i >= per * F) return;
    const int f = (int)(i / per);
    const int r = (int)(i - (int64_t)f * per);
    const int oy = r / Wo, ox = r - oy * Wo;
    const float2* fl = src + ((int64_t)f * Hs + ph) * Ws + pw;            // crop origin; element (y, x) at fl[y * Ws + x]
    const float cx = (float)ox * sx, cy = (float)oy * sy;                 // source coordinates inside the crop
    // grid_sample(align_corners=True) sees the coordinate after a round trip through [-1, 1]
    const float gx = 2.f * cx / (float)(nw - 1) - 1.f, gy = 2.f * cy / (float)(nh - 1) - 1.f;
    const float ix = (gx + 1.f) / 2.f * (float)(nw - 1), iy = (gy + 1.f) / 2.f * (float)(nh - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float vx = 0.f, vy = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                        // nw, ne, sw, se in ATen's order
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        if (xx >= 0 && xx < nw && yy >= 0 && yy < nh) {
            const float w = ((k & 1) ? wx1 : wx0) * ((k >> 1) ? wy1 : wy0);
            const float2 v = fl[(int64_t)yy * Ws + xx];
            vx += v.x * w;
            vy += v.y * w;
        }
    }
    // smoothness at the top-left source pixel of the (un-round-tripped) coordinate
    const int qx = (int)floorf(cx), qy = (int)floorf(cy);
    const float2 c0 = fl[(int64_t)qy * Ws + qx];
    bool keep = true;
    if (qx + 1 < nw) { const float2 c1 = fl[(int64_t)qy * Ws + qx + 1]; keep = keep && flow_pair_ok(c0.x, c0.y, c1.x, c1.y); }
    if (qy + 1 < nh) { const float2 c2 = fl[(int64_t)(qy + 1) * Ws + qx]; keep = keep && flow_pair_ok(c0.x, c0.y, c2.x, c2.y); }
    const float qnan = __builtin_nanf("");
    vx = keep ? vx / dvx : qnan;
    vy = keep ? vy / dvy : qnan;
    const float lx = (float)ox + vx, ly = (float)oy + vy;
    const bool outside = lx < 0.f || ly < 0.f || lx > (float)(Wo - 1) || ly > (float)(Ho - 1);
    dst[((int64_t)f * 2 + 0) * per + r] = outside ? qnan : vx;
    dst[((int64_t)f * 2 + 1) * per + r] = outside ? qnan : vy;
}

extern "C" int tcvom_flow_crop_resize(const float* src, float* dst, int32_t F, int32_t Hs, int32_t Ws, int32_t ph, int32_t pw, int32_t nh,
                                      int32_t nw, int32_t Ho, int32_t Wo, void* stream) {
    TCVOM_CHECK_ARG(src && dst && F > 0 && Hs > 0 && Ws > 0, "flow_crop_resize: bad args");
    TCVOM_CHECK_ARG(ph >= 0 && pw >= 0 && nh >= 2 && nw >= 2 && ph + nh <= Hs && pw + nw <= Ws, "flow_crop_resize: crop %d+%d x %d+%d outside a %d x %d frame",
                    ph, nh, pw, nw, Hs, Ws);
    TCVOM_CHECK_ARG(Ho >= 2 && Wo >= 2, "flow_crop_resize: the output needs at least 2 x 2 pixels (corner-aligned sampling)");
    // python doubles rounded to fp32 when they meet a fp32 tensor: (n - 1) / (out - 1) and n / out
    const float sx = (float)((double)(nw - 1) / (double)(Wo - 1)), sy = (float)((double)(nh - 1) / (double)(Ho - 1));
    const float dvx = (float)((double)nw / (double)Wo), dvy = (float)((double)nh / (double)Ho);
    const int64_t n = (int64_t)F * Ho * Wo;
    hipLaunchKernelGGL(flow_crop_resize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)src, dst, F, Hs, Ws, ph, pw,
                       nh, nw, Ho, Wo, sx, sy, dvx, dvy);
    TCVOM_LAUNCH_CHECK("flow_crop_resize");
    return TCVOM_OK;
}
