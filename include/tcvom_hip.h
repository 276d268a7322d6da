/* libtcvom_hip.so — C ABI of the MI355X (gfx950) kernels behind the TCVOM
 * per-frame-window hot path (GCA base + Temporal Attention Module).
 *
 * The reference (yunkezhang/TCVOM) is pure PyTorch and has no FFI of its own;
 * its operator boundary is the Python module API (models/model.py,
 * models/VMN/VMN_model.py, models/GCA/ops.py).  Each entry point below replaces
 * the chain of stock aten/cuDNN ops that the cited reference lines run; the
 * Python mirror of that API (package tcvom_amd, re-exported as models.X) binds
 * them with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned memory (PyTorch's
 *    caching allocator); the library never allocates, frees or synchronises;
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, <0 on error; tcvom_last_error() gives the message
 *    (thread-local); no C++ exception crosses the boundary;
 *  - activations are NHWC bf16 ("bf16" below = uint16 bit pattern), statistics,
 *    losses, logits and master weights are fp32;
 *  - re-entrant, no global mutable state.
 */
#ifndef TCVOM_HIP_H
#define TCVOM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCVOM_OK 0
#define TCVOM_ERR_ARG (-1)
#define TCVOM_ERR_LAUNCH (-2)

const char* tcvom_last_error(void);
int tcvom_abi_version(void);
/* The 16-bit storage type of this build: 0 = bf16 (libtcvom_hip.so), 1 = IEEE fp16 (libtcvom_hip_f16.so; same sources, same
 * ABI).  Wherever this header says "bf16" for an activation or a packed weight, read "the build's 16-bit type". */
int tcvom_act_dtype(void);

/* ------------------------------------------------------------------ implicit-GEMM convolution
 * One "phase" of a (possibly transposed) convolution on bf16 MFMA:
 *   out[n, i*out_step+out_off_h, j*out_step+out_off_w, k] =
 *        act( bias[k] + sum_{t<ntaps} sum_{c<C} in[n, i*in_step+dh[t], j*in_step+dw[t], c] * w[k][wslot[t]][c] )
 * for i<PH, j<PW (taps falling outside the input read zero).  Replaces
 * nn.Conv2d / nn.ConvTranspose2d calls of models/GCA/encoders/resnet_enc.py:33-49,
 * decoders/resnet_dec.py:33-59, res_gca_enc.py:20-55, VMN_model.py:13-15,63-66 and
 * (with ntaps==1) the dense GEMMs of GuidedCxtAtten (models/GCA/ops.py:177,204).
 * Optional fused epilogue: per-output-channel scale, "diagonal" subtraction
 * (GCA self-mask, ops.py:188), bias, ReLU, BatchNorm partial statistics
 * (sum, sum of squares per channel per 64-pixel group).                       */
#define TCVOM_MAX_TAPS 32
typedef struct {
    int32_t N, H, W, C;          /* input NHWC; C a multiple of 8 */
    int32_t OH, OW, K;           /* output tensor dims; K = output channels (multiple of 4) */
    int32_t PH, PW;              /* phase grid */
    int32_t in_step, out_step, out_off_h, out_off_w;
    int32_t ntaps;               /* the reduction runs over ntaps*C elements in 64-deep steps; a partial last step ends in zero-tap
                                    slots past the list: ((ntaps*C + 63) / 64 * 64 - 1) / C must stay below TCVOM_MAX_TAPS */
    int32_t tap_dh[TCVOM_MAX_TAPS], tap_dw[TCVOM_MAX_TAPS];
    int32_t tap_w[TCVOM_MAX_TAPS];   /* weight slot of tap t, -1 = zero tap (padding) */
    int32_t wt;                  /* weight slots: w is [K][wt][C] bf16 */
    int32_t ldo;                 /* output pixel stride in elements */
    int32_t act;                 /* 0 none, 1 ReLU (before store and statistics) */
    int32_t out_fp32;            /* 1: store fp32 instead of the 16-bit storage type (the dense GEMMs of GCA); 2: IEEE fp16 whatever the
                                    build stores (with in_f16 = 1: the fp16 island of the bf16 build) */
    int32_t stats_group_offset;  /* first statistics group written by this launch */
    int32_t batch;               /* >1: `batch` independent problems with the strides below: the dense GEMMs of GCA,
                                    and the S frames of a window through one conv layer (N samples per frame, own
                                    SpectralNorm'd weight copy per frame) */
    int32_t w_layout;            /* 0: w is [K][wt][C].  1: "fragment-major" (stride-1 3x3 layers with C == K in {64, 128}
                                    only, served by the weight-stationary kernel): element (k, slot, c) at
                                    ((((k/32)*wt + slot)*(C/16) + c/16)*64 + ((c/8)&1)*32 + k%32)*8 + c%8, i.e. every
                                    32 x 16 MFMA A fragment is one contiguous 1 KiB block in lane order */
    int32_t in_f16;              /* 1 (bf16 build only; the fp16 build IS that format and refuses the flag): `in` and `w` hold IEEE fp16
                                    and the product runs on v_mfma_f32_32x32x16_f16 -- the forward convs of the encoder stem, layer1 and
                                    layer2 ("fp16 island", tcvom_amd/gca_net.py); comes with out_fp32 == 2 and only with it.  Served by
                                    the halo-tile kernel, the weight-stationary kernel and the implicit GEMM */
    int64_t in_bstride, w_bstride, out_bstride, vec_bstride;
    int64_t stats_bstride;       /* statistics groups between batch elements (frame f writes groups
                                    stats_group_offset + f*stats_bstride + ...) */
} tcvom_conv_desc;

/* stats_partial: [groups][2][K] fp32 or NULL; mscale/mdiag/bias: [K] fp32 or NULL. */
int tcvom_conv_igemm(const void* in, const void* w, void* out, const float* bias,
                     const float* mscale, const float* mdiag, float* stats_partial,
                     const tcvom_conv_desc* d, void* stream);
/* The same for up to 4 phases (sub-pixel phases of a ConvTranspose2d / stride-2 data gradient) in ONE launch;
 * descs[i].stats_group_offset must be i * tcvom_conv_stats_groups(descs, nphase). */
int tcvom_conv_igemm_phases(const void* in, const void* w, void* out, const float* bias, float* stats_partial,
                            const tcvom_conv_desc* descs, int32_t nphase, void* stream);
/* Two dense products against ONE weight operand (out1 = in1 x w^T, out2 = in2 x w^T, both described by `desc`: ntaps = 1;
 * in2's batch stride is in2_bstride instead of desc->in_bstride) in one launch where the 256-tile GEMM takes the shape, else as two launches.  Replaces the two `torch.matmul`s autograd
 * runs for d(query) and d(key) of models/GCA/ops.py:177's F.conv2d (same correlation weights, two gradient operands). */
int tcvom_gemm_pair(const void* in1, const void* in2, const void* w, void* out1, void* out2,
                    const tcvom_conv_desc* desc, int64_t in2_bstride, void* stream);
/* name of the kernel instantiation a launch with these descriptors selects (bench / profile labels) */
const char* tcvom_conv_igemm_variant(const tcvom_conv_desc* d, int32_t nphase);
const char* tcvom_wgrad_igemm_variant(const tcvom_conv_desc* d);
/* number of statistics groups ONE phase of a launch of `nphase` phases like `d` writes */
int tcvom_conv_stats_groups(const tcvom_conv_desc* d, int32_t nphase);
/* profiling aid: with the environment variable TCVOM_CONV_TRACE set, the weight-stationary conv kernel records up to 64
 * shader-clock stamps of one workgroup of its latest launch (entry, halo issued, weights loaded, end, then per tile:
 * barrier passed, next halo issued, MFMAs done, halo landed); this copies the first n to the HOST array (synchronises). */
int tcvom_conv_trace_read(uint64_t* host, int32_t n);

/* Weight gradient of the same phase (reduction over pixels, both operands pixel-major):
 *   dw[k][wslot[t]][c] += sum_{n,i,j} dy[n, out pixel(i,j), k] * in[n, i*in_step+dh[t], j*in_step+dw[t], c]
 * dw is fp32 [K][wt][C], accumulated with atomics (caller zeroes it).  With
 * ntaps==1 this is the dense "both operands k-major" GEMM used by the GCA backward. */
int tcvom_wgrad_igemm(const void* dy, const void* in, float* dw, const tcvom_conv_desc* d,
                      int32_t ldy, void* stream);
int tcvom_wgrad_igemm_phases(const void* dy, const void* in, float* dw, const tcvom_conv_desc* descs,
                             int32_t nphase, int32_t ldy, void* stream);
/* the same for `nbatch` (1..8) problems of identical shape in ONE launch -- the S calls of one layer in a window
 * (same descriptors; dy / in / dw are HOST arrays of nbatch device pointers) */
int tcvom_wgrad_igemm_batched(const void* const* dy, const void* const* in, float* const* dw, int32_t nbatch,
                              const tcvom_conv_desc* descs, int32_t nphase, int32_t ldy, void* stream);
/* Problems of DIFFERENT descriptors in ONE launch of the same kernel (the weight gradients of the 1 x 1, stride-2, transposed and
 * small-channel convs of resnet_enc.py:76-111 / resnet_dec.py:33-72, which stay on the implicit-GEMM TT kernel: each a 20 - 100 us launch
 * of a few dozen workgroups on its own).  All problems of a launch share the tile (tm, tn) that tcvom_wgrad_igemm_variant names for them.
 *   tcvom_wgrad_igemm_hetero_plan: HOST planning, once per set of problems.  descs = the phase descriptors of all problems, concatenated
 *     (nphase[i] of them for problem i: that concatenation is also the DEVICE descriptor table the launch reads); ldy[i] = pixel stride of
 *     dy.  Writes 8 int32 per work item to `work` (HOST, capacity max_items items; NULL: count only) and returns the item count.
 *   tcvom_wgrad_igemm_hetero: the launch.  dys / ins / dws: HOST arrays of nprob DEVICE pointers (they may change from step to step);
 *     desc_table / work: DEVICE copies of `descs` / the planned work items (16-byte aligned, caller-owned).  dw accumulates (atomics). */
int tcvom_wgrad_igemm_hetero_plan(const tcvom_conv_desc* descs, const int32_t* nphase, const int32_t* ldy, int32_t nprob,
                                  int32_t tm, int32_t tn, int32_t* work, int32_t max_items);
int tcvom_wgrad_igemm_hetero(const void* const* dys, const void* const* ins, float* const* dws, int32_t nprob,
                             const tcvom_conv_desc* desc_table, const int32_t* work, int32_t nwork, int32_t tm, int32_t tn, void* stream);
int32_t tcvom_wgrad_igemm_hetero_max_problems(void);
/* Accumulator-stationary weight gradient (csrc/wgradws.hip) of `nprob` (1..tcvom_wgrad_ws_max_problems()) stride-1 3x3
 * convolutions of ONE geometry `d` (C a multiple of 64, K = 32 or a multiple of 64; K == ldy): the backward of the conv
 * layers of resnet_enc.py:33-49 / resnet_dec.py:43-59 for ALL layers and calls of that shape in a window -- with many
 * problems per launch a (problem, 64 k x 128 c x 9 tap) block of dw is owned by one or two workgroups and the atomic
 * partial sums shrink to ~1x the size of dw.  dy / in / dw are HOST arrays of nprob device pointers; dw accumulates.
 * tcvom_wgrad_igemm_batched routes eligible shapes here too (<= 8 problems).  TCVOM_ERR_ARG for other shapes. */
int tcvom_wgrad_ws_multi(const void* const* dy, const void* const* in, float* const* dw, int32_t nprob,
                         const tcvom_conv_desc* d, int32_t ldy, void* stream);
int32_t tcvom_wgrad_ws_max_problems(void);
/* The same for problems of up to tcvom_wgrad_ws_max_geometries() DIFFERENT geometries in one launch: problem i is a convolution
 * like descs[geo_index[i]] (dy pixel stride = its K).  All geometries must share the channel window of the kernel (C a multiple of
 * 128 for all of them, or for none) and the tap -> weight-slot map.  The atomic flush of the accumulators costs a launch the same
 * ~45 us however few problems it has, so the layers whose shape occurs only 1 - 9 times in a window (the channel-changing convs of
 * resnet_enc.py:76-84 / resnet_dec.py:61-72) ride in the launch of the big groups.  TCVOM_ERR_ARG for other shapes. */
int tcvom_wgrad_ws_hetero(const void* const* dy, const void* const* in, float* const* dw, int32_t nprob,
                          const tcvom_conv_desc* descs, int32_t ngeo, const int32_t* geo_index, void* stream);
int32_t tcvom_wgrad_ws_max_geometries(void);

/* ------------------------------------------------------------------ depthwise 3x3 (IndexNet / MobileNetV2 blocks)
 * Replaces nn.Conv2d(C, C, 3, 1, padding, dilation, groups=C, bias=False) of models/Index/net.py:38-61 (InvertedResidual, run
 * unpadded on the `fixed_padding` input) and models/Index/hlaspp.py:38-46 (ASPP branches, padding = dilation).
 * x [nframes][N][H][W][C] bf16 NHWC, w fp32 [9][C] tap-major (16-byte aligned), y [nframes][N][OH][OW][C] bf16 with
 * OH = H + 2 pad - 2 dilation.  stats (or NULL): [nframes][tcvom_dw3x3_stats_groups(N*OH*OW, C)][2][C] fp32 partial (sum, sum of
 * squares) of y per channel for the BatchNorm that follows (the layout tcvom_bn_finalize consumes).  flip != 0: taps reversed
 * (the data gradient: tcvom_dw3x3(dy, w, dx, NULL, N, OH, OW, C, dilation, 2 * dilation - pad, 1, ...)).
 * tcvom_dw3x3_wgrad: dw [9][C] fp32 = sum over all frames and pixels of dy * shifted x (overwritten). */
int tcvom_dw3x3_stats_groups(int64_t out_pixels, int32_t C);
int tcvom_dw3x3(const void* x, const float* w, void* y, float* stats, int32_t N, int32_t H, int32_t W, int32_t C,
                int32_t dilation, int32_t pad, int32_t flip, int32_t nframes, void* stream);
int tcvom_dw3x3_wgrad(const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t C,
                      int32_t dilation, int32_t pad, int32_t nframes, void* stream);

/* ------------------------------------------------------------------ IndexNet element-wise stages (csrc/indexnet.hip)
 * tcvom_index_pool_*: DepthwiseM2OIndexBlock's sigmoid / softmax-over-the-four-branches / pixel shuffle fused with the encoder's
 * indexed pooling (models/Index/hlindex.py:149-168, models/Index/net.py:203-205).  x1..x4 [N][h2][w2][C] bf16 (branch k = sub-pixel
 * (k / 2, k % 2) of a 2x2 cell), l [N][2 h2][2 w2][C] -> xe = idx_en * l and idx_de = sigmoid(x) at full resolution, pooled =
 * 4 * avg_pool2d(xe) at half resolution.  Backward: gradients of (xe, pooled, idx_de) (each may be NULL = zero) -> dx1..dx4, dl.
 * tcvom_index_up_*: IndexedUpsamlping's conv input (models/Index/hldecoder.py:128-133): out [N][H][W][C1 + C2] =
 * concat(idx * nearest_x2(enc), low) with enc [N][H/2][W/2][C1], idx [N][H][W][C1]; idx == NULL: enc is [N][H][W][C1] (plain concat). */
int tcvom_index_pool_fwd(const void* x1, const void* x2, const void* x3, const void* x4, const void* l, void* xe, void* pooled,
                         void* idx_de, int32_t N, int32_t h2, int32_t w2, int32_t C, void* stream);
int tcvom_index_pool_bwd(const void* x1, const void* x2, const void* x3, const void* x4, const void* l, const void* dxe,
                         const void* dpooled, const void* dde, void* dx1, void* dx2, void* dx3, void* dx4, void* dl,
                         int32_t N, int32_t h2, int32_t w2, int32_t C, void* stream);
int tcvom_index_up_fwd(const void* enc, const void* idx, const void* low, void* out, int32_t N, int32_t H, int32_t W,
                       int32_t C1, int32_t C2, void* stream);
int tcvom_index_up_bwd(const void* dout, const void* enc, const void* idx, void* denc, void* didx, void* dlow, int32_t N,
                       int32_t H, int32_t W, int32_t C1, int32_t C2, void* stream);
/* pred[1] of the IndexNet decoder: nn.Conv2d(1, 1, 5, padding=2, bias=False) (models/Index/net.py:21) on a fp32 [N][H][W] map,
 * w fp32 [25]; flip != 0: reversed taps (data gradient); _wgrad: dw[25] = sum dy * shifted x (overwritten). */
int tcvom_conv5x5_c1(const float* x, const float* w, float* y, int32_t N, int32_t H, int32_t W, int32_t flip, void* stream);
int tcvom_conv5x5_c1_wgrad(const float* dy, const float* x, float* dw, int32_t N, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------ BatchNorm around the convs
 * Replaces nn.BatchNorm2d (+ReLU / LeakyReLU(0.2) / residual add) of the BasicBlocks
 * (models/GCA/encoders/resnet_enc.py:33-49, decoders/resnet_dec.py:43-59).
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 LeakyReLU(0.01), 4 ReLU6.   z = act(y*scale + shift + res1) + res2   */
/* unbias_count: element count used for the unbiased running_var correction (0 = count); differs from
 * count when the statistics were taken before a nearest x2 up-sampling (resnet_dec.py:112-118) */
int tcvom_bn_finalize(const float* stats_partial, int32_t groups, int32_t C, int64_t count, int64_t unbias_count,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* scale_shift /*[2][C]*/, float* saved /*[2][C] mean,invstd*/,
                      double* scratch /* nframes * tcvom_bn_finalize_scratch_doubles(C) doubles, or NULL */,
                      int32_t nframes, int64_t slot_stride, void* stream);
/* nframes / slot_stride (above and below): a frame-batched call normalises `nframes` groups of `pixels` pixels each
 * (the S frames of a window in one launch; statistics stay per frame, as in the reference's per-frame encoder calls).
 * partial is then [nframes][groups][2][C], and frame f uses the vectors at scale_shift + f*slot_stride,
 * saved + f*slot_stride and coef + f*3*C.  nframes = 1, slot_stride = 0 is the plain call. */
int tcvom_bn_finalize_scratch_doubles(int32_t C);
/* running_mean/var EMA from the (mean, invstd) a train-mode tcvom_bn_finalize(running_mean=NULL) call saved */
int tcvom_bn_ema_update(const float* saved, float* running_mean, float* running_var, int32_t C, float momentum,
                        float eps, int64_t unbias_count, void* stream);
/* every deferred EMA of a window in one launch.  table: nbn rows of 6 int64 {running_mean ptr, running_var ptr,
 * address of call slot 0's saved (mean, invstd), C, floats between call slots, momentum | eps << 32 (fp32 bits)};
 * masks[b] bit s = call slot s of BatchNorm b was a train-mode call; unbias[b] = n / (n - 1).  masks / unbias are
 * HOST arrays (they travel as kernel arguments). */
int tcvom_bn_ema_multi(const int64_t* table, int32_t nbn, const uint32_t* masks, const float* unbias, void* stream);
int tcvom_bn_eval_coeffs(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale_shift, float* saved, void* stream);
/* y_fp32: 0 = the conv output y has the build's 16-bit storage type, 2 = IEEE fp16 whatever the build stores (the fp16 island of the
 * bf16 build) */
int tcvom_bn_apply(const void* y, const float* scale_shift, const void* res1, const void* res2, void* z,
                   int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes, int64_t slot_stride,
                   void* stream);
/* ... and `mask`: one byte per 8-channel vector ([nframes * pixels * C / 8]), bit k = the pre-activation value
 * norm(y) + res1 of channel k is positive.  The backward of a residual site (z = act(norm(y) + res1): the `out += identity` of
 * models/GCA/encoders/resnet_enc.py:33-49 and models/FBA/resnet_GN_WS.py:112-137) takes the activation's slope from these bits
 * (tcvom_bn_bwd_reduce_mask / tcvom_bn_bwd_apply_mask) instead of re-reading res1 in both passes.  act: ReLU / LeakyReLU / none. */
int tcvom_bn_apply_mask(const void* y, const float* scale_shift, const void* res1, const void* res2, void* z, uint8_t* mask,
                        int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes, int64_t slot_stride,
                        void* stream);
/* The apply pass inside the "fp16 island" of the bf16 build (the encoder stem, layer1 and layer2 of models/GCA/encoders/resnet_enc.py:70-84,
 * whose forward runs on IEEE fp16 operands: tcvom_conv_desc.in_f16): y is IEEE fp16 (y_fp32 must be 2); z is stored twice -- in the
 * build's type to `z` (what the backward, the weight gradient and every consumer outside the island read) and as IEEE fp16 to `z16`
 * (what the island's next forward conv and the residual input of its block read); res1 is IEEE fp16 when res1_f16 != 0;
 * mask as in tcvom_bn_apply_mask, or NULL. */
int tcvom_bn_apply_f16(const void* y, const float* scale_shift, const void* res1, int32_t res1_f16, const void* res2, void* z,
                       void* z16, uint8_t* mask, int64_t pixels, int32_t C, int32_t act, int32_t y_fp32, int32_t nframes,
                       int64_t slot_stride, void* stream);
int tcvom_bn_bwd_groups(int64_t pixels, int32_t C);                       /* = tcvom_bn_bwd_groups_n(pixels, C, 1) */
/* partial-sum groups per frame of a backward reduction over `nframes` frames (what tcvom_bn_bwd_reduce launches and writes) */
int tcvom_bn_bwd_groups_n(int64_t pixels, int32_t C, int32_t nframes);
/* dz2 (or NULL): a second addend of the incoming gradient, bf16 like dz -- the skip-branch gradient of a residual block
 * (`out += identity`, resnet_enc.py:45-47: the block input feeds conv1 AND the residual add), summed in fp32 on the fly
 * instead of by a separate element-wise pass */
int tcvom_bn_bwd_reduce(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                        const float* saved, float* partial /*[nframes][groups][2][C]*/, int64_t pixels, int32_t C,
                        int32_t act, int32_t y_fp32, int32_t nframes, int64_t slot_stride, void* stream);
/* _ranged: dz2 holds the frames dz2_f0 .. dz2_f1 - 1 of the batched call only ([dz2_f1 - dz2_f0][pixels][C], zero elsewhere): the
 * gradient a consumer that ran for the interior frames of a window (VMN_model.py:107-110) deposits with its producer */
int tcvom_bn_bwd_reduce_ranged(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                               const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act, int32_t y_fp32,
                               int32_t nframes, int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream);
int tcvom_bn_bwd_reduce_mask(const void* dz, const void* dz2, const void* y, const uint8_t* mask, const float* scale_shift,
                             const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act, int32_t y_fp32,
                             int32_t nframes, int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream);
/* three addends: dz (all frames) + dz2 and dz3, each holding its own frame range (NULL = absent): the outputs of the encoder stages
 * (resnet_enc.py:130-138 layer1 .. layer3, res_gca_enc.py:47-55 shortcuts) have three consumers -- the next stage's conv1, its
 * down-sampling branch and the shortcut branch.  res1 XOR mask (both may be NULL). */
int tcvom_bn_bwd_reduce3(const void* dz, const void* dz2, int32_t dz2_f0, int32_t dz2_f1, const void* dz3, int32_t dz3_f0,
                         int32_t dz3_f1, const void* y, const void* res1, const uint8_t* mask, const float* scale_shift,
                         const float* saved, float* partial, int64_t pixels, int32_t C, int32_t act, int32_t y_fp32,
                         int32_t nframes, int64_t slot_stride, void* stream);
/* Optional by-product of the BatchNorm backward of a SpectralNorm'd conv (models/GCA/ops.py:25-45: weight = weight_bar / sigma,
 * differentiated by autograd): SpectralNorm's backward needs <dW~, weight_bar> = sigma <dW~, W~>, and <dW~, W~> = d(loss)/d(alpha) of
 * y = conv(x, alpha W~) = <dy, y> -- which the two BatchNorm-backward sums determine per channel (training statistics:
 * gamma invstd^2 eps sum(g xhat); running statistics: gamma invstd (mean sum(g) + sum(g xhat) / invstd)).  The finalize kernels ADD
 * scale * sum_c <dy, y>_c of frame f to out[f * frame_stride]; tcvom_sn_backward multiplies by sigma.  Replaces a pass over the
 * weight gradients of these layers.  Bias-free convs only (y must be conv(x, W~) itself). */
typedef struct {
    float* out;               /* NULL: off */
    int64_t frame_stride;     /* elements between the slots of consecutive frames (0: all frames share one slot) */
    float eps;                /* the BatchNorm's eps */
    int32_t training;         /* != 0: batch statistics were used in the forward */
    float scale;              /* 1, or 1 / world under SyncBatchNorm (every rank holds the GLOBAL sums; the gradient all-reduce averages) */
} tcvom_sn_dot;
/* dgamma/dbeta are written, or accumulated when `accumulate` is set; coef is [3][C] scratch consumed by bn_bwd_apply */
int tcvom_bn_bwd_finalize(const float* partial, int32_t groups, int32_t C, int64_t count, const float* gamma,
                          const float* saved, float* dgamma, float* dbeta, float* coef /*[nframes][3][C]*/,
                          double* scratch /* nframes * tcvom_bn_finalize_scratch_doubles(C) doubles, or NULL */,
                          int32_t accumulate /* != 0: atomically ADD into dgamma/dbeta (required when nframes > 1) */,
                          int32_t nframes, int64_t slot_stride, const tcvom_sn_dot* dot /* or NULL */, void* stream);
/* GroupNorm (FBA base, models/FBA/layers_WS.py:26-27, nn.GroupNorm(32, C)) on the same partial sums: one sample per
 * "frame"; count = pixels of one sample.  The (scale, shift) / (mean, invstd) / coef vectors it writes drive
 * tcvom_bn_apply / tcvom_bn_bwd_reduce / tcvom_bn_bwd_apply unchanged.  dgamma / dbeta are ADDED to (atomics). */
int tcvom_gn_finalize(const float* stats_partial, int32_t groups, int32_t C, int64_t count, int32_t num_groups,
                      const float* gamma, const float* beta, float eps, float* scale_shift, float* saved,
                      double* scratch, int32_t nframes, int64_t slot_stride, void* stream);
int tcvom_gn_bwd_finalize(const float* partial, int32_t groups, int32_t C, int64_t count, int32_t num_groups,
                          const float* gamma, const float* saved, float* dgamma, float* dbeta, float* coef,
                          double* scratch, int32_t nframes, int64_t slot_stride, void* stream);
/* SyncBatchNorm (train_ddp.py:213 nn.SyncBatchNorm.convert_sync_batchnorm): the per-channel sums are produced
 * as an fp64 [nframes][2][C] vector (the frames of a frame-batched call are separate calls of the BatchNorm and keep
 * separate statistics), the host all-reduces it over the ranks in ONE collective (RCCL), and the *_sums finalizers
 * consume the summed vector with the global pixel count of one frame.  scale_shift / saved of frame f sit
 * f * slot_stride floats further, coef [nframes][3][C].  dgamma/dbeta come from the LOCAL sums (torch semantics),
 * summed over the frames. */
int tcvom_bn_reduce_sums(const float* partial, int32_t groups, int32_t C, double* sums /*[nframes][2][C]*/,
                         double* scratch /* nframes * tcvom_bn_finalize_scratch_doubles(C) doubles, or NULL */,
                         int32_t nframes, void* stream);
int tcvom_bn_finalize_sums(const double* sums, int32_t C, int64_t count, int64_t unbias_count, const float* gamma,
                           const float* beta, float eps, float* scale_shift, float* saved, int32_t nframes,
                           int64_t slot_stride, void* stream);
int tcvom_bn_bwd_finalize_sums(const double* sums_all, const double* sums_local, int32_t C, int64_t count,
                               const float* gamma, const float* saved, float* dgamma, float* dbeta, float* coef,
                               int32_t accumulate, int32_t nframes, int64_t slot_stride, const tcvom_sn_dot* dot, void* stream);
/* SyncBatchNorm WITHOUT a collective call (replaces the all_gather / all_reduce pairs nn.SyncBatchNorm issues per BatchNorm call
 * under train_ddp.py:271-280): every rank owns a mailbox of uncached device memory mapped into its peers
 * through hipIpc; the finalize kernel pushes its local fp64 sums into every peer's mailbox over xGMI as self-validating 8-byte
 * granules {32 data bits, 32-bit tag = seq}, polls its own mailbox for the `world` contributions and adds them in rank order
 * (bit-identical on all ranks).  Same launches as tcvom_bn_finalize / tcvom_bn_bwd_finalize; `count` is the pixel count of one
 * frame over ALL ranks.  Calls of one mailbox must be enqueued on ONE stream, in the same order and with consecutive `seq`
 * values on every rank.  A peer that never arrives does not hang the device: after timeout_ticks the kernel stores seq into
 * *status and carries on with partial sums (the host raises).
 * mailbox bytes = ring * world * capacity * 16. */
typedef struct {
    const void* peers;        /* DEVICE array of `world` pointers: base of rank r's mailbox as mapped in this process */
    int32_t world, rank;      /* world <= 32 */
    uint32_t seq;             /* tag of this exchange, != 0 */
    int32_t ring;             /* ring slots (>= 2); slot = seq % ring */
    int64_t capacity;         /* doubles per (slot, sender): nframes * 2 * C must fit */
    int64_t timeout_ticks;    /* 100 MHz ticks a poll may spin (0: 30 s) */
    int32_t* status;          /* device-visible word (pinned host memory), or NULL */
    int64_t* wait_ticks;      /* DEVICE int64[2] or NULL: [0] = longest pull (100 MHz ticks) seen so far (atomic max), [1] = sum of
                               * the pulls, one sample per (exchange, workgroup): how long this rank waited for its slowest peer */
} tcvom_bn_sync;
int tcvom_bn_finalize_sync(const float* stats_partial, int32_t groups, int32_t C, int64_t count, int64_t unbias_count,
                           const float* gamma, const float* beta, float eps, float* scale_shift, float* saved,
                           double* scratch, int32_t nframes, int64_t slot_stride, const tcvom_bn_sync* sync, void* stream);
int tcvom_bn_bwd_finalize_sync(const float* partial, int32_t groups, int32_t C, int64_t count,
                               const float* gamma, const float* saved, float* dgamma, float* dbeta, float* coef,
                               double* scratch, int32_t accumulate, int32_t nframes, int64_t slot_stride,
                               const tcvom_bn_sync* sync, const tcvom_sn_dot* dot, void* stream);
/* The mailbox memory: the only entry points that own memory (uncached + hipIpc-exportable, which the caller's allocator cannot
 * provide).  alloc: zero-filled, *handle64 = the 64-byte hipIpc handle to hand to the peers (all zero if the runtime cannot
 * export: a one-rank mailbox still works).  open / close: map / unmap a peer's mailbox.  These four synchronise the device. */
int tcvom_mbox_alloc(int64_t bytes, void** ptr, void* handle64);
int tcvom_mbox_open(const void* handle64, void** ptr);
int tcvom_mbox_close(void* ptr);
/* *device = the HIP device ordinal (in this process) the memory behind `ptr` lives on: a mapped peer mailbox that reports another
 * device than the caller's own shows that hipIpcOpenMemHandle crossed devices (diagnostic of bench.py --gpus N) */
int tcvom_mbox_device(const void* ptr, int32_t* device);
int tcvom_mbox_free(void* ptr);
/* in_relu != 0: y is the output of a fused ReLU (conv->ReLU->BN order, res_gca_enc.py:47-55) and the
 * gradient is additionally masked by y > 0 */
int tcvom_bn_bwd_apply(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                       const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                       int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                       int64_t slot_stride, void* stream);
int tcvom_bn_bwd_apply_ranged(const void* dz, const void* dz2, const void* y, const void* res1, const float* scale_shift,
                              const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                              int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                              int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream);
int tcvom_bn_bwd_apply_mask(const void* dz, const void* dz2, const void* y, const uint8_t* mask, const float* scale_shift,
                            const float* saved, const float* coef, void* dy, void* dres1, int64_t pixels,
                            int32_t C, int32_t act, int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes,
                            int64_t slot_stride, int32_t dz2_f0, int32_t dz2_f1, void* stream);
int tcvom_bn_bwd_apply3(const void* dz, const void* dz2, int32_t dz2_f0, int32_t dz2_f1, const void* dz3, int32_t dz3_f0,
                        int32_t dz3_f1, const void* y, const void* res1, const uint8_t* mask, const float* scale_shift,
                        const float* saved, const float* coef, void* dy, void* dres1,
                        void* dsum /* or NULL: dz + dz2 + dz3 itself, 16-bit (whole-tensor addends): the gradient of a residual added AFTER the activation */,
                        int64_t pixels, int32_t C, int32_t act,
                        int32_t training, int32_t in_relu, int32_t y_fp32, int32_t nframes, int64_t slot_stride, void* stream);

/* ------------------------------------------------------------------ batched SpectralNorm + weight packing
 * Replaces SpectralNorm._update_u_v/_noupdate_u_v (models/GCA/ops.py:25-45,74-80) for every wrapped
 * conv of the network at once; see tcvom_amd/csrc/spectral.hip for the table layout (24 int64 words
 * per layer) and tcvom_amd/weights.py for the host side that builds it.                      */
typedef struct {
    float* tvec;   /* [sum_wd]: ZERO before the first training call (every training call leaves it zero again) */
    float* svec;   /* [sum_h]  */
    float* sigma;  /* [max_calls][num_layers] */
    float* uhist;  /* [max_calls][sum_h]  */
    float* vhist;  /* [max_calls][sum_wd] */
    int64_t sum_h, sum_wd;
    int32_t num_layers;
} tcvom_sn_scratch;
int tcvom_sn_power_iteration(const int64_t* table, const tcvom_sn_scratch* s,
                             const int32_t* work_wtu, int32_t n_wtu, const int32_t* work_wv, int32_t n_wv,
                             const int32_t* sn_layers, int32_t n_sn, int32_t call, int32_t training, void* stream);
/* call >= 0: every work entry (layer, which, block) packs that power-iteration call; call < 0: the entry carries its own call in
 * bits 8.. of `which` (all calls of a window in one launch). */
int tcvom_sn_pack(const int64_t* table, const tcvom_sn_scratch* s, const int32_t* work_pack, int32_t n_pack,
                  int32_t call, void* fwd_arena, void* bwd_arena, int64_t fwd_call_stride,
                  int64_t bwd_call_stride, void* stream);
/* work_apply rows of one layer: (layer, block) for block < tcvom_sn_apply_blocks(kind, K, C, T, numel) */
int tcvom_sn_apply_blocks(int32_t kind, int32_t K, int32_t C, int32_t T, int64_t numel);
int tcvom_sn_backward(const int64_t* table, const tcvom_sn_scratch* s,
                      const int32_t* work_inner, int32_t n_inner, const int32_t* work_apply, int32_t n_apply,
                      const int32_t* ncalls, const float* dw_arena, int64_t dw_call_stride,
                      float* inner, int32_t max_calls, float* grad_arena, float out_scale /* multiplies every written gradient: 1 / loss scale of the fp16 build, else 1 */,
                      const float* dots /* [max_calls][layers]: <dy, y> of the layers flagged in dot_layers (tcvom_sn_dot), or NULL */,
                      const int32_t* dot_layers /* [layers]: != 0 -> <dW~, W_bar> = sigma * dots (no work_inner rows for that layer) */,
                      void* stream);

/* GuidedCxtAtten backward (models/GCA/ops.py:177-204 under autograd), the products that contract the ROW index of the N x N
 * matrices P (probabilities) and T (score gradient), which are read as they lie in memory -- k-major operand through the
 * transposing LDS read -- instead of transposed copies:
 *   tcvom_gca_dv:    dV[b][j][v]  = sum_{i < N} P[b][i][j] dO[b][i][v]           P: [batch][N][ld], dO: [batch][N][DV] (both k-major)
 *   tcvom_gca_dq_dk: dWq[b][i][d] = sum_j T[b][i][j] Gt[b][d][j],  Mp[b][j][d] = sum_i T[b][i][j] Gt[b][d][i]    T: [batch][N][ld]
 * and the forward value aggregation with V as it lies in memory (k-major A operand):
 *   tcvom_gca_pv:    O[b][i][v]   = sum_{j < N} P[b][i][j] V[b][j][v]            V: [batch][N][DV]
 * fp32 outputs [batch][N][DV] / [batch][N][D]; ld % 256 == 0 (pv: % 64), zeros in the padding columns N <= j < ld of P / T. */
int tcvom_gca_dv(const void* P, const void* dO, float* dV, int32_t N, int32_t DV, int64_t ld, int32_t batch, void* stream);
int tcvom_gca_pv(const void* P, const void* V, float* O, int32_t N, int32_t DV, int64_t ld, int32_t batch, void* stream);
int tcvom_gca_dq_dk(const void* T, const void* Gt, float* dWq, float* Mp, int32_t N, int32_t D, int64_t ld, int32_t batch, void* stream);

/* ------------------------------------------------------------------ layout / resampling helpers (NHWC bf16) */
int tcvom_avgpool2(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* the AvgPool2d(2, 2) of encoder layer2's downsample branch (resnet_enc.py:107-111) inside the fp16 island of the bf16 build: x16 is
 * IEEE fp16; the result is stored in the build's type (y) and as IEEE fp16 (y16) */
int tcvom_avgpool2_f16(const void* x16, void* y, void* y16, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_upsample2(const void* y, void* x, int32_t N, int32_t H, int32_t W, int32_t C, float scale, void* stream);
int tcvom_sumpool2(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, float scale, void* stream);
int tcvom_reflect_pad1(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_reflect_pad1_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_add(const void* a, const void* b, const void* c, void* z, int64_t numel, void* stream);
int tcvom_colsum(const void* dy, float* out, int64_t P, int32_t K, int32_t ld, void* stream);
int tcvom_transpose_bf16(const void* in, void* out, int32_t R, int32_t Cc, int64_t ldi, int64_t ldo,
                         int32_t batch, int64_t in_bstride, int64_t out_bstride, void* stream);
/* final conv C -> 1 with the output map fused: ksize 3 / mode 0 = (tanh + 1) / 2 (GCA decoder, resnet_dec.py:139-141),
 * ksize 5 / mode 1 = clamp(0, 1) (DIM alpha_pred, models/DIM/vggnet.py:76,123).  w is fp32 [ksize*ksize][C]. */
int tcvom_head_conv_fwd(const void* x, const float* w, const float* bias, float* alpha, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t ksize, int32_t mode, void* stream);
/* dw is [replicas][ksize*ksize][C] and db [replicas] fp32: the per-block partial sums are spread over `replicas`
 * copies (1..64; less atomic contention) which the caller adds up */
int tcvom_head_conv_bwd(const float* dalpha, const float* alpha, const void* x, const float* w, void* dx,
                        float* dpre, float* dw, float* db, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t ksize, int32_t mode, int32_t replicas, void* stream);

/* ------------------------------------------------------------------ Temporal Attention Module
 * Replaces FeatureAggregationModule._attention x2 + `v + xb + xf` (models/VMN/VMN_model.py:24-68).
 * q,kb,kf,v,out: NHWC bf16 [B,H,W,C]; mask uint8 [B,H,W]; attb/attf fp32 [B,window^2,H*W].      */
/* worklist: int32 [B*H*W + 1] device scratch; tam_fwd fills it with the number and the flat indices of the unknown pixels
 * (the attention runs on those only; the other pixels get out = v and zero logits) and tam_bwd reads it back */
int tcvom_tam_fwd(const void* q, const void* kb, const void* kf, const void* v, const uint8_t* mask,
                  void* out, float* attb, float* attf, int32_t* worklist, int32_t B, int32_t H, int32_t W, int32_t C,
                  int32_t window, void* stream);
/* pbuf, dsbuf: fp32 scratch [B][2][window^2][H*W]; dattb/dattf may be NULL */
int tcvom_tam_bwd(const void* q, const void* kb, const void* kf, const uint8_t* mask, const void* dout,
                  const float* dattb, const float* dattf, void* dq, void* dkb, void* dkf,
                  float* pbuf, float* dsbuf, const int32_t* worklist, int32_t B, int32_t H, int32_t W, int32_t C,
                  int32_t window, void* stream);

/* ------------------------------------------------------------------ Guided Contextual Attention pieces
 * (models/GCA/ops.py:106-229); the two N x N GEMMs go through tcvom_conv_igemm / tcvom_wgrad_igemm. */
int tcvom_gca_prepare(const void* g8, const uint8_t* unk8, void* G, float* scales, float* cvec, float* dvec,
                      float* nrm, int32_t B, int32_t h8, int32_t w8, int32_t CG, void* stream);
int tcvom_row_softmax(const float* S, void* P, int32_t rows, int32_t ncols, int64_t ld, int64_t ldp, void* stream);
int tcvom_row_softmax_bwd(const void* P, const float* dP, const float* cvec /*[rows/rows_per_batch][ncols]*/, void* T,
                          int32_t rows, int32_t ncols, int64_t ld, int64_t ldp, int32_t rows_per_batch, void* stream);
/* GuidedCxtAtten forward, scores + softmax in one pair of launches without the fp32 N x N score matrix (models/GCA/ops.py:177-190):
 * P[b][i][j] = softmax_j( c[b][j] <G[b][i], G[b][j]> - d[b][j] [i == j] ) as bf16 [batch][N][ld], zeros in the padding columns.
 * G bf16 [batch][N][D]; cvec / dvec fp32 [batch][N] (dvec may be NULL); stats fp32 scratch [batch][N][ld / 256][2].
 * tcvom_gca_scores_softmax_ok: 1 when the shape is served (N % 8 == 0, D % 64 == 0, ld % 256 == 0, ld <= 16384). */
int tcvom_gca_scores_softmax_ok(int32_t N, int32_t D, int64_t ld, int32_t batch);
int tcvom_gca_scores_softmax(const void* G, const float* cvec, const float* dvec, void* P, float* stats, int32_t N, int32_t D,
                             int64_t ld, int32_t batch, void* stream);
/* the two passes of tcvom_gca_scores_softmax as separate calls (the same kernels) */
int tcvom_gca_scores_exp(const void* G, const float* cvec, const float* dvec, void* P, float* stats, int32_t N, int32_t D,
                         int64_t ld, int32_t batch, void* stream);
int tcvom_gca_softmax_rescale(void* P, const float* stats, int32_t N, int64_t ld, int32_t batch, void* stream);
/* The same backward without the N x N fp32 dP matrix: one MFMA GEMM whose epilogue applies the softmax backward,
 *   T[b][i][j] = P[b][i][j] * (sum_v dO[b][i][v] V[b][j][v] - delta[b][i]) * cvec[b][j]   (bf16 [batch][N][ld], zero for j >= N),
 * with delta[b][i] = sum_j P dP = <dO[b][i], O[b][i]> from tcvom_rowdot_bf16 (O = the forward's P V, kept in fp32: for a
 * peaked softmax dP[i][i] - delta[i] cancels to zero and a bf16 O would leave its rounding error instead).  dO, V: bf16 [batch][N][DV].
 * Tt, Pt (both or neither; need ld % 256 == 0): bf16 [batch][ld][ld] transposed copies Tt[b][j][i] = T[b][i][j], Pt[b][j][i] =
 * P[b][i][j], written by the same epilogue -- the operands of the dV = P^T dO and M' = T^T G GEMMs. */
int tcvom_gca_dp_softmax_bwd(const void* dO, const void* V, const void* P, const float* delta, const float* cvec, void* T,
                             void* Tt, void* Pt, int32_t N, int32_t DV, int64_t ld, int32_t batch, void* stream);
/* out[r] = <a[r], b[r]> over `cols` columns (cols % 8 == 0); a is bf16, b is bf16 or (b_fp32 != 0) fp32 */
int tcvom_rowdot_bf16(const void* a, const void* b, int32_t b_fp32, float* out, int64_t rows, int32_t cols, void* stream);
int tcvom_gca_value_patches(const void* alpha, void* V, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream);
int tcvom_gca_value_patches_bwd(const float* dV, void* dalpha, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream);
int tcvom_gca_fold(const void* O, void* Y, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream);
int tcvom_gca_fold_f32(const float* O, void* Y, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream);   /* the same from an fp32 O */
int tcvom_gca_unfold(const void* dY, void* dO, int32_t B, int32_t h8, int32_t w8, int32_t C, void* stream);
int tcvom_gca_patches_bwd(float* dWq, const float* Mp, const void* G, const float* nrm, void* dg8, int32_t B,
                          int32_t h8, int32_t w8, int32_t CG, void* stream);

/* ------------------------------------------------------------------ DIM base (models/DIM/vggnet.py, config 1)
 * MaxPool2d(2, return_indices) / MaxUnpool2d(2) on NHWC bf16: idx holds the position (0..3 = dy*2+dx, first maximum
 * in scan order as torch) of every pooled element, one byte each.
 *   maxpool2_idx : x [N,H,W,C] -> y [N,H/2,W/2,C], idx            (vggnet.py:23,80)
 *   unpool2      : y, idx -> x (zeros elsewhere)                  (vggnet.py:61,104; also the max-pool gradient)
 *   pick2        : x [N,H,W,C], idx -> y [N,H/2,W/2,C] = x at idx (the unpool gradient)
 *   relu_bwd     : dy = y > 0 ? dz : 0 for conv+bias+ReLU layers without BatchNorm (vggnet.py:99-118) */
int tcvom_maxpool2_idx(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_unpool2(const void* y, const uint8_t* idx, void* x, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_pick2(const void* x, const uint8_t* idx, void* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_relu_bwd(const void* dz, const void* y, void* dy, int64_t numel, float negative_slope /* 0: ReLU */, void* stream);
/* im2col / col2im for conv6 (7x7, 512 -> 4096, vggnet.py:56), which runs as unfold + dense GEMM:
 *   unfold: u[p][t*C + c] = x[p + off_t][c];   fold: dx[q][c] = sum_t du[q - off_t][c*T + t]  (du is c-major) */
int tcvom_unfold(const void* x, void* u, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize, void* stream);
int tcvom_fold(const void* du, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ksize, void* stream);
/* DIM losses on one frame (models/model.py:94-127, utils/loss_func.py:9-22,42-59): with refine = mask ? pred : gt,
 *   acc[0] = sum_c |fg_c refine + bg_c (1 - refine) - img_c| mask       (L_comp numerator; fg/bg/img [B,3,HW] strided)
 *   acc[2] = sum |sqrt(dx(refine)^2 + dy(refine)^2 + e) - sqrt(dx(gt)^2 + dy(gt)^2 + e)| mask    (L_grad numerator)
 *   acc[1] = acc[3] = sum [mask > e]    (denominator count; two (sum, count) pairs for tcvom_loss_finalize)
 * and the gradient w.r.t. pred of  w_comp * L_comp + w_grad * L_grad  given the finished acc.  pred/gt/mask are
 * [B,HW] planes with batch strides p_stride / frame_stride, fg/bg/img batch stride rgb_stride. */
int tcvom_dim_losses_fwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg,
                         const float* img, float* comps /*[B,3,HW] or NULL*/, float* acc, int32_t B, int32_t H, int32_t W,
                         int64_t p_stride, int64_t frame_stride, int64_t rgb_stride, void* stream);
int tcvom_dim_losses_bwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg,
                         const float* img, const float* acc, const float* g_comp, const float* g_grad, float* dpred,
                         int32_t accumulate, int32_t B, int32_t H, int32_t W, int64_t p_stride, int64_t frame_stride,
                         int64_t rgb_stride, void* stream);

/* ------------------------------------------------------------------ facade: preprocessing, losses, optimizer
 * (models/model.py:54-127,285-345; utils/loss_func.py:9-22; train_ddp.py:296-297)               */
/* bg == NULL selects EvalModel.preprocess (models/model.py:360-386): `fg` holds the frames, `a` the user trimaps, no
 * compositing (imgs = flipped fg / 255) */
int tcvom_preprocess(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                     float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, float* trimask,
                     float* tris_vis, int64_t frames, int32_t H, int32_t W, int32_t dilate_radius, float eps,
                     int32_t tri_channels /* 3: one-hot {bg,unk,fg} (GCA); 1: 128/255-in-unknown plane (DIM, Index) */,
                     void* stream);
/* The same with one dilation radius PER CLIP (models/model.py:60-64: `kernel_rad` is drawn inside `for i in range(b)` when
 * DILATION_KERNEL is None).  frames = clips * frames_per_clip; clip_radii: HOST array of `clips` radii (read before the call
 * returns; consecutive clips of equal radius share a launch). */
int tcvom_preprocess_clips(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                           float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, float* trimask,
                           float* tris_vis, int32_t clips, int32_t frames_per_clip, int32_t H, int32_t W,
                           const int32_t* clip_radii, float eps, int32_t tri_channels, void* stream);
/* ... and the network input a second time as IEEE fp16 (x8_f16: same shape as x8; NULL = tcvom_preprocess_clips): what the first
 * conv of the fp16 island of the bf16 build reads */
int tcvom_preprocess_clips_f16(const float* a, const float* fg, const float* bg, float* gts, float* fgs, float* bgs,
                               float* imgs, uint8_t* unk_raw, uint8_t* unk_tmp, uint8_t* unk_dil, void* x8, void* x8_f16,
                               float* trimask, float* tris_vis, int32_t clips, int32_t frames_per_clip, int32_t H, int32_t W,
                               const int32_t* clip_radii, float eps, int32_t tri_channels, void* stream);
int tcvom_masked_l1_fwd(const float* p1, const float* g1, const float* m1, const float* p2, const float* g2,
                        const float* m2, const float* fgs, const float* bgs, float* alphas, float* comps,
                        float* acc, int64_t B, int64_t HW, int64_t p_stride, int64_t frame_stride, int64_t rgb_stride, void* stream);
int tcvom_masked_l1_bwd(const float* p1, const float* g1, const float* m1, const float* p2, const float* g2,
                        const float* m2, const float* acc, const float* gout, float weight, float* dp1,
                        float* dp2, int32_t accumulate, int64_t B, int64_t HW, int64_t p_stride, int64_t frame_stride, void* stream);
int tcvom_avgpool8(const float* g, float* out, int64_t frames, int32_t H, int32_t W, void* stream);
int tcvom_att_bce(const float* logits, const float* cg, const float* adj, const uint8_t* mask, float* dlogit,
                  float* acc, int32_t B, int32_t h, int32_t w, int32_t window, float thres, float smooth,
                  int64_t g_bstride, int32_t zero_acc, void* stream);
int tcvom_att_bce_bwd(const float* dlogit_unscaled, const float* acc, const float* gout, float weight,
                      float* dlogit, int64_t numel, int32_t window, void* stream);
int tcvom_loss_finalize(const float* acc, float* out, float weight, int32_t denom_mode, float numel_total,
                        int32_t window, int32_t accumulate, void* stream);
int tcvom_adam_mt(const int64_t* table, const int32_t* work, int32_t nblocks, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
/* The same with a device-side guard: when *skip_if_nonzero != 0 at execution time NOTHING is updated (the step of an
 * overflowed backward is dropped on the device, without a host synchronisation; torch.cuda.amp.GradScaler's rule). */
int tcvom_adam_mt_guarded(const int64_t* table, const int32_t* work, int32_t nblocks, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int64_t step, float grad_scale, const int32_t* skip_if_nonzero,
                          void* stream);
/* fp16 storage build: the conversions to the 16-bit type SATURATE at +-65504 (common.h h16_clamp).  `counters` (DEVICE int32[2]
 * or NULL; process-wide, read when a kernel is launched) receives: [0] += 1 per workgroup of a BatchNorm-backward reduction that
 * read a gradient element at the saturation value (an activation gradient overflowed under the current loss scale), [1] += 1 per
 * workgroup of a BatchNorm apply pass that read a saturated conv output.  The caller zeroes them.  No-op in the bf16 build. */
int tcvom_overflow_sink(int32_t* counters);

/* ------------------------------------------------------------------ FBA base (config 5: FullModel_VMD('vmn_fba'))
 * Weight standardisation (models/FBA/layers_WS.py:13-23) lives in the weight table of tcvom_sn_pack: layers with kind
 * bit 8 are packed as (w - mean_row) / (std_row + 1e-5); tcvom_ws_stats fills the per-row statistics (float4 per output
 * channel at table word 17), tcvom_ws_backward turns the gradient w.r.t. the standardised weight (already in the
 * parameter's layout inside grad_arena) into the gradient w.r.t. the raw weight, in place.  work_rows: (layer, row) pairs.
 * Kind bit 16 marks the 7x7 stride-2 stem, packed as a 4x4 stride-1 kernel over the 2x2 space-to-depth input. */
int tcvom_ws_stats(const int64_t* table, const int32_t* work_rows, int32_t n_rows, void* stream);
int tcvom_ws_backward(const int64_t* table, const int32_t* work_rows, int32_t n_rows, float* grad_arena, void* stream);
/* nn.MaxPool2d(3, 2, 1) (models/FBA/resnet_GN_WS.py:101) on NHWC bf16; idx = uint8 position of the maximum in its window */
int tcvom_maxpool3s2(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int tcvom_maxpool3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* pyramid pooling (models/VMN/VMN_FBA.py:23-31): nn.AdaptiveAvgPool2d(s) -> fp32 [N][s][s][C]; the backward sums the
 * gradients of up to 4 scales (dout / scales are HOST arrays) into dx (NHWC bf16) */
int tcvom_adaptive_avgpool(const void* x, float* out, int32_t N, int32_t h, int32_t w, int32_t C, int32_t s, void* stream);
/* every scale of the pyramid in one pass over x (outs / scales: HOST arrays of nscales <= 4 entries; outs[i]: fp32 [N][s_i][s_i][C]) */
int tcvom_adaptive_avgpool_multi(const void* x, float* const* outs, const int32_t* scales, int32_t nscales, int32_t N, int32_t h,
                                 int32_t w, int32_t C, void* stream);
/* the same without atomics: per-(cell, row split) sums into `scratch` (tcvom_adaptive_avgpool_scratch_floats floats; 0 = the scales have
 * too many bin boundaries), combined by a second launch in a fixed order; outs need no zeroing */
int tcvom_adaptive_avgpool_scratch_floats(const int32_t* scales, int32_t nscales, int32_t N, int32_t h, int32_t w, int32_t C);
int tcvom_adaptive_avgpool_multi_ws(const void* x, float* const* outs, const int32_t* scales, int32_t nscales, float* scratch, int32_t N,
                                    int32_t h, int32_t w, int32_t C, void* stream);
int tcvom_adaptive_avgpool_bwd(const float* const* dout, const int32_t* scales, int32_t nscales, void* dx, int32_t N,
                               int32_t h, int32_t w, int32_t C, void* stream);
/* ... + add[..., :C] (`add`: NHWC with add_ld channels per pixel): the second gradient of a tensor that feeds the pooling and the
 * pyramid concat (models/VMN/VMN_FBA.py:25-31) without a separate copy + add pass */
int tcvom_adaptive_avgpool_bwd_add(const float* const* dout, const int32_t* scales, int32_t nscales, void* dx, const void* add,
                                   int32_t add_ld, int32_t N, int32_t h, int32_t w, int32_t C, void* stream);
/* cat(F.interpolate(x, scale_factor=2, bilinear), skip) zero-padded to ld channels in one pass (models/VMN/VMN_FBA.py:37-48):
 * x [N][hs][ws][Cx], skip [N][2hs][2ws][Cs] -> dst [N][2hs][2ws][ld] */
int tcvom_up2_concat(const void* x, const void* skip, void* dst, int32_t N, int32_t hs, int32_t ws, int32_t Cx, int32_t Cs,
                     int32_t ld, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) between channel slices of NHWC bf16 tensors (pixel strides ld_*,
 * first channel c_*): any size ratio forward; backward for the exact x2 case (gather, bf16 out) and for small sources
 * such as the s x s pooled maps (fp32 out) */
int tcvom_bilinear(const void* src, void* dst, int32_t N, int32_t hs, int32_t ws, int32_t hd, int32_t wd, int32_t C,
                   int32_t ld_src, int32_t c_src, int32_t ld_dst, int32_t c_dst, void* stream);
int tcvom_bilinear_up2_bwd(const void* ddst, void* dsrc, int32_t N, int32_t hs, int32_t ws, int32_t C, int32_t ld_dst,
                           int32_t c_dst, void* stream);
int tcvom_bilinear_small_bwd(const void* ddst, float* dsrc, int32_t N, int32_t hs, int32_t ws, int32_t hd, int32_t wd,
                             int32_t C, int32_t ld_dst, int32_t c_dst, void* stream);
/* last decoder step (models/VMN/VMN_FBA.py:50-57): 1x1 conv 16 -> 7 (w [7][16], b [7] fp32), alpha = clamp, F / B =
 * sigmoid, fba_fusion (models/FBA/models.py:246-255).  x NHWC bf16 [N][HW][16]; img fp32 [N][3][HW] (image stride
 * img_stride); pred fp32 [N][7][HW] (stride pred_stride).  Backward: dw [replicas][7][16] and db [replicas][7] are
 * added to atomically (caller zeroes them and sums the replicas). */
int tcvom_fba_head_fwd(const void* x, const float* w, const float* b, const float* img, float* pred, int32_t N, int64_t HW,
                       int64_t img_stride, int64_t pred_stride, void* stream);
int tcvom_fba_head_bwd(const void* x, const float* w, const float* b, const float* img, const float* dpred, void* dx,
                       float* dw, float* db, int32_t replicas, int32_t N, int64_t HW, int64_t img_stride,
                       int64_t pred_stride, void* stream);
/* network input of the FBA base from the outputs of tcvom_preprocess (gts, dilated unknown mask, scaled RGB images):
 * make_trimap with 8 channels (models/model.py:71-77) incl. trimap_transform (utils/utils.py:12-39; exact Euclidean
 * distance transform, the reference calls cv2.distanceTransform on the host).  x2: bf16 [frames][H/2][W/2][64], the 2x2
 * space-to-depth form of cat(normalised RGB, 6 click maps, bg, fg) (16 channels per sub-pixel, 11 used); extras: bf16
 * [frames][H][W][8] = (normalised RGB, RGB, bg, fg); tris (optional): fp32 [frames][8][H][W]; edt_scratch: frames*2*H*W floats.
 * unk_dil == NULL: EvalModel.preprocess (models/model.py:380-385), the classes come from the user trimap `gts` alone */
int tcvom_fba_input(const float* gts, const uint8_t* unk_dil, const float* imgs, void* x2, void* extras, float* tris,
                    float* edt_scratch, int64_t frames, int32_t H, int32_t W, float eps, void* stream);

/* ------------------------------------------------------------------ FBA losses (models/model.py:129-197, utils/loss_func.py)
 * One interior frame of B samples, all tensors fp32 NCHW; *_stride = elements between consecutive samples of a tensor
 * that is a [:, c] slice of a [B, S, ...] window tensor.
 * point_fwd: refine / F / B selection by the unknown mask, the visualisation slices (alphas, comps, Fs, Bs), d0 [B][7][HW] =
 * (refine - gt, F - fg, B - bg), fb [B][6][HW] = (F, B), and acc[0..5] += sum |refine-gt|, |F gt + B (1-gt) - img|,
 * |fg refine + bg (1-refine) - img|, |F-fg|, |B-bg|, |grad-magnitude(refine) - grad-magnitude(gt)| (L1_mask / L1_grad sums).
 * point_bwd: dpred from coef[0..5] = d loss / d acc[i] plus the gradients w.r.t. d0 / fb of the two losses below. */
int tcvom_fba_point_fwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg, const float* img,
                        int64_t pred_stride, int64_t frame_stride, int64_t rgb_stride, float* d0, float* fb, float* alphas,
                        float* comps, float* Fs, float* Bs, float* acc, int32_t B, int32_t H, int32_t W, void* stream);
int tcvom_fba_point_bwd(const float* pred, const float* gt, const float* mask, const float* fg, const float* bg, const float* img,
                        int64_t pred_stride, int64_t frame_stride, int64_t rgb_stride, const float* coef, const float* g_d0,
                        const float* g_fb, float* dpred, int32_t B, int32_t H, int32_t W, void* stream);
/* exclusion_loss (utils/loss_func.py:63-90) on one pyramid level lvl [B][6][h][w] = (F, B): excl_abs: sums[0..3] += sum |gx F|,
 * |gx B|, |gy F|, |gy B|; excl_terms mode 0: out [B][2] += per-sample sum f(gF) f(a gB) in x / y (f(u) = (2 sigmoid(u) - 1)^2,
 * a = 2 mean|gF| / (mean|gB| + eps)); mode 1: out[0..1] += d L / d a given wts [B][2] = d L / d terms; excl_bwd: gradient w.r.t. the
 * level (+ 0.25 x the next level's gradient through the 2x2 average pooling). */
int tcvom_excl_abs(const float* lvl, float* sums, int32_t B, int32_t h, int32_t w, void* stream);
int tcvom_excl_terms(const float* lvl, const float* sums, const float* wts, float* out, int32_t mode, int32_t B, int32_t h,
                     int32_t w, void* stream);
int tcvom_avgpool2_f32(const float* x, float* y, int64_t planes, int32_t h, int32_t w, void* stream);
int tcvom_excl_bwd(const float* lvl, const float* sums, const float* wts, const float* dsum, const float* dcoarse, float* dlvl,
                   int32_t B, int32_t h, int32_t w, void* stream);
/* LapLoss (utils/loss_func.py:101-158) on the 7-channel difference image (the pyramid is linear): lap_down = reflect-padded 5x5
 * Gaussian + even sub-sampling; lap_resid: residual against the zero-interleaved, 4x Gaussian up-sampling, acc[c] += sum |resid|
 * per channel c = plane % 7, sgn = sign(resid); lap_bwd_coarse / lap_bwd_fine: the transposed operators, level by level. */
int tcvom_lap_down(const float* cur, float* down, int64_t planes, int32_t h, int32_t w, void* stream);
int tcvom_lap_resid(const float* cur, const float* down, int8_t* sgn, float* acc, int64_t planes, int32_t h, int32_t w, void* stream);
/* the scalar arithmetic between the FBA loss kernels (models/model.py:142-175) in one launch each: acc = the accumulator block of
 * tcvom_amd/fba_losses.py ([0..5] point sums, [6 + 7 l + c] Laplacian sums, per exclusion level sums[4] + terms[B][2]);
 * finish: out[3] = (L_alpha_comp, L_lap, L_grad); coefs: out = coef[6] | d L / d terms [3][B][2] | Laplacian level weights [5][7] from
 * the three incoming gradients (device scalars, NULL = 0).  excl_n* = 3 h w of the three exclusion levels. */
int tcvom_fba_loss_finish(const float* acc, int32_t B, float n1, float n3, float excl_n0, float excl_n1, float excl_n2, float* out,
                          void* stream);
int tcvom_fba_loss_coefs(const float* acc, const float* g_ac, const float* g_lap, const float* g_grad, int32_t B, float n1, float n3,
                         float excl_n0, float excl_n1, float excl_n2, float* out, void* stream);
int tcvom_lap_bwd_coarse(const int8_t* sgn, const float* coef, const float* gnext, float* r, int64_t planes, int32_t h, int32_t w,
                         void* stream);
int tcvom_lap_bwd_fine(const int8_t* sgn, const float* coef, const float* r, float* g, int64_t planes, int32_t h, int32_t w,
                       void* stream);

/* ------------------------------------------------------------------ evaluation metrics (calc_metric.py:22-46)
 * One frame, one launch: a / g predicted and ground-truth alpha (fp32 [H][W], 0..1), tri the uint8 trimap (unknown =
 * neither 0 nor 255), ha / hg the adjacent frame (or NULL), flow fp32 [2][H][W] (x then y displacement, NaN = invalid;
 * or NULL).  acc: double[8], zeroed by the caller, receives {unknown pixel count, sum |a-g|, sum (a-g)^2,
 * sum ((a-ha)-(g-hg))^2, MESSDdt sum |(a-g)-(pa-pg)|, sum |(a-g)^2-(pa-pg)^2|, valid flow pixels} where pa / pg are ha / hg
 * warped by the flow (utils/utils.py:70-123: bilinear, align_corners=True, zeros outside). */
int tcvom_matting_metrics(const float* a, const float* g, const uint8_t* tri, const float* ha, const float* hg,
                          const float* flow, double* acc, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------ data front-end (dataset/VMD.py)
 * The loader decodes PNGs into uint8 frames; everything after the decode runs on the device, S frames per launch.
 * tcvom_crop_resize_u8 = img_crop_and_resize (VMD.py:62-66): src uint8 [S][Hs][Ws][Cs] (HWC as decoded), crop window
 * rows ph..ph+nh-1 / columns pw..pw+nw-1, bilinear resize to Ho x Wo with align_corners=True, floor(x + 0.5); the nc
 * output planes take source channels chan[0..nc-1] (e.g. {2,1,0} turns RGB into the reference's BGR, {3} is the alpha of
 * an RGBA foreground).  dst fp32 [S][nc][Ho][Wo].  nh == Ho && nw == Wo is a plain crop + uint8 -> float conversion.
 * form selects which of ATen's two CPU bilinear kernels is reproduced bit for bit: 0 = the generic one (alpha planes; images in a
 * multi-threaded process), 1 = the channels-last one (3-channel images inside a DataLoader worker, which runs with one thread).
 * tcvom_count_unknown: counts[s] = #{0 < alpha < 255} of frame s of a fp32 [S][n] alpha (shape_aug's crop acceptance test,
 * VMD.py:146-150).  tcvom_pad_bottom_right = possible_pad (VMD.py:187-200): fp32 [S][C][H][W] -> [S][C][Ho][Wo], border
 * filled with value[c] (NULL = 0). */
int tcvom_crop_resize_u8(const void* src, float* dst, int32_t S, int32_t Hs, int32_t Ws, int32_t Cs, const int32_t* chan,
                         int32_t nc, int32_t ph, int32_t pw, int32_t nh, int32_t nw, int32_t Ho, int32_t Wo, int32_t form,
                         void* stream);
int tcvom_count_unknown(const float* alpha, int32_t S, int64_t n, int32_t* counts, void* stream);
/* flow_crop_and_resize of the loader's optical-flow branch (dataset/VMD.py:68-126): src fp32 [F][Hs][Ws][2] (x, y displacement
 * in pixels, NaN = invalid), crop (ph, pw, nh, nw) -> dst fp32 [F][2][Ho][Wo]: corner-aligned bilinear resampling, vectors
 * rescaled by nw / Wo and nh / Ho, NaN where the source is not smooth (VMD.py:76-99: neighbours within 45 degrees / 50 pixels)
 * or the vector leaves the new frame (VMD.py:119-124). */
int tcvom_flow_crop_resize(const float* src, float* dst, int32_t F, int32_t Hs, int32_t Ws, int32_t ph, int32_t pw, int32_t nh,
                           int32_t nw, int32_t Ho, int32_t Wo, void* stream);
int tcvom_pad_bottom_right(const float* src, float* dst, int32_t S, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                           const float* value, void* stream);

#ifdef __cplusplus
}
#endif
#endif
