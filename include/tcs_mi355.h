/*
 * tcs_mi355.h — C ABI of libtcs_mi355.so: the TC-Stereo inference hot path on MI355X (gfx950).
 *
 * The reference (jiaxiZeng/Temporally-Consistent-Stereo-Matching) has no FFI layer: its "operator
 * API" is Python call signatures over torch tensors, plus ONE native entry point, the CuPy-JIT CUDA
 * kernel `softsplat_out(n, tenIn, tenFlow, tenOut)` (core/utils/splatting/softsplat.py:285-290).
 * Every function below replaces one of those Python-level operators (cited per function) with a
 * hand-written HIP kernel sequence.  The Python mirror in
 * `temporally-consistent-stereo-matching_amd/core/` binds these with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - Every tensor is a raw DEVICE pointer to contiguous float32 in NCHW order unless stated.
 *  - The caller owns every buffer, including outputs and workspaces (`*_workspace_bytes`).
 *    Nothing is allocated, freed or synchronised inside; all work is enqueued on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream).  Safe under HIP graph capture.
 *  - Return value: 0 on success, negative TCS_E* otherwise.  Never throws, never exits.
 *  - Re-entrant per stream.  The only mutable global state is the set of device-side S16 domain-flag words, written by kernels on
 *    a domain violation and read-and-cleared by tcs_s16_flags / tcs_s16_flags_detail (which synchronise the device).
 */
#ifndef TCS_MI355_H
#define TCS_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCS_OK 0
#define TCS_EINVAL (-1)      /* bad argument (null pointer, non-positive size, unsupported shape) */
#define TCS_ELAUNCH (-2)     /* the HIP runtime rejected a launch / memset */
#define TCS_EUNSUPPORTED (-3)

typedef void* tcs_stream_t;

int tcs_abi_version(void);                 /* bumped when a signature changes (7: grouped launches, blend_warm_*) */
const char* tcs_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Correlation volume: build, pyramid, first-frame argmax, lookup           (core/corr.py)
 *
 * HBM layout of the pyramid ("skewed"): level i is stored as P_i[b][h][d][w1], d in [0, W_i),
 * W_i = W >> i, holding L_i[b][h][w1][j] at d = ((w1 >> i) - j) mod W_i.  For a smooth disparity
 * field the 2r+2 taps of horizontally adjacent pixels then sit in the same few rows `d`, contiguous
 * along w1, so one wavefront's loads coalesce into a handful of 256-byte segments instead of 64
 * scattered lines (the reference layout [b][h][w1][j] puts every pixel's window in its own row).
 * ---------------------------------------------------------------------------------------------- */

/* bytes of one skewed level i (i = 0..3) for a [B,C,H,W] feature map */
size_t tcs_corr_level_bytes(int B, int H, int W, int level);
/* scratch needed by tcs_corr_build (natural-layout level 0 + inverse norms) */
size_t tcs_corr_build_workspace_bytes(int B, int H, int W);

/*
 * CorrBlock1D.__init__ + CorrBlock1D.corr (core/corr.py:8-31,54-62) and, when sparse_* are given,
 * CorrBlock1D.argmax_disp (core/corr.py:67-79) fused into the same pass.
 *   fmap1, fmap2 : [B,C,H,W]
 *   pyr0..pyr3   : skewed levels (sizes from tcs_corr_level_bytes)                    (required)
 *   nat1..nat3   : natural-layout levels [B,H,W,W>>i] as the reference stores them    (nullable; tests)
 *                  (natural level 0 is always left in the workspace: see tcs_corr_ws_level0)
 *   cost_volume  : [B,W,H,W] masked volume cost[b,w2,h,w1] = V*(w2<=w1), corr.py:25-31 (nullable)
 *   sparse_disp, sparse_cost, sparse_mask : [B,1,H,W] each                             (all or none)
 */
int tcs_corr_build(const float* fmap1, const float* fmap2, int B, int C, int H, int W,
                   float* pyr0, float* pyr1, float* pyr2, float* pyr3,
                   float* nat1, float* nat2, float* nat3,
                   float* cost_volume,
                   float* sparse_disp, float* sparse_cost, float* sparse_mask,
                   void* workspace, tcs_stream_t stream);
/* pointer to the natural-layout level 0 volume [B,H,W,W] inside a build workspace */
float* tcs_corr_ws_level0(void* workspace);

/*
 * CorrBlock1D.__call__ (core/corr.py:33-52) with bilinear_sampler (core/utils/utils.py:82-97):
 *   coords [B,1,H,W] (x position in the right image) -> out [B, 4*(2*radius+1), H, W].
 * Channel = level*(2r+1) + tap, taps ascending in dx; samples outside a level read 0.
 */
int tcs_corr_lookup(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3,
                    const float* coords, int B, int H, int W, int radius, float* out,
                    unsigned long long* stamps, tcs_stream_t stream);
/* Measurement hook: when `stamps` is non-NULL (device uint64[2 * tcs_corr_lookup_blocks(B,H,W)], zeroed by the
 * caller) every workgroup records the device wall clock (s_memrealtime, 100 MHz) at its start and end, so the
 * launch duration = max(end) - min(start) can be read without a profiler.  NULL in production. */
int tcs_corr_lookup_blocks(int B, int H, int W);

/* ------------------------------------------------------------------------------------------------
 * Temporal warp                                 (core/utils/geo_utils.py, core/utils/splatting/softsplat.py)
 * ---------------------------------------------------------------------------------------------- */
size_t tcs_warp_workspace_bytes(int B, int C, int H, int W);

/*
 * warp() (geo_utils.py:158-198) = disparity -> 3-D -> relative pose -> reprojection, then
 * softsplat(..., 'soft-clipeps', valid) (softsplat.py:232-274, kernel softsplat_out :285-335),
 * optionally followed by the temporal matching cost of tc_stereo.py:139-140.
 *   prev_disp [B,1,H,W] (>= 0), prev_fmap [B,C,H,W]
 *   T_rel [B,4,4], K [B,3,3], K_inv [B,3,3] (K already scaled to the 1/4 grid), baseline [B]
 *   out_disp [B,1,H,W], out_mask [B,1,H,W]                                   (required)
 *   out_fmap [B,C,H,W]                                                       (nullable)
 *   cur_fmap [B,C,H,W] + out_cost [B,1,H,W]: cosine cost vs the current frame (both or none)
 * Float atomics are used for the splat, like the reference's atomicAdd: the summation order, and
 * therefore the last bits, can differ from run to run.
 */
int tcs_warp_forward(const float* prev_disp, const float* prev_fmap, const float* T_rel, const float* K,
                     const float* K_inv, const float* baseline, int B, int C, int H, int W,
                     float* out_disp, float* out_fmap, float* out_mask,
                     const float* cur_fmap, float* out_cost,
                     void* workspace, tcs_stream_t stream);

/* The pre-splat quantities of warp() (geo_utils.py:169-193), exposed for parity tests:
 * cur_disp, valid, flow [B,2,H,W], metric — all [B,1,H,W] unless noted. */
int tcs_warp_geometry(const float* prev_disp, const float* T_rel, const float* K, const float* K_inv,
                      const float* baseline, int B, int H, int W,
                      float* cur_disp, float* valid, float* flow, float* metric,
                      void* workspace, tcs_stream_t stream);

/* softsplat 'summation' core (softsplat.py:285-335): out[B,C,H,W] += bilinear splat of in along flow.
 * `out` must be zeroed by the caller.  One thread per source pixel; 4 x C float atomics each. */
int tcs_softsplat_sum(const float* in, const float* flow, int B, int C, int H, int W, float* out,
                      tcs_stream_t stream);

/* get_backward_grid (geo_utils.py:201-236): grid [B,2,H,W] of previous-frame pixel coordinates. */
int tcs_backward_grid(const float* disp, const float* T_rel, const float* K, const float* K_inv,
                      const float* baseline, int B, int H, int W, float* grid, tcs_stream_t stream);

/* Camera algebra of TCStereo.forward (tc_stereo.py:121-127,159) on the device, no host round trip:
 * K_scaled = K * [scale,scale,1]^T (rows 0,1), its inverse, and — when T / T_prev are given —
 * T_rel = T * inv(T_prev) (geo_utils.py:148-155) and T_back = T_prev * inv(T).  K [B,3,3], T [B,4,4]. */
int tcs_pose_prepare(const float* K, const float* T, const float* T_prev, float scale, int B,
                     float* K_scaled, float* K_scaled_inv, float* T_rel, float* T_back, tcs_stream_t stream);

/* bilinear_sampler (core/utils/utils.py:82-97): img [B,C,Hi,Wi] sampled at grid [B,2,Ho,Wo] (x,y in
 * pixels), zeros outside, align_corners=True -> out [B,C,Ho,Wo]. */
int tcs_bilinear_sample(const float* img, const float* grid, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        float* out, tcs_stream_t stream);

/* 0.5 * F.interpolate(grid, scale_factor=0.5, bilinear, align_corners=True) (tc_stereo.py:163):
 * grid [B,2,H,W] -> [B,2,H/2,W/2]. */
int tcs_grid_halve(const float* grid, int B, int H, int W, float* out, tcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Stencils of the refinement loop
 * ---------------------------------------------------------------------------------------------- */
/* coords bookkeeping (tc_stereo.py:188-189): coords1 += delta; disp_q = x - coords1 (in place on coords1). */
int tcs_flow_step(float* coords1, const float* delta, int B, int H, int W, float* disp_q, tcs_stream_t stream);

/* The three stencils that open an iteration's gradient stage, fused: disp_q = x - (coords1 + delta)
 * (tc_stereo.py:188-189; coords1 itself is left untouched), grad = scale * disp2disp_gradient_xy(disp_q)
 * (geo_utils.py:115-132, update.py:199), cands = disp2disp_grad_candidates(disp_q, level=2) (geo_utils.py:73-101).
 * Outputs equal tcs_flow_step + tcs_disp_gradient_xy + tcs_grad_candidates bit for bit.
 * coords1, delta, disp_q [B,1,H,W]; grad [B,2,H,W]; cands [B,32,H,W]. */
int tcs_flow_step_grads(const float* coords1, const float* delta, int B, int H, int W, float scale, float* disp_q, float* grad,
                        float* cands, tcs_stream_t stream);

/* disp2disp_gradient_xy (geo_utils.py:115-132) scaled by `scale` (the 5x of update.py:199):
 * disp [B,1,H,W] -> grad [B,2,H,W]. */
int tcs_disp_gradient_xy(const float* disp, int B, int H, int W, float scale, float* grad, tcs_stream_t stream);

/* disp2disp_grad_candidates(level=2) (geo_utils.py:73-101) flattened as update.py:202-204:
 * disp [B,1,H,W] -> cands [B,32,H,W], channel = comp*16 + k. */
int tcs_grad_candidates(const float* disp, int B, int H, int W, float* cands, tcs_stream_t stream);

/* DispRefine.propagate_disparity (update.py:259-289): grad [B,2,H,W], disp [B,1,H,W] ->
 * out [B,27,H,W] = cat(candidates(9), |grad diff|(18)) — the input of disp_f_stem (update.py:295). */
int tcs_propagate_disparity(const float* grad, const float* disp, int B, int H, int W, float* out27,
                            tcs_stream_t stream);

/* update.py:298-300 + tc_stereo.py:198-202: softmax over the 9 logits (max-subtracted), blend the 9
 * candidates (first 9 channels of a [B,cand_ctot,H,W] buffer), and emit
 *   refined [B,1,H,W], delta_disp = refined - disp_q [B,1,H,W] (nullable), coords1 = x - refined (nullable),
 *   and the next iteration's motion-encoder input coords1 - x (tc_stereo.py:180) twice (both nullable): flow_x [B,1,H,W]
 *   and flow_x_ch, one channel plane per batch element at a stride of flow_x_ch_bstride floats (channel 127 of the
 *   [B,128,H,W] motion feature buffer, update.py:126). */
int tcs_softmax_blend(const float* logits9, const float* cand, int cand_ctot, const float* disp_q,
                      int B, int H, int W, float* refined, float* delta_disp, float* coords1, float* flow_x, float* flow_x_ch,
                      long long flow_x_ch_bstride, tcs_stream_t stream);

/* TCStereo.upsample_flow (tc_stereo.py:75-88) with factor 4, applied to flow = -disp:
 * disp [B,1,H,W], mask [B,144,H,W] -> flow_up [B,1,4H,4W]; flow_q [B,1,H,W] = -disp (nullable).
 * clip != 0 additionally applies the torch.clip(., max=0) of the returned dict (tc_stereo.py:223-224)
 * to both outputs. */
int tcs_convex_upsample(const float* disp, const float* mask, int B, int H, int W, int clip, float* flow_up, float* flow_q,
                        tcs_stream_t stream);

/* pool2x = avg_pool2d(3, stride 2, pad 1), divisor 9 everywhere (update.py:114-115): [B,C,H,W] -> [B,C,Ho,Wo],
 * Ho = (H-1)/2+1. */
int tcs_avgpool3s2(const float* x, int B, int C, int H, int W, float* out, tcs_stream_t stream);

/* interp(): bilinear resize with align_corners=True (update.py:122-124): [B,C,H,W] -> [B,C,Ho,Wo]. */
int tcs_resize_bilinear(const float* x, int B, int C, int H, int W, int Ho, int Wo, float* out, tcs_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Convolutions on the matrix cores, fp32 NCHW tensors in and out: fp32 MFMA (TCS_MATH_F32) or fp16-split operands with fp32
 * accumulation (TCS_MATH_F16X3, the default of the Python mirror) — see tcs_conv_desc.math
 * ---------------------------------------------------------------------------------------------- */
#define TCS_ACT_NONE 0
#define TCS_ACT_RELU 1
#define TCS_ACT_SIGMOID 2
#define TCS_ACT_TANH 3
#define TCS_ACT_LEAKY 4          /* LeakyReLU(0.01) */
#define TCS_ACT_RELU_ADD_RELU 5  /* relu(relu(v) + addend): the tail of a residual block (extractor.py:44-58); needs addend */

#define TCS_EPI_LINEAR 0         /* out = act(conv + bias + addend) * post_scale                   */
#define TCS_EPI_GRU_ZR 1         /* first half: z = sigmoid(. + cz) -> out; second half: r = sigmoid(. + cr), out2 = r*h */
#define TCS_EPI_GRU_Q 2          /* q = tanh(. + cq); out = blend(z, h, q)                           */
#define TCS_EPI_BLEND9 4         /* tcs_conv2d_s16 only, Cout = 9: the nine outputs are the logits of DispRefine's softmax blend
                                    (core/update.py:298-300), which runs in the epilogue — see tcs_conv_s16_desc.blend_*     */
#define TCS_EPI_DECONV2X 3       /* ConvTranspose2d(k=4, stride 2, pad 1, no bias): weights from
                                    tcs_pack_deconv4x4s2_f16x3, out [B, Cout/4, 2H, 2W]   (F16X3 only)      */

#define TCS_MATH_F32 0           /* v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate                                  */
#define TCS_MATH_F16X3 1         /* fp16 hi/lo split, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate (~2^-21) */

#define TCS_MAX_SRC 4

typedef struct tcs_conv_desc {
    /* virtual concatenation of up to 4 NCHW sources along channels (replaces torch.cat, update.py:79-80) */
    const float* src[TCS_MAX_SRC];
    int src_ch[TCS_MAX_SRC];
    int n_src;
    const float* weight;     /* packed by tcs_pack_conv_weight */
    const float* bias;       /* [Cout] or NULL */
    int B, H, W;             /* stride-1 'same' convolution: output H,W = input H,W */
    int Cin, Cout, ksize;    /* ksize in {1,3,7} */
    int epilogue;            /* TCS_EPI_* */
    int act;                 /* TCS_ACT_*  (LINEAR epilogue only) */
    float post_scale;        /* LINEAR epilogue: multiplies the activated value (e.g. 0.25 of update.py:304) */
    const float* addend;     /* LINEAR: optional [B,Cout,H,W] added before the activation.
                                GRU_ZR: cz;  GRU_Q: cq  ([B,hidden,H,W], nullable)                   */
    const float* addend2;    /* GRU_ZR: cr (nullable) */
    const float* h;          /* GRU_ZR / GRU_Q: hidden state [B,hidden,H,W] */
    const float* z;          /* GRU_Q: update gate from the ZR pass */
    int blend_keep_z;        /* GRU_Q: 0 -> (1-z)h + zq (ConvGRU, update.py:85); 1 -> zh + (1-z)q (update.py:34,66) */
    float* out;              /* LINEAR: [B,out_ctot,H,W] written at channel offset out_coff. GRU_ZR: z. GRU_Q: new h */
    int out_ctot, out_coff;
    float* out2;             /* GRU_ZR: r*h */
    int stride;              /* 0 or 1: 'same' convolution; 2: 3x3 stride-2 pad-1 (F16X3 only), output (H-1)/2+1 x (W-1)/2+1 */
    int math;                /* TCS_MATH_F32: weights from tcs_pack_conv_weight; TCS_MATH_F16X3: from ..._f16x3 */
    float weight_unscale;    /* F16X3: 2^-scale_log2 given to tcs_pack_conv_weight_f16x3 */
    void* out16;             /* LINEAR, stride 1: optional S16 ("pre-split", see below) copy of the output, so that a layer fed by a
                                fp32 tensor (correlation features, disparity stencils) hands its result to tcs_conv2d_s16 consumers
                                without a conversion pass; `out` may then be NULL */
    int out16_groups, out16_group_offset;
    /* 7x7 RGB stem only (Cin = 3): the image preparation of TCStereo.forward folded into its input staging (core/tc_stereo.py:101-107).
     * in_transform = 1: every in-image sample x is read as 2 * (x / 255) - 1 (padding stays 0).  src_batch2 / batch_split: batch
     * elements >= batch_split come from this second tensor [B - batch_split, 3, H, W] instead of src[0] — torch.cat((image1, image2), 0)
     * without the copy.  0 / NULL = off. */
    int in_transform;
    const float* src_batch2;
    int batch_split;
} tcs_conv_desc;

/* packed weight size in floats for a [Cout,Cin,k,k] convolution */
size_t tcs_conv_packed_floats(int Cout, int Cin, int ksize);
/* OIHW device weights -> the kernel's layout ([Cin_pad][k*k][Cout_pad], zero padded) */
int tcs_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int ksize, float* packed, tcs_stream_t stream);
/* fp16-split weights: OIHW * 2^scale_log2, split into (hi, lo) halves, in the kernel's LDS image order.
 * Choose scale_log2 so that max|w| * 2^scale_log2 is ~2^10..2^14 (keeps the lo halves normal). ksize in {1,3}. */
size_t tcs_conv_packed_floats_f16x3(int Cout, int Cin, int ksize);
int tcs_pack_conv_weight_f16x3(const float* w_oihw, int Cout, int Cin, int ksize, int scale_log2, float* packed,
                               tcs_stream_t stream);
/* nn.ConvTranspose2d(Cin, Cout, 4, stride 2, padding 1, bias=False) weights [Cin,Cout,4,4] -> the F16X3 layout of
 * the equivalent 3x3 convolution with 4*Cout outputs (one group per output parity); use with TCS_EPI_DECONV2X,
 * desc.Cout = 4*Cout, ksize = 3.  (core/utils/basic_layers.py:17-21,57) */
size_t tcs_deconv_packed_floats_f16x3(int Cin, int Cout);
int tcs_pack_deconv4x4s2_f16x3(const float* w_iohw, int Cin, int Cout, int scale_log2, float* packed, float* scratch_oihw,
                               tcs_stream_t stream);
/* nn.InstanceNorm2d (affine=False, eps, biased variance) + activation + optional addend, per (b, c) plane:
 * out = act((x - mean) / sqrt(var + eps)) + addend      (basic_layers.py:28-35,65-76; update.py:325-367) */
int tcs_instance_norm(const float* x, int B, int C, int H, int W, float eps, int act, const float* addend, float* out,
                      tcs_stream_t stream);
/* nn.Conv2d(Cin, 1, 3, padding=1): single-output-channel 3x3 convolution as a plain fp32 reduction (FlowHead.conv2,
 * core/update.py:13).  w_oihw is the ORIGINAL [1,Cin,3,3] weight (no packing); out [B,1,H,W]. */
int tcs_conv3x3_cout1(const float* x, const float* w_oihw, const float* bias, int B, int Cin, int H, int W, float* out,
                      tcs_stream_t stream);
/* nn.Conv2d(k, padding=k/2) + fused epilogue; covers ConvGRU / Lightfuse / HiddenstateUpdater /
 * BasicMotionEncoder / FlowHead / the stride-1 convs of DispRefine and DispGradPredictor (core/update.py). */
int tcs_conv2d(const tcs_conv_desc* desc, tcs_stream_t stream);
/* n (1 or 2) INDEPENDENT convolutions of tcs_conv2d as one launch where a grouped kernel exists for them (two fp32-MFMA 3x3 layers of
 * <= 32-channel tiles: the first layers of DispGradPredictor's stems, core/update.py:200-205), otherwise one after the other; bit-equal
 * to n calls of tcs_conv2d either way (see tcs_conv2d_s16_group). */
int tcs_conv2d_group(const tcs_conv_desc* const* descs, int n, tcs_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Pre-split activations ("S16" tensors): the refinement loop's internal activation format.
 *
 * The convolutions contract fp16 hi/lo halves of every fp32 operand (TCS_MATH_F16X3).  Inside the refinement loop
 * (core/tc_stereo.py:176-210, core/update.py) the PRODUCER of an activation splits it once and stores it in operand
 * form, so that the consuming convolution stages it with plain 16-byte copies (LDS-DMA) instead of converting it again
 * for every output tile:
 *
 *     logical [B][C][H][W] fp32   ->   _Float16 [B][G][2][H+2][W+2][8]
 *         G = 2*ceil(C/16) channel groups of 8, [2] = {hi, lo} with x = hi + lo (+ <= 2^-22 |x|, |x| <= 65504),
 *         a one-pixel ZERO border (the 'same' padding of a 3x3 convolution), zero padding channels.
 *
 * The caller allocates the buffer ZERO-FILLED once (tcs_s16_bytes); producers only ever write interior pixels of real
 * channel groups, so the border stays zero for the buffer's lifetime.  fp32 NCHW remains the format of every reference-facing
 * entry point above; tcs_s16_from_f32 / tcs_s16_to_f32 convert at the loop's boundary.
 * ------------------------------------------------------------------------------------------------------------------- */
size_t tcs_s16_bytes(int B, int C, int H, int W);
/* Domain guard of the split (|x| <= 65504): every producer of S16 data marks a device-side flag word when it had to clamp a
 * finite value (bit 0) or met a NaN / Inf (bit 1) — the replacement of the reference's per-iteration NaN asserts
 * (core/update.py:27-35,58-67,78-86,155-158).  Reads and clears the word; SYNCHRONISES the device (call it once per frame or
 * sequence, outside any graph capture). */
int tcs_s16_flags(unsigned int* flags_out);
/* The same read-and-clear, one word per producing source file of the library, for locating a violation:
 * [0] tcs_conv_s16.hip (S16 convolutions, GRU / deconv / blend epilogues, tcs_s16_from_f32), [1] tcs_s16_ops.hip (pool, interp, InstanceNorm,
 * propagate stencil, fused HiddenstateUpdater), [2] tcs_conv.hip (fp32-MFMA and 7x7 convolutions with an S16 epilogue),
 * [3] tcs_conv_f16.hip (fp32-tensor fp16-split convolutions with an S16 epilogue), [4] tcs_stencil.hip (the stand-alone blend kernel's flow channel). */
#define TCS_S16_FLAG_UNITS 5
int tcs_s16_flags_detail(unsigned int* per_unit);
/* x [B,C,H,W] fp32 -> groups [group_offset, group_offset + 2*ceil(C/16)) of an S16 tensor with groups_total groups */
int tcs_s16_from_f32(const float* x, int B, int C, int H, int W, void* s16, int groups_total, int group_offset, tcs_stream_t stream);
/* channels [8*group_offset, 8*group_offset + C) of an S16 tensor -> out [B,C,H,W] fp32 (hi + lo) */
int tcs_s16_to_f32(const void* s16, int B, int C, int H, int W, int groups_total, int group_offset, float* out, tcs_stream_t stream);

typedef struct tcs_conv_s16_desc {
    /* virtual concatenation of up to 4 S16 sources along channels (torch.cat of update.py:79-80); every source but the
     * last must carry a multiple of 16 channels */
    const void* src[TCS_MAX_SRC];
    int src_ch[TCS_MAX_SRC];        /* logical channels per source */
    int src_groups[TCS_MAX_SRC];    /* groups allocated per source tensor */
    int n_src;
    const float* weight;     /* tcs_pack_conv_weight_f16x3 / tcs_pack_deconv4x4s2_f16x3 of the [Cout, Cin, k, k] weight */
    const float* bias;       /* [Cout] or NULL */
    int B, H, W;             /* INPUT grid */
    int Cin, Cout, ksize;    /* ksize in {1,3} */
    int stride;              /* 1, or 2 (3x3 pad 1 or 1x1 pad 0, LINEAR): output (H-1)/2+1 x (W-1)/2+1 */
    int epilogue;            /* TCS_EPI_* as for tcs_conv2d */
    int act;
    float post_scale;
    float weight_unscale;
    const float* addend;     /* fp32 NCHW: LINEAR addend [B,Cout,Ho,Wo]; GRU_ZR cz / GRU_Q cq [B,hidden,H,W] (nullable) */
    const float* addend2;    /* GRU_ZR: cr */
    const void* addend16;    /* LINEAR: S16 addend on the output grid (the skip of a residual block, extractor.py:44-58; use with
                                TCS_ACT_RELU_ADD_RELU), addend16_groups groups; nullable */
    int addend16_groups;
    const void* h;           /* GRU: hidden state, S16 with h_groups groups */
    int h_groups;
    const float* z;          /* GRU_Q: update gate, fp32 [B,hidden,H,W] */
    int blend_keep_z;
    void* out16;             /* S16 output (LINEAR: act(.)*post_scale; DECONV2X: [C/4 channels, 2H, 2W]; GRU_ZR: r*h;
                                GRU_Q: new h, may alias `h`), written at group offset out16_group_offset; nullable for LINEAR */
    int out16_groups, out16_group_offset;
    float* out32;            /* fp32 NCHW output: LINEAR (nullable; [B,out_ctot,Ho,Wo] at channel out_coff); GRU_ZR: z (required);
                                GRU_Q: optional fp32 copy of the new h */
    int out_ctot, out_coff;
    int tile_cfg;            /* 0 = choose by grid size; otherwise CSPLIT*100000 + RS*10000 + MT*1000 + ROWS*100 + KSTEPS*10 + NSTAGE
                                (benchmarks, tests; csrc/tcs_conv_s16.hip) */
    int addend_ctot;         /* channels per batch element of `addend` / `addend2` (0 = Cout, or hidden for the GRU epilogues).  A wider
                                value lets them be channel slices of one tensor: a convolution is linear in its input channels, so a
                                layer whose inputs become available at different times runs as partial convolutions that
                                accumulate into one fp32 [B, addend_ctot, H, W] tensor (LINEAR: addend = out32), and the last
                                partial applies the real epilogue with that sum as its addend (GRU_ZR: cz = sum[:, :hidden],
                                cr = sum[:, hidden:]) */
    /* TCS_EPI_BLEND9 (w_head's last 1x1 convolution + tcs_softmax_blend_s16 in one launch; same arithmetic, same outputs):
     * refined = sum_k softmax(logits)_k * cand[:, k]; delta = refined - disp; coords1 = x - refined; flow_x = coords1 - x, also
     * written into channel blend_flow16_channel of the S16 tensor blend_flow16 (the motion features, core/update.py:126). */
    const float* blend_cand;        /* [B, blend_cand_ctot >= 9, H, W]: the nine candidates are its first channels */
    int blend_cand_ctot;
    const float* blend_disp;        /* [B,1,H,W] */
    float* blend_refined;           /* [B,1,H,W], required */
    float* blend_delta;             /* [B,1,H,W] or NULL */
    float* blend_coords1;           /* [B,1,H,W] or NULL */
    float* blend_flow_x;            /* [B,1,H,W] or NULL */
    void* blend_flow16;             /* S16 tensor or NULL */
    int blend_flow16_groups, blend_flow16_channel;
    /* LINEAR: two layers that read the same input as ONE launch (weights concatenated along Cout): output channels
     * [out16_split, Cout) go to groups [0, ...) of the second S16 tensor `out16b` instead of `out16`
     * (DispGradPredictor's residual_head[0] and conv_out[0] both read x4_up, core/update.py:212-214).
     * out16_split must be a multiple of 32; NULL = single output. */
    void* out16b;
    int out16b_groups, out16_split;
    /* DECONV2X: InstanceNorm2d statistics of the output accumulated by the transposed convolution itself (the up-blocks are
     * ConvTranspose2d -> InstanceNorm2d -> LeakyReLU, core/utils/basic_layers.py:28-35,57): every workgroup ADDS (sum x, sum x^2) of its
     * tile, in 64-bit fixed point (2^20 / 2^16 units; integer atomics, so the result does not depend on the order of arrival), to
     * in_stats[b][channel][2]; tcs_instance_norm_apply_s16 turns them into mean / rstd.  in_stats: tcs_deconv_in_stats_bytes() bytes
     * (16 per (b, channel)), ZERO-FILLED by the caller before EVERY launch that accumulates into it; needs Cout/4 % 32 == 0 and an
     * output grid of <= 2^20 pixels.  NULL = off.  in_eps: unused since ABI 6 (the apply call takes eps). */
    void* in_stats;
    float in_eps;
    /* LINEAR, stride 1: "tap partials" of a FOLLOWING 3x3 convolution to tap_nout = 1 or 2 channels (FlowHead.conv2, core/update.py:13-17;
     * DispGradPredictor.residual_head[2], core/update.py:196,213), so that that convolution never runs as a launch: for each of the first
     * tap_tiles 32-channel output tiles this launch also writes tap_out[b][tile][o*9 + t][H][W] = sum_{c in tile} w2[o][c][t] * out[c]
     * (fp32), w2 packed by tcs_pack_tap_weights.  tcs_taps_sum / tcs_flow_taps_step_grads / tcs_taps_propagate_s16 finish the sum over
     * tiles and taps.  With tap_out the S16 / fp32 outputs of those tiles are optional (out16 may be NULL). */
    const float* tap_weights;
    float* tap_out;
    int tap_nout, tap_tiles;
    float tap_unscale;              /* 2^-scale_log2 given to tcs_pack_tap_weights */
    /* TCS_EPI_BLEND9, optional: the skewed correlation pyramid of the frame (tcs_corr_build's pyr0..3; [B,H,W>>i,W] each, this launch's
     * grid).  The lane that has just computed a pixel's new coords1 also issues the 4 x (2r+2) tap loads the NEXT tcs_corr_lookup will
     * make for that pixel and discards the values: between two lookups ~300 MB of convolution traffic push the pyramid rows out of
     * L2 and the Infinity Cache, and the lookup is three dependent memory round trips long, so where its taps come from decides
     * its duration.  No effect on any result.  NULL = off. */
    const float* blend_warm_pyr[4];
    int blend_warm_radius;
} tcs_conv_s16_desc;

/* S16 glue of the loop: pool2x / interp (core/update.py:114-124), the up-blocks' InstanceNorm + LeakyReLU + skip
 * (core/utils/basic_layers.py:28-35,65-76), DispRefine's candidate stencil (core/update.py:259-289: 27 channels into 4 groups
 * for the 1x1 stem, the 9 candidates also as fp32 for the blend), one fp32 channel into an S16 tensor, and the blend
 * kernel writing the next iteration's flow input into a channel of the S16 motion features (core/update.py:126). */
int tcs_avgpool3s2_s16(const void* x, int B, int groups, int H, int W, void* out, int out_groups, tcs_stream_t stream);
int tcs_resize_bilinear_s16(const void* x, int B, int groups, int H, int W, int Ho, int Wo, void* out, int out_groups, tcs_stream_t stream);
size_t tcs_instance_norm_s16_workspace_bytes(int B, int groups, int H, int W);
int tcs_instance_norm_s16(const void* x, int B, int groups, int H, int W, float eps, int act, const void* addend, int addend_groups,
                          void* out, int out_groups, void* workspace, tcs_stream_t stream);
/* The apply half of tcs_instance_norm_s16 for statistics accumulated by the producing transposed convolution
 * (tcs_conv_s16_desc.in_stats; C = its Cout/4 channels, [H,W] = its OUTPUT grid): out = act((x - mean) / sqrt(var + eps)) + addend,
 * biased variance as nn.InstanceNorm2d. */
size_t tcs_deconv_in_stats_bytes(int B, int C, int H_in, int W_in);
int tcs_instance_norm_apply_s16(const void* x, int B, int groups, int H, int W, int act, const void* addend, int addend_groups,
                                void* out, int out_groups, const void* in_stats, int C, float eps, tcs_stream_t stream);
/* Tap partials (tcs_conv_s16_desc.tap_*): weights [nout][C][3][3] * 2^scale_log2 -> fp16-split MFMA A fragments in the producer's
 * accumulator channel order, rows o*9 + t (tcs_tap_weights_floats floats; choose scale_log2 like tcs_pack_conv_weight_f16x3);
 * tcs_taps_sum: out[B,nout,H,W] = (addend + bias + sum over tiles and in-image taps) * scale (addend, bias nullable);
 * tcs_flow_taps_step_grads = tcs_flow_step_grads with delta = bias[0] + taps (FlowHead's output, also written to delta_out when
 * non-NULL); tcs_taps_propagate_s16 = tcs_propagate_disparity_s16 with grad = (g5 + bias2 + taps) * post_scale (core/update.py:213),
 * also written to grad_out [B,2,H,W] when non-NULL. */
size_t tcs_tap_weights_floats(int nout, int C);
int tcs_pack_tap_weights(const float* w_oihw, int nout, int C, int scale_log2, float* packed, tcs_stream_t stream);
int tcs_taps_sum(const float* taps, int ntile, int nout, const float* bias, const float* addend, float scale, int B, int H, int W, float* out,
                 tcs_stream_t stream);
int tcs_flow_taps_step_grads(const float* coords1, const float* taps, int ntile, const float* bias, int B, int H, int W, float scale,
                             float* disp_q, float* grad, float* cands, float* delta_out, tcs_stream_t stream);
int tcs_taps_propagate_s16(const float* taps, int ntile, const float* bias2, const float* g5, float post_scale, const float* disp, int B, int H,
                           int W, float* grad_out, float* cand9, void* out16, int out_groups, tcs_stream_t stream);
int tcs_propagate_disparity_s16(const float* grad, const float* disp, int B, int H, int W, float* cand9, void* out16, int out_groups,
                                tcs_stream_t stream);
int tcs_s16_set_channel(const float* x, int B, int H, int W, void* s16, int groups_total, int channel, tcs_stream_t stream);
int tcs_softmax_blend_s16(const float* logits9, const float* cand, int cand_ctot, const float* disp_q,
                          int B, int H, int W, float* refined, float* delta_disp, float* coords1, float* flow_x,
                          void* flow_x_s16, int flow_x_s16_groups, int flow_x_s16_channel, tcs_stream_t stream);

/* HiddenstateUpdater.forward (core/update.py:57-68) as one launch: x = W2.LeakyReLU(w1*delta + b1) + b2; z,r = sigmoid(Wzr.[h,x] + bzr);
 * q = tanh(Wq.[r*h,x] + bq); h <- z*h + (1-z)*q, in place on the S16 hidden state (128 channels).  The three 1x1 weight
 * matrices are A-fragment images from tcs_pack_weight_frags: `natural_channels` leading input channels in S16 order (operands read
 * from an S16 tensor), the rest in accumulator order (operands handed over in registers by the preceding layer):
 * W2: natural 64; Wzr [256 x 192]: natural 0 (h is fetched in accumulator order too); Wq [128 x 192]: natural 0. */
size_t tcs_weight_frags_bytes(int Cout, int Cin);
int tcs_pack_weight_frags(const float* w_oi, int Cout, int Cin, int natural_channels, int scale_log2, void* packed, tcs_stream_t stream);
int tcs_hidden_update_s16(void* h, int h_groups, const float* delta, const float* w1, const float* b1, const void* W2, const float* b2,
                          float unscale2, const void* Wzr, const float* bzr, float unscale_zr, const void* Wq, const float* bq,
                          float unscale_q, int B, int H, int W, tcs_stream_t stream);

/* nn.Conv2d / ConvGRU step on S16 activations (core/update.py:16-17,26-36,57-68,77-87,103-111,198-214,291-305) */
int tcs_conv2d_s16(const tcs_conv_s16_desc* desc, tcs_stream_t stream);

/* n (1 or 2) INDEPENDENT convolutions of tcs_conv2d_s16 — no output of one is an input of another; same batch size — issued as ONE
 * launch where a grouped kernel exists for their tile instances (the pairs of the refinement loop: BasicMotionEncoder.convc2 | convf2
 * core/update.py:105-108, the second layers of DispGradPredictor's stems :200-205, DispRefine.context_compress | disp_f_stem :293-297),
 * otherwise one after the other.  Results are bit-equal to n calls of tcs_conv2d_s16 in either case; what the grouped launch saves is the
 * cross-queue dependency a fork / join of two graph branches costs.  tcs_conv2d_s16_group_fused: 1 when the two
 * descriptors would run as one launch (no launch is made). */
int tcs_conv2d_s16_group(const tcs_conv_s16_desc* const* descs, int n, tcs_stream_t stream);
int tcs_conv2d_s16_group_fused(const tcs_conv_s16_desc* const* descs, int n);

#ifdef __cplusplus
}
#endif
#endif /* TCS_MI355_H */
