/* segsde_hip.h -- C ABI of libsegsde_hip.so: hand-written gfx950 (MI355X) kernels for the joint
 * segmentation + self-supervised-depth training hot path of
 * lhoyer/improving_segmentation_with_selfsupervised_depth.
 *
 * Boundary rules
 *   - plain pointers + sizes; every pointer is DEVICE memory (the caller -- PyTorch-ROCm in the Python host
 *     layer -- owns allocation); no torch types; `stream` is a hipStream_t passed as void*; nothing syncs.
 *   - return 0 on success, a SEGSDE_ERR_* code (<0) for bad arguments, or a positive hipError_t.
 *   - activations are NHWC fp32 ("rows" = pixels, `ld*` = floats between consecutive pixels, which lets a
 *     tensor be a channel slice of a wider buffer); images / disparities of the loss path are NCHW planar
 *     fp32 exactly as the reference's data loader hands them over.
 *   - workspaces are caller-provided; `*_workspace()` return the bytes needed.
 *
 * The reference is pure Python on torch.nn; it has no FFI.  Each entry point therefore cites the reference
 * call site(s) whose ATen op sequence it replaces (paths relative to /root/reference).  The ctypes binding a
 * maintainer would add is shown in INTEGRATION.md and shipped in
 * improving_segmentation_with_selfsupervised_depth_amd/_lib.py.
 */
#ifndef SEGSDE_HIP_H
#define SEGSDE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGSDE_ABI_VERSION 13

enum { SEGSDE_ERR_NULL = -1, SEGSDE_ERR_SHAPE = -2, SEGSDE_ERR_WORKSPACE = -3, SEGSDE_ERR_UNSUPPORTED = -4 };
enum { SEGSDE_ACT_NONE = 0, SEGSDE_ACT_RELU = 1, SEGSDE_ACT_ELU = 2, SEGSDE_ACT_SIGMOID = 3 };
enum { SEGSDE_DTYPE_F32 = 0, SEGSDE_DTYPE_I64 = 1, SEGSDE_DTYPE_U8 = 2 };
enum { SEGSDE_PAD_ZERO = 0, SEGSDE_PAD_REFLECT = 1, SEGSDE_PAD_REFLECT_ADJOINT = 2 };

int segsde_abi_version(void);

/* ------------------------------------------------------------------------------------------------ *
 * Convolutions (implicit GEMM on v_mfma_f32_32x32x2_f32)                                           *
 * ------------------------------------------------------------------------------------------------ */
typedef struct segsde_conv_desc {
  int B, H, W;      /* batch; spatial size of the (virtual) conv input: after the optional x2 upsample, before padding */
  int C0, C1;       /* channels of source 0 / source 1 (C1 = 0: single source); the conv sees [src0 | src1]           */
  int ld0, ld1;     /* pixel pitch (floats) of the two sources                                                          */
  int up0;          /* 1: source 0 is stored at (H/2, W/2) and nearest-upsampled x2 on the fly (depth_decoder.py:93-94) */
  int Ho, Wo, Cout; /* output size                                                                                       */
  int ldy, ldy2, nsplit; /* output pitch; optional second destination receiving channels >= nsplit (concat dgrad)       */
  int KH, KW, stride, dil, pad;
  int pad_mode;     /* SEGSDE_PAD_ZERO | SEGSDE_PAD_REFLECT (monodepth_layers.py:133-136) | SEGSDE_PAD_REFLECT_ADJOINT:
                       data-gradient of a reflection-padded 3x3/s1/p1 conv (x0 = dy, dgrad-packed weights, pad = 1)       */
  int in_div;       /* 1; >1 only for data-gradients of strided convs: input coordinate must divide by in_div          */
  int act;          /* fused epilogue activation applied after the bias: SEGSDE_ACT_*                                    */
  int sum2x2;       /* 1 (data-gradient through the x2 upsample): output channels < nsplit are summed over 2x2 pixel
                       blocks in registers and stored to y at (Ho/2, Wo/2); channels >= nsplit go to y2 per pixel.
                       Returns SEGSDE_ERR_UNSUPPORTED when the shape cannot take the fused path.                         */
  int accumulate;   /* 1: y += result instead of y = result (the gradient of a tensor with a second consumer -- a residual
                       block's input -- lands on the gradient that is already there; no separate add pass).  Single
                       destination, no activation; SEGSDE_ERR_UNSUPPORTED when the shape does not take the staged epilogue. */
  int compute;      /* 0: fp32 operands (exact fp32 products, the judged arithmetic).  1: the operands are rounded to fp16
                       inside the kernel and multiplied on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- what
                       torch.cuda.amp.autocast makes of the reference's convolutions under `amp: True` (train.py:468,502).
                       Launches whose shape does not take the LDS-DMA loop compute in fp32 whatever this says.            */
} segsde_conv_desc;

/* y[b,ho,wo,n] = act(bias[n] + sum_{kh,kw,c} x[b, ho*stride-pad+kh*dil, wo*stride-pad+kw*dil, c] * wpack[n][kh][kw][c])
 * Replaces: every nn.Conv2d on the path -- torchvision ResNet convs reached from models/resnet_encoder.py:93-99,
 * Conv3x3 (ReflectionPad2d+Conv2d, models/monodepth_layers.py:127-142) incl. the upsample+torch.cat in front of it
 * (models/depth_decoder.py:93-101), ASPP convs (models/model_parts.py:9-25), SelfAttention convs (:38-39),
 * segmentation heads (models/joint_segmentation_depth_decoder.py:35-53,111-116), PoseDecoder convs
 * (models/pose_decoder.py:29-33).  Run on dgrad-packed weights (segsde_pack_weight(for_dgrad=1)) with
 * pad' = (K-1)*dil - pad, stride 1 and in_div = stride it is the data-gradient of the same convolution. */
int segsde_conv2d_forward(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                          const float* bias, float* y, float* y2, void* stream);
/* Same, and additionally leaves the batch statistics of the output (no bias, no activation) as per-tile partials in
 * stats[rows][2][Cout] doubles (rows = segsde_conv2d_stats_rows(d) > 0; sum and sum of squares per row slice of a
 * 128-pixel tile): the statistics of the BatchNorm that follows (models/resnet_encoder.py conv -> bn pairs) without
 * another pass over the tensor.  Finish with segsde_bn_stats_from_partials.  Returns SEGSDE_ERR_UNSUPPORTED when the
 * shape cannot fuse (rows == 0). */
long segsde_conv2d_stats_rows(const segsde_conv_desc* d);
int segsde_conv2d_forward_stats(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                const float* bias, float* y, float* y2, double* stats, void* stream);
/* Data-gradient launch that also applies the backward of the activation which produced the tensor it differentiates with
 * respect to (models/monodepth_layers.py:108-124: ConvBlock = conv -> ELU; the next layer's data-gradient lands on the ELU
 * output): channels < nsplit of y are multiplied by act'(act_out) (act_out = that saved activation output, same row /
 * channel order as y, pitch act_ld, kind SEGSDE_ACT_RELU / ELU / SIGMOID) in the epilogue -- the separate activation-backward
 * pass over the tensor (segsde_act_backward) is not needed.  act_out = NULL: identical to segsde_conv2d_forward_stats.
 * SEGSDE_ERR_UNSUPPORTED when this shape cannot fuse it (the caller then runs segsde_act_backward itself). */
int segsde_conv2d_dgrad_actgrad(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                const float* bias, float* y, float* y2, double* stats, const float* act_out, int act_ld,
                                int act_kind, void* stream);

/* dW (OIHW, the state_dict layout) of the convolution described by d, given dy [B,Ho,Wo,Cout] (pitch lddy). */
size_t segsde_conv2d_wgrad_workspace(const segsde_conv_desc* d);
int segsde_conv2d_wgrad(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                        float* dw_oihw, float* workspace, size_t workspace_bytes, void* stream);

/* Upsample-folded route of Conv3x3 on [upsample(x0) | x1] (models/depth_decoder.py:93-101 with ReflectionPad2d(1) + 3x3,
 * monodepth_layers.py:127-142; d: up0 = 1, 3x3, stride 1, pad 1, SEGSDE_PAD_REFLECT): on the nearest-upsampled channels the
 * nine taps of an output pixel touch only 2x2 distinct low-resolution pixels, so per output parity class the upsampled half is
 * a 2x2 convolution of the low-resolution tensor with pre-summed weights (4 instead of 9 multiply-adds per upsampled channel;
 * mirrored padding becomes clamping).  Same results as the plain route up to the association of the sums; every entry returns
 * SEGSDE_ERR_UNSUPPORTED for shapes it does not take (the caller then uses the plain entry points).
 *   pack    : w_oihw [Cout][Ctot][3][3] -> wfold [4][Cout][2][2][C0] (forward classes (py, px) = (0,0),(0,1),(1,0),(1,1)) and
 *             wdfold [C0][4][4][Cout] (the 4x4 stride-2 kernel of the low-resolution data-gradient)
 *   forward : wpack = segsde_pack_weight(for_dgrad=0) of the same weight (its skip-channel slice is read in place)
 *   dgrad   : d = the FORWARD geometry; dy [B,H,W,Cout] pitch lddy; wdpack = segsde_pack_weight(for_dgrad=1); dx0
 *             [B,H/2,W/2,C0] dense (nullable), dx1 [B,H,W,C1] dense (nullable; accumulate_dx1 != 0: dx1 already holds
 *             another consumer's gradient of the skip tensor and this one is ADDED to it); act_out (nullable): saved activation output
 *             (pitch act_ld, kind SEGSDE_ACT_*) whose derivative multiplies dx0, as in segsde_conv2d_dgrad_actgrad
 *   wgrad   : dw_oihw [Cout][Ctot][3][3], deterministic (fixed-order reduction of the split partials) */
int segsde_upfold_pack(const float* w_oihw, int Cout, int C0, int Ctot, float* wfold, float* wdfold, void* stream);
int segsde_conv2d_forward_upfold(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                 const float* wfold, const float* bias, float* y, void* stream);
int segsde_conv2d_dgrad_upfold(const segsde_conv_desc* d, const float* dy, int lddy, const float* wdpack, const float* wfold,
                               const float* wdfold, float* dx0, float* dx1, int accumulate_dx1, const float* act_out, int act_ld,
                               int act_kind, void* stream);
size_t segsde_conv2d_wgrad_upfold_workspace(const segsde_conv_desc* d);
int segsde_conv2d_wgrad_upfold(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                               float* dw_oihw, float* workspace, size_t workspace_bytes, void* stream);

/* Network stems: nn.Conv2d(3 * n, 64, 7, stride 2, padding 3, bias=False) of torchvision's ResNet / ResNetMultiImageInput
 * (models/resnet_encoder.py:40-52, 90-93) on the LDS-DMA path.  xpad: the bordered input of segsde_nchw_to_nhwc_bordered with
 * pad_top = 3, pad_left = 3, Hp = H + 6, Wp = W + 8 and cp = 4 (3 planes) or 8 (6 planes) channels per pixel; a 32-float
 * reduction chunk is 8 / 4 neighbouring pixels of one input row ("virtual" 32-channel pixels with the real pixel pitch).
 *   pack     : w_oihw [Cout][C][7][7] -> wstem [Cout][7][8 * cp]  (8 pixels x cp channels per tap row; pixel 7 and channels >= C zero)
 *   forward  : y [B][Ho][Wo][Cout], Ho = (H-1)/2+1; stats (nullable): BatchNorm statistics partials, rows = segsde_stem7x7_stats_rows
 *   wgrad    : dw_oihw [Cout][C][7][7], deterministic; no data-gradient (the input is the image) */
int segsde_stem_pack(const float* w_oihw, int Cout, int C, int cp, float* wstem, void* stream);
long segsde_stem7x7_stats_rows(int B, int Hp, int Wp, int cp, int Cout);
int segsde_stem7x7_forward(const float* xpad, int B, int Hp, int Wp, int cp, const float* wstem, int Cout, float* y, double* stats,
                           void* stream);
size_t segsde_stem7x7_wgrad_workspace(int B, int Hp, int Wp, int cp, int Cout);
int segsde_stem7x7_wgrad(const float* xpad, int B, int Hp, int Wp, int cp, const float* dy, int lddy, int Cout, int C,
                         float* dw_oihw, float* workspace, size_t workspace_bytes, void* stream);

/* OIHW -> [O][KH][KW][I] (for_dgrad=0) or [I][KH][KW][O] spatially flipped (for_dgrad=1). */
int segsde_pack_weight(const float* w_oihw, float* out, int O, int I, int KH, int KW, int for_dgrad, void* stream);
/* both packs of one weight tensor in a single launch (training forward: the data-gradient pack is kept for backward) */
int segsde_pack_weight_both(const float* w_oihw, float* out_fwd, float* out_dgrad, int O, int I, int KH, int KW,
                            void* stream);
/* The same for many weights in one launch (a training step re-packs every convolution weight of the model).  jobs_device:
 * njobs + 1 entries in DEVICE memory; entry j packs one weight with blocks [block0, next entry's block0) of the launch, the
 * last entry is a sentinel whose block0 is total_blocks.  reserved = 1 selects the transposing path for a weight with at
 * most 9 taps: the job must then own exactly ceil(O/32) * ceil(I/32) blocks (one per 32 x 32 channel block); reserved = 0:
 * any number of blocks, one element per thread and trip. */
typedef struct segsde_pack_job {
  const float* w; float* fwd; float* dgrad;
  int O, I, KH, KW, block0, reserved;
} segsde_pack_job;
int segsde_pack_weight_both_multi(const segsde_pack_job* jobs_device, int njobs, int total_blocks, void* stream);

/* Adds to dx the gradient that entered the mirrored padding cells of a reflection-padded 3x3 stride-1 conv
 * (autograd of nn.ReflectionPad2d(1), models/monodepth_layers.py:134,140); wdpack = segsde_pack_weight(for_dgrad=1).
 * segsde_conv2d_forward(pad_mode = SEGSDE_PAD_REFLECT_ADJOINT) already includes it. */
int segsde_reflect_dgrad_fix(const float* dy, int lddy, const float* wdpack, float* dx, int lddx, float* dx2, int lddx2,
                             int nsplit, int B, int H, int W, int Cin, int Cout, void* stream);

/* ------------------------------------------------------------------------------------------------ *
 * BatchNorm / activations / pooling / resampling (HBM-bound, NHWC)                                 *
 * ------------------------------------------------------------------------------------------------ */
/* Training-mode batch statistics over M rows (nn.BatchNorm2d.forward in train(), every BN on the path):
 * mean[c], invstd[c] = 1/sqrt(biased var + eps); running stats updated with `momentum` using the unbiased var;
 * num_batches_tracked (nullable, device int64 scalar) is incremented by one, as nn.BatchNorm2d does per training forward. */
size_t segsde_bn_stats_workspace(long M, int C);
int segsde_bn_stats(const float* x, int ldx, long M, int C, float* mean, float* invstd, float* running_mean,
                    float* running_var, float momentum, float eps, int64_t* num_batches_tracked, void* workspace,
                    size_t workspace_bytes, void* stream);
/* Eval mode: mean = running_mean, invstd = 1/sqrt(running_var + eps). */
int segsde_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                         float* invstd, void* stream);
/* y = dropout(act(gamma*(x-mean)*invstd + beta + residual)); dropout keeps with prob 1-p and scales by 1/(1-p)
 * (nn.Dropout(0.5) in ASPP.project, models/model_parts.py:21-25), mask = hash(seed, element). */
int segsde_bn_apply(const float* x, int ldx, long M, int C, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, const float* residual, int ldr, float* y, int ldy, int act, float drop_p,
                    uint64_t seed, void* stream);
/* Backward of segsde_bn_apply + batch statistics.  Phase 1 reduces dgamma/dbeta (sums[0..C) = sum dz*xhat,
 * sums[C..2C) = sum dz) where dz = dy * dropout_mask * act'(y); phase 2 writes dx (and dres = dz if non-null).
 * batch_stats=0 (eval-mode BN): dx = gamma*invstd*dz. */
/* mean / invstd (+ running-statistics update, exactly as segsde_bn_stats) from the partial sums a
 * segsde_conv2d_forward_stats launch left behind; workspace: segsde_bn_stats_from_partials_workspace(C) bytes. */
size_t segsde_bn_stats_from_partials_workspace(int C);
int segsde_bn_stats_from_partials(const double* partials, long rows, long M, int C, float* mean, float* invstd,
                                  float* running_mean, float* running_var, float momentum, float eps,
                                  int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, void* stream);
size_t segsde_bn_backward_workspace(long M, int C);
/* y may be NULL ("remask"): for act = none, or act = ReLU with no residual / dropout (then beta is required with gamma),
 * the activation mask is recomputed from x exactly as the forward kernel formed it and the saved output is not read. */
int segsde_bn_backward(const float* dy, int lddy, const float* y, int ldy, const float* x, int ldx, long M, int C,
                       const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                       float drop_p, uint64_t seed, int batch_stats, float* dgamma, float* dbeta, float* dx, int lddx,
                       float* dres, int lddres, void* workspace, size_t workspace_bytes, void* stream);
/* dz = dy * act'(y) (ELU / ReLU / sigmoid via the saved output, as the reference's in-place ops do);
 * dbias[c] = sum_rows dz (nullable).  Replaces autograd of nn.ELU / nn.ReLU / torch.sigmoid + conv bias grad. */
size_t segsde_colsum_workspace(long M, int C);
int segsde_act_backward(const float* dy, int lddy, const float* y, int ldy, long M, int C, int act, float* dz, int lddz,
                        float* dbias, void* workspace, size_t workspace_bytes, void* stream);
int segsde_colsum(const float* x, int ldx, long M, int C, float* out, void* workspace, size_t workspace_bytes,
                  void* stream);

/* nn.MaxPool2d(3, 2, 1) (models/resnet_encoder.py:96); idx keeps the winning tap (first max in scan order). */
int segsde_maxpool3x3s2_forward(const float* x, int B, int H, int W, int C, float* y, uint8_t* idx, void* stream);
/* accumulate != 0: dx already holds another consumer's gradient of the pooled tensor and this one is added to it */
int segsde_maxpool3x3s2_backward(const float* dy, const uint8_t* idx, int B, int H, int W, int C, float* dx, int accumulate,
                                 void* stream);
/* adjoint of the nearest x2 upsample (models/monodepth_layers.py:202-205): dx[h,w] = sum of the 2x2 block of dy */
int segsde_upsample2x_backward(const float* dy, int lddy, int B, int h, int w, int C, float* dx, int lddx, void* stream);
/* upsample(x) of models/monodepth_layers.py:202-205 as a stand-alone call: x [B,h,w,C] -> y [B,2h,2w,C] */
int segsde_upsample2x_forward(const float* x, int ldx, int B, int h, int w, int C, float* y, int ldy, void* stream);
/* F.interpolate(mode="bilinear") and its adjoint, NHWC (joint_segmentation_depth_decoder.py:64-65,72-73,173-180;
 * torchvision ASPPPooling; loss/loss.py:22-23 with align_corners=1; loss/monodepth_loss.py:72-73 with C=1). */
int segsde_resize_bilinear_forward(const float* x, int ldx, int B, int Hi, int Wi, int C, float* y, int ldy, int Ho, int Wo,
                                   int align_corners, void* stream);
int segsde_resize_bilinear_backward(const float* dy, int lddy, int B, int Hi, int Wi, int C, float* dx, int lddx, int Ho,
                                    int Wo, int align_corners, void* stream);
/* nn.AdaptiveAvgPool2d(1) / out.mean(3).mean(2) (pose_decoder.py:49) and adjoint */
size_t segsde_global_avgpool_workspace(int B, long HW, int C);   /* bytes; 0: the launch needs none */
int segsde_global_avgpool_forward(const float* x, int ldx, int B, long HW, int C, float* y, void* ws, size_t ws_bytes,
                                  void* stream);
int segsde_global_avgpool_backward(const float* dy, int B, long HW, int C, float* dx, int lddx, void* stream);
/* SelfAttention gate y = f * sigmoid(a) (models/model_parts.py:44-46) and adjoint */
int segsde_gate_forward(const float* f, const float* a, long n, float* y, void* stream);
int segsde_gate_backward(const float* dy, const float* f, const float* a, long n, float* df, float* da, void* stream);
/* out = alpha*x + beta*y over n contiguous floats (PAD feature merge :160-161; EMA update train.py:346-358) */
int segsde_axpby(long n, float alpha, const float* x, float beta, const float* y, float* out, void* stream);
/* same with the scalars read from device memory (gradient scaling by upstream 0-dim tensors without a host sync) */
int segsde_axpby_dev(long n, const float* alpha, const float* x, const float* beta, const float* y, float* out, void* stream);
/* channel-slice copy dst[m, 0..C) = src[m, 0..C) (torch.cat of the ASPP branches, models/model_parts.py:31) */
int segsde_copy_channels(const float* src, int lds, float* dst, int ldd, long M, int C, void* stream);
/* nn.Dropout2d inside ConvBlock (models/monodepth_layers.py:117-119, depth_args.dropout > 0): y[b,p,c] = x[b,p,c] * scale[b,c]
 * with scale = 0 or 1/(1-p) per (sample, channel); the same call is its adjoint. */
int segsde_scale_channels(const float* x, int ldx, int B, long HW, int C, const float* scale, float* y, int ldy, void* stream);
/* nn.Dropout(p) on [M, C] rows (JointSegDepthDecoder's layer_dropout on the stacked features,
 * models/joint_segmentation_depth_decoder.py:50): y = x / (1 - p) where the counter-based draw of (seed, element index) keeps the
 * element, else 0; the same call with the same seed on the gradient is its adjoint. */
int segsde_dropout(const float* x, int ldx, long M, int C, float p, uint64_t seed, float* y, int ldy, void* stream);
/* NCHW image -> NHWC with the encoder's input normalisation (x - mean) / std (models/resnet_encoder.py:92);
 * mean = 0, std = 1 gives a plain layout change.  All ldy channels of every pixel are written: channels C..ldy-1 become
 * zero (the 3 / 6-channel network input padded to 4 / 8).  nhwc_to_nchw is the inverse layout change. */
int segsde_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float mean, float std, float* y, int ldy, void* stream);
/* The network-input edge for the stems: (x - mean) / sd of NCHW planes written as 4 / 8-channel pixels (ldy) into the interior of a
 * zero-bordered [B][Hp][Wp][ldy] tensor (interior at (pad_top, pad_left)): the stem's zero padding, materialised, so that
 * segsde_stem7x7_forward needs no padding logic (models/resnet_encoder.py:90-93: normalisation, then conv1 with padding 3). */
int segsde_nchw_to_nhwc_bordered(const float* x, int B, int C, int H, int W, float mean, float sd, float* y, int ldy, int pad_top,
                                 int pad_left, int Hp, int Wp, void* stream);
int segsde_nhwc_to_nchw(const float* x, int ldx, int B, int C, int H, int W, float* y, void* stream);

/* Winograd F(2x2,3x3) route of a stride-1 3x3 convolution with many channels whose padding equals its dilation (the conv2 of
 * every bottleneck, models/resnet_encoder.py:90-101 via torchvision's Bottleneck, dilation 2 in layer4 of the dilated ResNet;
 * the decoder's Conv3x3 on [x | skip] at the bottleneck resolution, models/depth_decoder.py:93-101).  The same call computes
 * the data-gradient of a zero-padded convolution when handed dY, the transposed geometry and the data-gradient pack.
 * segsde_winograd_pack: OIHW -> U = G g G^T as [16][Cout][Cin] (forward) and [16][Cin][Cout] of the flipped kernel
 * (data-gradient); either output may be NULL.  segsde_winograd_pack_multi: the packs of many weights in one launch (device-
 * resident job table; job j owns blocks [block0_j, block0_{j+1}), 2 * ceil(O * I / 256) each).  segsde_conv2d_winograd: input
 * transform (x1 nullable: channels C0.. of the concat), sixteen position GEMMs as one launch of the implicit-GEMM kernel,
 * output transform with bias (nullable) + activation d->act; `stats` (nullable) receives
 * [segsde_conv2d_winograd_stats_rows(d)][2][Cout] doubles for segsde_bn_stats_from_partials.  Zero or mirrored padding (mirrored:
 * dilation 1 only), H and W multiples of 2 * dilation, (C0 + C1) % 32 == 0, C0 % 4 == 0, Cout % 64 == 0; SEGSDE_ERR_UNSUPPORTED
 * otherwise (callers use segsde_conv2d_forward). */
typedef struct segsde_wino_job {
  const float* w;    /* OIHW 3x3 weight */
  float* u_fwd;      /* [16][O][I]  (reserved & 1: [16][I][O], the layout of segsde_conv2d_winograd_fused) */
  float* u_dgrad;    /* [16][I][O]  (reserved & 1: [16][O][I]) */
  int O, I, block0, reserved;
} segsde_wino_job;
size_t segsde_conv2d_winograd_workspace(const segsde_conv_desc* d);
long segsde_conv2d_winograd_stats_rows(const segsde_conv_desc* d);
int segsde_winograd_pack(const float* w_oihw, int Cout, int Cin, float* u_fwd, float* u_dgrad, void* stream);
int segsde_winograd_pack_multi(const segsde_wino_job* jobs_device, int njobs, int total_blocks, void* stream);
int segsde_conv2d_winograd(const segsde_conv_desc* d, const float* x0, const float* x1, const float* u_pack, const float* bias,
                           float* y, double* stats, float* v_keep, void* workspace, size_t workspace_bytes, void* stream);
/* the weight gradient on the same route: dU_p = V_p^T (A dY A^T)_p as sixteen position GEMMs in one launch of the
 * weight-gradient kernel (split boundaries on the position boundaries), dW = G^T dU G written as OIHW.  d: the FORWARD geometry.
 * v_keep of the forward call (nullable; 16 * Tp * (C0+C1) floats, Tp = B*H/2*W/2 rounded up to a multiple of 128) receives the
 * transformed input; handed back as v_saved
 * (nullable) the weight gradient skips its own input transform. */
size_t segsde_conv2d_wgrad_winograd_workspace(const segsde_conv_desc* d);
int segsde_conv2d_wgrad_winograd(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                                 const float* v_saved, float* dw_oihw, void* workspace, size_t workspace_bytes, void* stream);
/* Winograd F(2x2,3x3) with both transforms inside ONE kernel (no V / M tensors): the 64- and 128-channel conv2 of layer1 / layer2
 * (models/resnet_encoder.py:90-101 via torchvision's Bottleneck / BasicBlock) and the decoder's single-source Conv3x3
 * (models/monodepth_layers.py:127-142, reflect = 1: mirrored padding, forward only) -- 3x3, stride 1, padding 1, one source, H and W
 * even, C % 64 == 0, Cout % 64 == 0.  segsde_winograd_fused_pack: OIHW -> the transformed weights U (16 K N floats) in the layout
 * the kernel reads -- blocked: [transform row][k][block of 64 n][32 lanes][4 positions of the row][2 halves of the block], the
 * eight B operands of a lane and step contiguous (round 5; SEGSDE_WINO_FUSED_UBLK=0 in the environment of the process selects
 * the sixteen [K][N] planes of round 4 for packs and kernel alike); flip = 0: the forward pack
 * (K = Cin, N = Cout), flip = 1: the data-gradient pack of the spatially flipped kernel (K = Cout, N = Cin; call the convolution
 * with x = dY, C = Cout, Cout = Cin).  y = act(Y + bias) (bias nullable), or y += Y (accumulate = 1: no bias / activation /
 * statistics; the gradient collector of DESIGN.md 3.2f); stats (nullable):
 * [segsde_winograd_fused_stats_rows(B, H, W)][2][Cout] doubles for segsde_bn_stats_from_partials.  SEGSDE_ERR_UNSUPPORTED outside
 * these shapes (segsde_winograd_fused_ok tells beforehand). */
int segsde_winograd_fused_ok(int B, int H, int W, int C, int Cout);
long segsde_winograd_fused_stats_rows(int B, int H, int W);
int segsde_winograd_fused_pack(const float* w_oihw, int Cout, int Cin, int flip, float* u_kn, void* stream);
int segsde_conv2d_winograd_fused(const float* x, int ldx, int B, int H, int W, int C, int reflect, const float* u_kn, int Cout,
                                 const float* bias, int act, float* y, int ldy, int accumulate, double* stats, void* stream);
/* Round 5.  The same kernel on the decoder's virtual input [up2x?(x0) | x1] (models/depth_decoder.py:88-101: nearest-upsampled
 * previous block, encoder skip): x0 [B, H >> up0, W >> up0, C0], x1 [B, H, W, C1] or NULL; C0 % 64 == 0, (C0 + C1) % 64 == 0 --
 * upsampling, concat and mirrored padding are index arithmetic of the patch loader, all channels at 16 / 36 of the multiply-adds. */
int segsde_conv2d_winograd_fused2(const float* x0, int ld0, int C0, int up0, const float* x1, int ld1, int C1, int B, int H, int W,
                                  int reflect, const float* u_kn, int Cout, const float* bias, int act, float* y, int ldy,
                                  double* stats, void* stream);
/* Data-gradient on the one-kernel route with the direct route's epilogues: dx (+)= conv(dy, flipped pack) * act'(act_out)
 * (act_out nullable: the saved output of the activation that produced this convolution's input, ConvBlock's ELU,
 * monodepth_layers.py:108-125; accumulate: added onto what dx holds).  Zero padding; for the reflection-padded
 * Conv3x3 follow it with segsde_reflect_adjoint_borders (conv_igemm.hip): the gradient that entered the mirrored padding cells, as
 * four border launches + corner terms ADDED onto rows 1 / H-2 and columns 1 / W-2 of y (d = the descriptor of
 * segsde_conv2d_dgrad_actgrad with pad_mode = SEGSDE_PAD_REFLECT_ADJOINT; act_out as above).  ldu: row pitch of ud_kn (Cin; or,
 * for the gradient of ONE source of a two-source convolution -- the decoder's skip input, weight channels [C0, C0 + C1) -- the
 * full pack's width, ud_kn pointing at the slice's first 64-filter block (blocked layout: (first column / 64) * 256 floats into
 * the pack; the slice starts on a multiple of 64); segsde_reflect_adjoint_borders2 takes the matching slice of the forward pack
 * through ldw). */
int segsde_conv2d_winograd_fused_dgrad(const float* dy, int lddy, int B, int H, int W, int Cout, const float* ud_kn, int ldu, int Cin,
                                       float* dx, int lddx, int accumulate, const float* act_out, int act_ld, int act_kind, void* stream);
int segsde_reflect_adjoint_borders(const segsde_conv_desc* d, const float* dy, const float* wdpack, float* y, const float* act_out,
                                   int act_ld, int act_kind, void* stream);
/* The same mirrored-padding terms by a kernel of their own (csrc/winograd_fused.hip: two launches -- row lines, column lines +
 * corners -- of a 32-pixel x Cin MFMA kernel instead of five launches of the image-sized implicit-GEMM machinery); wpack = the
 * FORWARD pack [Cout][3][3][Cin] of segsde_pack_weight(for_dgrad = 0).  Cin % 32 == 0, Cout % 32 == 0, H, W >= 4. */
int segsde_reflect_adjoint_borders2(const float* dy, int lddy, const float* wpack, int ldw, float* dx, int lddx, const float* act_out,
                                    int act_ld, int act_kind, int B, int H, int W, int Cin, int Cout, void* stream);
/* 1 when segsde_reflect_adjoint_borders takes the descriptor (act_ld: pixel pitch of act_out, 0 without one) -- asked before the
 * zero-padded launch writes dx */
int segsde_reflect_adjoint_borders_ok(const segsde_conv_desc* d, int act_ld);
/* Weight gradient on the one-kernel Winograd scheme (csrc/winograd_wgrad.hip): dU_p = V_p^T (A dY A^T)_p with both operands formed
 * from the raw input patches and the raw output gradient in LDS, split over tile ranges into slabs, folded deterministically and
 * transformed back (dW = G^T dU G, OIHW).  d as for segsde_conv2d_wgrad: 3x3 / stride 1 / padding 1 / dilation 1, zero or mirrored
 * padding, one or two sources, up0 (the decoder's [upsample(x0) | x1], models/depth_decoder.py:88-101); C0 % 32 == 0, C1 % 32 == 0,
 * Cout % 64 == 0, H and W even.  Workspace 0 = shape not taken. */
size_t segsde_conv2d_wgrad_winograd_fused_workspace(const segsde_conv_desc* d);
int segsde_conv2d_wgrad_winograd_fused(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                                       float* dw_oihw, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ *
 * Pose: axis-angle + translation -> 4x4 (models/monodepth_layers.py:30-105)                         *
 * ------------------------------------------------------------------------------------------------ */
int segsde_pose_matrix_forward(const float* axisangle, const float* translation, int B, int stride, int invert, float* M,
                               void* stream);
int segsde_pose_matrix_backward(const float* axisangle, const float* translation, const float* dM, int B, int stride,
                                int invert, float* daxisangle, float* dtranslation, void* stream);

/* ------------------------------------------------------------------------------------------------ *
 * Monodepth photometric loss (loss/monodepth_loss.py, models/monodepth_layers.py:18-27,145-254)     *
 * ------------------------------------------------------------------------------------------------ */
/* generate_images_pred for one (scale, frame): bilinear-upsample disp_s (align_corners=False) -> depth ->
 * backproject with inv_K -> project with K*T -> grid_sample(src, bilinear, border, align_corners=True).
 * color: [B,3,H,W]; optional outputs grid [B,H,W,2] (normalised, as outputs[("sample",f,s)]) and depth [B,1,H,W]. */
int segsde_warp_forward(const float* disp, int hs, int ws, const float* inv_K, const float* K, const float* T,
                        const float* src, int B, int H, int W, float min_depth, float max_depth, float* color,
                        float* grid, float* depth, void* stream);
/* adjoint w.r.t. the upsampled disparity (g_disp_up [B,H,W], accumulated +=) and T (gT [B,4,4], accumulated +=). */
size_t segsde_warp_backward_workspace(int B, int H, int W);
int segsde_warp_backward(const float* gcolor, const float* disp, int hs, int ws, const float* inv_K, const float* K,
                         const float* T, const float* src, int B, int H, int W, float min_depth, float max_depth,
                         float* g_disp_up, float* gT, void* workspace, size_t workspace_bytes, void* stream);
/* compute_reprojection_loss: err[b,h,w] = 0.85*mean_c SSIM(pred,target) + 0.15*mean_c|target-pred| (or L1 only).
 * err / gerr are [B,H,W] planes with `bstride` floats between batches (a channel of a [B,n,H,W] tensor). */
int segsde_reprojection_error_forward(const float* pred, const float* target, int B, int H, int W, int no_ssim,
                                      float* err, long err_bstride, void* stream);
size_t segsde_reprojection_error_backward_workspace(int B, int H, int W);
int segsde_reprojection_error_backward(const float* pred, const float* target, const float* gerr, long gerr_bstride, int B,
                                       int H, int W, int no_ssim, float* gpred, void* workspace, size_t workspace_bytes,
                                       void* stream);
/* per-pixel min over [identity(+1e-5*noise) | reprojection] channels (monodepth_loss.py:136-177) for any number of source
 * frames n_reproj (1..8: the loop over frame_ids[1:]): reproj and ident are [B,n_reproj,H,W], noise [B, avg ? 1 : n_reproj, H, W].
 * ident/noise may be NULL (disable_automasking); avg=1 averages the channels of each group first.  n_ident of the adjoint: nonzero
 * if the forward had identity terms.
 * Outputs: sel [B,H,W] uint8 = argmin index in the combined order, identity_selection [B,H,W] float (nullable),
 * sum_out[0] = sum of the minima (the caller divides by B*H*W). */
size_t segsde_automask_workspace(int B, int H, int W);
int segsde_automask_min_forward(const float* ident, const float* noise, const float* reproj, int n_reproj, int avg, int B,
                                int H, int W, uint8_t* sel, float* identity_selection, float* sum_out, void* workspace,
                                size_t workspace_bytes, void* stream);
int segsde_automask_min_backward(const uint8_t* sel, int n_ident, int n_reproj, int avg, int B, int H, int W, float scale,
                                 float* greproj, void* stream);
/* edge-aware smoothness of the mean-normalised disparity (monodepth_loss.py:182-186, monodepth_layers.py:208-221):
 * out[0] = mean|dx d^|e^{-|dx I|} + mean|dy d^|e^{-|dy I|};  mean_disp [B] is kept for the backward. */
size_t segsde_smoothness_workspace(int B, int h, int w);
int segsde_smoothness_forward(const float* disp, const float* img, int B, int h, int w, float* mean_disp, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);
int segsde_smoothness_backward(const float* disp, const float* img, const float* mean_disp, int B, int h, int w,
                               float scale, float* gdisp, void* workspace, size_t workspace_bytes, void* stream);

/* The same layers by themselves, for callers outside the training path (a script that imports them by name from
 * models/monodepth_layers.py): get_smooth_loss WITHOUT the mean normalisation (monodepth_layers.py:208-221; gdisp is
 * accumulated into), SSIM.forward -> the per-channel loss map clamp((1 - SSIM) / 2, 0, 1) and its adjoint w.r.t. both
 * images (:224-254; gx / gy may be NULL), BackprojectDepth.forward -> cam_points [B,4,H*W] (:169-174) and
 * Project3D.forward -> pix_coords [B,H,W,2] in [-1, 1] (:188-199). */
size_t segsde_smooth_loss_workspace(int B, int h, int w);
int segsde_smooth_loss_forward(const float* disp, const float* img, int B, int h, int w, float* out, void* workspace,
                               size_t workspace_bytes, void* stream);
int segsde_smooth_loss_backward(const float* disp, const float* img, int B, int h, int w, float scale, float* gdisp,
                                void* workspace, size_t workspace_bytes, void* stream);
int segsde_ssim_map_forward(const float* x, const float* y, int B, int C, int H, int W, float* out, void* stream);
int segsde_ssim_map_backward(const float* x, const float* y, const float* gout, int B, int C, int H, int W, float* gx,
                             float* gy, void* stream);
int segsde_backproject_depth(const float* depth, const float* inv_K, int B, int H, int W, float* cam_points, void* stream);
int segsde_project3d(const float* points, const float* K, const float* T, int B, int H, int W, float eps, float* pix_coords,
                     void* stream);
/* Their adjoints (the reference layers are plain differentiable torch code, :169-174 / :188-199; the photometric gradient
 * reaches the depth through BackprojectDepth and the pose through Project3D): d depth [B,H*W] from d cam_points [B,4,H*W];
 * d points [B,4,H*W] (nullable) and d T [B,4,4] (nullable; needs the workspace: per-block partial sums in double, folded in
 * block order) from d pix_coords [B,H,W,2].  The intrinsics are data: no gradient for K / inv_K. */
int segsde_backproject_depth_backward(const float* g_cam_points, const float* inv_K, int B, int H, int W, float* g_depth,
                                      void* stream);
size_t segsde_project3d_backward_workspace(int B, int H, int W);
int segsde_project3d_backward(const float* points, const float* K, const float* T, const float* g_pix, int B, int H, int W,
                              float eps, float* g_points, float* g_T, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ *
 * Segmentation loss and DepthMix / ClassMix (loss/loss.py:17-37, loader/transformsgpu.py:33-47,     *
 * loader/transformmasks.py:27-41, train.py:585-604)                                                 *
 * ------------------------------------------------------------------------------------------------ */
/* F.cross_entropy over NHWC logits (pitch ld) with ignore_index; out[0] = sum_i w_i*nll_i, out[1] = sum_i w_i over
 * non-ignored pixels (w_i = class_weight[t_i] or 1; times pixel_weights[i] if given). */
size_t segsde_cross_entropy_workspace(long M);
int segsde_cross_entropy_forward(const float* logits, int ld, long M, int C, const int64_t* target, int64_t ignore_index,
                                 const float* class_weight, const float* pixel_weights, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* dlogits[i,c] = scale * w_i * (softmax_c - [c == t_i]) (0 for ignored pixels) */
int segsde_cross_entropy_backward(const float* logits, int ld, long M, int C, const int64_t* target, int64_t ignore_index,
                                  const float* class_weight, const float* pixel_weights, const float* scale,
                                  float* dlogits, int lddl, void* stream);
/* out_i = m_i*x_i + (1-m_i)*x_{(i+1)%B}; mask is int64 or float32 [Bm,H,W] (Bm = B, or B/2: paired-halves branch);
 * element strides let x be NCHW or channels-last.  Bit-exact with the reference's fp32 op sequence. */
int segsde_mix(const void* mask, int mask_is_int64, int Bm, const float* x, int B, int C, int H, int W, long sb, long sc,
               long sh, long sw, float* out, void* stream);
int segsde_mix_labels(const int64_t* mask, const int64_t* target, int B, int H, int W, int64_t* out, void* stream);
/* depthcomp: m_i = (d_i >= d_{(i+1)%B} - margin) * (d_i >= ft_i) -> int64 [B,H,W] (train.py:585-604, generalised partner);
 * ft_i = fg_threshold_per_sample[i] (nullable DEVICE [B]: the per-image draw of train.py:592-599) or fg_threshold for all. */
int segsde_depthcomp_mask(const float* depths, int B, long HW, float margin, float fg_threshold,
                          const float* fg_threshold_per_sample, int64_t* mask, void* stream);
/* generate_depth_mask: depth >= thr (one threshold), or (depth >= min(t)) <= max(t) as the reference writes it */
int segsde_depth_threshold_mask(const float* depth, long n, float t1, float t2, int two_thresholds, float* mask, void* stream);
/* generate_class_mask: N[h,w] = #{k : pred[h,w] == classes[k]} */
int segsde_class_mask(const int64_t* pred, long n, const int64_t* classes, int n_classes, int64_t* mask, void* stream);


/* ---- SURVEY.md 8(f) "next" rows: the trainer-side callers of the path ------------------------------------------------ */
/* One chunk of a multi-tensor operation: n <= 65536 contiguous floats of one tensor. */
typedef struct segsde_mt_chunk { float* dst; const float* src; long n; } segsde_mt_chunk;
/* EMA teacher update, train.py:346-358 (Trainer.update_ema_variables): dst = alpha*dst + one_minus_alpha*src for every
 * chunk of the DEVICE-resident table, one launch for all parameter tensors.  alpha / one_minus_alpha: the fp32 roundings
 * of the reference's Python doubles (min(1 - 1/(iteration+1), alpha_teacher) and 1 minus that). */
int segsde_multi_tensor_lerp(const segsde_mt_chunk* table_dev, int nchunks, float alpha, float one_minus_alpha, void* stream);
/* Pseudo labels of Trainer.calc_pseudo_label_loss, train.py:644-651: label = argmax_c prob (first maximum), ignore_index
 * where the maximum is 0; count[0] = #{max >= threshold}; max_prob (nullable) = the maxima; pixel_weight (nullable) =
 * count/(B*HW) broadcast to every pixel (the reference's unlabeled_weight * ones, without the host round trip). */
int segsde_pseudo_label(const float* prob_nchw, int B, int C, long HW, float threshold, int64_t ignore_index, int64_t* label,
                        float* max_prob, unsigned long long* count, float* pixel_weight, void* stream);
/* runningScore.update / _fast_hist (evaluation/metrics.py:12-25): hist[C*gt + pred] += 1 for every pixel with 0 <= gt < C,
 * accumulated into the DEVICE-resident hist[C*C].  Either pred (int64, as train.py:848 computes it) or logits (then the
 * argmax over the C class planes is fused; element strides sb / sc / sp for batch, class and pixel let the tensor be NCHW
 * or channels-last).  Integer atomics: exact. */
int segsde_confusion_update(const float* logits, long sb, long sc, long sp, const int64_t* pred, const int64_t* gt, int B,
                            long HW, int C, unsigned long long* hist, void* stream);

/* Teacher softmax of the unlabeled step, train.py:666 (torch.softmax(logits_u_w, dim=1)): NHWC logits rows (pitch ld)
 * -> class probabilities in NCHW planar layout, the layout segsde_mix / segsde_pseudo_label read. */
int segsde_softmax_nhwc_to_nchw(const float* logits, int ld, int B, long HW, int C, float* out_nchw, void* stream);
/* mix_use_gt, train.py:667-672 (``softmax_u_w[i] = unlabeled_inputs["onehot_lbl"][i]`` for every sample i of the unlabeled batch
 * whose ``is_labeled[i]`` is set): prob_nchw [B,C,HW] is overwritten IN PLACE with the one-hot planes (the loader's layout and
 * dtype, loader/sequence_segmentation_loader.py:237-246: int64 [C,H,W], all zero on ignored pixels; SEGSDE_DTYPE_*) of the
 * flagged samples; is_labeled is a DEVICE uint8 [B] (no host synchronisation).  Unflagged samples are not touched. */
int segsde_onehot_select(float* prob_nchw, const void* onehot, int onehot_dtype, const uint8_t* is_labeled, int B, int C,
                         long HW, void* stream);
/* Online-depth normalisation for the depthcomp mask, train.py:690-697, and the stored depth estimates of
 * DepthEstimator.prepare_depth_estimates, loader/depth_estimator.py:83-91: per sample b, out = (x - min_b) / (max_b - min_b)
 * (the reference's clamp to [min, max] is the identity); minmax (nullable) receives [B][2] = {min_b, max_b}; out_u8
 * (nullable) receives the 8-bit image ToPILImage makes of it (mul(255).byte()).  At least one of out / out_u8. */
size_t segsde_minmax_normalize_workspace(int B, long HW);
int segsde_minmax_normalize(const float* x, int B, long HW, float* out, float* minmax, uint8_t* out_u8, void* workspace,
                            size_t workspace_bytes, void* stream);
/* Fused photometric loss of one scale, both source frames per launch (loss/monodepth_loss.py:104-177 with
 * models/monodepth_layers.py:224-254): the per-stage entry points above chained through LDS tiles and registers.
 *   identity : err(src_f, target) for f = 0, 1 -> ident [B,2,H,W]  (identical for all scales: computed once, :139-147)
 *   forward  : err(pred_f, target), min over [ident (+ noise * 1e-5) | err] (avg: channel means first, :149-156), first
 *              minimum wins -> sel (uint8 index into the concatenation), identity_selection (nullable, :175-177),
 *              sum_out[0] = sum of the minima.  ident / noise nullable (disable_automasking).
 *   backward : d(scale * sum of minima) / d pred_f is formed in the tile (SSIM window statistics recomputed, reflection
 *              fold included) and pushed straight through the warp adjoint: g_disp_up [B,H,W] is WRITTEN (both frames
 *              summed), gT_f [B,4,4] += weight[0] * dL/dT_f (weight: nullable device scalar = upstream gradient).
 * pred_f are the warped frames [B,3,H,W] (segsde_warp_forward); T_f / src_f the pose and source frame of frame f. */
size_t segsde_photometric_workspace(int B, int H, int W);
int segsde_photometric_identity(const float* src0, const float* src1, const float* target, int B, int H, int W, int no_ssim,
                                float* ident, void* stream);
int segsde_photometric_forward(const float* pred0, const float* pred1, const float* target, const float* ident,
                               const float* noise, int B, int H, int W, int no_ssim, int avg, uint8_t* sel,
                               float* identity_selection, float* sum_out, void* workspace, size_t workspace_bytes,
                               void* stream);
int segsde_photometric_backward(const float* pred0, const float* pred1, const float* target, const uint8_t* sel, int n_ident,
                                const float* disp, int hs, int ws, const float* inv_K, const float* K, const float* T0,
                                const float* T1, const float* src0, const float* src1, int B, int H, int W, float min_depth,
                                float max_depth, int no_ssim, int avg, float scale, const float* weight, float* g_disp_up,
                                float* gT0, float* gT1, void* workspace, size_t workspace_bytes, void* stream);
/* Test-time depth, MonodepthLoss.generate_depth_test_pred, loss/monodepth_loss.py:54-62: bilinear upsample of disp
 * [B,1,hs,ws] to H x W (align_corners=False) and disp_to_depth (monodepth_layers.py:18-27) with the test depth range. */
int segsde_disp_to_depth(const float* disp, int hs, int ws, int B, int H, int W, float min_depth, float max_depth,
                         float* depth, void* stream);

/* strongTransform's colour jitter, loader/transformsgpu.py:10-17 (kornia 0.4.0 ColorJitter(s, s, s, s); third party, parity
 * unpinned): x, y [B,3,HW] planar; params DEVICE [B][4] = {brightness, contrast, saturation, hue} factors; order HOST int[4],
 * a permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue}: clamp(x + (bf - 1)), clamp(x * cf), HSV s *= sf (clamped),
 * HSV h = fmod(h + 2 pi hf, 2 pi). */
int segsde_color_jitter(const float* x, int B, long HW, const float* params, const int* order, float* y, void* stream);
/* strongTransform's blur, loader/transformsgpu.py:20-30 (kornia 0.4.0 GaussianBlur2d, reflect border; parity unpinned):
 * separable -- column pass with the ny (odd) taps wy, then row pass with the nx taps wx -- over `planes` H x W planes;
 * tmp: planes*H*W floats; taps are DEVICE arrays (normalised Gaussian weights, possibly truncated to their non-zero support). */
int segsde_gaussian_blur(const float* x, int planes, int H, int W, const float* wy, int ny, const float* wx, int nx, float* tmp,
                         float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGSDE_HIP_H */
