// Teacher-side pieces of the DepthMix unlabeled step (Trainer.train_step_segmentation_unlabeled, train.py:653-724):
// the class softmax of the teacher's logits (train.py:666) and the per-sample min-max normalisation of the student's
// online disparity that feeds the depthcomp mask (train.py:690-697).  HBM-bound, one pass each.
#include "segsde_common.h"

namespace {
#define ST(s) static_cast<hipStream_t>(s)
constexpr int SM_PIX = 256;

// logits: NHWC rows of C floats at pitch ld (what the segmentation head produces) -> softmax in NCHW planar layout (what
// transformsgpu.mix / calc_pseudo_label_loss consume).  A block stages 256 consecutive pixels through LDS (16-byte loads
// when the rows are dense), one thread per pixel works on its row in LDS (row stride C: odd for the 19 classes, no bank
// conflicts), stores are coalesced per class plane.
__global__ __launch_bounds__(256) void softmax_nhwc_to_nchw_kernel(const float* logits, int ld, long HW, int C, float* out) {
  SEGSDE_SMEM;
  float* sx = reinterpret_cast<float*>(segsde_smem);      // [SM_PIX][C]
  const int b = blockIdx.y;
  const float* lb = logits + (long)b * HW * ld;
  float* ob = out + (long)b * C * HW;
  for (long p0 = (long)blockIdx.x * SM_PIX; p0 < HW; p0 += (long)gridDim.x * SM_PIX) {
    const int np = (int)(HW - p0 < SM_PIX ? HW - p0 : SM_PIX);
    __syncthreads();
    if (ld == C) {
      const float* src = lb + p0 * C;
      const int n = np * C, n4 = n >> 2;
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<float4*>(sx)[e] = reinterpret_cast<const float4*>(src)[e];
        for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) sx[e] = src[e];
      } else {
        for (int e = threadIdx.x; e < n; e += 256) sx[e] = src[e];
      }
    } else {
      for (int e = threadIdx.x; e < np * C; e += 256) { const int r = e / C, c = e - r * C; sx[e] = lb[(p0 + r) * ld + c]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < np) {
      float* x = sx + threadIdx.x * C;
      float mx = x[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) { const float e = expf(x[c] - mx); x[c] = e; se += e; }
      const long p = p0 + threadIdx.x;
      for (int c = 0; c < C; ++c) ob[(long)c * HW + p] = x[c] / se;
    }
  }
}

__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* x, long HW, float* part /*[B][nblk][2]*/) {
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);   // [2][4]
  const int b = blockIdx.y;
  const float* xb = x + (long)b * HW;
  float mn = INFINITY, mx = -INFINITY;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < HW; e += (long)gridDim.x * 256) {
    const float v = xb[e];
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = mn; sh[4 + (threadIdx.x >> 6)] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) { mn = fminf(mn, sh[i]); mx = fmaxf(mx, sh[4 + i]); }
    part[((long)b * gridDim.x + blockIdx.x) * 2] = mn;
    part[((long)b * gridDim.x + blockIdx.x) * 2 + 1] = mx;
  }
}
// every block folds the (<= 512) partials of its sample itself, then out = (x - min) / (max - min): the reference's two
// fp32 ops (train.py:696; its clamp to [min, max] is the identity)
__global__ __launch_bounds__(256) void minmax_apply_kernel(const float* x, long HW, const float* part, int nblk, float* out,
                                                          float* minmax /*[B][2] or null*/, uint8_t* out_u8 /*or null*/) {
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);
  const int b = blockIdx.y;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    mn = fminf(mn, part[((long)b * nblk + i) * 2]); mx = fmaxf(mx, part[((long)b * nblk + i) * 2 + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = mn; sh[4 + (threadIdx.x >> 6)] = mx; }
  __syncthreads();
  mn = fminf(fminf(sh[0], sh[1]), fminf(sh[2], sh[3]));
  mx = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
  if (minmax && blockIdx.x == 0 && threadIdx.x == 0) { minmax[2 * b] = mn; minmax[2 * b + 1] = mx; }
  const float range = mx - mn;
  const float* xb = x + (long)b * HW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < HW; e += (long)gridDim.x * 256) {
    const float v = (xb[e] - mn) / range;
    if (out) out[(long)b * HW + e] = v;
    // the stored depth estimate: torchvision ToPILImage on a float tensor = mul(255).byte() (truncation)
    if (out_u8) out_u8[(long)b * HW + e] = (uint8_t)(v * 255.f);
  }
}
// mix_use_gt (train.py:667-672): for the samples of the unlabeled batch that carry a label, the teacher's softmax planes are
// replaced by the one-hot ground truth (the loader's int64 [C][H][W] planes, all zero on ignored pixels) before the
// argmax / mix / pseudo-label.  The per-sample switch is read from a DEVICE flag array: no host round trip; unlabeled
// samples are left untouched (their blocks return at once), so the result is bit-exact in both branches.
template <typename T>
__global__ __launch_bounds__(256) void onehot_select_kernel(float* prob, const T* onehot, const uint8_t* is_labeled, long CHW) {
  const int b = blockIdx.y;
  if (!is_labeled[b]) return;
  float* pb = prob + (long)b * CHW;
  const T* ob = onehot + (long)b * CHW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < CHW; e += (long)gridDim.x * 256) pb[e] = (float)ob[e];
}
inline int plane_blocks(long HW) { long nb = (HW + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb)); }
}  // namespace

extern "C" int segsde_softmax_nhwc_to_nchw(const float* logits, int ld, int B, long HW, int C, float* out, void* stream) {
  if (!logits || !out) return SEGSDE_ERR_NULL;
  if (B <= 0 || HW <= 0 || C <= 0 || C > 160 || ld < C) return SEGSDE_ERR_SHAPE;
  long nb = (HW + SM_PIX - 1) / SM_PIX;
  nb = nb > 2048 ? 2048 : nb;
  hipLaunchKernelGGL(softmax_nhwc_to_nchw_kernel, dim3((unsigned)nb, B), dim3(256), (size_t)SM_PIX * C * sizeof(float),
                     ST(stream), logits, ld, HW, C, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_onehot_select(float* prob_nchw, const void* onehot, int onehot_dtype, const uint8_t* is_labeled, int B,
                                    int C, long HW, void* stream) {
  if (!prob_nchw || !onehot || !is_labeled) return SEGSDE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SEGSDE_ERR_SHAPE;
  const long CHW = (long)C * HW;
  long nb = (CHW + 1023) / 1024;
  nb = nb > 4096 ? 4096 : nb;
  const dim3 grid((unsigned)nb, B), block(256);
  if (onehot_dtype == SEGSDE_DTYPE_I64)
    hipLaunchKernelGGL(onehot_select_kernel<int64_t>, grid, block, 0, ST(stream), prob_nchw, (const int64_t*)onehot, is_labeled, CHW);
  else if (onehot_dtype == SEGSDE_DTYPE_F32)
    hipLaunchKernelGGL(onehot_select_kernel<float>, grid, block, 0, ST(stream), prob_nchw, (const float*)onehot, is_labeled, CHW);
  else if (onehot_dtype == SEGSDE_DTYPE_U8)
    hipLaunchKernelGGL(onehot_select_kernel<uint8_t>, grid, block, 0, ST(stream), prob_nchw, (const uint8_t*)onehot, is_labeled, CHW);
  else
    return SEGSDE_ERR_SHAPE;
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t segsde_minmax_normalize_workspace(int B, long HW) {
  return (size_t)B * plane_blocks(HW) * 2 * sizeof(float);
}

extern "C" int segsde_minmax_normalize(const float* x, int B, long HW, float* out, float* minmax, uint8_t* out_u8, void* ws,
                                       size_t ws_bytes, void* stream) {
  if (!x || (!out && !out_u8) || !ws) return SEGSDE_ERR_NULL;
  if (B <= 0 || HW <= 0) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_minmax_normalize_workspace(B, HW)) return SEGSDE_ERR_WORKSPACE;
  const int nb = plane_blocks(HW);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb, B), dim3(256), 64, ST(stream), x, HW, (float*)ws);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(minmax_apply_kernel, dim3(nb, B), dim3(256), 64, ST(stream), x, HW, (const float*)ws, nb, out, minmax, out_u8);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
