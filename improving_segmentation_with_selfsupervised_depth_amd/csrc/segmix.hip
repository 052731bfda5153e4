// Segmentation cross-entropy (loss/loss.py:17-37) over NHWC logits, and the DepthMix / ClassMix mask +
// composite kernels (loader/transformsgpu.py:33-47, loader/transformmasks.py:27-41, train.py:585-604).
// Mask / composite arithmetic reproduces the reference's fp32 op sequence exactly (file is compiled with
// -ffp-contract=off: m*x + (1-m)*y must not fuse), so outputs are bit-identical.
#include "segsde_common.h"

namespace {
#define ST(s) static_cast<hipStream_t>(s)
inline int flat_blocks(long n) { long nb = (n + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb)); }
inline int ce_blocks(long n) { long nb = (n + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb)); }

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, int ld, long M, int C, const int64_t* target,
                                                     int64_t ignore, const float* cw, const float* pw, double* part) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  double num = 0.0, den = 0.0;
  for (long m = blockIdx.x * 256L + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const int64_t t = target[m];
    if (t == ignore) continue;
    const float* x = logits + m * ld;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    const float nll = (mx + logf(se)) - x[t];
    float w = cw ? cw[t] : 1.f;
    den += (double)w;
    if (pw) w *= pw[m];
    num += (double)(w * nll);
  }
  const double a = segsde_block_sum(num, sh);
  const double b = segsde_block_sum(den, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
}
__global__ __launch_bounds__(64) void ce_finalize_kernel(const double* part, int n, float* out) {
  if (threadIdx.x != 0) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < n; ++i) { a += part[2 * i]; b += part[2 * i + 1]; }
  out[0] = (float)a; out[1] = (float)b;
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, int ld, long M, int C, const int64_t* target,
                                                     int64_t ignore, const float* cw, const float* pw,
                                                     const float* scale, float* dl, int lddl) {
  const float sc = scale[0];
  for (long m = blockIdx.x * 256L + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const int64_t t = target[m];
    float* o = dl + m * lddl;
    if (t == ignore) { for (int c = 0; c < C; ++c) o[c] = 0.f; continue; }
    const float* x = logits + m * ld;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    float w = (cw ? cw[t] : 1.f) * sc;
    if (pw) w *= pw[m];
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) o[c] = w * (expf(x[c] - mx) * inv - (c == t ? 1.f : 0.f));
  }
}

// Dense logits (row pitch == C, the 19-class head): a block stages 256 consecutive pixels -- one contiguous span -- through
// LDS with 16-byte accesses; one thread per pixel then works on its row in LDS (row stride C = 19 floats: odd, no bank
// conflicts).  The per-thread global version above moves 4 bytes per lane at a 76-byte stride (7x off the HBM roofline).
constexpr int CE_PIX = 256;
__device__ __forceinline__ void ce_stage_in(const float* src, int n, float* sh) {
  const int n4 = n >> 2;
  for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<float4*>(sh)[e] = reinterpret_cast<const float4*>(src)[e];
  for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) sh[e] = src[e];
}

__global__ __launch_bounds__(256) void ce_fwd_dense_kernel(const float* logits, long M, int C, const int64_t* target,
                                                           int64_t ignore, const float* cw, const float* pw, double* part) {
  SEGSDE_SMEM;
  float* sx = reinterpret_cast<float*>(segsde_smem);                    // [CE_PIX][C]
  double* sh = reinterpret_cast<double*>(sx + CE_PIX * C + (CE_PIX * C & 1));   // 8-byte aligned scratch for the block sums
  double num = 0.0, den = 0.0;
  for (long m0 = (long)blockIdx.x * CE_PIX; m0 < M; m0 += (long)gridDim.x * CE_PIX) {
    const int np = (int)(M - m0 < CE_PIX ? M - m0 : CE_PIX);
    __syncthreads();
    ce_stage_in(logits + m0 * C, np * C, sx);
    __syncthreads();
    const long m = m0 + threadIdx.x;
    if ((int)threadIdx.x >= np) continue;
    const int64_t t = target[m];
    if (t == ignore) continue;
    const float* x = sx + threadIdx.x * C;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    const float nll = (mx + logf(se)) - x[t];
    float w = cw ? cw[t] : 1.f;
    den += (double)w;
    if (pw) w *= pw[m];
    num += (double)(w * nll);
  }
  __syncthreads();
  const double a = segsde_block_sum(num, sh);
  const double b = segsde_block_sum(den, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
}

__global__ __launch_bounds__(256) void ce_bwd_dense_kernel(const float* logits, long M, int C, const int64_t* target,
                                                           int64_t ignore, const float* cw, const float* pw,
                                                           const float* scale, float* dl) {
  SEGSDE_SMEM;
  float* sx = reinterpret_cast<float*>(segsde_smem);     // [CE_PIX][C], overwritten in place with the gradients
  const float sc = scale[0];
  for (long m0 = (long)blockIdx.x * CE_PIX; m0 < M; m0 += (long)gridDim.x * CE_PIX) {
    const int np = (int)(M - m0 < CE_PIX ? M - m0 : CE_PIX);
    __syncthreads();
    ce_stage_in(logits + m0 * C, np * C, sx);
    __syncthreads();
    if ((int)threadIdx.x < np) {
      const long m = m0 + threadIdx.x;
      const int64_t t = target[m];
      float* x = sx + threadIdx.x * C;
      if (t == ignore) {
        for (int c = 0; c < C; ++c) x[c] = 0.f;
      } else {
        float mx = x[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
        float w = (cw ? cw[t] : 1.f) * sc;
        if (pw) w *= pw[m];
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) x[c] = w * (expf(x[c] - mx) * inv - (c == t ? 1.f : 0.f));
      }
    }
    __syncthreads();
    float* dst = dl + m0 * C;
    const int n = np * C, n4 = n >> 2;
    for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(sx)[e];
    for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) dst[e] = sx[e];
  }
}

template <class MT>
__global__ __launch_bounds__(256) void mix_kernel(const MT* mask, int Bm, const float* x, int B, int C, int H, int W,
                                                  long sb, long sc, long sh, long sw, float* out) {
  const long total = (long)B * C * H * W;
  const bool half = (Bm != B);
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    // enumerate in the memory order of x (smallest stride innermost) for coalescing: assume sw or sc is 1
    int b, c, h, w;
    if (sc == 1) { c = (int)(e % C); long t = e / C; w = (int)(t % W); t /= W; h = (int)(t % H); b = (int)(t / H); }
    else { w = (int)(e % W); long t = e / W; h = (int)(t % H); t /= H; c = (int)(t % C); b = (int)(t / C); }
    const long off = c * sc + h * sh + w * sw;
    int ia, ib, mi; bool swap = false;
    if (!half) { ia = b; ib = (b + 1) % B; mi = b; }
    else { mi = b % Bm; ia = 2 * mi; ib = 2 * mi + 1; swap = b >= Bm; }
    const MT m = mask[((long)mi * H + h) * W + w];
    const float mf = (float)m, omf = (float)((MT)1 - m);
    const float xa = x[ia * sb + off], xb = x[ib * sb + off];
    float r;
    if (!swap) { const float p = mf * xa, q = omf * xb; r = p + q; }
    else { const float p = omf * xa, q = mf * xb; r = p + q; }
    out[b * sb + off] = r;
  }
}
__global__ __launch_bounds__(256) void mix_labels_kernel(const int64_t* mask, const int64_t* t, int B, long HW, int64_t* out) {
  const long total = (long)B * HW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / HW, p = e - b * HW;
    const int64_t m = mask[e];
    out[e] = m * t[e] + (1 - m) * t[((b + 1) % B) * HW + p];
  }
}
__global__ __launch_bounds__(256) void depthcomp_kernel(const float* d, int B, long HW, float margin, float ft_all,
                                                        const float* ft_dev /*[B] or null*/, int64_t* mask) {
  const long total = (long)B * HW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / HW, p = e - b * HW;
    const float ft = ft_dev ? ft_dev[b] : ft_all;    // train.py:592-599: one foreground threshold PER IMAGE when a range is configured
    const float own = d[e], other = d[((b + 1) % B) * HW + p];
    const float thr = other - margin;
    const int64_t fg = own >= thr ? 1 : 0;
    mask[e] = fg * (own >= ft ? 1 : 0);
  }
}
__global__ __launch_bounds__(256) void depth_thr_kernel(const float* d, long n, float t1, float t2, int two, float* mask) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    if (!two) mask[e] = d[e] >= t1 ? 1.f : 0.f;
    else {
      // reference: depth.ge(t1).le(t2).float() -- the boolean (0/1) of the first test is compared with t2
      const float ge = d[e] >= t1 ? 1.f : 0.f;
      mask[e] = ge <= t2 ? 1.f : 0.f;
    }
  }
}
__global__ __launch_bounds__(256) void class_mask_kernel(const int64_t* pred, long n, const int64_t* classes, int nc, int64_t* mask) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int64_t v = pred[e];
    int64_t s = 0;
    for (int k = 0; k < nc; ++k) s += (v == classes[k]) ? 1 : 0;
    mask[e] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// the callers on either side of the path (train.py)
// ---------------------------------------------------------------------------------------------------
// EMA teacher update (train.py:346-358): ema = alpha * ema + (1 - alpha) * p over every parameter tensor, ONE launch
// for the ~880 tensors instead of a Python loop of three elementwise kernels each.  The table splits tensors into
// chunks of at most MT_CHUNK elements; block b owns chunk b.  Same fp32 operation order as the reference
// (two multiplies, one add; alpha and 1-alpha rounded to fp32 on the host exactly as torch does): bit-exact.
__global__ __launch_bounds__(256) void multi_tensor_lerp_kernel(const segsde_mt_chunk* table, float alpha, float one_minus_alpha) {
  const segsde_mt_chunk c = table[blockIdx.x];
  float* dst = c.dst; const float* src = c.src;
  const int n = (int)c.n;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
    const int n4 = n >> 2;
    for (int e = threadIdx.x; e < n4; e += 256) {
      float4 a = reinterpret_cast<float4*>(dst)[e];
      const float4 b = reinterpret_cast<const float4*>(src)[e];
      a.x = alpha * a.x + one_minus_alpha * b.x; a.y = alpha * a.y + one_minus_alpha * b.y;
      a.z = alpha * a.z + one_minus_alpha * b.z; a.w = alpha * a.w + one_minus_alpha * b.w;
      reinterpret_cast<float4*>(dst)[e] = a;
    }
    for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) dst[e] = alpha * dst[e] + one_minus_alpha * src[e];
  } else {
    for (int e = threadIdx.x; e < n; e += 256) dst[e] = alpha * dst[e] + one_minus_alpha * src[e];
  }
}

// Pseudo labels from the (mixed) teacher softmax (train.py:644-651): per pixel max / first argmax over the C class planes of
// an NCHW tensor, label = ignore_index where the max is exactly 0 (pixels the mix left empty), and the number of
// pixels whose confidence reaches the threshold (the reference's unlabeled_weight numerator) -- one pass over the
// 19 x H x W tensor instead of max + compare + masked assignment + ge + sum.
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* prob, int C, long HW, long total, float thr,
                                                           int64_t ignore_index, int64_t* label, float* maxp,
                                                           unsigned long long* count) {
  unsigned long long mine = 0;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / HW, p = e - b * HW;
    const float* src = prob + b * C * HW + p;
    float best = src[0]; int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = src[(long)c * HW];
      if (v > best) { best = v; arg = c; }     // strict: the first maximum wins, as torch.max(dim) does on the CPU
    }
    label[e] = best == 0.f ? ignore_index : (int64_t)arg;
    if (maxp) maxp[e] = best;
    mine += best >= thr ? 1ull : 0ull;
  }
  // block count -> one atomic per block (integer: order-independent, deterministic)
  SEGSDE_SMEM;
  unsigned long long* sh = reinterpret_cast<unsigned long long*>(segsde_smem);
  sh[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sh[0]) atomicAdd(count, sh[0]);
}

// pixel weights of calc_pseudo_label_loss: every pixel gets count / total (kept on the device: no .item() round trip)
__global__ __launch_bounds__(256) void fill_fraction_kernel(const unsigned long long* count, long total, float* out) {
  // the reference divides two Python numbers in double and fills a float32 tensor with the result
  const float w = (float)((double)count[0] / (double)total);
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) out[e] = w;
}

// Validation: runningScore._fast_hist, evaluation/metrics.py:12-25 -- hist[n*gt + pred] += 1 over the
// pixels with 0 <= gt < n -- fused with the argmax over the class planes (train.py:848 `semantics.data.max(1)[1]`) when
// logits are given instead of predictions.  Block-private LDS histogram, one integer atomic per touched bin per block:
// exact and order-independent.  logits strides let the tensor be NCHW or channels-last.
__global__ __launch_bounds__(256) void confusion_kernel(const float* logits, long sb, long sc, long sp, const int64_t* pred,
                                                        const int64_t* gt, long HW, long total, int C,
                                                        unsigned long long* hist) {
  SEGSDE_SMEM;
  unsigned* sh = reinterpret_cast<unsigned*>(segsde_smem);    // [C*C]
  const int bins = C * C;
  for (int i = threadIdx.x; i < bins; i += 256) sh[i] = 0u;
  __syncthreads();
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int64_t t = gt[e];
    if (t < 0 || t >= C) continue;
    int64_t pr;
    if (logits) {
      const long b = e / HW, p = e - b * HW;
      const float* src = logits + b * sb + p * sp;
      float best = src[0]; int arg = 0;
      for (int c = 1; c < C; ++c) {
        const float v = src[(long)c * sc];
        if (v > best) { best = v; arg = c; }
      }
      pr = arg;
    } else {
      pr = pred[e];
      if (pr < 0 || pr >= C) continue;        // np.bincount would raise / misplace; the reference never produces these
    }
    atomicAdd(&sh[(int)t * C + (int)pr], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += 256)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}
}  // namespace

extern "C" size_t segsde_cross_entropy_workspace(long M) { return (size_t)ce_blocks(M) * 2 * sizeof(double); }

extern "C" int segsde_cross_entropy_forward(const float* logits, int ld, long M, int C, const int64_t* target,
                                            int64_t ignore_index, const float* class_weight, const float* pixel_weights,
                                            float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !target || !out || !ws) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0 || ld < C) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_cross_entropy_workspace(M)) return SEGSDE_ERR_WORKSPACE;
  const int nb = ce_blocks(M);
  const bool dense = ld == C && C <= 64 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
  if (dense)
    hipLaunchKernelGGL(ce_fwd_dense_kernel, dim3(nb), dim3(256), (size_t)(CE_PIX * C + 2) * sizeof(float) + 64, ST(stream), logits,
                       M, C, target, ignore_index, class_weight, pixel_weights, (double*)ws);
  else
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(256), 64, ST(stream), logits, ld, M, C, target, ignore_index, class_weight,
                       pixel_weights, (double*)ws);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), (const double*)ws, nb, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_cross_entropy_backward(const float* logits, int ld, long M, int C, const int64_t* target,
                                             int64_t ignore_index, const float* class_weight, const float* pixel_weights,
                                             const float* scale, float* dlogits, int lddl, void* stream) {
  if (!logits || !target || !scale || !dlogits) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0 || ld < C || lddl < C) return SEGSDE_ERR_SHAPE;
  const bool dense = ld == C && lddl == C && C <= 64 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
  if (dense)
    hipLaunchKernelGGL(ce_bwd_dense_kernel, dim3(ce_blocks(M)), dim3(256), (size_t)CE_PIX * C * sizeof(float), ST(stream), logits, M, C,
                       target, ignore_index, class_weight, pixel_weights, scale, dlogits);
  else
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(ce_blocks(M)), dim3(256), 0, ST(stream), logits, ld, M, C, target, ignore_index,
                       class_weight, pixel_weights, scale, dlogits, lddl);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_mix(const void* mask, int mask_is_int64, int Bm, const float* x, int B, int C, int H, int W, long sb,
                          long sc, long sh, long sw, float* out, void* stream) {
  if (!mask || !x || !out) return SEGSDE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SEGSDE_ERR_SHAPE;
  if (Bm != B && !(2 * Bm == B)) return SEGSDE_ERR_SHAPE;
  if (sc != 1 && sw != 1) return SEGSDE_ERR_UNSUPPORTED;
  const long total = (long)B * C * H * W;
  if (mask_is_int64)
    hipLaunchKernelGGL(mix_kernel<int64_t>, dim3(flat_blocks(total)), dim3(256), 0, ST(stream), (const int64_t*)mask, Bm, x, B,
                       C, H, W, sb, sc, sh, sw, out);
  else
    hipLaunchKernelGGL(mix_kernel<float>, dim3(flat_blocks(total)), dim3(256), 0, ST(stream), (const float*)mask, Bm, x, B, C,
                       H, W, sb, sc, sh, sw, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_mix_labels(const int64_t* mask, const int64_t* target, int B, int H, int W, int64_t* out, void* stream) {
  if (!mask || !target || !out) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(mix_labels_kernel, dim3(flat_blocks((long)B * H * W)), dim3(256), 0, ST(stream), mask, target, B,
                     (long)H * W, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_depthcomp_mask(const float* depths, int B, long HW, float margin, float fg_threshold,
                                     const float* fg_threshold_per_sample, int64_t* mask, void* stream) {
  if (!depths || !mask) return SEGSDE_ERR_NULL;
  if (B < 1 || HW <= 0) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(depthcomp_kernel, dim3(flat_blocks((long)B * HW)), dim3(256), 0, ST(stream), depths, B, HW, margin,
                     fg_threshold, fg_threshold_per_sample, mask);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_depth_threshold_mask(const float* depth, long n, float t1, float t2, int two_thresholds, float* mask,
                                           void* stream) {
  if (!depth || !mask) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(depth_thr_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST(stream), depth, n, t1, t2, two_thresholds, mask);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_class_mask(const int64_t* pred, long n, const int64_t* classes, int n_classes, int64_t* mask, void* stream) {
  if (!pred || !classes || !mask) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(class_mask_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST(stream), pred, n, classes, n_classes, mask);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_multi_tensor_lerp(const segsde_mt_chunk* table_dev, int nchunks, float alpha, float one_minus_alpha,
                                        void* stream) {
  if (!table_dev) return SEGSDE_ERR_NULL;
  if (nchunks <= 0) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(multi_tensor_lerp_kernel, dim3(nchunks), dim3(256), 0, ST(stream), table_dev, alpha, one_minus_alpha);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_pseudo_label(const float* prob_nchw, int B, int C, long HW, float threshold, int64_t ignore_index,
                                   int64_t* label, float* max_prob, unsigned long long* count, float* pixel_weight,
                                   void* stream) {
  if (!prob_nchw || !label || !count) return SEGSDE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SEGSDE_ERR_SHAPE;
  const long total = (long)B * HW;
  if (hipMemsetAsync(count, 0, sizeof(unsigned long long), ST(stream)) != hipSuccess) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(pseudo_label_kernel, dim3(flat_blocks(total)), dim3(256), 256 * sizeof(unsigned long long), ST(stream),
                     prob_nchw, C, HW, total, threshold, ignore_index, label, max_prob, count);
  SEGSDE_CHECK_LAUNCH();
  if (pixel_weight) {
    hipLaunchKernelGGL(fill_fraction_kernel, dim3(flat_blocks(total)), dim3(256), 0, ST(stream), (const unsigned long long*)count,
                       total, pixel_weight);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int segsde_confusion_update(const float* logits, long sb, long sc, long sp, const int64_t* pred, const int64_t* gt,
                                       int B, long HW, int C, unsigned long long* hist, void* stream) {
  if ((!logits && !pred) || !gt || !hist) return SEGSDE_ERR_NULL;
  if (B <= 0 || HW <= 0 || C <= 0 || C > 64) return SEGSDE_ERR_SHAPE;
  const long total = (long)B * HW;
  long nb = (total + 256L * 64 - 1) / (256L * 64);       // >= 64 pixels per thread: the block histogram flush is amortised
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)nb), dim3(256), (size_t)C * C * sizeof(unsigned), ST(stream), logits, sb, sc,
                     sp, pred, gt, HW, total, C, hist);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
