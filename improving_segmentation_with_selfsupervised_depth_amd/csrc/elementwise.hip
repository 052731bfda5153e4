// HBM-bound NHWC kernels around the convolutions: BatchNorm (train / eval) forward + backward, activation
// backward + bias gradient, max-pool, nearest-upsample adjoint, bilinear resize (+adjoint), global average
// pool, SelfAttention gate, axpby, channel-slice copy and the NCHW<->NHWC edges.
//
// Design: every tensor is "M rows x C channels" with a pixel pitch ld.  Per-channel reductions use one thread
// per channel inside a 64-channel slab (a wave reads one 256-byte row segment per instruction), double
// accumulators per thread, a deterministic two-level tree (block partials -> finalize kernel): no atomics, so
// results are run-to-run reproducible.  Pure elementwise kernels take a float4 path when C, the pitches and
// the base pointers allow it.
#include <stdlib.h>
#include "segsde_common.h"
#include <type_traits>
#include <utility>
#include <cstdlib>
#include <cstring>

namespace {

template <int VW> struct VecF { float v[VW]; };
template <int VW> __device__ __forceinline__ VecF<VW> ldv(const float* p) {
  VecF<VW> r;
  if (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1 % VW] = t.y; r.v[2 % VW] = t.z; r.v[3 % VW] = t.w; }
  else r.v[0] = p[0];
  return r;
}
template <int VW> __device__ __forceinline__ void stv(float* p, const VecF<VW>& r) {
  if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % VW], r.v[2 % VW], r.v[3 % VW]);
  else p[0] = r.v[0];
}

constexpr int SLAB = 64;      // channels per block column
constexpr int RLANES = 4;     // row lanes per block (256 threads = 64 channels x 4 rows)

__host__ __device__ inline int red_blocks(long M) {
  long nb = (M + 255) / 256;
  return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}

// generic column reduction: Op::row(m, c, s0, s1) accumulates two sums for channel c over this block's rows.
// VW = 4: each thread owns 4 consecutive channels (16-byte loads; 16 channel-quads x 16 row lanes per block),
// VW = 1: one channel per thread (64 channels x 4 row lanes) for odd channel counts / unaligned pitches.
// TX channel lanes x TY = 256 / TX row lanes.  Wide tensors: TX = 64 / VW lanes of VW channels each.  Narrow ones (C < 64:
// the 1-channel disparity heads, the 19-class logits) shrink TX to the next power of two >= C so that all 256 threads
// stay busy -- with 64 channel lanes a 1-channel reduction ran on 4 threads per block.
// an Op may provide per-channel context (ctx<VW>(c)) that its row() takes: loaded once per thread instead of once per row
template <class Op, int VW, class = void> struct has_ctx : std::false_type {};
template <class Op, int VW>
struct has_ctx<Op, VW, std::void_t<decltype(std::declval<const Op&>().template ctx<VW>(0))>> : std::true_type {};

// Fin: finalize inside the reduction kernel (round-3 EXPERIMENT, off by default -- see launch_colreduce).  The
// block that draws the LAST ticket of its channel group folds the group's partial rows -- in the same fixed lane order as
// pair_finalize_kernel, so the result is bit-identical to the two-kernel path -- and writes the float sums.
struct Fin { unsigned* tickets; float* out0; float* out1; };

template <class Op, int VW, int TX>
__global__ __launch_bounds__(256) void colreduce_kernel(Op op, long M, int C, double* part, Fin fin) {
  constexpr int TY = 256 / TX, CW = TX * VW;          // channels per block
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);  // [2][TY][CW]
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int c = blockIdx.y * CW + tx * VW;
  const int nb = gridDim.x;
  const long rows_per = (M + nb - 1) / nb;
  const long r_begin = blockIdx.x * rows_per;
  long r_end = r_begin + rows_per; if (r_end > M) r_end = M;
  double s0[VW], s1[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
  if (c < C) {
    if constexpr (has_ctx<Op, VW>::value) {
      const auto k = op.template ctx<VW>(c);
      for (long m = r_begin + ty; m < r_end; m += TY) op.template row<VW>(m, c, k, s0, s1);
    } else {
      // the loads of the next rows do not depend on the sums: unrolled, they are in flight together (same order of adds).
      // Bias-gradient column sums 86 -> 63 us, the single-column one 846 -> 273 us; the BatchNorm reduction above (three loads
      // and a mask per row) lost 3 % with it and stays rolled.
#pragma unroll 4
      for (long m = r_begin + ty; m < r_end; m += TY) op.template row<VW>(m, c, s0, s1);
    }
  }
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    sh[ty * CW + tx * VW + j] = s0[j];
    sh[(TY + ty) * CW + tx * VW + j] = s1[j];
  }
  __syncthreads();
  const int cc = threadIdx.x;   // the first CW threads finish one channel each
  if (cc < CW && blockIdx.y * CW + cc < C) {
    double a = 0.0, b2 = 0.0;
    for (int j = 0; j < TY; ++j) { a += sh[j * CW + cc]; b2 += sh[(TY + j) * CW + cc]; }
    part[((long)blockIdx.x * 2 + 0) * C + blockIdx.y * CW + cc] = a;
    part[((long)blockIdx.x * 2 + 1) * C + blockIdx.y * CW + cc] = b2;
  }
  if (!fin.tickets) return;
  __threadfence();
  __syncthreads();
  unsigned* flag = reinterpret_cast<unsigned*>(sh);
  if (threadIdx.x == 0) flag[0] = atomicAdd(&fin.tickets[blockIdx.y], 1u) == (unsigned)(nb - 1) ? 1u : 0u;
  __syncthreads();
  const bool last = flag[0] != 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // pair_finalize_kernel's order: 16 partial-lanes (lane l takes rows l, l + 16, ...), then lanes 0..15 in sequence
  for (int t = threadIdx.x; t < 16 * CW; t += 256) {
    const int l = t / CW, ch = t - l * CW, cg = blockIdx.y * CW + ch;
    double a = 0.0, b2 = 0.0;
    if (cg < C)
      for (int i = l; i < nb; i += 16) { a += part[((long)i * 2) * C + cg]; b2 += part[((long)i * 2 + 1) * C + cg]; }
    sh[l * CW + ch] = a; sh[(16 + l) * CW + ch] = b2;
  }
  __syncthreads();
  if (cc < CW && blockIdx.y * CW + cc < C) {
    double a = 0.0, b2 = 0.0;
    for (int j = 0; j < 16; ++j) { a += sh[j * CW + cc]; b2 += sh[(16 + j) * CW + cc]; }
    if (fin.out0) fin.out0[blockIdx.y * CW + cc] = (float)a;
    if (fin.out1) fin.out1[blockIdx.y * CW + cc] = (float)b2;
  }
  if (threadIdx.x == 0) fin.tickets[blockIdx.y] = 0u;
}

template <class Op, int TX>
void launch_colreduce_scalar(Op op, long M, int C, double* part, hipStream_t s, Fin fin) {
  const dim3 grid(red_blocks(M), (C + TX - 1) / TX);
  // LDS: [2][TY][TX] block partials, and [2][16][TX] lane sums of the in-kernel finalize
  const size_t smem = 2 * (size_t)(256 > 16 * TX ? 256 : 16 * TX) * sizeof(double);
  hipLaunchKernelGGL((colreduce_kernel<Op, 1, TX>), grid, dim3(256), smem, s, op, M, C, part, fin);
}

// fin: out0 / out1 (nullable) receive the two column sums as floats from the reduction kernel itself (its last block per
// channel group); returns true in *done when it did (the caller then skips pair_finalize_kernel)
template <class Op>
int launch_colreduce(Op op, long M, int C, double* part, bool vec, hipStream_t s, float* out0 = nullptr, float* out1 = nullptr,
                     bool* done = nullptr) {
  Fin fin{nullptr, out0, out1};
  // off by default (SEGSDE_TUNE="cfin=1"): the device-scope fence costs more than the ~7 us finalize launch it replaces
  // (colreduce<BnBwdOp> 53 -> 222 us per launch on the 8-XCD part; profiles/experiments_r03.md)
  static const bool enabled = [] { const char* e = getenv("SEGSDE_TUNE"); return e && strstr(e, "cfin=1"); }();
  if (done && enabled && (out0 || out1)) {
    const int cw = vec ? SLAB : (C <= 1 ? 1 : C <= 2 ? 2 : C <= 4 ? 4 : C <= 8 ? 8 : C <= 16 ? 16 : C <= 32 ? 32 : 64);   // channels per block
    fin.tickets = segsde_ticket_slice((C + cw - 1) / cw);             // one ticket per channel group (grid.y)
  }
  if (vec) {
    const dim3 grid(red_blocks(M), (C + SLAB - 1) / SLAB);
    hipLaunchKernelGGL((colreduce_kernel<Op, 4, SLAB / 4>), grid, dim3(256), 2 * 16 * SLAB * sizeof(double), s, op, M, C, part, fin);
  } else if (C <= 1) launch_colreduce_scalar<Op, 1>(op, M, C, part, s, fin);
  else if (C <= 2) launch_colreduce_scalar<Op, 2>(op, M, C, part, s, fin);
  else if (C <= 4) launch_colreduce_scalar<Op, 4>(op, M, C, part, s, fin);
  else if (C <= 8) launch_colreduce_scalar<Op, 8>(op, M, C, part, s, fin);
  else if (C <= 16) launch_colreduce_scalar<Op, 16>(op, M, C, part, s, fin);
  else if (C <= 32) launch_colreduce_scalar<Op, 32>(op, M, C, part, s, fin);
  else launch_colreduce_scalar<Op, 64>(op, M, C, part, s, fin);
  SEGSDE_CHECK_LAUNCH();
  if (done) *done = fin.tickets != nullptr;
  return 0;
}

inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t part_bytes(long M, int C) { return (size_t)red_blocks(M) * 2 * C * sizeof(double); }

// ------------------------------------------------------------------ BatchNorm forward
struct StatsOp {
  const float* x; int ld;
  template <int VW> __device__ void row(long m, int c, double* s0, double* s1) const {
    const VecF<VW> t = ldv<VW>(x + m * ld + c);
#pragma unroll
    for (int j = 0; j < VW; ++j) { const double v = (double)t.v[j]; s0[j] += v; s1[j] += v * v; }
  }
};

// partial-slab combine shared by the finalize kernels: 16 channels x 16 partial-lanes per block, fixed order
__device__ __forceinline__ void combine_partials(const double* part, int nb, int C, int c, int pl, double* sh /*[2][16][16]*/,
                                                 double& s, double& q) {
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int i = pl; i < nb; i += 16) { a += part[((long)i * 2) * C + c]; b += part[((long)i * 2 + 1) * C + c]; }
  const int cl = threadIdx.x & 15;
  sh[pl * 16 + cl] = a; sh[256 + pl * 16 + cl] = b;
  __syncthreads();
  s = 0.0; q = 0.0;
  if (pl == 0)
    for (int j = 0; j < 16; ++j) { s += sh[j * 16 + cl]; q += sh[256 + j * 16 + cl]; }
}

__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* part, int nb, long M, int C, float eps,
                                                                float momentum, float* mean, float* invstd,
                                                                float* running_mean, float* running_var, int64_t* nbt) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;   // nn.BatchNorm2d.num_batches_tracked, without its own launch
  double s, q;
  combine_partials(part, nb, C, c, pl, sh, s, q);
  if (pl != 0 || c >= C) return;
  const double mu = s / (double)M;
  double var = q / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* rm, const float* rv, int C, float eps,
                                                            float* mean, float* invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  invstd[c] = 1.0f / sqrtf(rv[c] + eps);
}

template <int VW>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, int ldx, long M, int C, const float* mean,
                                                       const float* invstd, const float* gamma, const float* beta,
                                                       const float* res, int ldr, float* y, int ldy, int act,
                                                       float drop_p, uint64_t seed, int cv_shift) {
  const int CV = C / VW;
  const long total = M * CV;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  // Round 3: when the grid stride is a multiple of the channel-vector count (power-of-two channel counts, host-chosen grid) a
  // thread meets the SAME channels in every trip: mean / invstd / gamma / beta are loaded once, not four 16-byte loads next
  // to every 16 bytes of data (those loads hit L1 but took 4/5 of the kernel's vector-memory issue slots: 5.0 TB/s before).
  const long stride = (long)gridDim.x * 256;
  const bool fixed_c = cv_shift >= 0 && (stride & (long)(CV - 1)) == 0;
  VecF<VW> mu, is, ga, be;
  if (fixed_c) {
    const int c = (int)((blockIdx.x * 256L + threadIdx.x) & (long)(CV - 1)) * VW;
    mu = ldv<VW>(mean + c); is = ldv<VW>(invstd + c);
    if (gamma) { ga = ldv<VW>(gamma + c); be = ldv<VW>(beta + c); }
  }
  auto finish = [&](long m, int c, const VecF<VW>& xv, const VecF<VW>& rv) {
    VecF<VW> o;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      float t = (xv.v[j] - mu.v[j]) * is.v[j];
      if (gamma) t = t * ga.v[j] + be.v[j];
      if (res) t += rv.v[j];
      t = segsde_act(t, act);
      if (drop_p > 0.f) t = segsde_uniform01(seed, (uint64_t)(m * C + c + j)) >= drop_p ? t * keep_scale : 0.f;
      o.v[j] = t;
    }
    stv<VW>(y + m * ldy + c, o);
  };
  long e = blockIdx.x * 256L + threadIdx.x;
  // (four trips per thread at a time, all loads in flight together, measured SLOWER: 44.3 -> 49.0 us per launch; the column
  // sums, which only load, gained from the same unrolling)
  for (; e < total; e += stride) {
    // channel counts are powers of two on the whole ResNet / decoder path: shift + mask instead of a 64-bit division
    const long m = cv_shift >= 0 ? (e >> cv_shift) : e / CV;
    const int c = (int)(e - m * CV) * VW;
    const VecF<VW> xv = ldv<VW>(x + m * ldx + c);
    if (!fixed_c) {   // per-channel parameters as one 16-byte load each
      mu = ldv<VW>(mean + c); is = ldv<VW>(invstd + c);
      if (gamma) { ga = ldv<VW>(gamma + c); be = ldv<VW>(beta + c); }
    }
    VecF<VW> rv;
    if (res) rv = ldv<VW>(res + m * ldr + c);
    finish(m, c, xv, rv);
  }
}

// ------------------------------------------------------------------ BatchNorm backward
// dz = dy * dropout_mask * act'(y)
__device__ __forceinline__ float bn_dz(float dy, float y, int act, float drop_p, uint64_t seed, uint64_t idx) {
  float g = dy;
  if (drop_p > 0.f) {
    // y is the post-dropout value; the pre-dropout activation output is y*(1-p) where kept
    const bool keep = segsde_uniform01(seed, idx) >= drop_p;
    if (!keep) return 0.f;
    const float ks = 1.f / (1.f - drop_p);
    g *= ks;
    y = y / ks;
  }
  return g * segsde_act_grad_from_out(y, act);
}

// y == nullptr ("remask"): the saved output is not read.  Allowed for act = none (y is not needed at all) and for a ReLU
// with no residual and no dropout, whose mask is recomputed from x exactly as bn_apply formed its argument
// ((x - mean) * invstd, then * gamma + beta, same operation order, -ffp-contract=off) -- one of the three tensor reads of
// each backward pass saved on two of every three BatchNorms of a bottleneck.
__device__ __forceinline__ float bn_dz_remask(float dy, float xh, float gamma, float beta, bool affine, int act) {
  if (act != SEGSDE_ACT_RELU) return dy;
  float t = xh;
  if (affine) t = t * gamma + beta;
  return t > 0.f ? dy : 0.f;
}

struct BnBwdOp {
  const float* dy; int lddy; const float* y; int ldy; const float* x; int ldx; const float* mean; const float* invstd;
  int act, C; float drop_p; uint64_t seed; const float* gamma; const float* beta;
  // per-channel parameters, loaded ONCE per thread (a thread of colreduce_kernel keeps its channels for all its rows)
  template <int VW> struct Ctx { VecF<VW> mu, is, ga, be; };
  template <int VW> __device__ Ctx<VW> ctx(int c) const {
    Ctx<VW> k;
    k.mu = ldv<VW>(mean + c); k.is = ldv<VW>(invstd + c);
    if (!y && gamma) { k.ga = ldv<VW>(gamma + c); k.be = ldv<VW>(beta + c); }
    return k;
  }
  template <int VW> __device__ void row(long m, int c, const Ctx<VW>& k, double* s0, double* s1) const {
    const VecF<VW> g = ldv<VW>(dy + m * lddy + c), xx = ldv<VW>(x + m * ldx + c);
    VecF<VW> yy;
    if (y) yy = ldv<VW>(y + m * ldy + c);
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float xh = (xx.v[j] - k.mu.v[j]) * k.is.v[j];
      const float dz = y ? bn_dz(g.v[j], yy.v[j], act, drop_p, seed, (uint64_t)(m * C + c + j))
                         : bn_dz_remask(g.v[j], xh, gamma ? k.ga.v[j] : 1.f, gamma ? k.be.v[j] : 0.f, gamma != nullptr, act);
      s0[j] += (double)dz * (double)xh; s1[j] += (double)dz;
    }
  }
};

__global__ __launch_bounds__(256) void pair_finalize_kernel(const double* part, int nb, int C, float* out0, float* out1) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  double a, b;
  combine_partials(part, nb, C, c, pl, sh, a, b);
  if (pl != 0 || c >= C) return;
  if (out0) out0[c] = (float)a;
  if (out1) out1[c] = (float)b;
}

template <int VW>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, int lddy, const float* y, int ldy,
                                                           const float* x, int ldx, long M, int C, const float* mean,
                                                           const float* invstd, const float* gamma, const float* beta,
                                                           int act, float drop_p, uint64_t seed, int batch_stats,
                                                           const float* dgamma, const float* dbeta, float* dx, int lddx,
                                                           float* dres, int lddres, int cv_shift) {
  const int CV = C / VW;
  const long total = M * CV;
  const float invM = 1.f / (float)M;
  const bool remask = y == nullptr;
  // per-channel parameters once per thread when every trip of the grid-stride loop meets the same channels (see bn_apply_kernel):
  // up to six 16-byte parameter loads next to the two or three data loads of a trip before
  const long stride = (long)gridDim.x * 256;
  const bool fixed_c = cv_shift >= 0 && (stride & (long)(CV - 1)) == 0;
  VecF<VW> ga, be, mu, is, dg, db;
  auto load_params = [&](int c) {
    if (dx || remask) {
      is = ldv<VW>(invstd + c);
      if (gamma) ga = ldv<VW>(gamma + c);
      if (gamma && remask) be = ldv<VW>(beta + c);
      if (batch_stats || remask) mu = ldv<VW>(mean + c);
      if (batch_stats && dx) { dg = ldv<VW>(dgamma + c); db = ldv<VW>(dbeta + c); }
    }
  };
  if (fixed_c) load_params((int)((blockIdx.x * 256L + threadIdx.x) & (long)(CV - 1)) * VW);
  auto finish = [&](long m, int c, const VecF<VW>& g, const VecF<VW>& xx, const VecF<VW>& yy) {
    VecF<VW> dz, o;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      if (remask) dz.v[j] = bn_dz_remask(g.v[j], (xx.v[j] - mu.v[j]) * is.v[j], gamma ? ga.v[j] : 1.f, gamma ? be.v[j] : 0.f, gamma != nullptr, act);
      else dz.v[j] = bn_dz(g.v[j], yy.v[j], act, drop_p, seed, (uint64_t)(m * C + c + j));
      if (dx) {
        const float gm = gamma ? ga.v[j] : 1.f;
        if (batch_stats) {
          const float xh = (xx.v[j] - mu.v[j]) * is.v[j];
          o.v[j] = gm * is.v[j] * (dz.v[j] - db.v[j] * invM - xh * dg.v[j] * invM);
        } else {
          o.v[j] = gm * is.v[j] * dz.v[j];
        }
      }
    }
    if (dres) stv<VW>(dres + m * lddres + c, dz);
    if (dx) stv<VW>(dx + m * lddx + c, o);
  };
  long e = blockIdx.x * 256L + threadIdx.x;
  for (; e < total; e += stride) {
    const long m = cv_shift >= 0 ? (e >> cv_shift) : e / CV;
    const int c = (int)(e - m * CV) * VW;
    const VecF<VW> g = ldv<VW>(dy + m * lddy + c);
    VecF<VW> xx, yy;
    if (!remask) yy = ldv<VW>(y + m * ldy + c);
    if ((dx && batch_stats) || remask) xx = ldv<VW>(x + m * ldx + c);
    if (!fixed_c) load_params(c);
    finish(m, c, g, xx, yy);
  }
}

// ------------------------------------------------------------------ activation backward + bias gradient
struct ActBwdOp {
  const float* dy; int lddy; const float* y; int ldy; float* dz; int lddz; int act;
  template <int VW> __device__ void row(long m, int c, double* s0, double* s1) const {
    const VecF<VW> g = ldv<VW>(dy + m * lddy + c), yy = ldv<VW>(y + m * ldy + c);
    VecF<VW> o;
#pragma unroll
    for (int j = 0; j < VW; ++j) { o.v[j] = g.v[j] * segsde_act_grad_from_out(yy.v[j], act); s0[j] += (double)o.v[j]; }
    if (dz) stv<VW>(dz + m * lddz + c, o);
  }
};
struct ColsumOp {
  const float* x; int ld;
  template <int VW> __device__ void row(long m, int c, double* s0, double* s1) const {
    const VecF<VW> t = ldv<VW>(x + m * ld + c);
#pragma unroll
    for (int j = 0; j < VW; ++j) s0[j] += (double)t.v[j];
  }
};

// ------------------------------------------------------------------ max-pool 3x3 s2 p1
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, int B, int H, int W, int C, int Ho, int Wo,
                                                          float* y, uint8_t* idx) {
  const long total = (long)B * Ho * Wo * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); long t = e / C;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); const int b = (int)(t / Ho);
    float best = -INFINITY; int bi = 0; bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * ho - 1 + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int w = 2 * wo - 1 + kw;
        if (w < 0 || w >= W) continue;
        const float v = x[((long)(b * H + h) * W + w) * C + c];
        if (first || v > best || v != v) { best = v; bi = kh * 3 + kw; first = false; }
      }
    }
    y[e] = best;
    idx[e] = (uint8_t)bi;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dy, const uint8_t* idx, int B, int H, int W, int C,
                                                          int Ho, int Wo, float* dx, int accumulate) {
  const long total = (long)B * H * W * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); long t = e / C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int b = (int)(t / H);
    float s = 0.f;
    // output windows (ho, kh) with 2*ho - 1 + kh == h
    for (int kh = 0; kh < 3; ++kh) {
      const int hn = h + 1 - kh;
      if (hn < 0 || (hn & 1)) continue;
      const int ho = hn >> 1; if (ho >= Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int wn = w + 1 - kw;
        if (wn < 0 || (wn & 1)) continue;
        const int wo = wn >> 1; if (wo >= Wo) continue;
        const long o = ((long)(b * Ho + ho) * Wo + wo) * C + c;
        if (idx[o] == kh * 3 + kw) s += dy[o];
      }
    }
    dx[e] = accumulate ? dx[e] + s : s;
  }
}

// four channels per thread, 32-bit index arithmetic (the scalar version spends its time in three 64-bit divisions per element)
// four channels per thread: nine 16-byte loads, one 16-byte store and one 4-byte index store per thread (the scalar kernel
// above moved 4 bytes per load: 0.35 ms for the 0.7 GB of the stem's pooling, 2.5x its HBM time)
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* x, int B, int H, int W, int C4, int Ho, int Wo,
                                                           float* y, uint8_t* idx) {
  const long total = (long)B * Ho * Wo * C4;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c4 = (int)(e % C4); long t = e / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); const int b = (int)(t / Ho);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * ho - 1 + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int w = 2 * wo - 1 + kw;
        if (w < 0 || w >= W) continue;
        const float4 v4 = reinterpret_cast<const float4*>(x)[((long)(b * H + h) * W + w) * C4 + c4];
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (first || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = kh * 3 + kw; }
        first = false;
      }
    }
    reinterpret_cast<float4*>(y)[e] = make_float4(best[0], best[1], best[2], best[3]);
    reinterpret_cast<uchar4*>(idx)[e] = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
  }
}
__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* dy, const uint8_t* idx, int B, int H, int W, int C4,
                                                           int Ho, int Wo, float* dx, int accumulate) {
  const int total = B * H * W * C4;       // host guarantees < 2^31
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c4 = e % C4; int t = e / C4;
    const int w = t % W; t /= W;
    const int h = t % H; const int b = t / H;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < 3; ++kh) {
      const int hn = h + 1 - kh;
      if (hn < 0 || (hn & 1)) continue;
      const int ho = hn >> 1; if (ho >= Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int wn = w + 1 - kw;
        if (wn < 0 || (wn & 1)) continue;
        const int wo = wn >> 1; if (wo >= Wo) continue;
        const long o = ((long)(b * Ho + ho) * Wo + wo) * C4 + c4;     // in units of 4 channels
        const uchar4 k = reinterpret_cast<const uchar4*>(idx)[o];
        const float4 g = reinterpret_cast<const float4*>(dy)[o];
        const unsigned char tap = (unsigned char)(kh * 3 + kw);
        s.x += k.x == tap ? g.x : 0.f; s.y += k.y == tap ? g.y : 0.f; s.z += k.z == tap ? g.z : 0.f; s.w += k.w == tap ? g.w : 0.f;
      }
    }
    if (accumulate) {
      const float4 o = reinterpret_cast<const float4*>(dx)[e];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    reinterpret_cast<float4*>(dx)[e] = s;
  }
}

// ------------------------------------------------------------------ nearest x2 upsample adjoint
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* dy, int lddy, int B, int h, int w, int C,
                                                             float* dx, int lddx) {
  const long total = (long)B * h * w * C;
  const int W2 = 2 * w;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); long t = e / C;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h); const int b = (int)(t / h);
    const float* p = dy + ((long)(b * 2 * h + 2 * y) * W2 + 2 * x) * lddy + c;
    const float s = (p[0] + p[lddy]) + (p[(long)W2 * lddy] + p[(long)W2 * lddy + lddy]);
    dx[((long)(b * h + y) * w + x) * lddx + c] = s;
  }
}

// nearest x2 upsample itself (monodepth_layers.py:202-205 as a stand-alone call; the decoders never materialise it)
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* x, int ldx, int B, int h, int w, int C, float* y,
                                                             int ldy) {
  const long total = (long)B * 2 * h * 2 * w * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); long t = e / C;
    const int X = (int)(t % (2 * w)); t /= 2 * w;
    const int Y = (int)(t % (2 * h)); const int b = (int)(t / (2 * h));
    y[((long)(b * 2 * h + Y) * 2 * w + X) * ldy + c] = x[((long)(b * h + (Y >> 1)) * w + (X >> 1)) * ldx + c];
  }
}

// ------------------------------------------------------------------ bilinear resize (ATen upsample_bilinear2d semantics)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_src(int dst, int in, int out, int align_corners) {
  float scale, src;
  if (align_corners) { scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; src = scale * (float)dst; }
  else { scale = (float)in / (float)out; src = scale * ((float)dst + 0.5f) - 0.5f; if (src < 0.f) src = 0.f; }
  Lerp r;
  r.i0 = (int)src; if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0; r.l0 = 1.f - r.l1;
  return r;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* x, int ldx, int B, int Hi, int Wi, int C, float* y,
                                                         int ldy, int Ho, int Wo, int ac) {
  const long total = (long)B * Ho * Wo * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); long t = e / C;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); const int b = (int)(t / Ho);
    const Lerp lh = lerp_src(ho, Hi, Ho, ac), lw = lerp_src(wo, Wi, Wo, ac);
    const float* base = x + (long)b * Hi * Wi * ldx + c;
    const float v00 = base[((long)lh.i0 * Wi + lw.i0) * ldx], v01 = base[((long)lh.i0 * Wi + lw.i1) * ldx];
    const float v10 = base[((long)lh.i1 * Wi + lw.i0) * ldx], v11 = base[((long)lh.i1 * Wi + lw.i1) * ldx];
    y[((long)(b * Ho + ho) * Wo + wo) * ldy + c] =
        lh.l0 * (lw.l0 * v00 + lw.l1 * v01) + lh.l1 * (lw.l0 * v10 + lw.l1 * v11);
  }
}

// adjoint as a deterministic gather: each input pixel visits the (conservatively bounded) destination range
// that can reference it and re-derives the forward weights
__device__ __forceinline__ void dst_range(int i, int in, int out, int ac, int& lo, int& hi) {
  float inv;
  if (ac) inv = in > 1 ? (float)(out - 1) / (float)(in - 1) : (float)out;
  else inv = (float)out / (float)in;
  lo = (int)floorf(((float)i - 1.5f) * inv) - 2;
  hi = (int)ceilf(((float)i + 1.5f) * inv) + 2;
  if (in == 1 || out == 1) { lo = 0; hi = out - 1; }
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* dy, int lddy, int B, int Hi, int Wi, int C,
                                                         float* dx, int lddx, int Ho, int Wo, int ac) {
  const long total = (long)B * Hi * Wi * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); long t = e / C;
    const int wi = (int)(t % Wi); t /= Wi;
    const int hi = (int)(t % Hi); const int b = (int)(t / Hi);
    int hlo, hhi, wlo, whi;
    dst_range(hi, Hi, Ho, ac, hlo, hhi);
    dst_range(wi, Wi, Wo, ac, wlo, whi);
    float s = 0.f;
    for (int ho = hlo; ho <= hhi; ++ho) {
      const Lerp lh = lerp_src(ho, Hi, Ho, ac);
      float wh = 0.f;
      if (lh.i0 == hi) wh += lh.l0;
      if (lh.i1 == hi) wh += lh.l1;
      if (wh == 0.f) continue;
      float rs = 0.f;
      for (int wo = wlo; wo <= whi; ++wo) {
        const Lerp lw = lerp_src(wo, Wi, Wo, ac);
        float ww = 0.f;
        if (lw.i0 == wi) ww += lw.l0;
        if (lw.i1 == wi) ww += lw.l1;
        if (ww != 0.f) rs += ww * dy[((long)(b * Ho + ho) * Wo + wo) * lddy + c];
      }
      s += wh * rs;
    }
    dx[((long)(b * Hi + hi) * Wi + wi) * lddx + c] = s;
  }
}

// ------------------------------------------------------------------ global average pool
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* x, int ldx, long HW, int C, double scale, float* y, int ldy,
                                                      double* part) {
  // y[b][c] = scale * sum over the image's HW pixels; one block per (b, 64-channel slab, pixel slice), 4 row lanes, eight
  // independent accumulators per lane (a single chain of HW/4 dependent loads was pure latency: 216 us for 2048 pixels).
  // gridDim.z > 1 (few images x few channels, e.g. batch 2 at 1024x2048: 4 blocks used to read 134 MB): every slice leaves
  // its double partial sum in part[b][z][c]; gap_finalize_kernel adds the slices in fixed order.
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * SLAB + tx, b = blockIdx.x, nz = gridDim.z;
  const long per = (HW + nz - 1) / nz, m_lo = per * blockIdx.z, m_hi = m_lo + per < HW ? m_lo + per : HW;
  double s8[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s8[u] = 0.0;
  if (c < C) {
    const float* xb = x + (long)b * HW * ldx + c;
    for (long m = m_lo + ty; m < m_hi; m += RLANES * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long mm = m + (long)u * RLANES;
        if (mm < m_hi) s8[u] += (double)xb[mm * ldx];
      }
    }
  }
  sh[ty * SLAB + tx] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  __syncthreads();
  if (ty == 0 && c < C) {
    const double t = sh[tx] + sh[SLAB + tx] + sh[2 * SLAB + tx] + sh[3 * SLAB + tx];
    if (nz == 1) y[(long)b * ldy + c] = (float)(t * scale);
    else part[((long)b * nz + blockIdx.z) * C + c] = t;
  }
}
// the same with four channels per thread (16-byte loads): 16 channel quads x 16 row lanes per block, four independent
// accumulator sets per lane
__global__ __launch_bounds__(256) void gap_fwd4_kernel(const float* x, int ldx, long HW, int C, double scale, float* y, int ldy,
                                                       double* part) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);   // [16 row lanes][64 channels]
  const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.y * SLAB + 4 * tq, b = blockIdx.x, nz = gridDim.z;
  const long per = (HW + nz - 1) / nz, m_lo = per * blockIdx.z, m_hi = m_lo + per < HW ? m_lo + per : HW;
  double s[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[u][j] = 0.0;
  if (c < C) {
    const float* xb = x + (long)b * HW * ldx + c;
    for (long m = m_lo + ty; m < m_hi; m += 16 * 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long mm = m + (long)u * 16;
        if (mm < m_hi) {
          const float4 v = *reinterpret_cast<const float4*>(xb + mm * ldx);
          s[u][0] += (double)v.x; s[u][1] += (double)v.y; s[u][2] += (double)v.z; s[u][3] += (double)v.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sh[ty * SLAB + 4 * tq + j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
  __syncthreads();
  if (threadIdx.x < SLAB) {
    const int cc = blockIdx.y * SLAB + threadIdx.x;
    if (cc < C) {
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += sh[r * SLAB + threadIdx.x];
      if (nz == 1) y[(long)b * ldy + cc] = (float)(t * scale);
      else part[((long)b * nz + blockIdx.z) * C + cc] = t;
    }
  }
}
__global__ __launch_bounds__(256) void gap_finalize_kernel(const double* part, int nz, int B, int C, double scale, float* y, int ldy) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)B * C) return;
  const int b = (int)(e / C), c = (int)(e - (long)b * C);
  double t = 0.0;
  for (int z = 0; z < nz; ++z) t += part[((long)b * nz + z) * C + c];
  y[(long)b * ldy + c] = (float)(t * scale);
}
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* dy, int B, long HW, int C, float* dx, int lddx) {
  const long total = (long)B * HW * C;
  const float inv = 1.f / (float)HW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); const long m = e / C; const long b = m / HW;
    dx[m * lddx + c] = dy[b * C + c] * inv;
  }
}

// ------------------------------------------------------------------ small elementwise ops
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* f, const float* a, long n, float* y) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256)
    y[e] = f[e] * (1.f / (1.f + expf(-a[e])));
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* dy, const float* f, const float* a, long n, float* df,
                                                       float* da) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float s = 1.f / (1.f + expf(-a[e]));
    df[e] = dy[e] * s;
    da[e] = dy[e] * f[e] * s * (1.f - s);
  }
}
__global__ __launch_bounds__(256) void axpby_kernel(long n, float alpha, const float* x, float beta, const float* y,
                                                    float* out) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256)
    out[e] = alpha * x[e] + (y ? beta * y[e] : 0.f);
}
__global__ __launch_bounds__(256) void axpby_dev_kernel(long n, const float* alpha, const float* x, const float* beta,
                                                        const float* y, float* out) {
  const float a = alpha[0], b = y ? beta[0] : 0.f;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256)
    out[e] = a * x[e] + (y ? b * y[e] : 0.f);
}
template <int VW>
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* src, int lds, float* dst, int ldd, long M, int C) {
  const int CV = C / VW;
  const long total = M * CV;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long m = e / CV; const int c = (int)(e - m * CV) * VW;
    if (VW == 4) *reinterpret_cast<float4*>(dst + m * ldd + c) = *reinterpret_cast<const float4*>(src + m * lds + c);
    else dst[m * ldd + c] = src[m * lds + c];
  }
}
// y[b, p, c] = x[b, p, c] * scale[b, c]   (nn.Dropout2d: whole channel maps dropped / rescaled; its own adjoint)
__global__ __launch_bounds__(256) void scale_channels_kernel(const float* x, int ldx, const float* scale, long HW, int C, long total,
                                                             float* y, int ldy) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); const long m = e / C; const long b = m / HW;
    y[m * ldy + c] = x[m * ldx + c] * scale[b * C + c];
  }
}
// nn.Dropout on an NHWC tensor: y = keep(e) ? x / (1 - p) : 0 with the counter-based mask of (seed, element index) -- the same
// draw the BatchNorm-fused dropout uses (segsde_uniform01), regenerated by the adjoint (the same call on the gradient)
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, int ldx, long M, int C, float p, uint64_t seed, float* y, int ldy) {
  const float ks = 1.f / (1.f - p);
  const long total = M * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C); const long m = e / C;
    y[m * ldy + c] = segsde_uniform01(seed, (uint64_t)e) >= p ? x[m * ldx + c] * ks : 0.f;
  }
}
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, int B, int C, int H, int W, float mean, float sd,
                                                           float* y, int ldy) {
  // every one of the ldy channels of a pixel is written (channels past C: zeros -- the padded network input needs no
  // separate memset), so the stores of a wave are one contiguous run
  const long HW = (long)H * W;
  if (ldy <= 8 && (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    // network inputs (3 / 6 planes -> 4 / 8 channels): one pixel per thread -- neighbouring threads read neighbouring
    // pixels of each plane and store neighbouring 16 / 32-byte pixels
    const long npix = (long)B * HW;
    for (long m = blockIdx.x * 256L + threadIdx.x; m < npix; m += (long)gridDim.x * 256) {
      const long b = m / HW, p = m - b * HW;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = (c < C && c < ldy) ? (x[(b * C + c) * HW + p] - mean) / sd : 0.f;
      float4* q = reinterpret_cast<float4*>(y + m * ldy);
      q[0] = make_float4(v[0], v[1], v[2], v[3]);
      if (ldy == 8) q[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    return;
  }
  const long total = (long)B * HW * ldy;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % ldy); const long m = e / ldy; const long b = m / HW, p = m - b * HW;
    y[e] = c < C ? (x[(b * C + c) * HW + p] - mean) / sd : 0.f;
  }
}
// the same edge for the network stems (csrc/conv_igemm.hip, "Network stems"): 4 / 8 channels per pixel, written into the interior
// of a [B][H + pt + pb][W + pl + pr][ldy] tensor whose border is zero (the normalised image's zero padding, materialised)
__global__ __launch_bounds__(256) void nchw_to_nhwc_border_kernel(const float* x, int B, int C, int H, int W, float mean, float sd,
                                                                  float* y, int ldy, int pt, int pl, int Hp, int Wp) {
  const long npix = (long)B * Hp * Wp, HW = (long)H * W;
  for (long m = blockIdx.x * 256L + threadIdx.x; m < npix; m += (long)gridDim.x * 256) {
    const long b = m / ((long)Hp * Wp); const long r = m - b * Hp * Wp;
    const int hp = (int)(r / Wp), wp = (int)(r - (long)hp * Wp), h = hp - pt, w = wp - pl;
    const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (in && c < C && c < ldy) ? (x[(b * C + c) * HW + (long)h * W + w] - mean) / sd : 0.f;
    float4* q = reinterpret_cast<float4*>(y + m * ldy);
    q[0] = make_float4(v[0], v[1], v[2], v[3]);
    if (ldy == 8) q[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* x, int ldx, int B, int C, int H, int W, float* y) {
  const long HW = (long)H * W, total = (long)B * HW * C;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e % HW; long t = e / HW; const int c = (int)(t % C); const long b = t / C;
    y[e] = x[(b * HW + p) * ldx + c];
  }
}

inline int pow2_shift(int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; }
inline int ew_blocks(long total) { long nb = (total + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb)); }
// grid of a channel-vector kernel whose stride (blocks x 256) is a multiple of the CV channel vectors per row whenever CV is a
// power of two (<= 256: any block count; 512, 1024, ...: the block count rounded up to a multiple of CV / 256)
inline int ew_blocks_cv(long total, int cv) {
  int nb = ew_blocks(total);
  if (cv > 256 && (cv & (cv - 1)) == 0) { const int q = cv / 256; nb = ((nb + q - 1) / q) * q; }
  return nb;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
#define ST(s) static_cast<hipStream_t>(s)

}  // namespace

extern "C" size_t segsde_bn_stats_workspace(long M, int C) { return part_bytes(M, C); }
extern "C" size_t segsde_bn_backward_workspace(long M, int C) { return part_bytes(M, C); }
extern "C" size_t segsde_colsum_workspace(long M, int C) { return part_bytes(M, C); }

extern "C" int segsde_bn_stats(const float* x, int ldx, long M, int C, float* mean, float* invstd, float* running_mean,
                               float* running_var, float momentum, float eps, int64_t* num_batches_tracked, void* ws,
                               size_t ws_bytes, void* stream) {
  if (!x || !mean || !invstd || !ws) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0 || ldx < C) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < part_bytes(M, C)) return SEGSDE_ERR_WORKSPACE;
  StatsOp op{x, ldx};
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && al16p(x);
  if (int e = launch_colreduce(op, M, C, (double*)ws, vec, ST(stream))) return e;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 15) / 16), dim3(256), 4096, ST(stream), (const double*)ws,
                     red_blocks(M), M, C, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

// conv-epilogue partials [rows][2][C] doubles -> partials [nb][2][C] in the colreduce format: block (b, cg) sums the rows
// b, b + nb, ... of 64 channels with 4 row-lanes, fixed order
__global__ __launch_bounds__(256) void bn_partials_reduce_kernel(const double* part, long rows, int C, int nb, double* out) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);   // [2][4][64]
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.y * 64 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (long r = blockIdx.x + (long)rl * nb; r < rows; r += 4L * nb) {
      a += part[(r * 2) * C + c]; b += part[(r * 2 + 1) * C + c];
    }
  sh[rl * 64 + cl] = a; sh[256 + rl * 64 + cl] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
    out[((long)blockIdx.x * 2) * C + c] = (sh[cl] + sh[64 + cl]) + (sh[128 + cl] + sh[192 + cl]);
    out[((long)blockIdx.x * 2 + 1) * C + c] = (sh[256 + cl] + sh[320 + cl]) + (sh[384 + cl] + sh[448 + cl]);
  }
}

// conv-epilogue partials -> mean / invstd / running statistics in ONE launch when there are only a few partial rows:
// 16 channels x 16 row-lanes per block, fixed order
__global__ __launch_bounds__(256) void bn_stats_from_partials_kernel(const double* part, long rows, long M, int C, float eps,
                                                                     float momentum, float* mean, float* invstd,
                                                                     float* running_mean, float* running_var, int64_t* nbt) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), pl = threadIdx.x >> 4;
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
  double s, q;
  combine_partials(part, (int)rows, C, c, pl, sh, s, q);
  if (pl != 0 || c >= C) return;
  const double mu = s / (double)M;
  double var = q / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// the same in one launch for a few hundred partial rows (the 32 x 64 and 64 x 128 maps: most BatchNorms of a ResNet-101 step):
// 4 channels (one 32-byte sector of doubles per row) x 64 row-lanes per block, every thread sums rows rl, rl + 64, ... in that
// order, the lanes meet in LDS and are folded in lane order -- C / 4 blocks, no second launch (round 5; the two-launch path took
// 7.9 + 4.8 us per BatchNorm, 148 of them per step)
__global__ __launch_bounds__(256) void bn_stats_from_partials_wide_kernel(const double* part, long rows, long M, int C, float eps,
                                                                          float momentum, float* mean, float* invstd,
                                                                          float* running_mean, float* running_var, int64_t* nbt) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);   // [2][64][4]
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (long r = rl; r < rows; r += 64) { a += part[(r * 2) * C + c]; b += part[(r * 2 + 1) * C + c]; }
  sh[rl * 4 + cl] = a; sh[256 + rl * 4 + cl] = b;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  double s = 0.0, q = 0.0;
  for (int j = 0; j < 64; ++j) { s += sh[j * 4 + cl]; q += sh[256 + j * 4 + cl]; }
  const double mu = s / (double)M;
  double var = q / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

namespace {
constexpr int PARTIALS_NB = 64;
long partials_wide_max() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("SEGSDE_BN_PARTIALS_WIDE_MAX"); v = e ? atol(e) : 1024; }
  return v;
}
}

extern "C" size_t segsde_bn_stats_from_partials_workspace(int C) { return (size_t)PARTIALS_NB * 2 * (C > 0 ? C : 1) * sizeof(double); }

extern "C" int segsde_bn_stats_from_partials(const double* partials, long rows, long M, int C, float* mean, float* invstd,
                                             float* running_mean, float* running_var, float momentum, float eps,
                                             int64_t* num_batches_tracked, void* ws, size_t ws_bytes, void* stream) {
  if (!partials || !mean || !invstd || !ws) return SEGSDE_ERR_NULL;
  if (rows <= 0 || M <= 0 || C <= 0) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_bn_stats_from_partials_workspace(C)) return SEGSDE_ERR_WORKSPACE;
  if (rows <= 64) {   // (tried for rows <= 2048: 16 serial row-lanes took 18.6 us against 8.7 + 4.8 us of the two parallel launches)
    hipLaunchKernelGGL(bn_stats_from_partials_kernel, dim3((C + 15) / 16), dim3(256), 4096, ST(stream), partials, rows, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  if (rows <= partials_wide_max()) {
    hipLaunchKernelGGL(bn_stats_from_partials_wide_kernel, dim3((C + 3) / 4), dim3(256), 4096, ST(stream), partials, rows, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  const int nb = rows < PARTIALS_NB ? (int)rows : PARTIALS_NB;
  hipLaunchKernelGGL(bn_partials_reduce_kernel, dim3(nb, (C + 63) / 64), dim3(256), 4096, ST(stream), partials, rows, C, nb,
                     (double*)ws);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 15) / 16), dim3(256), 4096, ST(stream), (const double*)ws, nb, M, C,
                     eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_bn_eval_stats(const float* rm, const float* rv, int C, float eps, float* mean, float* invstd,
                                    void* stream) {
  if (!rm || !rv || !mean || !invstd) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, ST(stream), rm, rv, C, eps, mean, invstd);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_bn_apply(const float* x, int ldx, long M, int C, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* residual, int ldr, float* y, int ldy,
                               int act, float drop_p, uint64_t seed, void* stream) {
  if (!x || !mean || !invstd || !y) return SEGSDE_ERR_NULL;
  if ((gamma == nullptr) != (beta == nullptr)) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0 || drop_p < 0.f || drop_p >= 1.f) return SEGSDE_ERR_SHAPE;
  const bool v4 = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && al16(x) && al16(y) &&
                  (!residual || ((ldr % 4 == 0) && al16(residual)));
  if (v4)
    hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(ew_blocks_cv(M * C / 4, C / 4)), dim3(256), 0, ST(stream), x, ldx, M, C, mean,
                       invstd, gamma, beta, residual, ldr, y, ldy, act, drop_p, seed, pow2_shift(C / 4));
  else
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(ew_blocks_cv(M * C, C)), dim3(256), 0, ST(stream), x, ldx, M, C, mean, invstd,
                       gamma, beta, residual, ldr, y, ldy, act, drop_p, seed, pow2_shift(C));
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_bn_backward(const float* dy, int lddy, const float* y, int ldy, const float* x, int ldx, long M,
                                  int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                  int act, float drop_p, uint64_t seed, int batch_stats, float* dgamma, float* dbeta,
                                  float* dx, int lddx, float* dres, int lddres, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !x || !mean || !invstd || !dgamma || !dbeta || !ws) return SEGSDE_ERR_NULL;
  if (!y) {   // remask mode: only where the mask is a function of x alone
    if (!(act == SEGSDE_ACT_NONE || (act == SEGSDE_ACT_RELU && drop_p <= 0.f && !dres && (!gamma || beta)))) return SEGSDE_ERR_NULL;
  }
  if (M <= 0 || C <= 0) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < part_bytes(M, C)) return SEGSDE_ERR_WORKSPACE;
  if (!y) ldy = ldx;
  BnBwdOp op{dy, lddy, y, ldy, x, ldx, mean, invstd, act, C, drop_p, seed, gamma, beta};
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (lddy % 4 == 0) && al16p(x) && (!y || al16p(y)) && al16p(dy) &&
                   (!gamma || (al16p(gamma) && (!beta || al16p(beta)))) && al16p(mean) && al16p(invstd);
  bool fin = false;
  if (int e = launch_colreduce(op, M, C, (double*)ws, vec, ST(stream), dgamma, dbeta, &fin)) return e;
  if (!fin) {
    hipLaunchKernelGGL(pair_finalize_kernel, dim3((C + 15) / 16), dim3(256), 4096, ST(stream), (const double*)ws,
                       red_blocks(M), C, dgamma, dbeta);
    SEGSDE_CHECK_LAUNCH();
  }
  if (dx || dres) {
    // (dgamma / dbeta may be slices of a gradient bucket: ddp.py aligns them, callers of the C ABI need not)
    const bool v4 = vec && (!dx || ((lddx % 4 == 0) && al16p(dx))) && (!dres || ((lddres % 4 == 0) && al16p(dres))) &&
                    al16p(dgamma) && al16p(dbeta);
    if (v4)
      hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_blocks_cv(M * C / 4, C / 4)), dim3(256), 0, ST(stream), dy, lddy, y, ldy, x,
                         ldx, M, C, mean, invstd, gamma, beta, act, drop_p, seed, batch_stats, (const float*)dgamma,
                         (const float*)dbeta, dx, lddx, dres, lddres, pow2_shift(C / 4));
    else
      hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_blocks_cv(M * C, C)), dim3(256), 0, ST(stream), dy, lddy, y, ldy, x, ldx,
                         M, C, mean, invstd, gamma, beta, act, drop_p, seed, batch_stats, (const float*)dgamma,
                         (const float*)dbeta, dx, lddx, dres, lddres, pow2_shift(C));
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int segsde_act_backward(const float* dy, int lddy, const float* y, int ldy, long M, int C, int act, float* dz,
                                   int lddz, float* dbias, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !y || !ws) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < part_bytes(M, C)) return SEGSDE_ERR_WORKSPACE;
  ActBwdOp op{dy, lddy, y, ldy, dz, lddz, act};
  const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && (lddy % 4 == 0) && al16p(y) && al16p(dy) &&
                   (!dz || ((lddz % 4 == 0) && al16p(dz)));
  bool fin = false;
  if (int e = launch_colreduce(op, M, C, (double*)ws, vec, ST(stream), dbias, nullptr, &fin)) return e;
  if (dbias && !fin) {
    hipLaunchKernelGGL(pair_finalize_kernel, dim3((C + 15) / 16), dim3(256), 4096, ST(stream), (const double*)ws,
                       red_blocks(M), C, dbias, (float*)nullptr);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int segsde_colsum(const float* x, int ldx, long M, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out || !ws) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < part_bytes(M, C)) return SEGSDE_ERR_WORKSPACE;
  ColsumOp op{x, ldx};
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && al16p(x);
  bool fin = false;
  if (int e = launch_colreduce(op, M, C, (double*)ws, vec, ST(stream), out, nullptr, &fin)) return e;
  if (!fin) {
    hipLaunchKernelGGL(pair_finalize_kernel, dim3((C + 15) / 16), dim3(256), 4096, ST(stream), (const double*)ws,
                       red_blocks(M), C, out, (float*)nullptr);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int segsde_maxpool3x3s2_forward(const float* x, int B, int H, int W, int C, float* y, uint8_t* idx, void* stream) {
  if (!x || !y || !idx) return SEGSDE_ERR_NULL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (C % 4 == 0 && al16(x) && al16(y) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0) {
    hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(ew_blocks((long)B * Ho * Wo * (C / 4))), dim3(256), 0, ST(stream), x, B, H, W,
                       C / 4, Ho, Wo, y, idx);
  } else {
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks((long)B * Ho * Wo * C)), dim3(256), 0, ST(stream), x, B, H, W, C,
                       Ho, Wo, y, idx);
  }
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_maxpool3x3s2_backward(const float* dy, const uint8_t* idx, int B, int H, int W, int C, float* dx,
                                            int accumulate, void* stream) {
  if (!dy || !idx || !dx) return SEGSDE_ERR_NULL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (C % 4 == 0 && (long)B * H * W * (C / 4) < (1L << 31) && al16p(dy) && al16p(dx) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0) {
    hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(ew_blocks((long)B * H * W * (C / 4))), dim3(256), 0, ST(stream), dy, idx, B, H, W,
                       C / 4, Ho, Wo, dx, accumulate);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0, ST(stream), dy, idx, B, H, W,
                     C, Ho, Wo, dx, accumulate);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_upsample2x_backward(const float* dy, int lddy, int B, int h, int w, int C, float* dx, int lddx,
                                          void* stream) {
  if (!dy || !dx) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_blocks((long)B * h * w * C)), dim3(256), 0, ST(stream), dy, lddy, B, h,
                     w, C, dx, lddx);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_upsample2x_forward(const float* x, int ldx, int B, int h, int w, int C, float* y, int ldy, void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_blocks((long)B * 4 * h * w * C)), dim3(256), 0, ST(stream), x, ldx, B, h, w, C,
                     y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_resize_bilinear_forward(const float* x, int ldx, int B, int Hi, int Wi, int C, float* y, int ldy,
                                              int Ho, int Wo, int ac, void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(ew_blocks((long)B * Ho * Wo * C)), dim3(256), 0, ST(stream), x, ldx, B, Hi, Wi,
                     C, y, ldy, Ho, Wo, ac);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_resize_bilinear_backward(const float* dy, int lddy, int B, int Hi, int Wi, int C, float* dx, int lddx,
                                               int Ho, int Wo, int ac, void* stream) {
  if (!dy || !dx) return SEGSDE_ERR_NULL;
  if (Hi == 1 && Wi == 1) {
    // a 1x1 source is broadcast by the forward pass (ASPP image-pooling branch): its gradient is the plain per-image sum
    if (C % 4 == 0 && lddy % 4 == 0 && al16(dy))
      hipLaunchKernelGGL(gap_fwd4_kernel, dim3(B, (C + SLAB - 1) / SLAB), dim3(256), 16 * SLAB * sizeof(double), ST(stream), dy,
                         lddy, (long)Ho * Wo, C, 1.0, dx, lddx, (double*)nullptr);
    else
      hipLaunchKernelGGL(gap_fwd_kernel, dim3(B, (C + SLAB - 1) / SLAB), dim3(256), RLANES * SLAB * sizeof(double), ST(stream),
                         dy, lddy, (long)Ho * Wo, C, 1.0, dx, lddx, (double*)nullptr);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(ew_blocks((long)B * Hi * Wi * C)), dim3(256), 0, ST(stream), dy, lddy, B, Hi,
                     Wi, C, dx, lddx, Ho, Wo, ac);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
namespace {
// pixel slices per image so that the launch has about a thousand blocks (1: the single-kernel path)
int gap_slices(int B, long HW, int C) {
  const long blocks = (long)B * ((C + SLAB - 1) / SLAB);
  long nz = (1024 + blocks - 1) / blocks;
  const long maxz = HW / (RLANES * 8 * 4);        // at least four trips of the inner loop per slice
  if (nz > maxz) nz = maxz;
  if (nz > 256) nz = 256;
  return nz < 1 ? 1 : (int)nz;
}
}  // namespace
extern "C" size_t segsde_global_avgpool_workspace(int B, long HW, int C) {
  const int nz = gap_slices(B, HW, C);
  return nz > 1 ? (size_t)B * nz * C * sizeof(double) : 0;
}
extern "C" int segsde_global_avgpool_forward(const float* x, int ldx, int B, long HW, int C, float* y, void* ws, size_t ws_bytes,
                                             void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  if (B <= 0 || HW <= 0 || C <= 0) return SEGSDE_ERR_SHAPE;
  const int nz = gap_slices(B, HW, C);
  if (nz > 1 && (!ws || ws_bytes < segsde_global_avgpool_workspace(B, HW, C))) return SEGSDE_ERR_WORKSPACE;
  if (C % 4 == 0 && ldx % 4 == 0 && al16(x)) {
    hipLaunchKernelGGL(gap_fwd4_kernel, dim3(B, (C + SLAB - 1) / SLAB, nz), dim3(256), 16 * SLAB * sizeof(double), ST(stream), x,
                       ldx, HW, C, 1.0 / (double)HW, y, C, (double*)ws);
  } else {
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(B, (C + SLAB - 1) / SLAB, nz), dim3(256), RLANES * SLAB * sizeof(double), ST(stream),
                       x, ldx, HW, C, 1.0 / (double)HW, y, C, (double*)ws);
  }
  SEGSDE_CHECK_LAUNCH();
  if (nz > 1) {
    hipLaunchKernelGGL(gap_finalize_kernel, dim3(segsde_cdiv((long)B * C, 256)), dim3(256), 0, ST(stream), (const double*)ws, nz,
                       B, C, 1.0 / (double)HW, y, C);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}
extern "C" int segsde_global_avgpool_backward(const float* dy, int B, long HW, int C, float* dx, int lddx, void* stream) {
  if (!dy || !dx) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_blocks((long)B * HW * C)), dim3(256), 0, ST(stream), dy, B, HW, C, dx, lddx);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_gate_forward(const float* f, const float* a, long n, float* y, void* stream) {
  if (!f || !a || !y) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), f, a, n, y);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_gate_backward(const float* dy, const float* f, const float* a, long n, float* df, float* da,
                                    void* stream) {
  if (!dy || !f || !a || !df || !da) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), dy, f, a, n, df, da);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_axpby(long n, float alpha, const float* x, float beta, const float* y, float* out, void* stream) {
  if (!x || !out) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), n, alpha, x, beta, y, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_axpby_dev(long n, const float* alpha, const float* x, const float* beta, const float* y, float* out,
                                void* stream) {
  if (!alpha || !x || !out || (y && !beta)) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(axpby_dev_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST(stream), n, alpha, x, beta, y, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_copy_channels(const float* src, int lds, float* dst, int ldd, long M, int C, void* stream) {
  if (!src || !dst) return SEGSDE_ERR_NULL;
  const bool v4 = (C % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && al16(src) && al16(dst);
  if (v4) hipLaunchKernelGGL(copy_channels_kernel<4>, dim3(ew_blocks(M * C / 4)), dim3(256), 0, ST(stream), src, lds, dst, ldd, M, C);
  else hipLaunchKernelGGL(copy_channels_kernel<1>, dim3(ew_blocks(M * C)), dim3(256), 0, ST(stream), src, lds, dst, ldd, M, C);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_scale_channels(const float* x, int ldx, int B, long HW, int C, const float* scale, float* y, int ldy,
                                     void* stream) {
  if (!x || !scale || !y) return SEGSDE_ERR_NULL;
  if (B <= 0 || HW <= 0 || C <= 0 || ldx < C || ldy < C) return SEGSDE_ERR_SHAPE;
  const long total = (long)B * HW * C;
  hipLaunchKernelGGL(scale_channels_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST(stream), x, ldx, scale, HW, C, total, y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_dropout(const float* x, int ldx, long M, int C, float p, uint64_t seed, float* y, int ldy, void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  if (M <= 0 || C <= 0 || ldx < C || ldy < C || p < 0.f || p >= 1.f) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks(M * C)), dim3(256), 0, ST(stream), x, ldx, M, C, p, seed, y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float mean, float sd, float* y, int ldy,
                                   void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks((long)B * ldy * H * W)), dim3(256), 0, ST(stream), x, B, C, H, W,
                     mean, sd, y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_nchw_to_nhwc_bordered(const float* x, int B, int C, int H, int W, float mean, float sd, float* y, int ldy,
                                            int pad_top, int pad_left, int Hp, int Wp, void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  if ((ldy != 4 && ldy != 8) || C > ldy || pad_top < 0 || pad_left < 0 || Hp < H + pad_top || Wp < W + pad_left ||
      (reinterpret_cast<uintptr_t>(y) & 15))
    return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(nchw_to_nhwc_border_kernel, dim3(ew_blocks((long)B * Hp * Wp)), dim3(256), 0, ST(stream), x, B, C, H, W, mean, sd, y,
                     ldy, pad_top, pad_left, Hp, Wp);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_nhwc_to_nchw(const float* x, int ldx, int B, int C, int H, int W, float* y, void* stream) {
  if (!x || !y) return SEGSDE_ERR_NULL;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks((long)B * C * H * W)), dim3(256), 0, ST(stream), x, ldx, B, C, H, W, y);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
