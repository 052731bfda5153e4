// Monodepth photometric loss kernels (loss/monodepth_loss.py, models/monodepth_layers.py:18-27,145-254):
// disparity upsample -> depth -> back-projection -> projection -> border-clamped bilinear warp, SSIM+L1
// reprojection error, auto-masking min, edge-aware smoothness -- forward and hand-derived backward.
// All tensors are NCHW planar fp32 exactly as the reference's loader hands them over (no layout copies).
// HBM-bound: every kernel is one coalesced pass over pixel planes; cross-pixel reductions are two-level and
// deterministic (block partials in double -> a finalize kernel), the pose-matrix gradient included.
#include "segsde_common.h"
#include <cstdlib>

namespace {
#define ST(s) static_cast<hipStream_t>(s)

struct Lerp { int i0, i1; float l0, l1; };
// F.interpolate(..., mode="bilinear", align_corners=False) source coordinates (ATen area_pixel_compute_source_index)
__device__ __forceinline__ Lerp lerp_half(int dst, int in, int out) {
  const float scale = (float)in / (float)out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp r;
  r.i0 = (int)src; if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0; r.l0 = 1.f - r.l1;
  return r;
}

struct Geo {  // per-pixel geometry shared by warp forward and backward
  float depth, xn, yn, zn;   // depth and inv_K[:3,:3] * (u, v, 1)
  float p0, p1, p2;          // P * [cam; 1]
  float x, y;                // pixel coordinates in the source frame
  float gx, gy;              // normalised grid as stored in outputs[("sample", f, s)]
  float ix, iy;              // un-normalised + border-clamped sampling position
  bool in_x, in_y;           // clamp inactive (gradient passes)
};

__device__ __forceinline__ Geo geometry(const float* disp_b, int hs, int ws, int H, int W, int h, int w,
                                        const float* iK /*[16] inv_K*/, const float* P /*[12] (K*T)[:3]*/,
                                        float min_disp, float max_disp) {
  Geo g;
  const Lerp lh = lerp_half(h, hs, H), lw = lerp_half(w, ws, W);
  const float d = lh.l0 * (lw.l0 * disp_b[lh.i0 * ws + lw.i0] + lw.l1 * disp_b[lh.i0 * ws + lw.i1]) +
                  lh.l1 * (lw.l0 * disp_b[lh.i1 * ws + lw.i0] + lw.l1 * disp_b[lh.i1 * ws + lw.i1]);
  const float scaled = min_disp + (max_disp - min_disp) * d;       // monodepth_layers.py:23-26
  g.depth = 1.f / scaled;
  const float u = (float)w, v = (float)h;                            // :155-167 pixel-centre grid
  g.xn = iK[0] * u + iK[1] * v + iK[2];
  g.yn = iK[4] * u + iK[5] * v + iK[6];
  g.zn = iK[8] * u + iK[9] * v + iK[10];
  const float cx = g.depth * g.xn, cy = g.depth * g.yn, cz = g.depth * g.zn;   // :170-172
  g.p0 = P[0] * cx + P[1] * cy + P[2] * cz + P[3];
  g.p1 = P[4] * cx + P[5] * cy + P[6] * cz + P[7];
  g.p2 = P[8] * cx + P[9] * cy + P[10] * cz + P[11];
  const float den = g.p2 + 1e-7f;                                    // :193
  g.x = g.p0 / den; g.y = g.p1 / den;
  g.gx = (g.x / (float)(W - 1) - 0.5f) * 2.f;                         // :196-198
  g.gy = (g.y / (float)(H - 1) - 0.5f) * 2.f;
  // grid_sample(align_corners=True) un-normalisation and padding_mode="border" clamp
  float ix = ((g.gx + 1.f) / 2.f) * (float)(W - 1);
  float iy = ((g.gy + 1.f) / 2.f) * (float)(H - 1);
  g.in_x = ix > 0.f && ix < (float)(W - 1);
  g.in_y = iy > 0.f && iy < (float)(H - 1);
  ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  if (!(ix == ix)) ix = 0.f;   // NaN guard (ATen clamps NaN through fmin/fmax the same way: result max())
  if (!(iy == iy)) iy = 0.f;
  g.ix = ix; g.iy = iy;
  return g;
}

// P = (K @ T)[:3, :]   (monodepth_layers.py:189), computed by 12 threads per block into LDS
__device__ __forceinline__ void load_P(const float* K, const float* T, float* P, float* iKs, const float* inv_K) {
  const int t = threadIdx.x;
  if (t < 12) {
    const int r = t >> 2, c = t & 3;
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += K[r * 4 + k] * T[k * 4 + c];
    P[t] = s;
  }
  if (t >= 16 && t < 32) iKs[t - 16] = inv_K[t - 16];
  __syncthreads();
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* disp, int hs, int ws, const float* inv_K,
                                                       const float* K, const float* T, const float* src, int H, int W,
                                                       float min_disp, float max_disp, float* color, float* grid,
                                                       float* depth) {
  SEGSDE_SMEM;
  float* P = reinterpret_cast<float*>(segsde_smem);
  float* iK = P + 16;
  const int b = blockIdx.y;
  load_P(K + b * 16, T + b * 16, P, iK, inv_K + b * 16);
  const long HW = (long)H * W;
  const float* disp_b = disp + (long)b * hs * ws;
  const float* src_b = src + (long)b * 3 * HW;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)((unsigned)p / (unsigned)W), w = (int)p - h * W;   // p < H*W < 2^31: 32-bit division
    const Geo g = geometry(disp_b, hs, ws, H, W, h, w, iK, P, min_disp, max_disp);
    if (depth) depth[b * HW + p] = g.depth;
    if (grid) { grid[(b * HW + p) * 2] = g.gx; grid[(b * HW + p) * 2 + 1] = g.gy; }
    const float fx = floorf(g.ix), fy = floorf(g.iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = g.ix - fx, wx0 = (float)x1 - g.ix, wy1 = g.iy - fy, wy0 = (float)y1 - g.iy;
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    const bool vx1 = x1 <= W - 1, vy1 = y1 <= H - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* s = src_b + c * HW;
      float v = s[(long)y0 * W + x0] * nw;
      if (vx1) v += s[(long)y0 * W + x1] * ne;
      if (vy1) v += s[(long)y1 * W + x0] * sw;
      if (vx1 && vy1) v += s[(long)y1 * W + x1] * se;
      color[(b * 3 + c) * HW + p] = v;
    }
  }
}

// generate_depth_test_pred (loss/monodepth_loss.py:54-62): upsample + disp_to_depth only
__global__ __launch_bounds__(256) void disp_to_depth_kernel(const float* disp, int hs, int ws, int H, int W, float min_disp,
                                                            float max_disp, float* depth) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  const float* disp_b = disp + (long)b * hs * ws;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)((unsigned)p / (unsigned)W), w = (int)p - h * W;
    const Lerp lh = lerp_half(h, hs, H), lw = lerp_half(w, ws, W);
    const float d = lh.l0 * (lw.l0 * disp_b[lh.i0 * ws + lw.i0] + lw.l1 * disp_b[lh.i0 * ws + lw.i1]) +
                    lh.l1 * (lw.l0 * disp_b[lh.i1 * ws + lw.i0] + lw.l1 * disp_b[lh.i1 * ws + lw.i1]);
    depth[b * HW + p] = 1.f / (min_disp + (max_disp - min_disp) * d);
  }
}

// adjoint: gcolor [B,3,H,W] -> g_disp_up [B,H,W] (+=) and per-block partial sums of dL/dP (3x4) per batch element
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* gcolor, const float* disp, int hs, int ws,
                                                       const float* inv_K, const float* K, const float* T,
                                                       const float* src, int H, int W, float min_disp, float max_disp,
                                                       float* g_disp_up, double* gP_part) {
  SEGSDE_SMEM;
  float* P = reinterpret_cast<float*>(segsde_smem);
  float* iK = P + 16;
  double* sh = reinterpret_cast<double*>(segsde_smem + 256);
  const int b = blockIdx.y;
  load_P(K + b * 16, T + b * 16, P, iK, inv_K + b * 16);
  const long HW = (long)H * W;
  const float* disp_b = disp + (long)b * hs * ws;
  const float* src_b = src + (long)b * 3 * HW;
  double acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)((unsigned)p / (unsigned)W), w = (int)p - h * W;   // p < H*W < 2^31: 32-bit division
    const Geo g = geometry(disp_b, hs, ws, H, W, h, w, iK, P, min_disp, max_disp);
    const float fx = floorf(g.ix), fy = floorf(g.iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx1 = x1 <= W - 1, vy1 = y1 <= H - 1;
    float gix = 0.f, giy = 0.f;   // ATen grid_sampler_2d_backward
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* s = src_b + c * HW;
      const float go = gcolor[(b * 3 + c) * HW + p];
      const float vnw = s[(long)y0 * W + x0];
      const float vne = vx1 ? s[(long)y0 * W + x1] : 0.f;
      const float vsw = vy1 ? s[(long)y1 * W + x0] : 0.f;
      const float vse = (vx1 && vy1) ? s[(long)y1 * W + x1] : 0.f;
      gix -= vnw * ((float)y1 - g.iy) * go; giy -= vnw * ((float)x1 - g.ix) * go;
      gix += vne * ((float)y1 - g.iy) * go; giy -= vne * (g.ix - (float)x0) * go;
      gix -= vsw * (g.iy - (float)y0) * go; giy += vsw * ((float)x1 - g.ix) * go;
      gix += vse * (g.iy - (float)y0) * go; giy += vse * (g.ix - (float)x0) * go;
    }
    // d ix / d x = ((W-1)/2) * (2/(W-1)) = 1 where the border clamp is inactive, else 0
    const float g_x = g.in_x ? gix : 0.f, g_y = g.in_y ? giy : 0.f;
    const float den = g.p2 + 1e-7f;
    const float gp0 = g_x / den, gp1 = g_y / den, gp2 = -(g_x * g.x + g_y * g.y) / den;
    const float cx = g.depth * g.xn, cy = g.depth * g.yn, cz = g.depth * g.zn;
    acc[0] += gp0 * cx; acc[1] += gp0 * cy; acc[2] += gp0 * cz; acc[3] += gp0;
    acc[4] += gp1 * cx; acc[5] += gp1 * cy; acc[6] += gp1 * cz; acc[7] += gp1;
    acc[8] += gp2 * cx; acc[9] += gp2 * cy; acc[10] += gp2 * cz; acc[11] += gp2;
    const float gcx = P[0] * gp0 + P[4] * gp1 + P[8] * gp2;
    const float gcy = P[1] * gp0 + P[5] * gp1 + P[9] * gp2;
    const float gcz = P[2] * gp0 + P[6] * gp1 + P[10] * gp2;
    const float gdepth = gcx * g.xn + gcy * g.yn + gcz * g.zn;
    const float gscaled = -gdepth * g.depth * g.depth;
    g_disp_up[b * HW + p] += gscaled * (max_disp - min_disp);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const double r = segsde_block_sum(acc[i], sh);
    if (threadIdx.x == 0) gP_part[((long)b * gridDim.x + blockIdx.x) * 12 + i] = r;
  }
}

// gT[b] += K[b]^T (rows 0..2) * gP[b]
// 12 sums over the block partials per batch item: 12 elements x 16 lanes in parallel, fixed order (a single thread per
// element took 214 us per call)
__global__ __launch_bounds__(256) void warp_bwd_finalize_kernel(const double* gP_part, int nblk, const float* K, int B,
                                                                float* gT) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);   // [12][16] lane sums, then [12] totals at sh + 192
  double* gp = sh + 192;
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < 192) {
    const int e = t >> 4, l = t & 15;
    double a = 0.0;
    for (int i = l; i < nblk; i += 16) a += gP_part[((long)b * nblk + i) * 12 + e];
    sh[e * 16 + l] = a;
  }
  __syncthreads();
  if (t < 12) {
    double a = 0.0;
    for (int l = 0; l < 16; ++l) a += sh[t * 16 + l];
    gp[t] = a;
  }
  __syncthreads();
  if (t >= 16) return;
  const int k = t >> 2, j = t & 3;
  double s = 0.0;
  for (int r = 0; r < 3; ++r) s += (double)K[b * 16 + r * 4 + k] * gp[r * 4 + j];
  gT[b * 16 + t] += (float)s;
}

// ---------------------------------------------------------------------------------------- SSIM + L1
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

struct Stats { float mx, my, sxx, syy, sxy; };
__device__ __forceinline__ Stats window_stats(const float* x, const float* y, int H, int W, int h, int w) {
  float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
  for (int dh = -1; dh <= 1; ++dh) {
    const int hh = refl(h + dh, H);
#pragma unroll
    for (int dw = -1; dw <= 1; ++dw) {
      const int ww = refl(w + dw, W);
      const float a = x[(long)hh * W + ww], b = y[(long)hh * W + ww];
      sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
    }
  }
  Stats s;
  s.mx = sx / 9.f; s.my = sy / 9.f; s.sxx = sxx / 9.f; s.syy = syy / 9.f; s.sxy = sxy / 9.f;
  return s;
}

__global__ __launch_bounds__(256) void reproj_err_fwd_kernel(const float* pred, const float* target, int H, int W,
                                                             int no_ssim, float* err, long err_bs) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)((unsigned)p / (unsigned)W), w = (int)p - h * W;   // p < H*W < 2^31: 32-bit division
    float l1 = 0.f, ss = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* x = pred + (b * 3 + c) * HW;
      const float* y = target + (b * 3 + c) * HW;
      l1 += fabsf(y[p] - x[p]);
      if (!no_ssim) {
        const Stats s = window_stats(x, y, H, W, h, w);
        const float sig_x = s.sxx - s.mx * s.mx, sig_y = s.syy - s.my * s.my, sig_xy = s.sxy - s.mx * s.my;
        const float n = (2.f * s.mx * s.my + C1) * (2.f * sig_xy + C2);
        const float d = (s.mx * s.mx + s.my * s.my + C1) * (sig_x + sig_y + C2);
        ss += fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
      }
    }
    l1 = l1 / 3.f;
    err[b * err_bs + p] = no_ssim ? l1 : (0.85f * (ss / 3.f) + 0.15f * l1);
  }
}

// pass 1 of the backward: per window centre q and channel, coefficients (a, bx, by) such that
// d err_q / d x_p = a + bx * x_p + by * y_p for every padded cell p of q's 3x3 window (already times gerr_q)
__global__ __launch_bounds__(256) void ssim_coef_kernel(const float* pred, const float* target, const float* gerr,
                                                        long gerr_bs, int H, int W, float* coef /*[B][3][3][H][W]*/) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)((unsigned)p / (unsigned)W), w = (int)p - h * W;   // p < H*W < 2^31: 32-bit division
    const float gq = gerr[b * gerr_bs + p] * (0.85f / 3.f) * (-0.5f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* x = pred + (b * 3 + c) * HW;
      const float* y = target + (b * 3 + c) * HW;
      const Stats s = window_stats(x, y, H, W, h, w);
      const float sig_x = s.sxx - s.mx * s.mx, sig_y = s.syy - s.my * s.my, sig_xy = s.sxy - s.mx * s.my;
      const float A1 = 2.f * s.mx * s.my + C1, A2 = 2.f * sig_xy + C2;
      const float B1 = s.mx * s.mx + s.my * s.my + C1, B2 = sig_x + sig_y + C2;
      const float n = A1 * A2, d = B1 * B2, r = n / d;
      const float raw = (1.f - r) / 2.f;
      const float m = (raw >= 0.f && raw <= 1.f) ? gq / (9.f * d) : 0.f;   // clamp passes gradient on [0,1]
      float* o = coef + ((long)(b * 3 + c) * 3) * HW + p;
      o[0] = m * (2.f * s.my * A2 - 2.f * A1 * s.my - r * (2.f * s.mx * B2 - 2.f * B1 * s.mx));
      o[HW] = m * (-r * 2.f * B1);
      o[2 * HW] = m * (2.f * A1);
    }
  }
}

// pass 2: gather the coefficient maps over every (padded pre-image of p) x (window centre) pair, add the L1 term
__global__ __launch_bounds__(256) void reproj_err_bwd_kernel(const float* pred, const float* target, const float* gerr,
                                                             long gerr_bs, const float* coef, int H, int W, int no_ssim,
                                                             float* gpred) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)((unsigned)p / (unsigned)W), w = (int)p - h * W;   // p < H*W < 2^31: 32-bit division
    int hp[3], wp[3], nh = 0, nw = 0;
    hp[nh++] = h; if (h == 1) hp[nh++] = -1; if (h == H - 2) hp[nh++] = H;
    wp[nw++] = w; if (w == 1) wp[nw++] = -1; if (w == W - 2) wp[nw++] = W;
    const float gl1 = gerr[b * gerr_bs + p] * (no_ssim ? (1.f / 3.f) : (0.15f / 3.f));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = pred[(b * 3 + c) * HW + p], yv = target[(b * 3 + c) * HW + p];
      float g = 0.f;
      if (!no_ssim && h >= 2 && h <= H - 3 && w >= 2 && w <= W - 3) {
        // interior pixel: exactly the nine windows around it, no mirrored pre-images, no range checks -- fully unrolled
        const float* o = coef + ((long)(b * 3 + c) * 3) * HW + p;
        float sa = 0.f, sbx = 0.f, sby = 0.f;
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
          for (int dw = -1; dw <= 1; ++dw) {
            const long q = (long)dh * W + dw;
            sa += o[q]; sbx += o[HW + q]; sby += o[2 * HW + q];
          }
        g = sa + sbx * xv + sby * yv;
      } else if (!no_ssim) {
        const float* o = coef + ((long)(b * 3 + c) * 3) * HW;
        float sa = 0.f, sbx = 0.f, sby = 0.f;
        for (int a = 0; a < nh; ++a)
          for (int bb = 0; bb < nw; ++bb)
            for (int dh = -1; dh <= 1; ++dh) {
              const int qh = hp[a] + dh;
              if (qh < 0 || qh >= H) continue;
              for (int dw = -1; dw <= 1; ++dw) {
                const int qw = wp[bb] + dw;
                if (qw < 0 || qw >= W) continue;
                const long q = (long)qh * W + qw;
                sa += o[q]; sbx += o[HW + q]; sby += o[2 * HW + q];
              }
            }
        g = sa + sbx * xv + sby * yv;
      }
      const float df = yv - xv;   // d|t - x|/dx = -sign(t - x)
      g += gl1 * (df > 0.f ? -1.f : (df < 0.f ? 1.f : 0.f));
      gpred[(b * 3 + c) * HW + p] = g;
    }
  }
}

// ---------------------------------------------------------------------------------------- auto-mask min
// n source frames (monodepth_loss.py:136-177 loops over frame_ids[1:], whatever their number): ident / reproj are [B,n,H,W], noise
// [B, avg ? 1 : n, H, W]; avg = the channel mean of each group first (torch's mean: the sum in channel order, divided by n)
#define SEGSDE_AUTOMASK_MAX_FRAMES 8
__global__ __launch_bounds__(256) void automask_fwd_kernel(const float* ident, const float* noise, const float* reproj,
                                                           int n_reproj, int avg, long total, long HW, uint8_t* sel,
                                                           float* isel, double* part) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int ni = ident ? (avg ? 1 : n_reproj) : 0;
  const float fn = (float)n_reproj;
  double acc = 0.0;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / HW, p = e - b * HW;
    float best = 0.f; int bi = 0, n = 0;
    auto offer = [&](float v) { if (n == 0 || v < best) { best = v; bi = n; } ++n; };   // the first minimum wins (torch.min)
    if (ident) {
      if (avg) {
        float a = ident[(b * n_reproj) * HW + p];
        for (int j = 1; j < n_reproj; ++j) a += ident[(b * n_reproj + j) * HW + p];
        a = a / fn;
        if (noise) a += noise[b * HW + p] * 0.00001f;
        offer(a);
      } else {
        for (int j = 0; j < n_reproj; ++j) {
          float a = ident[(b * n_reproj + j) * HW + p];
          if (noise) a += noise[(b * n_reproj + j) * HW + p] * 0.00001f;
          offer(a);
        }
      }
    }
    if (avg && n_reproj > 1) {
      float a = reproj[(b * n_reproj) * HW + p];
      for (int j = 1; j < n_reproj; ++j) a += reproj[(b * n_reproj + j) * HW + p];
      offer(a / fn);
    } else {
      for (int j = 0; j < n_reproj; ++j) offer(reproj[(b * n_reproj + j) * HW + p]);
    }
    sel[e] = (uint8_t)bi;
    if (isel) isel[e] = bi > ni - 1 ? 1.f : 0.f;
    acc += (double)best;
  }
  const double r = segsde_block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const double* part, int n, float* out, int out_idx, double scale) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  const double r = segsde_block_sum(a, sh);
  if (threadIdx.x == 0) out[out_idx] = (float)(r * scale);
}

__global__ __launch_bounds__(256) void automask_bwd_kernel(const uint8_t* sel, int ni, int n_reproj, int avg, long total,
                                                           long HW, float scale, float* greproj) {
  const float ga = scale / (float)n_reproj;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / HW, p = e - b * HW;
    const int s = sel[e];
    if (avg && n_reproj > 1) {
      const float g = (s == ni) ? ga : 0.f;
      for (int j = 0; j < n_reproj; ++j) greproj[(b * n_reproj + j) * HW + p] = g;
    } else {
      for (int j = 0; j < n_reproj; ++j) greproj[(b * n_reproj + j) * HW + p] = (s == ni + j) ? scale : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void plane_sum_kernel(const float* x, long n, double* part) {
  // grid (nblk, B): per-batch plane sums
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const float* xb = x + blockIdx.y * n;
  double a = 0.0;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) a += (double)xb[e];
  const double r = segsde_block_sum(a, sh);
  if (threadIdx.x == 0) part[(long)blockIdx.y * gridDim.x + blockIdx.x] = r;
}
__global__ __launch_bounds__(64) void plane_mean_finalize_kernel(const double* part, int nblk, double inv_n, float* out) {
  if (threadIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[(long)blockIdx.x * nblk + i];
  out[blockIdx.x] = (float)(s * inv_n);
}

__device__ __forceinline__ float edge_w(const float* img, long HW, long p, long q) {
  const float g = (fabsf(img[p] - img[q]) + fabsf(img[HW + p] - img[HW + q]) + fabsf(img[2 * HW + p] - img[2 * HW + q])) / 3.f;
  return expf(-g);
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* disp, const float* img, const float* mean_disp,
                                                         int h, int w, double* part /*[B][nblk][2]*/) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int b = blockIdx.y;
  const long HW = (long)h * w;
  const float* d = disp + b * HW;
  const float* im = img + (long)b * 3 * HW;
  const float inv = mean_disp[b] + 1e-7f;
  double ax = 0.0, ay = 0.0;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int y = (int)((unsigned)p / (unsigned)w), x = (int)p - y * w;   // p < h*w < 2^31: 32-bit division
    const float dn = d[p] / inv;
    if (x + 1 < w) ax += (double)(fabsf(dn - d[p + 1] / inv) * edge_w(im, HW, p, p + 1));
    if (y + 1 < h) ay += (double)(fabsf(dn - d[p + w] / inv) * edge_w(im, HW, p, p + w));
  }
  const double rx = segsde_block_sum(ax, sh);
  const double ry = segsde_block_sum(ay, sh);
  if (threadIdx.x == 0) {
    part[((long)b * gridDim.x + blockIdx.x) * 2] = rx;
    part[((long)b * gridDim.x + blockIdx.x) * 2 + 1] = ry;
  }
}
// 256 lanes sum strided subsets of the block partials, then a fixed-order combine (one thread took 383 us per call)
__global__ __launch_bounds__(256) void smooth_finalize_kernel(const double* part, int n, double inv_nx, double inv_ny,
                                                              float* out) {
  SEGSDE_SMEM;
  double* shx = reinterpret_cast<double*>(segsde_smem);
  double* shy = shx + 256;
  const int t = threadIdx.x;
  double sx = 0.0, sy = 0.0;
  for (int i = t; i < n; i += 256) { sx += part[2 * i]; sy += part[2 * i + 1]; }
  shx[t] = sx; shy[t] = sy;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { shx[t] += shx[t + o]; shy[t] += shy[t + o]; }
    __syncthreads();
  }
  if (t == 0) out[0] = (float)(shx[0] * inv_nx) + (float)(shy[0] * inv_ny);
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// pass A: G = dL/d(normalised disparity) per pixel -> tmp, and per-block partials of sum(G * disp)
__global__ __launch_bounds__(256) void smooth_bwd_a_kernel(const float* disp, const float* img, const float* mean_disp,
                                                           int h, int w, float sx, float sy, float* tmp, double* part) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int b = blockIdx.y;
  const long HW = (long)h * w;
  const float* d = disp + b * HW;
  const float* im = img + (long)b * 3 * HW;
  const float inv = mean_disp[b] + 1e-7f;
  double acc = 0.0;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int y = (int)((unsigned)p / (unsigned)w), x = (int)p - y * w;   // p < h*w < 2^31: 32-bit division
    const float dn = d[p] / inv;
    float g = 0.f;
    if (x + 1 < w) g += sx * sgn(dn - d[p + 1] / inv) * edge_w(im, HW, p, p + 1);
    if (x > 0) g -= sx * sgn(d[p - 1] / inv - dn) * edge_w(im, HW, p - 1, p);
    if (y + 1 < h) g += sy * sgn(dn - d[p + w] / inv) * edge_w(im, HW, p, p + w);
    if (y > 0) g -= sy * sgn(d[p - w] / inv - dn) * edge_w(im, HW, p - w, p);
    tmp[b * HW + p] = g;
    acc += (double)g * (double)d[p];
  }
  const double r = segsde_block_sum(acc, sh);
  if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = r;
}
// pass B: gdisp += G/(m+eps) - S/((m+eps)^2 * n)
__global__ __launch_bounds__(256) void smooth_bwd_b_kernel(const float* tmp, const float* mean_disp, const double* part,
                                                           int nblk, long HW, float* gdisp) {
  const int b = blockIdx.y;
  double S = 0.0;
  for (int i = 0; i < nblk; ++i) S += part[(long)b * nblk + i];
  const float inv = mean_disp[b] + 1e-7f;
  const float corr = (float)(S / ((double)inv * (double)inv * (double)HW));
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256)
    gdisp[b * HW + p] += tmp[b * HW + p] / inv - corr;
}

// ======================================================================================== fused photometric kernels
// One forward and one backward launch per scale (both source frames at once) instead of eleven per-stage launches:
// target / warped frames are staged once per 32x8 pixel tile (+ halo, reflection applied while staging) in LDS, the
// 3x3 SSIM windows read LDS, SSIM+L1 -> auto-mask minimum stay in registers, and the backward pass recomputes the window
// statistics in the tile instead of reading a 9-plane coefficient workspace: per-window-centre coefficients live in LDS
// for the tile + 1 halo, the 3x3 gather with the reflection fold runs on LDS, and the resulting d err / d warped feeds
// the warp adjoint (source taps, d/d disparity, d/d pose partial sums) in the same thread -- neither the error planes,
// nor their gradients, nor the coefficient maps ever exist in HBM.
constexpr int PT_W = 32, PT_H = 8;                 // pixel tile of a 256-thread block
constexpr int PF_W = PT_W + 2, PF_H = PT_H + 2;    // forward staging: 1-pixel halo
constexpr int PB_W = PT_W + 4, PB_H = PT_H + 4;    // backward staging: 2-pixel halo (statistics for tile + 1)
constexpr int PC_W = PT_W + 2, PC_H = PT_H + 2;    // coefficient tile: tile + 1

__device__ __forceinline__ int refl_clamp(int i, int n) {
  i = i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
  return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);       // only positions outside a partial tile's image part get clamped
}

// Staging of the 9 planes [target c0..2 | pred0 c0..2 | pred1 c0..2][rows][cols] of one tile (+ halo) into LDS, reflection
// applied while staging so the window code never sees a border.  A thread owns the same one or two (row, col) positions
// of the staged rectangle for every tile of its strip: the row part of the address is computed once per block, the
// column part once per tile, and the nine plane loads of a position differ by constant offsets.
struct StagePos { int e[2]; int c[2]; long rowoff[2]; };
__device__ __forceinline__ StagePos stage_setup(int rows, int cols, int h_top, int H, int W) {
  StagePos sp;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int r = e / cols;
    sp.e[i] = e < rows * cols ? e : -1;
    sp.c[i] = e - r * cols;
    sp.rowoff[i] = (long)refl_clamp(h_top + r, H) * W;
  }
  return sp;
}
__device__ __forceinline__ void stage_tile(float* sm, const StagePos& sp, int per, const float* target_b, const float* pred0_b,
                                           const float* pred1_b, long HW, int w_left, int W) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (sp.e[i] < 0) continue;
    const long off = sp.rowoff[i] + refl_clamp(w_left + sp.c[i], W);
    float v[9];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { v[ch] = target_b[ch * HW + off]; v[3 + ch] = pred0_b[ch * HW + off]; v[6 + ch] = pred1_b[ch * HW + off]; }
#pragma unroll
    for (int pl = 0; pl < 9; ++pl) sm[pl * per + sp.e[i]] = v[pl];
  }
}

// SSIM + L1 error of one (pred, target) pair at tile position (r, c) of staged planes with row pitch `cols`
// (arithmetic identical to reproj_err_fwd_kernel / window_stats: same summation order)
__device__ __forceinline__ float tile_error(const float* px, const float* py, int cols, int per, int r, int c, int no_ssim) {
  float l1 = 0.f, ss = 0.f;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* x = px + ch * per;
    const float* y = py + ch * per;
    l1 += fabsf(y[r * cols + c] - x[r * cols + c]);
    if (!no_ssim) {
      float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
      for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
          const float a = x[(r + dh) * cols + c + dw], b = y[(r + dh) * cols + c + dw];
          sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
      const float mx = sx / 9.f, my = sy / 9.f;
      const float sig_x = sxx / 9.f - mx * mx, sig_y = syy / 9.f - my * my, sig_xy = sxy / 9.f - mx * my;
      const float n = (2.f * mx * my + C1) * (2.f * sig_xy + C2);
      const float d = (mx * mx + my * my + C1) * (sig_x + sig_y + C2);
      ss += fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
    }
  }
  l1 = l1 / 3.f;
  return no_ssim ? l1 : (0.85f * (ss / 3.f) + 0.15f * l1);
}

// XCD-aware block order for the tiled photometric kernels: the hardware deals consecutive workgroups round-robin over the
// 8 XCDs, each with its own L2; after the remap every XCD owns a contiguous range of (image, tile row, strip) triples, so the
// vertically adjacent strips that share two (forward: one) halo rows hit the same L2 instead of fetching them through eight
// (PMC FETCH_SIZE before: 2.5x the algorithmic bytes of the backward launch)
struct PhotoBlk { int bx, by, bz; };
__device__ __forceinline__ PhotoBlk photo_block() {
  const int nb = gridDim.x * gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int t = segsde_xcd_remap(lin, nb);
  PhotoBlk r;
  r.bx = t % (int)gridDim.x;
  const int q = t / (int)gridDim.x;
  r.by = q % (int)gridDim.y; r.bz = q / (int)gridDim.y;
  return r;
}

// IDENT = true : err planes of (src0, target), (src1, target) -> out_err [B,2,H,W]         (monodepth_loss.py:139-147)
// IDENT = false: errors of the two warped frames + auto-mask minimum -> sel, identity_selection, block partial sums
template <bool IDENT>
__global__ __launch_bounds__(256) void photometric_fwd_kernel(const float* pred0, const float* pred1, const float* target,
                                                              const float* ident, const float* noise, int H, int W,
                                                              int no_ssim, int avg, int tiles_per_block, float* out_err,
                                                              uint8_t* sel, float* isel, double* part) {
  SEGSDE_SMEM;
  float* sm = reinterpret_cast<float*>(segsde_smem);           // [9][PF_H][PF_W]: target 0-2, pred0 3-5, pred1 6-8
  double* sh = reinterpret_cast<double*>(sm + 9 * PF_H * PF_W + ((9 * PF_H * PF_W) & 1));
  const PhotoBlk pb = photo_block();
  const int b = pb.bz, h0 = pb.by * PT_H;
  const long HW = (long)H * W;
  const float* tb = target + (long)b * 3 * HW;
  const float* p0b = pred0 + (long)b * 3 * HW;
  const float* p1b = pred1 + (long)b * 3 * HW;
  const int per = PF_H * PF_W;
  const StagePos sp = stage_setup(PF_H, PF_W, h0 - 1, H, W);
  const int r = threadIdx.x / PT_W, c = threadIdx.x - r * PT_W;
  const int h = h0 + r;
  const int ntx = (W + PT_W - 1) / PT_W;
  double acc = 0.0;
  for (int tx = pb.bx * tiles_per_block; tx < ntx && tx < (pb.bx + 1) * tiles_per_block; ++tx) {
    const int w0 = tx * PT_W, w = w0 + c;
    __syncthreads();                                           // the previous tile's window reads are done
    stage_tile(sm, sp, per, tb, p0b, p1b, HW, w0 - 1, W);
    __syncthreads();
    if (h < H && w < W) {
      const long p = (long)h * W + w;
      const float e0 = tile_error(sm + 3 * per, sm, PF_W, per, r + 1, c + 1, no_ssim);
      const float e1 = tile_error(sm + 6 * per, sm, PF_W, per, r + 1, c + 1, no_ssim);
      if (IDENT) {
        out_err[((long)b * 2) * HW + p] = e0;
        out_err[((long)b * 2 + 1) * HW + p] = e1;
      } else {
        // combined = cat(identity (+ noise * 1e-5), reprojection); min over dim 1; first minimum wins (monodepth_loss.py:147-177)
        float v[4]; int n = 0;
        const int ni = ident ? (avg ? 1 : 2) : 0;
        if (ident) {
          const float i0 = ident[((long)b * 2) * HW + p], i1 = ident[((long)b * 2 + 1) * HW + p];
          if (avg) { v[n] = (i0 + i1) / 2.f; if (noise) v[n] += noise[(long)b * HW + p] * 0.00001f; ++n; }
          else {
            v[n] = i0; if (noise) v[n] += noise[((long)b * 2) * HW + p] * 0.00001f; ++n;
            v[n] = i1; if (noise) v[n] += noise[((long)b * 2 + 1) * HW + p] * 0.00001f; ++n;
          }
        }
        if (avg) v[n++] = (e0 + e1) / 2.f;
        else { v[n++] = e0; v[n++] = e1; }
        float best = v[0]; int bi = 0;
        for (int j = 1; j < n; ++j) if (v[j] < best) { best = v[j]; bi = j; }
        sel[(long)b * HW + p] = (uint8_t)bi;
        if (isel) isel[(long)b * HW + p] = bi > ni - 1 ? 1.f : 0.f;
        acc += (double)best;
      }
    }
  }
  if (!IDENT) {
    const double t = segsde_block_sum(acc, sh);
    if (threadIdx.x == 0) part[((long)b * gridDim.y + pb.by) * gridDim.x + pb.bx] = t;
  }
}

// ---- packed variants (round 4).  PMC (profiles/pmc_r0{3,4}_sq_waits.txt) shows these kernels bound by VALU issue, not by HBM:
// 1 116 (forward) / 1 893 (backward) vector instructions per wave and tile = 0.88 / 0.71 of the issue slots of the launch.  The two
// source frames of a pixel run the same arithmetic on different data against the SAME target window, so the packed kernels keep
// (frame 0, frame 1) in the two halves of a 64-bit register pair -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do both frames in
// one issue slot -- and compute the target's window sums once instead of once per frame.  Same operations in the same order
// per frame as the per-stage kernels (bit-identical errors, hence selections).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat2(float v) { f2 r; r.x = v; r.y = v; return r; }
// x / c for a constant c, correctly rounded like the IEEE division it replaces (Markstein: q = RN(x * RN(1/c)), the exact
// residual x - c q through one fma, one fma to apply it) in 3 issue slots instead of the ~10 of v_div_scale / v_rcp /
// v_div_fmas / v_div_fixup.  Checked against x / 9.f and x / 3.f for EVERY non-negative float, denormals included: identical.
template <int C> __device__ __forceinline__ f2 div_by(f2 x) {
  constexpr float c = (float)C, rc = 1.f / (float)C;
  const f2 q = x * rc;
  return __builtin_elementwise_fma(__builtin_elementwise_fma(splat2(-c), q, x), splat2(rc), q);
}
template <int C> __device__ __forceinline__ float div_by(float x) {
  constexpr float c = (float)C, rc = 1.f / (float)C;
  const float q = x * rc;
  return __builtin_fmaf(__builtin_fmaf(-c, q, x), rc, q);
}

// staging for the packed kernels: target planes [3][per] floats, then the two frames interleaved [3][per] f2
__device__ __forceinline__ void stage_tile2(float* smt, f2* smp, const StagePos& sp, int per, const float* target_b,
                                            const float* pred0_b, const float* pred1_b, long HW, int w_left, int W) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (sp.e[i] < 0) continue;
    const long off = sp.rowoff[i] + refl_clamp(w_left + sp.c[i], W);
    float t[3]; f2 q[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { t[ch] = target_b[ch * HW + off]; q[ch].x = pred0_b[ch * HW + off]; q[ch].y = pred1_b[ch * HW + off]; }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { smt[ch * per + sp.e[i]] = t[ch]; smp[ch * per + sp.e[i]] = q[ch]; }
  }
}

// (error of frame 0, error of frame 1) at tile position (r, c): tile_error for both frames at once
__device__ __forceinline__ f2 tile_error2(const f2* px, const float* py, int cols, int per, int r, int c, int no_ssim) {
  f2 l1 = splat2(0.f), ss = splat2(0.f);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const f2* x = px + ch * per;
    const float* y = py + ch * per;
    l1 += __builtin_elementwise_abs(y[r * cols + c] - x[r * cols + c]);
    if (!no_ssim) {
      f2 sx = splat2(0.f), sxx = splat2(0.f), sxy = splat2(0.f);
      float sy = 0.f, syy = 0.f;
#pragma unroll
      for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
          const f2 a = x[(r + dh) * cols + c + dw];
          const float b = y[(r + dh) * cols + c + dw];
          sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
      const f2 mx = div_by<9>(sx);
      const float my = div_by<9>(sy);
      const f2 sig_x = div_by<9>(sxx) - mx * mx, sig_xy = div_by<9>(sxy) - mx * my;
      const float sig_y = div_by<9>(syy) - my * my;
      const f2 n = (2.f * mx * my + C1) * (2.f * sig_xy + C2);
      const f2 d = (mx * mx + my * my + C1) * (sig_x + sig_y + C2);
      ss += __builtin_elementwise_min(__builtin_elementwise_max((1.f - n / d) / 2.f, splat2(0.f)), splat2(1.f));
    }
  }
  l1 = div_by<3>(l1);
  return no_ssim ? l1 : (0.85f * div_by<3>(ss) + 0.15f * l1);
}

template <bool IDENT>
__global__ __launch_bounds__(256) void photometric_fwd2_kernel(const float* pred0, const float* pred1, const float* target,
                                                               const float* ident, const float* noise, int H, int W,
                                                               int no_ssim, int avg, int tiles_per_block, float* out_err,
                                                               uint8_t* sel, float* isel, double* part) {
  SEGSDE_SMEM;
  const int per = PF_H * PF_W;                                   // even: the f2 planes stay 8-byte aligned
  float* smt = reinterpret_cast<float*>(segsde_smem);            // [3][per] target
  f2* smp = reinterpret_cast<f2*>(smt + 3 * per + ((3 * per) & 1));   // [3][per] (frame 0, frame 1)
  double* sh = reinterpret_cast<double*>(reinterpret_cast<float*>(smp) + 6 * per);
  const PhotoBlk pb = photo_block();
  const int b = pb.bz, h0 = pb.by * PT_H;
  const long HW = (long)H * W;
  const float* tb = target + (long)b * 3 * HW;
  const float* p0b = pred0 + (long)b * 3 * HW;
  const float* p1b = pred1 + (long)b * 3 * HW;
  const StagePos sp = stage_setup(PF_H, PF_W, h0 - 1, H, W);
  const int r = threadIdx.x / PT_W, c = threadIdx.x - r * PT_W;
  const int h = h0 + r;
  const int ntx = (W + PT_W - 1) / PT_W;
  double acc = 0.0;
  for (int tx = pb.bx * tiles_per_block; tx < ntx && tx < (pb.bx + 1) * tiles_per_block; ++tx) {
    const int w0 = tx * PT_W, w = w0 + c;
    __syncthreads();                                           // the previous tile's window reads are done
    stage_tile2(smt, smp, sp, per, tb, p0b, p1b, HW, w0 - 1, W);
    __syncthreads();
    if (h < H && w < W) {
      const long p = (long)h * W + w;
      const f2 e = tile_error2(smp, smt, PF_W, per, r + 1, c + 1, no_ssim);
      const float e0 = e.x, e1 = e.y;
      if (IDENT) {
        out_err[((long)b * 2) * HW + p] = e0;
        out_err[((long)b * 2 + 1) * HW + p] = e1;
      } else {
        float v[4]; int n = 0;
        const int ni = ident ? (avg ? 1 : 2) : 0;
        if (ident) {
          const float i0 = ident[((long)b * 2) * HW + p], i1 = ident[((long)b * 2 + 1) * HW + p];
          if (avg) { v[n] = (i0 + i1) / 2.f; if (noise) v[n] += noise[(long)b * HW + p] * 0.00001f; ++n; }
          else {
            v[n] = i0; if (noise) v[n] += noise[((long)b * 2) * HW + p] * 0.00001f; ++n;
            v[n] = i1; if (noise) v[n] += noise[((long)b * 2 + 1) * HW + p] * 0.00001f; ++n;
          }
        }
        if (avg) v[n++] = (e0 + e1) / 2.f;
        else { v[n++] = e0; v[n++] = e1; }
        float best = v[0]; int bi = 0;
        for (int j = 1; j < n; ++j) if (v[j] < best) { best = v[j]; bi = j; }
        sel[(long)b * HW + p] = (uint8_t)bi;
        if (isel) isel[(long)b * HW + p] = bi > ni - 1 ? 1.f : 0.f;
        acc += (double)best;
      }
    }
  }
  if (!IDENT) {
    const double t = segsde_block_sum(acc, sh);
    if (threadIdx.x == 0) part[((long)b * gridDim.y + pb.by) * gridDim.x + pb.bx] = t;
  }
}

struct PhotoBwdP {
  const float* pred[2]; const float* target; const uint8_t* sel; const float* disp; const float* inv_K; const float* K;
  const float* T[2]; const float* src[2];
  int hs, ws, H, W, no_ssim, avg, ni;      // ni: number of identity entries in front of the reprojection ones (0, 1, 2)
  int tiles_per_block;                     // a block walks this many horizontally adjacent tiles (one reduction at the end)
  float scale, min_disp, max_disp;
  float* g_disp_up; double* gP_part;       // [B][nblk][2][12]
};

__global__ __launch_bounds__(256) void photometric_bwd_kernel(PhotoBwdP a) {
  SEGSDE_SMEM;
  float* sm = reinterpret_cast<float*>(segsde_smem);      // [9][PB_H][PB_W] pixels, then [2][3][3][PC_H][PC_W] coefficients
  float* cf = sm + 9 * PB_H * PB_W;
  float* gpx = cf + 18 * PC_H * PC_W;                      // [6][PT_H][PT_W]: d err / d warped pixel per (frame, channel)
  float* geo = gpx + 6 * PT_H * PT_W;                      // P0[12] pad P1[12] pad iK[16]
  double* sh = reinterpret_cast<double*>(geo + 48);
  uint8_t* ssel = reinterpret_cast<uint8_t*>(sh + 4);      // [PC_H][PC_W]: selection of tile + 1 (255 outside the image)
  const PhotoBlk pb = photo_block();
  const int b = pb.bz, h0 = pb.by * PT_H;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  {
    const int t = threadIdx.x;
    if (t < 24) {
      const int f = t / 12, e = t - f * 12, r = e >> 2, c = e & 3;
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.K[b * 16 + r * 4 + k] * a.T[f][b * 16 + k * 4 + c];
      geo[f * 16 + e] = s;
    }
    if (t >= 32 && t < 48) geo[t] = a.inv_K[b * 16 + t - 32];
  }
  const int perp = PB_H * PB_W, perc = PC_H * PC_W;
  const float* tb = a.target + (long)b * 3 * HW;
  const float* p0b = a.pred[0] + (long)b * 3 * HW;
  const float* p1b = a.pred[1] + (long)b * 3 * HW;
  const StagePos sp = stage_setup(PB_H, PB_W, h0 - 2, H, W);
  const int r = threadIdx.x / PT_W, c = threadIdx.x - r * PT_W;
  const int h = h0 + r;
  const int ntx = (W + PT_W - 1) / PT_W;
  // a thread adds at most tiles_per_block (<= 16) values per entry in fp32; from the block reduction on the sums are double
  float acc[2][12];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[f][i] = 0.f;
  for (int tx = pb.bx * a.tiles_per_block; tx < ntx && tx < (pb.bx + 1) * a.tiles_per_block; ++tx) {
  const int w0 = tx * PT_W, w = w0 + c;
  __syncthreads();                                             // the previous tile's LDS reads are done (geo is visible)
  stage_tile(sm, sp, perp, tb, p0b, p1b, HW, w0 - 2, W);
  __syncthreads();
  // ---- per-window-centre coefficients for tile + 1: d err_q / d x_cell = a + bx * x_cell + by * y_cell (times upstream).
  // Round 3: one thread per (column of tile + 1, frame, channel) WALKS DOWN the rows with rolling three-row sums -- each new
  // staged row costs 3 + 3 LDS reads for the five row sums and a window is the sum of the last three row sums, instead of
  // 9 + 9 reads per window centre (the 3x3 windows of vertically adjacent centres share two of their three rows).
  // sel of tile + 1 is staged first (one byte per centre).
  for (int e = threadIdx.x; e < perc; e += 256) {
    const int qr = e / PC_W, qc = e - qr * PC_W;
    const int qh = h0 - 1 + qr, qw = w0 - 1 + qc;
    ssel[e] = (qh >= 0 && qh < H && qw >= 0 && qw < W) ? a.sel[(long)b * HW + (long)qh * W + qw] : (uint8_t)255;
  }
  __syncthreads();
  if (threadIdx.x < PC_W * 6) {
    const int qc = threadIdx.x % PC_W, fc = threadIdx.x / PC_W, f = fc / 3, ch = fc - 3 * f;
    float* o = cf + (fc * 3) * perc + qc;
    if (a.no_ssim) {
      for (int qr = 0; qr < PC_H; ++qr) { o[qr * PC_W] = 0.f; o[perc + qr * PC_W] = 0.f; o[2 * perc + qr * PC_W] = 0.f; }
    } else {
      const float* x = sm + (3 + fc) * perp + qc;        // planes 3..8 are pred0 c0..2, pred1 c0..2 = 3 + 3 f + ch
      const float* y = sm + ch * perp + qc;
      float rx[3], ry[3], rxx[3], ryy[3], rxy[3];
#pragma unroll
      for (int r = 0; r < PC_H + 2; ++r) {               // staged rows 0 .. PB_H-1; window of centre qr = rows qr .. qr+2
        const float x0 = x[r * PB_W], x1 = x[r * PB_W + 1], x2 = x[r * PB_W + 2];
        const float y0 = y[r * PB_W], y1 = y[r * PB_W + 1], y2 = y[r * PB_W + 2];
        const int k = r % 3;
        rx[k] = x0 + x1 + x2; ry[k] = y0 + y1 + y2;
        rxx[k] = x0 * x0 + x1 * x1 + x2 * x2; ryy[k] = y0 * y0 + y1 * y1 + y2 * y2; rxy[k] = x0 * y0 + x1 * y1 + x2 * y2;
        if (r < 2) continue;
        const int qr = r - 2;
        const int sl = ssel[qr * PC_W + qc];
        float up = 0.f;
        if (a.avg) { if (sl == a.ni) up = 0.5f * a.scale; }
        else if (sl == a.ni + f) up = a.scale;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (up != 0.f) {
          const float sx = rx[0] + rx[1] + rx[2], sy = ry[0] + ry[1] + ry[2];
          const float sxx = rxx[0] + rxx[1] + rxx[2], syy = ryy[0] + ryy[1] + ryy[2], sxy = rxy[0] + rxy[1] + rxy[2];
          // gradient path (not part of the bit-exact selection): reciprocal multiplies instead of IEEE divisions by 9
          constexpr float R9 = 1.f / 9.f;
          const float gq = up * (0.85f / 3.f) * (-0.5f);
          const float mx = sx * R9, my = sy * R9;
          const float sig_x = sxx * R9 - mx * mx, sig_y = syy * R9 - my * my, sig_xy = sxy * R9 - mx * my;
          const float A1 = 2.f * mx * my + C1, A2 = 2.f * sig_xy + C2;
          const float B1 = mx * mx + my * my + C1, B2 = sig_x + sig_y + C2;
          const float n = A1 * A2, d = B1 * B2, rd = 1.f / d, rr = n * rd;
          const float raw = (1.f - rr) * 0.5f;
          const float m = (raw >= 0.f && raw <= 1.f) ? gq * R9 * rd : 0.f;   // clamp passes gradient on [0,1]
          c0 = m * (2.f * my * A2 - 2.f * A1 * my - rr * (2.f * mx * B2 - 2.f * B1 * mx));
          c1 = m * (-rr * 2.f * B1);
          c2 = m * (2.f * A1);
        }
        o[qr * PC_W] = c0; o[perc + qr * PC_W] = c1; o[2 * perc + qr * PC_W] = c2;
      }
    }
  }
  __syncthreads();
  // ---- d err / d warped pixel: the 3x3 gather of the coefficient windows (reflection fold as multiplicities) + the L1 term,
  // again by column walkers: one thread per (tile column, frame, channel), rolling three-row sums of the column-weighted
  // coefficient rows; the result goes to LDS for the pixel threads of the warp adjoint below
  if (threadIdx.x < PT_W * 6) {
    const int c_ = threadIdx.x % PT_W, fc = threadIdx.x / PT_W, f = fc / 3, ch = fc - 3 * f;
    const int wq = w0 + c_;
    // how often window centre p + d contains a padded cell that maps to p: the mirrored cell -1 (for w == 1) lies in the
    // window of centre 0 only, the mirrored cell W (for w == W-2) in the window of centre W-1 only
    const float mw0 = wq - 1 >= 0 ? (wq == 1 ? 2.f : 1.f) : 0.f, mw2 = wq + 1 <= W - 1 ? (wq == W - 2 ? 2.f : 1.f) : 0.f;
    const float* o = cf + (fc * 3) * perc + c_;          // coefficient column c_ .. c_+2 <-> dw = -1 .. +1
    float ra[3], rbx[3], rby[3];
#pragma unroll
    for (int qr = 0; qr < PC_H; ++qr) {                  // coefficient rows 0 .. PT_H+1 <-> image rows h0-1 ..
      const float* q = o + qr * PC_W;
      const int k = qr % 3;
      ra[k] = mw0 * q[0] + q[1] + mw2 * q[2];
      rbx[k] = mw0 * q[perc] + q[perc + 1] + mw2 * q[perc + 2];
      rby[k] = mw0 * q[2 * perc] + q[2 * perc + 1] + mw2 * q[2 * perc + 2];
      if (qr < 2) continue;
      const int r_ = qr - 2, hq = h0 + r_;               // pixel row; its window rows are coefficient rows r_ .. r_+2
      const float mh0 = hq - 1 >= 0 ? (hq == 1 ? 2.f : 1.f) : 0.f, mh2 = hq + 1 <= H - 1 ? (hq == H - 2 ? 2.f : 1.f) : 0.f;
      const int k0 = r_ % 3, k1 = (r_ + 1) % 3, k2 = (r_ + 2) % 3;
      const float sa = mh0 * ra[k0] + ra[k1] + mh2 * ra[k2];
      const float sbx = mh0 * rbx[k0] + rbx[k1] + mh2 * rbx[k2];
      const float sby = mh0 * rby[k0] + rby[k1] + mh2 * rby[k2];
      const float xv = sm[(3 + fc) * perp + (r_ + 2) * PB_W + c_ + 2], yv = sm[ch * perp + (r_ + 2) * PB_W + c_ + 2];
      const int sl = ssel[(r_ + 1) * PC_W + c_ + 1];
      float upp = 0.f;
      if (a.avg) { if (sl == a.ni) upp = 0.5f * a.scale; }
      else if (sl == a.ni + f) upp = a.scale;
      const float gl1 = upp * (a.no_ssim ? (1.f / 3.f) : (0.15f / 3.f));
      const float df = yv - xv;   // d|t - x|/dx = -sign(t - x)
      gpx[fc * (PT_W * PT_H) + r_ * PT_W + c_] = sa + sbx * xv + sby * yv + gl1 * (df > 0.f ? -1.f : (df < 0.f ? 1.f : 0.f));
    }
  }
  __syncthreads();
  // ---- per pixel: warp adjoint
  if (h < H && w < W) {
    const long p = (long)h * W + w;
    float gdisp = 0.f;
    const float* disp_b = a.disp + (long)b * a.hs * a.ws;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float gp[3];
      bool any = false;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        gp[ch] = gpx[(f * 3 + ch) * (PT_W * PT_H) + r * PT_W + c];
        any = any || (gp[ch] != 0.f);
      }
      if (!any) continue;          // masked out here and in every neighbouring window: no contribution from this frame
      const float* P = geo + f * 16;
      const float* iK = geo + 32;
      const Geo g = geometry(disp_b, a.hs, a.ws, H, W, h, w, iK, P, a.min_disp, a.max_disp);
      const float fx = floorf(g.ix), fy = floorf(g.iy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const bool vx1 = x1 <= W - 1, vy1 = y1 <= H - 1;
      const float* src_b = a.src[f] + (long)b * 3 * HW;
      float gix = 0.f, giy = 0.f;   // ATen grid_sampler_2d_backward
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* sp = src_b + ch * HW;
        const float go = gp[ch];
        const float vnw = sp[(long)y0 * W + x0];
        const float vne = vx1 ? sp[(long)y0 * W + x1] : 0.f;
        const float vsw = vy1 ? sp[(long)y1 * W + x0] : 0.f;
        const float vse = (vx1 && vy1) ? sp[(long)y1 * W + x1] : 0.f;
        gix -= vnw * ((float)y1 - g.iy) * go; giy -= vnw * ((float)x1 - g.ix) * go;
        gix += vne * ((float)y1 - g.iy) * go; giy -= vne * (g.ix - (float)x0) * go;
        gix -= vsw * (g.iy - (float)y0) * go; giy += vsw * ((float)x1 - g.ix) * go;
        gix += vse * (g.iy - (float)y0) * go; giy += vse * (g.ix - (float)x0) * go;
      }
      const float g_x = g.in_x ? gix : 0.f, g_y = g.in_y ? giy : 0.f;
      const float den = g.p2 + 1e-7f;
      const float gp0 = g_x / den, gp1 = g_y / den, gp2 = -(g_x * g.x + g_y * g.y) / den;
      const float cx = g.depth * g.xn, cy = g.depth * g.yn, cz = g.depth * g.zn;
      acc[f][0] += gp0 * cx; acc[f][1] += gp0 * cy; acc[f][2] += gp0 * cz; acc[f][3] += gp0;
      acc[f][4] += gp1 * cx; acc[f][5] += gp1 * cy; acc[f][6] += gp1 * cz; acc[f][7] += gp1;
      acc[f][8] += gp2 * cx; acc[f][9] += gp2 * cy; acc[f][10] += gp2 * cz; acc[f][11] += gp2;
      const float gcx = P[0] * gp0 + P[4] * gp1 + P[8] * gp2;
      const float gcy = P[1] * gp0 + P[5] * gp1 + P[9] * gp2;
      const float gcz = P[2] * gp0 + P[6] * gp1 + P[10] * gp2;
      const float gdepth = gcx * g.xn + gcy * g.yn + gcz * g.zn;
      gdisp += (-gdepth * g.depth * g.depth) * (a.max_disp - a.min_disp);
    }
    a.g_disp_up[(long)b * HW + p] = gdisp;
  }
  }   // tiles of the strip
  const long blk = ((long)b * gridDim.y + pb.by) * gridDim.x + pb.bx;
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const double t = segsde_block_sum((double)acc[f][i], sh);
      if (threadIdx.x == 0) a.gP_part[(blk * 2 + f) * 12 + i] = t;
    }
}

// Packed backward: same phases as photometric_bwd_kernel with (frame 0, frame 1) in register pairs.
//  * coefficient walkers: one thread per (column of tile + 1, channel) -- 102 instead of 204 -- the target's row sums once;
//  * gather walkers: one thread per (tile column, channel), 96 instead of 192;
//    the two walker phases of a tile run on opposite halves of the block and swap halves with the tile's parity, so that the
//    four SIMDs of a CU see the same load over a strip;
//  * warp adjoint: the geometry of a pixel is evaluated by the same scalar code as in the forward pass (identical sampling
//    cells), everything behind it -- tap gradients, d/dP sums, d/d depth -- is packed; the taps of both frames are requested
//    before either is used;
//  * sel of tile + 1 is staged together with the pixels (one barrier less), the reciprocals of the gradient path are v_rcp_f32.
__device__ __forceinline__ f2 rcp2(f2 v) { f2 r; r.x = __builtin_amdgcn_rcpf(v.x); r.y = __builtin_amdgcn_rcpf(v.y); return r; }

// SPLIT: the walkers of a column are two threads, one per half of the rows (two warm-up rows each): all four waves walk, the
// serial chain of a phase is 7 / 6 row steps instead of 12 / 10
template <bool SPLIT>
__global__ __launch_bounds__(256) void photometric_bwd2_kernel(PhotoBwdP a) {
  SEGSDE_SMEM;
  const int perp = PB_H * PB_W, perc = PC_H * PC_W, npt = PT_H * PT_W;
  float* smt = reinterpret_cast<float*>(segsde_smem);     // [3][perp] target
  f2* smp = reinterpret_cast<f2*>(smt + 3 * perp);        // [3][perp] (frame 0, frame 1); 3 * perp is even
  f2* cf = smp + 3 * perp;                                // [3 channels][3 coefficients][perc]
  f2* gpx = cf + 9 * perc;                                // [3][npt]: d err / d warped pixel
  float* geo = reinterpret_cast<float*>(gpx + 3 * npt);   // P0[12] pad P1[12] pad iK[16], then the P's interleaved: f2[12]
  f2* P2 = reinterpret_cast<f2*>(geo + 48);
  double* sh = reinterpret_cast<double*>(geo + 72);
  uint8_t* ssel = reinterpret_cast<uint8_t*>(sh + 4);     // [PC_H][PC_W]: selection of tile + 1 (255 outside the image)
  const PhotoBlk pb = photo_block();
  const int b = pb.bz, h0 = pb.by * PT_H;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  {
    const int t = threadIdx.x;
    if (t < 24) {
      const int f = t / 12, e = t - f * 12, r = e >> 2, c = e & 3;
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.K[b * 16 + r * 4 + k] * a.T[f][b * 16 + k * 4 + c];
      geo[f * 16 + e] = s;
      geo[48 + e * 2 + f] = s;
    }
    if (t >= 32 && t < 48) geo[t] = a.inv_K[b * 16 + t - 32];
  }
  const float* tb = a.target + (long)b * 3 * HW;
  const float* p0b = a.pred[0] + (long)b * 3 * HW;
  const float* p1b = a.pred[1] + (long)b * 3 * HW;
  const float* s0b = a.src[0] + (long)b * 3 * HW;
  const float* s1b = a.src[1] + (long)b * 3 * HW;
  const float* disp_b = a.disp + (long)b * a.hs * a.ws;
  const StagePos sp = stage_setup(PB_H, PB_W, h0 - 2, H, W);
  const int r = threadIdx.x / PT_W, c = threadIdx.x - r * PT_W;
  const int h = h0 + r;
  const int ntx = (W + PT_W - 1) / PT_W;
  const float up_one = a.avg ? 0.5f * a.scale : a.scale;
  // a thread adds at most tiles_per_block (<= 16) values per entry in fp32; from the block reduction on the sums are double
  f2 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = splat2(0.f);
  for (int tx = pb.bx * a.tiles_per_block; tx < ntx && tx < (pb.bx + 1) * a.tiles_per_block; ++tx) {
  const int w0 = tx * PT_W, w = w0 + c;
  __syncthreads();                                             // the previous tile's LDS reads are done (geo is visible)
  stage_tile2(smt, smp, sp, perp, tb, p0b, p1b, HW, w0 - 2, W);
  for (int e = threadIdx.x; e < perc; e += 256) {
    const int qr = e / PC_W, qc = e - qr * PC_W;
    const int qh = h0 - 1 + qr, qw = w0 - 1 + qc;
    ssel[e] = (qh >= 0 && qh < H && qw >= 0 && qw < W) ? a.sel[(long)b * HW + (long)qh * W + qw] : (uint8_t)255;
  }
  __syncthreads();
  const int half = (tx & 1) << 7;
  // ---- per-window-centre coefficients for tile + 1 (see photometric_bwd_kernel): column walkers with rolling three-row sums
  const int tb_ = SPLIT ? (int)threadIdx.x : (int)((threadIdx.x + half) & 255);
  if (tb_ < PC_W * 3 * (SPLIT ? 2 : 1)) {
    const int grp = tb_ / (PC_W * 3), tb1 = tb_ - grp * (PC_W * 3);
    const int ch = tb1 / PC_W, qc = tb1 - ch * PC_W;
    constexpr int NB = SPLIT ? PC_H / 2 + 2 : PC_H + 2;   // staged rows a walker visits
    const int rb = grp * (PC_H / 2);                      // its first staged row = its first centre
    f2* o = cf + (ch * 3) * perc + qc;
    if (a.no_ssim) {
      for (int qr = rb; qr < rb + NB - 2; ++qr) { o[qr * PC_W] = splat2(0.f); o[perc + qr * PC_W] = splat2(0.f); o[2 * perc + qr * PC_W] = splat2(0.f); }
    } else {
      const f2* x = smp + ch * perp + rb * PB_W + qc;
      const float* y = smt + ch * perp + rb * PB_W + qc;
      f2 rx[3], rxx[3], rxy[3];
      float ry[3], ryy[3];
#pragma unroll
      for (int rr_ = 0; rr_ < NB; ++rr_) {                 // staged rows rb .. ; window of centre qr = staged rows qr .. qr+2
        const f2 x0 = x[rr_ * PB_W], x1 = x[rr_ * PB_W + 1], x2 = x[rr_ * PB_W + 2];
        const float y0 = y[rr_ * PB_W], y1 = y[rr_ * PB_W + 1], y2 = y[rr_ * PB_W + 2];
        const int k = rr_ % 3;
        rx[k] = x0 + x1 + x2; ry[k] = y0 + y1 + y2;
        rxx[k] = x0 * x0 + x1 * x1 + x2 * x2; ryy[k] = y0 * y0 + y1 * y1 + y2 * y2; rxy[k] = x0 * y0 + x1 * y1 + x2 * y2;
        if (rr_ < 2) continue;
        const int qr = rb + rr_ - 2;
        const int sl = ssel[qr * PC_W + qc];
        f2 up = splat2(0.f);
        if (a.avg) { if (sl == a.ni) up = splat2(up_one); }
        else { if (sl == a.ni) up.x = up_one; if (sl == a.ni + 1) up.y = up_one; }
        f2 c0 = splat2(0.f), c1 = splat2(0.f), c2 = splat2(0.f);
        if (up.x != 0.f || up.y != 0.f) {
          const f2 sx = rx[0] + rx[1] + rx[2], sxx = rxx[0] + rxx[1] + rxx[2], sxy = rxy[0] + rxy[1] + rxy[2];
          const float sy = ry[0] + ry[1] + ry[2], syy = ryy[0] + ryy[1] + ryy[2];
          constexpr float R9 = 1.f / 9.f;
          const f2 gq = up * (0.85f / 3.f) * (-0.5f);
          const f2 mx = sx * R9;
          const float my = sy * R9;
          const f2 sig_x = sxx * R9 - mx * mx, sig_xy = sxy * R9 - mx * my;
          const float sig_y = syy * R9 - my * my;
          const f2 A1 = 2.f * mx * my + C1, A2 = 2.f * sig_xy + C2;
          const f2 B1 = mx * mx + my * my + C1, B2 = sig_x + sig_y + C2;
          const f2 n = A1 * A2, d = B1 * B2, rd = rcp2(d), q = n * rd;
          const f2 raw = (1.f - q) * 0.5f;
          f2 m = gq * R9 * rd;                                   // clamp passes gradient on [0,1]
          if (!(raw.x >= 0.f && raw.x <= 1.f)) m.x = 0.f;
          if (!(raw.y >= 0.f && raw.y <= 1.f)) m.y = 0.f;
          c0 = m * (2.f * my * A2 - 2.f * A1 * my - q * (2.f * mx * B2 - 2.f * B1 * mx));
          c1 = m * (-q * 2.f * B1);
          c2 = m * (2.f * A1);
        }
        o[qr * PC_W] = c0; o[perc + qr * PC_W] = c1; o[2 * perc + qr * PC_W] = c2;
      }
    }
  }
  __syncthreads();
  // ---- d err / d warped pixel: 3x3 gather of the coefficient windows (reflection fold as multiplicities) + the L1 term
  const int tc_ = SPLIT ? (int)threadIdx.x : (int)((threadIdx.x + 128 + half) & 255);
  if (tc_ < PT_W * 3 * (SPLIT ? 2 : 1)) {
    const int grp = tc_ / (PT_W * 3), tc1 = tc_ - grp * (PT_W * 3);
    const int ch = tc1 / PT_W, c_ = tc1 - ch * PT_W;
    constexpr int NC = SPLIT ? PT_H / 2 + 2 : PT_H + 2;   // coefficient rows a walker visits
    const int rb = grp * (PT_H / 2);                      // its first coefficient row = its first pixel row
    const int wq = w0 + c_;
    const float mw0 = wq - 1 >= 0 ? (wq == 1 ? 2.f : 1.f) : 0.f, mw2 = wq + 1 <= W - 1 ? (wq == W - 2 ? 2.f : 1.f) : 0.f;
    const f2* o = cf + (ch * 3) * perc + rb * PC_W + c_;   // coefficient column c_ .. c_+2 <-> dw = -1 .. +1
    f2 ra[3], rbx[3], rby[3];
#pragma unroll
    for (int qr = 0; qr < NC; ++qr) {                    // coefficient rows rb .. <-> image rows h0-1+rb ..
      const f2* q = o + qr * PC_W;
      const int k = qr % 3;
      ra[k] = mw0 * q[0] + q[1] + mw2 * q[2];
      rbx[k] = mw0 * q[perc] + q[perc + 1] + mw2 * q[perc + 2];
      rby[k] = mw0 * q[2 * perc] + q[2 * perc + 1] + mw2 * q[2 * perc + 2];
      if (qr < 2) continue;
      const int r_ = rb + qr - 2, hq = h0 + r_;          // pixel row; its window rows are coefficient rows r_ .. r_+2
      const float mh0 = hq - 1 >= 0 ? (hq == 1 ? 2.f : 1.f) : 0.f, mh2 = hq + 1 <= H - 1 ? (hq == H - 2 ? 2.f : 1.f) : 0.f;
      const int k0 = (qr - 2) % 3, k1 = (qr - 1) % 3, k2 = qr % 3;
      const f2 sa = mh0 * ra[k0] + ra[k1] + mh2 * ra[k2];
      const f2 sbx = mh0 * rbx[k0] + rbx[k1] + mh2 * rbx[k2];
      const f2 sby = mh0 * rby[k0] + rby[k1] + mh2 * rby[k2];
      const f2 xv = smp[ch * perp + (r_ + 2) * PB_W + c_ + 2];
      const float yv = smt[ch * perp + (r_ + 2) * PB_W + c_ + 2];
      const int sl = ssel[(r_ + 1) * PC_W + c_ + 1];
      f2 upp = splat2(0.f);
      if (a.avg) { if (sl == a.ni) upp = splat2(up_one); }
      else { if (sl == a.ni) upp.x = up_one; if (sl == a.ni + 1) upp.y = up_one; }
      const f2 gl1 = upp * (a.no_ssim ? (1.f / 3.f) : (0.15f / 3.f));
      const f2 df = yv - xv;      // d|t - x|/dx = -sign(t - x)
      f2 sg;
      sg.x = df.x > 0.f ? -1.f : (df.x < 0.f ? 1.f : 0.f);
      sg.y = df.y > 0.f ? -1.f : (df.y < 0.f ? 1.f : 0.f);
      gpx[ch * npt + r_ * PT_W + c_] = sa + sbx * xv + sby * yv + gl1 * sg;
    }
  }
  __syncthreads();
  // ---- per pixel: warp adjoint of both frames
  {
    const bool live = h < H && w < W;
    const long p = (long)h * W + w;
    f2 gp[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) gp[ch] = gpx[ch * npt + r * PT_W + c];
    const bool any0 = live && (gp[0].x != 0.f || gp[1].x != 0.f || gp[2].x != 0.f);
    const bool any1 = live && (gp[0].y != 0.f || gp[1].y != 0.f || gp[2].y != 0.f);
    // a frame nobody in the wave needs (auto-mask regions select one frame, or the identity, over large areas) is not
    // projected at all; the votes sit in converged code
    const bool wave0 = __any(any0) != 0, wave1 = __any(any1) != 0;
    float gdisp = 0.f;
    if (any0 || any1) {     // otherwise masked out here and in every neighbouring window, for both frames
      const float* iK = geo + 32;
      Geo g0, g1;
      if (wave0) g0 = geometry(disp_b, a.hs, a.ws, H, W, h, w, iK, geo, a.min_disp, a.max_disp);
      if (wave1) g1 = geometry(disp_b, a.hs, a.ws, H, W, h, w, iK, geo + 16, a.min_disp, a.max_disp);
      if (!wave0) g0 = g1;
      if (!wave1) g1 = g0;
      const int x00 = (int)floorf(g0.ix), y00 = (int)floorf(g0.iy), x01 = (int)floorf(g1.ix), y01 = (int)floorf(g1.iy);
      const bool vx0 = x00 + 1 <= W - 1, vy0 = y00 + 1 <= H - 1, vx1 = x01 + 1 <= W - 1, vy1 = y01 + 1 <= H - 1;
      // ATen grid_sampler_2d_backward; taps outside the image are never dereferenced (their weight is zero: border clamp)
      const long o0 = (long)y00 * W + x00, o1 = (long)y01 * W + x01;
      const long e0 = vx0 ? 1 : 0, s0 = vy0 ? W : 0, e1 = vx1 ? 1 : 0, s1 = vy1 ? W : 0;
      f2 vnw[3], vne[3], vsw[3], vse[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* q0 = s0b + ch * HW + o0;
        const float* q1 = s1b + ch * HW + o1;
        vnw[ch].x = q0[0]; vne[ch].x = q0[e0]; vsw[ch].x = q0[s0]; vse[ch].x = q0[s0 + e0];
        vnw[ch].y = q1[0]; vne[ch].y = q1[e1]; vsw[ch].y = q1[s1]; vse[ch].y = q1[s1 + e1];
      }
      f2 ix, iy, fx1, fy1, fx0, fy0, mE, mS;
      ix.x = g0.ix; ix.y = g1.ix; iy.x = g0.iy; iy.y = g1.iy;
      fx0.x = (float)x00; fx0.y = (float)x01; fy0.x = (float)y00; fy0.y = (float)y01;
      fx1 = fx0 + 1.f; fy1 = fy0 + 1.f;
      mE.x = vx0 ? 1.f : 0.f; mE.y = vx1 ? 1.f : 0.f; mS.x = vy0 ? 1.f : 0.f; mS.y = vy1 ? 1.f : 0.f;
      const f2 wy0 = fy1 - iy, wy1 = iy - fy0, wx0 = fx1 - ix, wx1 = ix - fx0;
      f2 gix = splat2(0.f), giy = splat2(0.f);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const f2 go = gp[ch];
        const f2 ne = vne[ch] * mE, sw = vsw[ch] * mS, se = vse[ch] * (mE * mS);
        gix -= vnw[ch] * wy0 * go; giy -= vnw[ch] * wx0 * go;
        gix += ne * wy0 * go;      giy -= ne * wx1 * go;
        gix -= sw * wy1 * go;      giy += sw * wx0 * go;
        gix += se * wy1 * go;      giy += se * wx1 * go;
      }
      f2 g_x, g_y, den, px_, py_;
      g_x.x = (g0.in_x && any0) ? gix.x : 0.f; g_x.y = (g1.in_x && any1) ? gix.y : 0.f;
      g_y.x = (g0.in_y && any0) ? giy.x : 0.f; g_y.y = (g1.in_y && any1) ? giy.y : 0.f;
      den.x = g0.p2 + 1e-7f; den.y = g1.p2 + 1e-7f;
      px_.x = g0.x; px_.y = g1.x; py_.x = g0.y; py_.y = g1.y;
      const f2 rden = rcp2(den);
      f2 gp0 = g_x * rden, gp1 = g_y * rden, gp2 = -(g_x * px_ + g_y * py_) * rden;
      if (!any0) { gp0.x = 0.f; gp1.x = 0.f; gp2.x = 0.f; }
      if (!any1) { gp0.y = 0.f; gp1.y = 0.f; gp2.y = 0.f; }
      const float cx = g0.depth * g0.xn, cy = g0.depth * g0.yn, cz = g0.depth * g0.zn;
      acc[0] += gp0 * cx; acc[1] += gp0 * cy; acc[2] += gp0 * cz; acc[3] += gp0;
      acc[4] += gp1 * cx; acc[5] += gp1 * cy; acc[6] += gp1 * cz; acc[7] += gp1;
      acc[8] += gp2 * cx; acc[9] += gp2 * cy; acc[10] += gp2 * cz; acc[11] += gp2;
      const f2 gcx = P2[0] * gp0 + P2[4] * gp1 + P2[8] * gp2;
      const f2 gcy = P2[1] * gp0 + P2[5] * gp1 + P2[9] * gp2;
      const f2 gcz = P2[2] * gp0 + P2[6] * gp1 + P2[10] * gp2;
      const f2 gdepth = gcx * g0.xn + gcy * g0.yn + gcz * g0.zn;
      const f2 gd = (-gdepth * g0.depth * g0.depth) * (a.max_disp - a.min_disp);
      gdisp = gd.x + gd.y;
    }
    if (live) a.g_disp_up[(long)b * HW + p] = gdisp;
  }
  }   // tiles of the strip
  const long blk = ((long)b * gridDim.y + pb.by) * gridDim.x + pb.bx;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const double t0 = segsde_block_sum((double)acc[i].x, sh);
    const double t1 = segsde_block_sum((double)acc[i].y, sh);
    if (threadIdx.x == 0) { a.gP_part[(blk * 2 + 0) * 12 + i] = t0; a.gP_part[(blk * 2 + 1) * 12 + i] = t1; }
  }
}

// gT_f[b] += weight * K[b]^T (rows 0..2) * sum_blocks gP_f[b]      (both frames: grid (B, 2))
__global__ __launch_bounds__(256) void photometric_bwd_finalize_kernel(const double* gP_part, int nblk, const float* K,
                                                                       const float* weight, float* gT0, float* gT1) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);   // [12][16] lane sums, then [12] totals at sh + 192
  double* gp = sh + 192;
  const int b = blockIdx.x, f = blockIdx.y, t = threadIdx.x;
  if (t < 192) {
    const int e = t >> 4, l = t & 15;
    double s = 0.0;
    for (int i = l; i < nblk; i += 16) s += gP_part[(((long)b * nblk + i) * 2 + f) * 12 + e];
    sh[e * 16 + l] = s;
  }
  __syncthreads();
  if (t < 12) {
    double s = 0.0;
    for (int l = 0; l < 16; ++l) s += sh[t * 16 + l];
    gp[t] = s;
  }
  __syncthreads();
  if (t >= 16) return;
  const int k = t >> 2, j = t & 3;
  double s = 0.0;
  for (int r = 0; r < 3; ++r) s += (double)K[b * 16 + r * 4 + k] * gp[r * 4 + j];
  float* gT = f ? gT1 : gT0;
  gT[b * 16 + t] += (weight ? weight[0] : 1.f) * (float)s;
}

inline int plane_blocks(long HW) { long nb = (HW + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb)); }
inline int flat_blocks(long n) { long nb = (n + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb)); }

}  // namespace

extern "C" int segsde_warp_forward(const float* disp, int hs, int ws, const float* inv_K, const float* K, const float* T,
                                   const float* src, int B, int H, int W, float min_depth, float max_depth, float* color,
                                   float* grid, float* depth, void* stream) {
  if (!disp || !inv_K || !K || !T || !src || !color) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || hs <= 0 || ws <= 0 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  // up to 512 blocks per image, grid-stride: one pixel per thread (2048 blocks at 512x1024) measured 131 against 117 us in the
  // step's kernel trace (SEGSDE_WARP_BLOCKS=2048, profiles/experiments_r04.md)
  static const int cap = [] { const char* e = getenv("SEGSDE_WARP_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  const long nbw = ((long)H * W + 255) / 256;
  hipLaunchKernelGGL(warp_fwd_kernel, dim3((unsigned)(nbw < 1 ? 1 : (nbw > cap ? cap : nbw)), B), dim3(256), 512, ST(stream), disp, hs, ws, inv_K, K,
                     T, src, H, W, 1.f / max_depth, 1.f / min_depth, color, grid, depth);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_disp_to_depth(const float* disp, int hs, int ws, int B, int H, int W, float min_depth, float max_depth,
                                    float* depth, void* stream) {
  if (!disp || !depth) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 1 || W < 1 || hs <= 0 || ws <= 0 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(disp_to_depth_kernel, dim3(plane_blocks((long)H * W), B), dim3(256), 0, ST(stream), disp, hs, ws, H, W,
                     1.f / max_depth, 1.f / min_depth, depth);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t segsde_warp_backward_workspace(int B, int H, int W) {
  return (size_t)B * plane_blocks((long)H * W) * 12 * sizeof(double);
}

extern "C" int segsde_warp_backward(const float* gcolor, const float* disp, int hs, int ws, const float* inv_K,
                                    const float* K, const float* T, const float* src, int B, int H, int W, float min_depth,
                                    float max_depth, float* g_disp_up, float* gT, void* ws_, size_t ws_bytes, void* stream) {
  if (!gcolor || !disp || !inv_K || !K || !T || !src || !g_disp_up || !gT || !ws_) return SEGSDE_ERR_NULL;
  if (ws_bytes < segsde_warp_backward_workspace(B, H, W)) return SEGSDE_ERR_WORKSPACE;
  const int nblk = plane_blocks((long)H * W);
  hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblk, B), dim3(256), 512, ST(stream), gcolor, disp, hs, ws, inv_K, K, T, src, H,
                     W, 1.f / max_depth, 1.f / min_depth, g_disp_up, (double*)ws_);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(warp_bwd_finalize_kernel, dim3(B), dim3(256), 204 * sizeof(double), ST(stream), (const double*)ws_, nblk, K, B, gT);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_reprojection_error_forward(const float* pred, const float* target, int B, int H, int W, int no_ssim,
                                                 float* err, long err_bstride, void* stream) {
  if (!pred || !target || !err) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(reproj_err_fwd_kernel, dim3(plane_blocks((long)H * W), B), dim3(256), 0, ST(stream), pred, target, H,
                     W, no_ssim, err, err_bstride);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t segsde_reprojection_error_backward_workspace(int B, int H, int W) {
  return (size_t)B * 9 * H * W * sizeof(float);
}

extern "C" int segsde_reprojection_error_backward(const float* pred, const float* target, const float* gerr,
                                                  long gerr_bstride, int B, int H, int W, int no_ssim, float* gpred,
                                                  void* ws_, size_t ws_bytes, void* stream) {
  if (!pred || !target || !gerr || !gpred) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  const dim3 grid(plane_blocks((long)H * W), B);
  if (!no_ssim) {
    if (!ws_) return SEGSDE_ERR_NULL;
    if (ws_bytes < segsde_reprojection_error_backward_workspace(B, H, W)) return SEGSDE_ERR_WORKSPACE;
    hipLaunchKernelGGL(ssim_coef_kernel, grid, dim3(256), 0, ST(stream), pred, target, gerr, gerr_bstride, H, W, (float*)ws_);
    SEGSDE_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(reproj_err_bwd_kernel, grid, dim3(256), 0, ST(stream), pred, target, gerr, gerr_bstride,
                     (const float*)ws_, H, W, no_ssim, gpred);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t segsde_automask_workspace(int B, int H, int W) {
  return (size_t)flat_blocks((long)B * H * W) * sizeof(double);
}

extern "C" int segsde_automask_min_forward(const float* ident, const float* noise, const float* reproj, int n_reproj,
                                           int avg, int B, int H, int W, uint8_t* sel, float* identity_selection,
                                           float* sum_out, void* ws_, size_t ws_bytes, void* stream) {
  if (!reproj || !sel || !sum_out || !ws_) return SEGSDE_ERR_NULL;
  if (n_reproj < 1 || n_reproj > SEGSDE_AUTOMASK_MAX_FRAMES) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_automask_workspace(B, H, W)) return SEGSDE_ERR_WORKSPACE;
  const long total = (long)B * H * W;
  const int nb = flat_blocks(total);
  hipLaunchKernelGGL(automask_fwd_kernel, dim3(nb), dim3(256), 64, ST(stream), ident, noise, reproj, n_reproj, avg, total,
                     (long)H * W, sel, identity_selection, (double*)ws_);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 64, ST(stream), (const double*)ws_, nb, sum_out, 0, 1.0);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_automask_min_backward(const uint8_t* sel, int n_ident, int n_reproj, int avg, int B, int H, int W,
                                            float scale, float* greproj, void* stream) {
  if (!sel || !greproj) return SEGSDE_ERR_NULL;
  if (n_reproj < 1 || n_reproj > SEGSDE_AUTOMASK_MAX_FRAMES) return SEGSDE_ERR_SHAPE;
  const long total = (long)B * H * W;
  const int ni = n_ident ? (avg ? 1 : n_reproj) : 0;
  hipLaunchKernelGGL(automask_bwd_kernel, dim3(flat_blocks(total)), dim3(256), 0, ST(stream), sel, ni, n_reproj, avg, total,
                     (long)H * W, scale, greproj);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t segsde_smoothness_workspace(int B, int h, int w) {
  return (size_t)B * plane_blocks((long)h * w) * 2 * sizeof(double) + (size_t)B * h * w * sizeof(float) + 64;
}

extern "C" int segsde_smoothness_forward(const float* disp, const float* img, int B, int h, int w, float* mean_disp,
                                         float* out, void* ws_, size_t ws_bytes, void* stream) {
  if (!disp || !img || !mean_disp || !out || !ws_) return SEGSDE_ERR_NULL;
  if (B <= 0 || h < 2 || w < 2) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_smoothness_workspace(B, h, w)) return SEGSDE_ERR_WORKSPACE;
  const long HW = (long)h * w;
  const int nb = plane_blocks(HW);
  double* part = (double*)ws_;
  hipLaunchKernelGGL(plane_sum_kernel, dim3(nb, B), dim3(256), 64, ST(stream), disp, HW, part);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(plane_mean_finalize_kernel, dim3(B), dim3(64), 0, ST(stream), (const double*)part, nb, 1.0 / (double)HW,
                     mean_disp);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nb, B), dim3(256), 64, ST(stream), disp, img, (const float*)mean_disp, h, w, part);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_finalize_kernel, dim3(1), dim3(256), 512 * sizeof(double), ST(stream), (const double*)part, B * nb,
                     1.0 / ((double)B * h * (w - 1)), 1.0 / ((double)B * (h - 1) * w), out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_smoothness_backward(const float* disp, const float* img, const float* mean_disp, int B, int h, int w,
                                          float scale, float* gdisp, void* ws_, size_t ws_bytes, void* stream) {
  if (!disp || !img || !mean_disp || !gdisp || !ws_) return SEGSDE_ERR_NULL;
  if (ws_bytes < segsde_smoothness_workspace(B, h, w)) return SEGSDE_ERR_WORKSPACE;
  const long HW = (long)h * w;
  const int nb = plane_blocks(HW);
  double* part = (double*)ws_;
  float* tmp = (float*)((char*)ws_ + (((size_t)B * nb * 2 * sizeof(double) + 63) / 64) * 64);
  const float sx = scale / ((float)B * h * (w - 1)), sy = scale / ((float)B * (h - 1) * w);
  hipLaunchKernelGGL(smooth_bwd_a_kernel, dim3(nb, B), dim3(256), 64, ST(stream), disp, img, mean_disp, h, w, sx, sy, tmp, part);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_bwd_b_kernel, dim3(nb, B), dim3(256), 0, ST(stream), (const float*)tmp, mean_disp,
                     (const double*)part, nb, HW, gdisp);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

// get_smooth_loss(disp, img) by itself (models/monodepth_layers.py:208-221), i.e. WITHOUT the mean normalisation
// MonodepthLoss puts in front of it: the same kernels with a "mean" m for which (m + 1e-7f) is exactly 1.0f.
constexpr float SMOOTH_UNIT_MEAN = 0.99999988f;
static_assert(SMOOTH_UNIT_MEAN + 1e-7f == 1.0f, "the un-normalised smoothness relies on (m + 1e-7f) == 1.0f");
__global__ __launch_bounds__(64) void fill_kernel(float* p, int n, float v) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = v;
}
__global__ __launch_bounds__(256) void smooth_bwd_plain_kernel(const float* tmp, long HW, float* gdisp) {
  const int b = blockIdx.y;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) gdisp[b * HW + p] += tmp[b * HW + p];
}

extern "C" size_t segsde_smooth_loss_workspace(int B, int h, int w) { return segsde_smoothness_workspace(B, h, w) + 256; }

extern "C" int segsde_smooth_loss_forward(const float* disp, const float* img, int B, int h, int w, float* out, void* ws_,
                                          size_t ws_bytes, void* stream) {
  if (!disp || !img || !out || !ws_) return SEGSDE_ERR_NULL;
  if (B <= 0 || B > 64 || h < 2 || w < 2) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_smooth_loss_workspace(B, h, w)) return SEGSDE_ERR_WORKSPACE;
  const long HW = (long)h * w;
  const int nb = plane_blocks(HW);
  float* unit = (float*)ws_;
  double* part = (double*)((char*)ws_ + 256);
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, ST(stream), unit, B, SMOOTH_UNIT_MEAN);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nb, B), dim3(256), 64, ST(stream), disp, img, (const float*)unit, h, w, part);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_finalize_kernel, dim3(1), dim3(256), 512 * sizeof(double), ST(stream), (const double*)part, B * nb,
                     1.0 / ((double)B * h * (w - 1)), 1.0 / ((double)B * (h - 1) * w), out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_smooth_loss_backward(const float* disp, const float* img, int B, int h, int w, float scale, float* gdisp,
                                           void* ws_, size_t ws_bytes, void* stream) {
  if (!disp || !img || !gdisp || !ws_) return SEGSDE_ERR_NULL;
  if (B <= 0 || B > 64 || h < 2 || w < 2) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_smooth_loss_workspace(B, h, w)) return SEGSDE_ERR_WORKSPACE;
  const long HW = (long)h * w;
  const int nb = plane_blocks(HW);
  float* unit = (float*)ws_;
  double* part = (double*)((char*)ws_ + 256);
  float* tmp = (float*)((char*)part + (((size_t)B * nb * 2 * sizeof(double) + 63) / 64) * 64);
  const float sx = scale / ((float)B * h * (w - 1)), sy = scale / ((float)B * (h - 1) * w);
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, ST(stream), unit, B, SMOOTH_UNIT_MEAN);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_bwd_a_kernel, dim3(nb, B), dim3(256), 64, ST(stream), disp, img, (const float*)unit, h, w, sx, sy, tmp,
                     part);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(smooth_bwd_plain_kernel, dim3(nb, B), dim3(256), 0, ST(stream), (const float*)tmp, HW, gdisp);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------ fused photometric entry points
// a block walks `tiles` horizontally adjacent tiles so that its (double precision, 24-value) reduction is paid once per
// strip; strips shrink while the launch would otherwise fall below ~8 blocks per CU
static inline int photo_tiles_per_block(int B, int H, int W) {
  const int ntx = (W + PT_W - 1) / PT_W, nty = (H + PT_H - 1) / PT_H;
  if (const char* e = getenv("SEGSDE_PHOTO_TILES")) {            // experiment / test knob: fixed strip length
    const int t = atoi(e);
    if (t >= 1) return t < ntx ? t : ntx;
  }
  int t = ntx < 16 ? ntx : 16;
  while (t > 1 && (long)B * nty * ((ntx + t - 1) / t) < 2048) t >>= 1;
  return t < 1 ? 1 : t;
}
static inline dim3 photo_grid(int B, int H, int W) {
  const int t = photo_tiles_per_block(B, H, W), ntx = (W + PT_W - 1) / PT_W;
  return dim3((ntx + t - 1) / t, (H + PT_H - 1) / PT_H, B);
}
static inline long photo_blocks(int B, int H, int W) { const dim3 g = photo_grid(B, H, W); return (long)g.x * g.y * g.z; }
// experiment knob: SEGSDE_PHOTO_PACKED=0 keeps the one-frame-per-lane kernels of round 3 (A/B in profiles/experiments_r04.md)
static inline bool photo_packed() {
  static const bool v = [] { const char* e = getenv("SEGSDE_PHOTO_PACKED"); return !(e && e[0] == '0'); }();
  return v;
}
static inline bool photo_split() {   // SEGSDE_PHOTO_SPLIT=0: the packed backward with whole-column walkers on half of the block
  static const bool v = [] { const char* e = getenv("SEGSDE_PHOTO_SPLIT"); return !(e && e[0] == '0'); }();
  return v;
}

extern "C" size_t segsde_photometric_workspace(int B, int H, int W) {
  return (size_t)photo_blocks(B, H, W) * 24 * sizeof(double);
}

extern "C" int segsde_photometric_identity(const float* src0, const float* src1, const float* target, int B, int H, int W,
                                           int no_ssim, float* ident, void* stream) {
  if (!src0 || !src1 || !target || !ident) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(photo_packed() ? photometric_fwd2_kernel<true> : photometric_fwd_kernel<true>, photo_grid(B, H, W), dim3(256),
                     (9 * PF_H * PF_W + 1) * sizeof(float) + 64, ST(stream), src0, src1, target, (const float*)nullptr,
                     (const float*)nullptr, H, W, no_ssim, 0, photo_tiles_per_block(B, H, W), ident, (uint8_t*)nullptr,
                     (float*)nullptr, (double*)nullptr);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_photometric_forward(const float* pred0, const float* pred1, const float* target, const float* ident,
                                          const float* noise, int B, int H, int W, int no_ssim, int avg, uint8_t* sel,
                                          float* identity_selection, float* sum_out, void* ws_, size_t ws_bytes,
                                          void* stream) {
  if (!pred0 || !pred1 || !target || !sel || !sum_out || !ws_) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_photometric_workspace(B, H, W)) return SEGSDE_ERR_WORKSPACE;
  hipLaunchKernelGGL(photo_packed() ? photometric_fwd2_kernel<false> : photometric_fwd_kernel<false>, photo_grid(B, H, W), dim3(256),
                     (9 * PF_H * PF_W + 1) * sizeof(float) + 64, ST(stream), pred0, pred1, target, ident, noise, H, W, no_ssim,
                     avg, photo_tiles_per_block(B, H, W), (float*)nullptr, sel, identity_selection, (double*)ws_);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 64, ST(stream), (const double*)ws_, (int)photo_blocks(B, H, W),
                     sum_out, 0, 1.0);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_photometric_backward(const float* pred0, const float* pred1, const float* target, const uint8_t* sel,
                                           int n_ident, const float* disp, int hs, int ws, const float* inv_K, const float* K,
                                           const float* T0, const float* T1, const float* src0, const float* src1, int B, int H,
                                           int W, float min_depth, float max_depth, int no_ssim, int avg, float scale,
                                           const float* weight, float* g_disp_up, float* gT0, float* gT1, void* ws_,
                                           size_t ws_bytes, void* stream) {
  if (!pred0 || !pred1 || !target || !sel || !disp || !inv_K || !K || !T0 || !T1 || !src0 || !src1 || !g_disp_up || !gT0 ||
      !gT1 || !ws_) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2 || hs <= 0 || ws <= 0 || (long)H * W >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  if (ws_bytes < segsde_photometric_workspace(B, H, W)) return SEGSDE_ERR_WORKSPACE;
  PhotoBwdP a;
  a.pred[0] = pred0; a.pred[1] = pred1; a.target = target; a.sel = sel; a.disp = disp; a.inv_K = inv_K; a.K = K;
  a.T[0] = T0; a.T[1] = T1; a.src[0] = src0; a.src[1] = src1;
  a.hs = hs; a.ws = ws; a.H = H; a.W = W; a.no_ssim = no_ssim; a.avg = avg;
  a.ni = n_ident ? (avg ? 1 : 2) : 0;
  a.tiles_per_block = photo_tiles_per_block(B, H, W);
  a.scale = scale; a.min_disp = 1.f / max_depth; a.max_disp = 1.f / min_depth;
  a.g_disp_up = g_disp_up; a.gP_part = (double*)ws_;
  const size_t lds = (9 * PB_H * PB_W + 18 * PC_H * PC_W + 6 * PT_H * PT_W + 48) * sizeof(float) + 64 + ((PC_H * PC_W + 15) / 16) * 16;
  if (photo_packed())
    hipLaunchKernelGGL(photo_split() ? photometric_bwd2_kernel<true> : photometric_bwd2_kernel<false>, photo_grid(B, H, W), dim3(256),
                       lds + 24 * sizeof(float), ST(stream), a);
  else
    hipLaunchKernelGGL(photometric_bwd_kernel, photo_grid(B, H, W), dim3(256), lds, ST(stream), a);
  SEGSDE_CHECK_LAUNCH();
  const int nblk = (int)(photo_blocks(B, H, W) / B);
  hipLaunchKernelGGL(photometric_bwd_finalize_kernel, dim3(B, 2), dim3(256), 204 * sizeof(double), ST(stream),
                     (const double*)ws_, nblk, K, weight, gT0, gT1);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
