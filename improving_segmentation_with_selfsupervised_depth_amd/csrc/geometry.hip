// Stand-alone versions of the geometry / SSIM layers of models/monodepth_layers.py (:145-254: BackprojectDepth, Project3D,
// SSIM, get_smooth_loss) for callers OUTSIDE the training path.  MonodepthLoss never launches these: its fused kernels
// (loss.hip) fold the same arithmetic into one pass per scale.  They exist so that a user script that imports the layers by
// name keeps working on device tensors: one thread per output element, plain coalesced loads (every operand is read once
// or a handful of times through L1/L2; nothing here is on the benchmarked step).
#include "segsde_common.h"

namespace {

__device__ __forceinline__ int refl1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

struct WinStat { float mx, my, sx, sy, sxy; };
// 3x3 window of the reflection-padded planes around (h, w): means and (co)variances as SSIM.forward forms them
__device__ __forceinline__ WinStat win_stat(const float* x, const float* y, int H, int W, int h, int w) {
  float ax = 0.f, ay = 0.f, axx = 0.f, ayy = 0.f, axy = 0.f;
  for (int dh = -1; dh <= 1; ++dh) {
    const int hh = refl1(h + dh, H);
    for (int dw = -1; dw <= 1; ++dw) {
      const int ww = refl1(w + dw, W);
      const float xv = x[(long)hh * W + ww], yv = y[(long)hh * W + ww];
      ax += xv; ay += yv; axx += xv * xv; ayy += yv * yv; axy += xv * yv;
    }
  }
  WinStat s;
  s.mx = ax / 9.f; s.my = ay / 9.f;
  s.sx = axx / 9.f - s.mx * s.mx; s.sy = ayy / 9.f - s.my * s.my; s.sxy = axy / 9.f - s.mx * s.my;
  return s;
}
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

__global__ __launch_bounds__(256) void ssim_map_fwd_kernel(const float* x, const float* y, int H, int W, long total, float* out) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= total) return;
  const long HW = (long)H * W, plane = e / HW;
  const int p = (int)(e - plane * HW), h = p / W, w = p - h * W;
  const WinStat s = win_stat(x + plane * HW, y + plane * HW, H, W, h, w);
  const float n = (2.f * s.mx * s.my + SSIM_C1) * (2.f * s.sxy + SSIM_C2);
  const float d = (s.mx * s.mx + s.my * s.my + SSIM_C1) * (s.sx + s.sy + SSIM_C2);
  const float v = (1.f - n / d) * 0.5f;
  out[e] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
}

// gather form of the adjoint: input pixel p collects from every window centre q whose (reflected) 3x3 footprint contains p,
// with the footprint's multiplicity (a border pixel's mirror image can put p into a window twice per axis)
__global__ __launch_bounds__(256) void ssim_map_bwd_kernel(const float* x, const float* y, const float* gout, int H, int W,
                                                           long total, float* gx, float* gy) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= total) return;
  const long HW = (long)H * W, plane = e / HW;
  const int p = (int)(e - plane * HW), ph = p / W, pw = p - ph * W;
  const float* xp = x + plane * HW;
  const float* yp = y + plane * HW;
  const float* gp = gout + plane * HW;
  const float xv = xp[p], yv = yp[p];
  float accx = 0.f, accy = 0.f;
  for (int qh = ph - 2; qh <= ph + 2; ++qh) {
    if (qh < 0 || qh >= H) continue;
    int mh = 0;
    for (int d = -1; d <= 1; ++d) mh += refl1(qh + d, H) == ph;
    if (!mh) continue;
    for (int qw = pw - 2; qw <= pw + 2; ++qw) {
      if (qw < 0 || qw >= W) continue;
      int mw = 0;
      for (int d = -1; d <= 1; ++d) mw += refl1(qw + d, W) == pw;
      if (!mw) continue;
      const float g = gp[(long)qh * W + qw];
      if (g == 0.f) continue;
      const WinStat s = win_stat(xp, yp, H, W, qh, qw);
      const float n1 = 2.f * s.mx * s.my + SSIM_C1, n2 = 2.f * s.sxy + SSIM_C2;
      const float d1 = s.mx * s.mx + s.my * s.my + SSIM_C1, d2 = s.sx + s.sy + SSIM_C2;
      const float N = n1 * n2, D = d1 * d2, v = (1.f - N / D) * 0.5f;
      if (v < 0.f || v > 1.f) continue;                      // torch.clamp passes the gradient inside [min, max] only
      const float k = -0.5f * g * (float)(mh * mw) / (9.f * D * D);
      // d/dx_p: mu_x -> 1/9, sigma_x -> 2 (x_p - mu_x) / 9, sigma_xy -> (y_p - mu_y) / 9 (the 1/9 sits in k)
      const float dNx = 2.f * s.my * n2 + n1 * 2.f * (yv - s.my), dDx = 2.f * s.mx * d2 + d1 * 2.f * (xv - s.mx);
      const float dNy = 2.f * s.mx * n2 + n1 * 2.f * (xv - s.mx), dDy = 2.f * s.my * d2 + d1 * 2.f * (yv - s.my);
      accx += k * (dNx * D - N * dDx);
      accy += k * (dNy * D - N * dDy);
    }
  }
  if (gx) gx[e] = accx;
  if (gy) gy[e] = accy;
}

__global__ __launch_bounds__(256) void backproject_kernel(const float* depth, const float* inv_K, int H, int W, float* out) {
  const int b = blockIdx.y;
  const long HW = (long)H * W, n = blockIdx.x * 256L + threadIdx.x;
  if (n >= HW) return;
  const float* k = inv_K + b * 16;
  const float px = (float)(n % W), py = (float)(n / W), d = depth[b * HW + n];
  float* o = out + (long)b * 4 * HW;
  for (int r = 0; r < 3; ++r) o[r * HW + n] = d * (k[4 * r] * px + k[4 * r + 1] * py + k[4 * r + 2]);
  o[3 * HW + n] = 1.f;
}

__global__ __launch_bounds__(256) void project3d_kernel(const float* pts, const float* K, const float* T, int H, int W, float eps,
                                                        float* out) {
  const int b = blockIdx.y;
  const long HW = (long)H * W, n = blockIdx.x * 256L + threadIdx.x;
  if (n >= HW) return;
  float P[12];                                               // (K @ T)[:3, :]
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
      for (int j = 0; j < 4; ++j) a += K[b * 16 + 4 * r + j] * T[b * 16 + 4 * j + c];
      P[4 * r + c] = a;
    }
  const float* q = pts + (long)b * 4 * HW;
  float cam[3];
  for (int r = 0; r < 3; ++r) {
    float a = 0.f;
    for (int j = 0; j < 4; ++j) a += P[4 * r + j] * q[j * HW + n];
    cam[r] = a;
  }
  const float den = cam[2] + eps;
  out[((long)b * HW + n) * 2] = (cam[0] / den / (float)(W - 1) - 0.5f) * 2.f;
  out[((long)b * HW + n) * 2 + 1] = (cam[1] / den / (float)(H - 1) - 0.5f) * 2.f;
}


// adjoint of backproject_kernel w.r.t. the depth: d depth[p] = sum_r g[r][p] * (inv_K[r][:3] . (u, v, 1))
__global__ __launch_bounds__(256) void backproject_bwd_kernel(const float* g, const float* inv_K, int H, int W, float* gdepth) {
  const int b = blockIdx.y;
  const long HW = (long)H * W, n = blockIdx.x * 256L + threadIdx.x;
  if (n >= HW) return;
  const float* k = inv_K + b * 16;
  const float px = (float)(n % W), py = (float)(n / W);
  const float* gp = g + (long)b * 4 * HW;
  float a = 0.f;
  for (int r = 0; r < 3; ++r) a += gp[r * HW + n] * (k[4 * r] * px + k[4 * r + 1] * py + k[4 * r + 2]);
  gdepth[b * HW + n] = a;
}

// adjoint of project3d_kernel: d points [B,4,HW] (nullable) and, per block, the twelve partial sums of d P = d cam . points^T
// (doubles, part[b][block][12]); project3d_bwd_finalize_kernel folds them in block order and applies K^T: d T = K[:3,:]^T d P
__global__ __launch_bounds__(256) void project3d_bwd_kernel(const float* pts, const float* K, const float* T, const float* gout,
                                                            int H, int W, float eps, float* gpts, double* part) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);
  const int b = blockIdx.y;
  const long HW = (long)H * W, n = blockIdx.x * 256L + threadIdx.x;
  float P[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
      for (int j = 0; j < 4; ++j) a += K[b * 16 + 4 * r + j] * T[b * 16 + 4 * j + c];
      P[4 * r + c] = a;
    }
  float dc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < HW) {
    const float* qp = pts + (long)b * 4 * HW;
    for (int j = 0; j < 4; ++j) q[j] = qp[j * HW + n];
    float cam[3];
    for (int r = 0; r < 3; ++r) {
      float a = 0.f;
      for (int j = 0; j < 4; ++j) a += P[4 * r + j] * q[j];
      cam[r] = a;
    }
    const float den = cam[2] + eps;
    const float gx = gout[((long)b * HW + n) * 2] * 2.f / (float)(W - 1), gy = gout[((long)b * HW + n) * 2 + 1] * 2.f / (float)(H - 1);
    dc[0] = gx / den; dc[1] = gy / den;
    dc[2] = -(gx * cam[0] + gy * cam[1]) / (den * den);
    if (gpts) {
      float* o = gpts + (long)b * 4 * HW;
      for (int j = 0; j < 4; ++j) o[j * HW + n] = P[j] * dc[0] + P[4 + j] * dc[1] + P[8 + j] * dc[2];
    }
  }
  if (part) {
    for (int r = 0; r < 3; ++r)
      for (int j = 0; j < 4; ++j) {
        const double s = segsde_block_sum((double)dc[r] * (double)q[j], sh);
        if (threadIdx.x == 0) part[((long)b * gridDim.x + blockIdx.x) * 12 + 4 * r + j] = s;
      }
  }
}

__global__ __launch_bounds__(64) void project3d_bwd_finalize_kernel(const double* part, const float* K, int nblk, float* gT) {
  const int b = blockIdx.x, e = threadIdx.x;
  SEGSDE_SMEM;
  double* dP = reinterpret_cast<double*>(segsde_smem);
  if (e < 12) {
    double a = 0.0;
    for (int i = 0; i < nblk; ++i) a += part[((long)b * nblk + i) * 12 + e];
    dP[e] = a;
  }
  __syncthreads();
  if (e < 16) {                                            // d T[k][c] = sum_r K[r][k] d P[r][c], r < 3
    const int k = e >> 2, c = e & 3;
    double a = 0.0;
    for (int r = 0; r < 3; ++r) a += (double)K[b * 16 + 4 * r + k] * dP[4 * r + c];
    gT[b * 16 + e] = (float)a;
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int segsde_ssim_map_forward(const float* x, const float* y, int B, int C, int H, int W, float* out, void* stream) {
  if (!x || !y || !out) return SEGSDE_ERR_NULL;
  if (B <= 0 || C <= 0 || H < 2 || W < 2) return SEGSDE_ERR_SHAPE;
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(ssim_map_fwd_kernel, dim3(segsde_cdiv(total, 256)), dim3(256), 0, ST(stream), x, y, H, W, total, out);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_ssim_map_backward(const float* x, const float* y, const float* gout, int B, int C, int H, int W, float* gx,
                                        float* gy, void* stream) {
  if (!x || !y || !gout || (!gx && !gy)) return SEGSDE_ERR_NULL;
  if (B <= 0 || C <= 0 || H < 2 || W < 2) return SEGSDE_ERR_SHAPE;
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(ssim_map_bwd_kernel, dim3(segsde_cdiv(total, 256)), dim3(256), 0, ST(stream), x, y, gout, H, W, total, gx, gy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_backproject_depth(const float* depth, const float* inv_K, int B, int H, int W, float* cam_points,
                                        void* stream) {
  if (!depth || !inv_K || !cam_points) return SEGSDE_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(backproject_kernel, dim3(segsde_cdiv((long)H * W, 256), B), dim3(256), 0, ST(stream), depth, inv_K, H, W,
                     cam_points);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_project3d(const float* points, const float* K, const float* T, int B, int H, int W, float eps,
                                float* pix_coords, void* stream) {
  if (!points || !K || !T || !pix_coords) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(project3d_kernel, dim3(segsde_cdiv((long)H * W, 256), B), dim3(256), 0, ST(stream), points, K, T, H, W, eps,
                     pix_coords);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_backproject_depth_backward(const float* g_cam_points, const float* inv_K, int B, int H, int W, float* g_depth,
                                                 void* stream) {
  if (!g_cam_points || !inv_K || !g_depth) return SEGSDE_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(backproject_bwd_kernel, dim3(segsde_cdiv((long)H * W, 256), B), dim3(256), 0, ST(stream), g_cam_points, inv_K,
                     H, W, g_depth);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t segsde_project3d_backward_workspace(int B, int H, int W) {
  if (B <= 0 || H < 2 || W < 2) return 0;
  return (size_t)B * segsde_cdiv((long)H * W, 256) * 12 * sizeof(double);
}

extern "C" int segsde_project3d_backward(const float* points, const float* K, const float* T, const float* g_pix, int B, int H, int W,
                                         float eps, float* g_points, float* g_T, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (!points || !K || !T || !g_pix || (!g_points && !g_T)) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 2 || W < 2) return SEGSDE_ERR_SHAPE;
  if (g_T && (!workspace || workspace_bytes < segsde_project3d_backward_workspace(B, H, W))) return SEGSDE_ERR_WORKSPACE;
  const int nblk = segsde_cdiv((long)H * W, 256);
  double* part = g_T ? static_cast<double*>(workspace) : nullptr;
  hipLaunchKernelGGL(project3d_bwd_kernel, dim3(nblk, B), dim3(256), 4 * sizeof(double), ST(stream), points, K, T, g_pix, H, W, eps, g_points, part);
  SEGSDE_CHECK_LAUNCH();
  if (g_T) {
    hipLaunchKernelGGL(project3d_bwd_finalize_kernel, dim3(B), dim3(64), 12 * sizeof(double), ST(stream), part, K, nblk, g_T);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}
