// Axis-angle + translation -> 4x4 camera transform and its adjoint
// (models/monodepth_layers.py:30-105: transformation_from_parameters, rot_from_axisangle,
// get_translation_matrix).  B is tiny: one thread per batch element; the backward uses forward-mode duals
// over the 6 inputs, contracted with dM.
#include "segsde_common.h"

namespace {

struct D6 { float v; float d[6]; };
__device__ __forceinline__ D6 cst(float v) { D6 r; r.v = v; for (int i = 0; i < 6; ++i) r.d[i] = 0.f; return r; }
__device__ __forceinline__ D6 var(float v, int i) { D6 r = cst(v); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D6 operator+(const D6& a, const D6& b) { D6 r; r.v = a.v + b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D6 operator-(const D6& a, const D6& b) { D6 r; r.v = a.v - b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D6 operator-(const D6& a) { D6 r; r.v = -a.v; for (int i = 0; i < 6; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ D6 operator*(const D6& a, const D6& b) { D6 r; r.v = a.v * b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D6 operator/(const D6& a, const D6& b) { D6 r; r.v = a.v / b.v; for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
__device__ __forceinline__ D6 dsqrt0(const D6& a) {  // sqrt with the zero sub-gradient torch.norm uses at 0
  D6 r; r.v = sqrtf(a.v);
  for (int i = 0; i < 6; ++i) r.d[i] = r.v > 0.f ? a.d[i] / (2.f * r.v) : 0.f;
  return r;
}
__device__ __forceinline__ D6 dsin(const D6& a) { D6 r; r.v = sinf(a.v); const float c = cosf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = c * a.d[i]; return r; }
__device__ __forceinline__ D6 dcos(const D6& a) { D6 r; r.v = cosf(a.v); const float s = -sinf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = s * a.d[i]; return r; }

// M[16] as duals of (axisangle[3], translation[3])
__device__ __forceinline__ void pose_dual(const float* aa, const float* tr, int invert, D6* M) {
  const D6 v0 = var(aa[0], 0), v1 = var(aa[1], 1), v2 = var(aa[2], 2);
  D6 t[3] = {var(tr[0], 3), var(tr[1], 4), var(tr[2], 5)};
  const D6 angle = dsqrt0(v0 * v0 + v1 * v1 + v2 * v2);
  const D6 den = angle + cst(1e-7f);
  const D6 x = v0 / den, y = v1 / den, z = v2 / den;
  const D6 ca = dcos(angle), sa = dsin(angle), C = cst(1.f) - ca;
  const D6 xs = x * sa, ys = y * sa, zs = z * sa, xC = x * C, yC = y * C, zC = z * C;
  const D6 xyC = x * yC, yzC = y * zC, zxC = z * xC;
  D6 R[3][3] = {{x * xC + ca, xyC - zs, zxC + ys}, {xyC + zs, y * yC + ca, yzC - xs}, {zxC - ys, yzC + xs, z * zC + ca}};
  for (int i = 0; i < 16; ++i) M[i] = cst(0.f);
  M[15] = cst(1.f);
  if (invert) {
    // M = R^T @ T(-t)
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[j][i];
      M[i * 4 + 3] = R[0][i] * (-t[0]) + R[1][i] * (-t[1]) + R[2][i] * (-t[2]);
    }
  } else {
    // M = T(t) @ R
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i][j];
      M[i * 4 + 3] = t[i];
    }
  }
}

__global__ __launch_bounds__(64) void pose_fwd_kernel(const float* aa, const float* tr, int B, int stride, int invert, float* M) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  D6 m[16];
  pose_dual(aa + (long)b * stride, tr + (long)b * stride, invert, m);
  for (int i = 0; i < 16; ++i) M[b * 16 + i] = m[i].v;
}
__global__ __launch_bounds__(64) void pose_bwd_kernel(const float* aa, const float* tr, const float* dM, int B, int stride,
                                                      int invert, float* daa, float* dtr) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  D6 m[16];
  pose_dual(aa + (long)b * stride, tr + (long)b * stride, invert, m);
  float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 6; ++k) g[k] += dM[b * 16 + i] * m[i].d[k];
  for (int k = 0; k < 3; ++k) { daa[(long)b * stride + k] = g[k]; dtr[(long)b * stride + k] = g[3 + k]; }
}
}  // namespace

extern "C" int segsde_pose_matrix_forward(const float* aa, const float* tr, int B, int stride, int invert, float* M,
                                          void* stream) {
  if (!aa || !tr || !M) return SEGSDE_ERR_NULL;
  if (B <= 0 || stride < 3) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(pose_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), aa, tr, B, stride, invert, M);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int segsde_pose_matrix_backward(const float* aa, const float* tr, const float* dM, int B, int stride, int invert,
                                           float* daa, float* dtr, void* stream) {
  if (!aa || !tr || !dM || !daa || !dtr) return SEGSDE_ERR_NULL;
  if (B <= 0 || stride < 3) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(pose_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), aa, tr, dM, B, stride, invert, daa, dtr);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
