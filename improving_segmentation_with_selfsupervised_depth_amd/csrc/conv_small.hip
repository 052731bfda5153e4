// Single-output-channel 3x3 convolution (the disparity heads: Conv3x3(num_ch_dec[s], 1) + sigmoid,
// models/depth_decoder.py:68-69,107-110) -- forward, data-gradient and weight-gradient.
//
// With Cout = 1 the op is a 9*C-tap stencil-reduce per pixel: 2*9*C FLOP against 4*C bytes, i.e. HBM-bound; running it
// through the 128x32 MFMA tile wastes 31/32 of the matrix work (2.8 TFLOP/s measured).  Here C/4 lanes own one pixel
// (one float4 of channels each, so a pixel's row is one coalesced 16*C/4-byte access), the 9 weight float4s live in
// registers, and the per-pixel dot product is finished with wave shuffles.
#include "segsde_common.h"
#include "conv_small.h"

namespace {
#define ST(s) static_cast<hipStream_t>(s)

__device__ __forceinline__ int refl1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// LP = lanes per pixel = C/4 (16, 32 or 64)
template <int LP>
__global__ __launch_bounds__(256) void c1_fwd_kernel(const float* x, int ldx, int B, int H, int W, const float* wp /*[9][C]*/,
                                                     const float* bias, int reflect, int act, float* y, int ldy) {
  constexpr int PPB = 256 / LP;                  // pixels per block pass
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4*>(wp + t * (LP * 4) + 4 * lane_c);
  const float b0 = bias ? bias[0] : 0.f;
  const long npix = (long)B * H * W;
  for (long p0 = (long)blockIdx.x * PPB; p0 < npix; p0 += (long)gridDim.x * PPB) {
    const long p = p0 + slot;
    const bool live = p < npix;
    const long pp = live ? p : 0;
    const int wq = (int)(pp % W); const long t2 = pp / W; const int hq = (int)(t2 % H); const long b = t2 / H;
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int hi = hq + kh - 1; bool okh = true;
      if (reflect) hi = refl1(hi, H); else okh = (unsigned)hi < (unsigned)H;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int wi = wq + kw - 1; bool ok = okh;
        if (reflect) wi = refl1(wi, W); else ok = ok && (unsigned)wi < (unsigned)W;
        if (ok) {
          const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + hi) * W + wi) * ldx + 4 * lane_c);
          const float4 ww = w[kh * 3 + kw];
          acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
        }
      }
    }
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_c == 0 && live) y[p * ldy] = segsde_act(acc + b0, act);
  }
}

// dx[pix][c] = sum over padded pre-images of pix and taps of dz[q] * wd[c][kh'][kw']   (wd = dgrad pack, flipped)
template <int LP>
__global__ __launch_bounds__(256) void c1_dgrad_kernel(const float* dz, int lddz, int B, int H, int W,
                                                       const float* wd /*[C][9]*/, int adjoint, float* dx, int lddx,
                                                       float* dx2, int lddx2, int nsplit) {
  constexpr int PPB = 256 / LP;
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float w[4][9];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) w[j][t] = wd[(4 * lane_c + j) * 9 + t];
  const long npix = (long)B * H * W;
  for (long p0 = (long)blockIdx.x * PPB; p0 < npix; p0 += (long)gridDim.x * PPB) {
    const long p = p0 + slot;
    if (p >= npix) continue;
    const int wq = (int)(p % W); const long t2 = p / W; const int hq = (int)(t2 % H); const long b = t2 / H;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    // pre-images per axis: the pixel itself (+ the mirrored padding cell when it sits next to the border)
    int hp[3], wpp[3], nh = 0, nw = 0;
    hp[nh++] = hq; wpp[nw++] = wq;
    if (adjoint) {
      if (hq == 1) hp[nh++] = -1;
      if (hq == H - 2) hp[nh++] = H;
      if (wq == 1) wpp[nw++] = -1;
      if (wq == W - 2) wpp[nw++] = W;
    }
    for (int a = 0; a < nh; ++a)
      for (int bb = 0; bb < nw; ++bb)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int yh = hp[a] + kh - 1;        // flipped tap index kh' reads dz at +kh'-1
          if ((unsigned)yh >= (unsigned)H) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int yw = wpp[bb] + kw - 1;
            if ((unsigned)yw >= (unsigned)W) continue;
            const float g = dz[((b * H + yh) * W + yw) * lddz];
            const int t = kh * 3 + kw;
            a0 += g * w[0][t]; a1 += g * w[1][t]; a2 += g * w[2][t]; a3 += g * w[3][t];
          }
        }
    const int c = 4 * lane_c;
    float* dst = c < nsplit ? dx + p * lddx + c : dx2 + p * lddx2 + (c - nsplit);
    *reinterpret_cast<float4*>(dst) = make_float4(a0, a1, a2, a3);
  }
}

// per-block partial dw[tap][c] = sum over this block's pixels of x[pad(pix + tap)][c] * dz[pix]
template <int LP>
__global__ __launch_bounds__(256) void c1_wgrad_kernel(const float* x, int ldx, int B, int H, int W, const float* dz,
                                                       int lddz, int reflect, float* part /*[nblk][9][C]*/) {
  constexpr int PPB = 256 / LP;
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);    // [PPB][9][C]
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long npix = (long)B * H * W;
  for (long p0 = (long)blockIdx.x * PPB; p0 < npix; p0 += (long)gridDim.x * PPB) {
    const long p = p0 + slot;
    if (p >= npix) continue;
    const int wq = (int)(p % W); const long t2 = p / W; const int hq = (int)(t2 % H); const long b = t2 / H;
    const float g = dz[p * lddz];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int hi = hq + kh - 1; bool okh = true;
      if (reflect) hi = refl1(hi, H); else okh = (unsigned)hi < (unsigned)H;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int wi = wq + kw - 1; bool ok = okh;
        if (reflect) wi = refl1(wi, W); else ok = ok && (unsigned)wi < (unsigned)W;
        if (ok) {
          const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + hi) * W + wi) * ldx + 4 * lane_c);
          float4& a = acc[kh * 3 + kw];
          a.x += v.x * g; a.y += v.y * g; a.z += v.z * g; a.w += v.w * g;
        }
      }
    }
  }
  constexpr int C = LP * 4;
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(sh + (slot * 9 + t) * C + 4 * lane_c) = acc[t];
  __syncthreads();
  for (int e = threadIdx.x; e < 9 * C; e += 256) {
    float s = 0.f;
    for (int q = 0; q < PPB; ++q) s += sh[q * 9 * C + e];
    part[(long)blockIdx.x * 9 * C + e] = s;
  }
}

__global__ __launch_bounds__(256) void c1_wgrad_reduce_kernel(const float* part, int nblk, int C, float* dw /*[1][C][3][3]*/) {
  const int e = blockIdx.x * 256 + threadIdx.x;     // e = tap * C + c
  if (e >= 9 * C) return;
  float s = 0.f;
  for (int z = 0; z < nblk; ++z) s += part[(long)z * 9 * C + e];
  const int tap = e / C, c = e - tap * C;
  dw[c * 9 + tap] = s;
}

inline int c1_blocks(long npix, int ppb) { long nb = (npix + ppb - 1) / ppb; return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb)); }
}  // namespace

bool segsde_c1_supported(int C, int ldx) { return (C == 64 || C == 128 || C == 256) && (ldx % 4 == 0); }

size_t segsde_c1_wgrad_workspace(int C) { return (size_t)1024 * 9 * C * sizeof(float); }

#define C1_DISPATCH(KERNEL, grid, smem, ...)                                                       \
  do {                                                                                             \
    if (C == 64) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);  \
    else if (C == 128) hipLaunchKernelGGL((KERNEL<32>), grid, dim3(256), smem, ST(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);          \
  } while (0)

int segsde_c1_forward(const float* x, int ldx, int B, int H, int W, int C, const float* wpack, const float* bias, int reflect,
                      int act, float* y, int ldy, void* stream) {
  const dim3 grid(c1_blocks((long)B * H * W, 256 / (C / 4)));
  C1_DISPATCH(c1_fwd_kernel, grid, 0, x, ldx, B, H, W, wpack, bias, reflect, act, y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_c1_dgrad(const float* dz, int lddz, int B, int H, int W, int C, const float* wdpack, int adjoint, float* dx, int lddx,
                    float* dx2, int lddx2, int nsplit, void* stream) {
  if (!dx2) { dx2 = dx; lddx2 = lddx; nsplit = C; }
  const dim3 grid(c1_blocks((long)B * H * W, 256 / (C / 4)));
  C1_DISPATCH(c1_dgrad_kernel, grid, 0, dz, lddz, B, H, W, wdpack, adjoint, dx, lddx, dx2, lddx2, nsplit);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_c1_wgrad(const float* x, int ldx, int B, int H, int W, int C, const float* dz, int lddz, int reflect, float* dw,
                    float* workspace, void* stream) {
  const int ppb = 256 / (C / 4);
  long nb = ((long)B * H * W + ppb - 1) / ppb;
  const int nblk = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
  const size_t smem = (size_t)ppb * 9 * C * sizeof(float);
  C1_DISPATCH(c1_wgrad_kernel, dim3(nblk), smem, x, ldx, B, H, W, dz, lddz, reflect, workspace);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, ST(stream), (const float*)workspace, nblk,
                     C, dw);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
