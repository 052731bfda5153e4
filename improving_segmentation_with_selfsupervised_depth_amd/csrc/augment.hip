// The rest of the reference's strongTransform (train.py:654-659): ColorJitter and GaussianBlur of the mixed images,
// loader/transformsgpu.py:10-30.  The reference delegates both to kornia 0.4.0 (requirements.txt:16), which is neither
// under /root/reference nor installed here: the arithmetic below restates kornia 0.4.0's published algorithm
// (kornia.augmentation.ColorJitter -> color.adjust_{brightness,contrast,saturation,hue} with rgb_to_hsv / hsv_to_rgb;
// kornia.filters.GaussianBlur2d -> get_gaussian_kernel2d + filter2D with reflect borders) -- PARITY UNPINNED, see DESIGN.md.
// HBM-bound, one pass (jitter) / two separable passes (blur) over [B,3,H,W] planar images.
#include "segsde_common.h"

namespace {
#define ST(s) static_cast<hipStream_t>(s)
constexpr float TWO_PI = 6.283185307179586f;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float floor_mod(float a, float m) { const float r = fmodf(a, m); return (r != 0.f && ((r < 0.f) != (m < 0.f))) ? r + m : r; }

// kornia.color.rgb_to_hsv: h in radians [0, 2 pi), s = (max - min) / max (0 where max == 0), v = max
__device__ __forceinline__ void rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  v = maxc;
  const float deltac = maxc - minc;
  s = deltac / v;
  if (!(s == s)) s = 0.f;                                   // 0 / 0 for a black pixel
  const float dc = deltac == 0.f ? 1.f : deltac;
  const float rc = (maxc - r) / dc, gc = (maxc - g) / dc, bc = (maxc - b) / dc;
  float hh = 4.f + gc - rc;
  if (g == maxc) hh = 2.f + rc - bc;
  if (r == maxc) hh = bc - gc;
  if (minc == maxc) hh = 0.f;
  hh = floor_mod(hh / 6.f, 1.f);
  h = TWO_PI * hh;
}
// kornia.color.hsv_to_rgb (sector table v,q,p,p,t,v / t,v,v,q,p,p / p,p,t,v,v,q)
__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
  const float hh = h / TWO_PI;
  const float hi = floor_mod(floorf(hh * 6.f), 6.f);
  const float f = floor_mod(hh * 6.f, 6.f) - hi;
  const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  const int i = (int)hi;
  r = i == 0 ? v : (i == 1 ? q : (i == 2 ? p : (i == 3 ? p : (i == 4 ? t : v))));
  g = i == 0 ? t : (i == 1 ? v : (i == 2 ? v : (i == 3 ? q : (i == 4 ? p : p))));
  b = i == 0 ? p : (i == 1 ? p : (i == 2 ? t : (i == 3 ? v : (i == 4 ? v : q))));
}

// params [B][4] = {brightness_factor, contrast_factor, saturation_factor, hue_factor}; the four adjustments are applied
// in the order `order` (a permutation of 0 brightness, 1 contrast, 2 saturation, 3 hue), each on the result of the last:
//   brightness: clamp(x + (bf - 1), 0, 1);  contrast: clamp(x * cf, 0, 1);
//   saturation: HSV, s <- clamp(s * sf, 0, 1);  hue: HSV, h <- fmod(h + 2 pi hf, 2 pi)
__global__ __launch_bounds__(256) void color_jitter_kernel(const float* x, float* y, long HW, const float* params, int o0,
                                                           int o1, int o2, int o3) {
  const int b = blockIdx.y;
  const float bf = params[b * 4], cf = params[b * 4 + 1], sf = params[b * 4 + 2], hf = params[b * 4 + 3];
  const float* xb = x + (long)b * 3 * HW;
  float* yb = y + (long)b * 3 * HW;
  const int order[4] = {o0, o1, o2, o3};
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    float r = xb[p], g = xb[HW + p], bl = xb[2 * HW + p];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int op = order[k];
      if (op == 0) { const float a = bf - 1.f; r = clamp01(r + a); g = clamp01(g + a); bl = clamp01(bl + a); }
      else if (op == 1) { r = clamp01(r * cf); g = clamp01(g * cf); bl = clamp01(bl * cf); }
      else {
        float h, s, v;
        rgb_to_hsv(r, g, bl, h, s, v);
        if (op == 2) s = clamp01(s * sf);
        else h = fmodf(h + hf * TWO_PI, TWO_PI);
        hsv_to_rgb(h, s, v, r, g, bl);
      }
    }
    yb[p] = r; yb[HW + p] = g; yb[2 * HW + p] = bl;
  }
}

// one separable pass: out[.., i, ..] = sum_t w[t] * x[.., reflect(i + t - R), ..] along rows (axis 0) or columns (axis 1)
__global__ __launch_bounds__(256) void blur_pass_kernel(const float* x, float* y, int H, int W, const float* w, int ntaps,
                                                        int axis) {
  const long HW = (long)H * W;
  const float* xp = x + (long)blockIdx.y * HW;     // one of the B*3 planes
  float* yp = y + (long)blockIdx.y * HW;
  const int R = ntaps >> 1;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int h = (int)(p / W), c = (int)(p - (long)h * W);
    float acc = 0.f;
    for (int t = 0; t < ntaps; ++t) {
      if (axis == 0) {
        int i = h + t - R; i = i < 0 ? -i : (i >= H ? 2 * H - 2 - i : i);
        acc += w[t] * xp[(long)i * W + c];
      } else {
        int i = c + t - R; i = i < 0 ? -i : (i >= W ? 2 * W - 2 - i : i);
        acc += w[t] * xp[(long)h * W + i];
      }
    }
    yp[p] = acc;
  }
}
inline int plane_blocks(long HW) { long nb = (HW + 255) / 256; return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb)); }
}  // namespace

extern "C" int segsde_color_jitter(const float* x, int B, long HW, const float* params, const int* order, float* y,
                                   void* stream) {
  if (!x || !params || !order || !y) return SEGSDE_ERR_NULL;
  if (B <= 0 || HW <= 0) return SEGSDE_ERR_SHAPE;
  int seen = 0;
  for (int k = 0; k < 4; ++k) { if (order[k] < 0 || order[k] > 3) return SEGSDE_ERR_SHAPE; seen |= 1 << order[k]; }
  if (seen != 15) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(color_jitter_kernel, dim3(plane_blocks(HW), B), dim3(256), 0, ST(stream), x, y, HW, params, order[0],
                     order[1], order[2], order[3]);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_gaussian_blur(const float* x, int planes, int H, int W, const float* wy, int ny, const float* wx, int nx,
                                    float* tmp, float* y, void* stream) {
  if (!x || !wy || !wx || !tmp || !y) return SEGSDE_ERR_NULL;
  if (planes <= 0 || H < 1 || W < 1 || ny < 1 || nx < 1 || !(ny & 1) || !(nx & 1)) return SEGSDE_ERR_SHAPE;
  if (ny / 2 >= H || nx / 2 >= W) return SEGSDE_ERR_SHAPE;      // reflection needs the radius inside the image
  const dim3 grid(plane_blocks((long)H * W), planes);
  hipLaunchKernelGGL(blur_pass_kernel, grid, dim3(256), 0, ST(stream), x, tmp, H, W, wy, ny, 0);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(blur_pass_kernel, grid, dim3(256), 0, ST(stream), (const float*)tmp, y, H, W, wx, nx, 1);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
