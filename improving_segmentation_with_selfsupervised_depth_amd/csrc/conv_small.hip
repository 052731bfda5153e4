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
                                                       float* dx2, int lddx2, int nsplit, const float* agy, int agld,
                                                       int agkind) {
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
    if (agy && c < nsplit) {   // the input of this conv is the output of an activation: its derivative is applied here
      const float4 yv = *reinterpret_cast<const float4*>(agy + p * agld + c);
      a0 *= segsde_act_grad_from_out(yv.x, agkind); a1 *= segsde_act_grad_from_out(yv.y, agkind);
      a2 *= segsde_act_grad_from_out(yv.z, agkind); a3 *= segsde_act_grad_from_out(yv.w, agkind);
    }
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

// 16 lanes per output element split the block partials between them (fixed order), then a shuffle tree: with one thread
// per element the 9*C outputs were 576 threads walking 1024 partials each -- 300 us of pure latency
__global__ __launch_bounds__(256) void c1_wgrad_reduce_kernel(const float* part, int nblk, int C, float* dw /*[1][C][3][3]*/) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;     // e = tap * C + c
  const bool live = e < 9 * C;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (int z = l; z < nblk; z += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (z + 16 * u < nblk) s4[u] += part[(long)(z + 16 * u) * 9 * C + e];
    }
  float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
#pragma unroll
  for (int d = 8; d > 0; d >>= 1) s += __shfl_xor(s, d);
  if (live && l == 0) {
    const int tap = e / C, c = e - tap * C;
    dw[c * 9 + tap] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// Strip versions (W % 8 == 0, every real layer): a group of LP lanes owns S = 8 consecutive pixels of one row.  The
// 3 x (S+2) input window is loaded once and slid across the S outputs (3.75 instead of 9 pixel loads per output: the
// per-pixel kernels are bound by L1 re-reads, not HBM) and the (b, h, w) decode is one 32-bit division per strip
// instead of two 64-bit divisions per pixel.
// ---------------------------------------------------------------------------------------------------
constexpr int CS = 8;

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

struct Strip { int b, h, w0; bool live; };
__device__ __forceinline__ Strip strip_decode(int sid, int nstrips, int nsr, int H) {
  Strip t;
  t.live = sid < nstrips;
  const int s = t.live ? sid : 0;
  const int row = s / nsr;
  t.w0 = (s - row * nsr) * CS;
  t.b = row / H;
  t.h = row - t.b * H;
  return t;
}

template <int LP>
__global__ __launch_bounds__(256) void c1s_fwd_kernel(const float* x, int ldx, int B, int H, int W, const float* wp /*[9][C]*/,
                                                      const float* bias, int reflect, int act, float* y, int ldy) {
  constexpr int G = 256 / LP;
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4*>(wp + t * (LP * 4) + 4 * lane_c);
  const float b0 = bias ? bias[0] : 0.f;
  const int nsr = W / CS, nstrips = B * H * nsr;
  for (int s0 = blockIdx.x * G; s0 < nstrips; s0 += gridDim.x * G) {
    const Strip st = strip_decode(s0 + slot, nstrips, nsr, H);
    float acc[CS];
#pragma unroll
    for (int o = 0; o < CS; ++o) acc[o] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int hi = st.h + r - 1; bool okh = st.live;
      if (reflect) hi = refl1(hi, H); else okh = okh && (unsigned)hi < (unsigned)H;
      const float* rp = x + (size_t)(st.b * H + (okh ? hi : 0)) * W * ldx + 4 * lane_c;
      float4 v[CS + 2];
#pragma unroll
      for (int j = 0; j < CS + 2; ++j) {
        int wi = st.w0 - 1 + j; bool ok = okh;
        if (reflect) wi = refl1(wi, W); else ok = ok && (unsigned)wi < (unsigned)W;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v[j] = *reinterpret_cast<const float4*>(rp + (size_t)wi * ldx);
      }
#pragma unroll
      for (int j = 0; j < CS + 2; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int o = j - kw;
          if (o >= 0 && o < CS) acc[o] += dot4(v[j], w[r * 3 + kw]);
        }
    }
    float mine = 0.f;
#pragma unroll
    for (int o = 0; o < CS; ++o) {
      float a = acc[o];
#pragma unroll
      for (int d = LP / 2; d > 0; d >>= 1) a += __shfl_xor(a, d);
      mine = lane_c == o ? a : mine;
    }
    if (lane_c < CS && st.live) y[((size_t)(st.b * H + st.h) * W + st.w0 + lane_c) * ldy] = segsde_act(mine + b0, act);
  }
}

template <int LP>
__global__ __launch_bounds__(256) void c1s_wgrad_kernel(const float* x, int ldx, int B, int H, int W, const float* dz, int lddz,
                                                        int reflect, float* part /*[nblk][9][C]*/) {
  constexpr int G = 256 / LP;
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);    // [G][9][C]
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nsr = W / CS, nstrips = B * H * nsr;
  for (int s0 = blockIdx.x * G; s0 < nstrips; s0 += gridDim.x * G) {
    const Strip st = strip_decode(s0 + slot, nstrips, nsr, H);
    if (!st.live) continue;
    float g[CS];
    const float* gp = dz + ((size_t)(st.b * H + st.h) * W + st.w0) * lddz;
#pragma unroll
    for (int o = 0; o < CS; ++o) g[o] = gp[(size_t)o * lddz];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int hi = st.h + r - 1; bool okh = true;
      if (reflect) hi = refl1(hi, H); else okh = (unsigned)hi < (unsigned)H;
      const float* rp = x + (size_t)(st.b * H + (okh ? hi : 0)) * W * ldx + 4 * lane_c;
      float4 v[CS + 2];
#pragma unroll
      for (int j = 0; j < CS + 2; ++j) {
        int wi = st.w0 - 1 + j; bool ok = okh;
        if (reflect) wi = refl1(wi, W); else ok = ok && (unsigned)wi < (unsigned)W;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v[j] = *reinterpret_cast<const float4*>(rp + (size_t)wi * ldx);
      }
#pragma unroll
      for (int j = 0; j < CS + 2; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int o = j - kw;
          if (o >= 0 && o < CS) {
            float4& a = acc[r * 3 + kw];
            a.x += v[j].x * g[o]; a.y += v[j].y * g[o]; a.z += v[j].z * g[o]; a.w += v[j].w * g[o];
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
    for (int q = 0; q < G; ++q) s += sh[q * 9 * C + e];
    part[(long)blockIdx.x * 9 * C + e] = s;
  }
}

template <int LP>
__global__ __launch_bounds__(256) void c1s_dgrad_kernel(const float* dz, int lddz, int B, int H, int W, const float* wd /*[C][9]*/,
                                                        int adjoint, float* dx, int lddx, float* dx2, int lddx2, int nsplit,
                                                        const float* agy, int agld, int agkind) {
  constexpr int G = 256 / LP;
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float w[4][9];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) w[j][t] = wd[(4 * lane_c + j) * 9 + t];
  const int c = 4 * lane_c;
  const int nsr = W / CS, nstrips = B * H * nsr;
  for (int s0 = blockIdx.x * G; s0 < nstrips; s0 += gridDim.x * G) {
    const Strip st = strip_decode(s0 + slot, nstrips, nsr, H);
    if (!st.live) continue;
    const size_t pix0 = (size_t)(st.b * H + st.h) * W + st.w0;
    float* dst = c < nsplit ? dx + pix0 * lddx + c : dx2 + pix0 * lddx2 + (c - nsplit);
    const int ldd = c < nsplit ? lddx : lddx2;
    const float* ag = (agy && c < nsplit) ? agy + pix0 * agld + c : nullptr;
    auto put = [&](int o, float a0, float a1, float a2, float a3) {
      if (ag) {   // derivative of the activation whose output this conv read, from that saved output
        const float4 yv = *reinterpret_cast<const float4*>(ag + (size_t)o * agld);
        a0 *= segsde_act_grad_from_out(yv.x, agkind); a1 *= segsde_act_grad_from_out(yv.y, agkind);
        a2 *= segsde_act_grad_from_out(yv.z, agkind); a3 *= segsde_act_grad_from_out(yv.w, agkind);
      }
      *reinterpret_cast<float4*>(dst + (size_t)o * ldd) = make_float4(a0, a1, a2, a3);
    };
    // Reflection adjoint: rows 1 / H-2 collect the mirrored padding rows -- whole strips, the general loop below.  Columns 1 /
    // W-2 collect the mirrored padding COLUMNS: one pixel of the first / last strip of a row, through the taps kw = 2 / kw = 0
    // on gradient values the strip has loaded anyway -- handled inline (those strips used to take the general loop and, four
    // strips to a wave, slowed 6 % of all waves by an order of magnitude: 1.66 ms per call against 0.85 ms of traffic).
    const bool border = adjoint && (st.h == 1 || st.h == H - 2);
    const bool colfirst = adjoint && st.w0 == 0, collast = adjoint && st.w0 + CS == W;
    if (!border) {
      // the saved activation outputs of the strip are requested up front (the stores below may alias them as far as the
      // compiler knows: left inside put() every pixel paid a full load latency before its store -- 1.7 ms per call at
      // 16 x 512 x 1024 x 64 against 0.85 ms of traffic)
      float4 yv[CS];
      if (ag) {
#pragma unroll
        for (int o = 0; o < CS; ++o) yv[o] = *reinterpret_cast<const float4*>(ag + (size_t)o * agld);
      }
      float g[3][CS + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yh = st.h + r - 1;
        const bool okh = (unsigned)yh < (unsigned)H;
        const float* gp = dz + (size_t)(st.b * H + (okh ? yh : 0)) * W * lddz;
#pragma unroll
        for (int j = 0; j < CS + 2; ++j) {
          const int yw = st.w0 - 1 + j;
          const bool ok = okh && (unsigned)yw < (unsigned)W;
          g[r][j] = 0.f;
          if (ok) g[r][j] = gp[(size_t)yw * lddz];
        }
      }
#pragma unroll
      for (int o = 0; o < CS; ++o) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float gv = g[kh][o + kw];
            const int t = kh * 3 + kw;
            a0 += gv * w[0][t]; a1 += gv * w[1][t]; a2 += gv * w[2][t]; a3 += gv * w[3][t];
          }
        if ((colfirst && o == 1) || (collast && o == CS - 2)) {
          // pre-image in the mirrored column -1 (window column kw = 2 lands on column 0 = g[.][1]) / W (kw = 0 on column W-1 = g[.][CS])
          const int j = o == 1 ? 1 : CS, kw = o == 1 ? 2 : 0;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float gv = g[kh][j];
            const int t = kh * 3 + kw;
            a0 += gv * w[0][t]; a1 += gv * w[1][t]; a2 += gv * w[2][t]; a3 += gv * w[3][t];
          }
        }
        if (ag) {
          a0 *= segsde_act_grad_from_out(yv[o].x, agkind); a1 *= segsde_act_grad_from_out(yv[o].y, agkind);
          a2 *= segsde_act_grad_from_out(yv[o].z, agkind); a3 *= segsde_act_grad_from_out(yv[o].w, agkind);
        }
        *reinterpret_cast<float4*>(dst + (size_t)o * ldd) = make_float4(a0, a1, a2, a3);
      }
    } else {
      // strips that touch row 1 / H-2 or column 1 / W-2: pixels also collect the mirrored padding cells
      for (int o = 0; o < CS; ++o) {
        const int hq = st.h, wq = st.w0 + o;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int hp[3], wpp[3], nh = 0, nw = 0;
        hp[nh++] = hq; wpp[nw++] = wq;
        if (hq == 1) hp[nh++] = -1;
        if (hq == H - 2) hp[nh++] = H;
        if (wq == 1) wpp[nw++] = -1;
        if (wq == W - 2) wpp[nw++] = W;
        for (int a = 0; a < nh; ++a)
          for (int bb = 0; bb < nw; ++bb)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int yh = hp[a] + kh - 1;
              if ((unsigned)yh >= (unsigned)H) continue;
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const int yw = wpp[bb] + kw - 1;
                if ((unsigned)yw >= (unsigned)W) continue;
                const float gv = dz[((size_t)(st.b * H + yh) * W + yw) * lddz];
                const int t = kh * 3 + kw;
                a0 += gv * w[0][t]; a1 += gv * w[1][t]; a2 += gv * w[2][t]; a3 += gv * w[3][t];
              }
            }
        put(o, a0, a1, a2, a3);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// "Skinny" 1x1 convolutions: a GEMM side of at most 32 (the 19-class segmentation head, joint_segmentation_depth_decoder.py
// heads): its data-gradient (19 gradient channels in, 64/128/256 out) is HBM-bound (2*19*4 FLOP per 16 output bytes),
// but the MFMA tile pads 19 -> 32 on a scalar gather (9 TFLOP/s).  (A register kernel for the weight-gradient measured
// slower than the MFMA path and was dropped.)  Here C/4 lanes
// own a pixel (one float4 of the wide side each), the narrow side's values are wave-broadcast scalar loads and the
// weights live in registers.
// ---------------------------------------------------------------------------------------------------
// The narrow side must be dense (row pitch == K): a block stages the K values of 64 consecutive pixels -- one contiguous
// span of memory -- into LDS with coalesced 16-byte loads, and the pixel groups then read them as LDS broadcasts.  (Per-pixel
// scalar global loads made the kernels latency-bound: 19 load instructions per pixel per wave.)
constexpr int SKP = 64;   // pixels staged per block pass

__device__ __forceinline__ void skinny_stage(const float* src, long p0, int n, int K, float* sh) {
  const float* base = src + p0 * K;            // p0 % 4 == 0  =>  16-byte aligned whenever src is
  const int total = n * K, t4 = total & ~3;
  for (int e = threadIdx.x * 4; e < t4; e += 1024) *reinterpret_cast<float4*>(sh + e) = *reinterpret_cast<const float4*>(base + e);
  for (int e = t4 + threadIdx.x; e < total; e += 256) sh[e] = base[e];
}

// y[p][c] = sum_k a[p][k] * w[c][k]     (a: [M][K] dense, K <= KB; w: [C][K] row-major)
template <int LP, int KB>
__global__ __launch_bounds__(256) void skinny_nk_kernel(const float* a, int K, const float* w, long M, float* y, int ldy) {
  constexpr int G = 256 / LP;
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);    // [SKP][K]
  const int lane_c = threadIdx.x % LP, slot = threadIdx.x / LP;
  float wr[KB][4];
#pragma unroll
  for (int k = 0; k < KB; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[k][j] = k < K ? w[(long)(4 * lane_c + j) * K + k] : 0.f;
  for (long p0 = (long)blockIdx.x * SKP; p0 < M; p0 += (long)gridDim.x * SKP) {
    const int n = (int)(M - p0 < SKP ? M - p0 : SKP);
    __syncthreads();
    skinny_stage(a, p0, n, K, sh);
    __syncthreads();
    for (int q = slot; q < n; q += G) {
      const float* gp = sh + q * K;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const float g = k < K ? gp[k] : 0.f;
        a0 += g * wr[k][0]; a1 += g * wr[k][1]; a2 += g * wr[k][2]; a3 += g * wr[k][3];
      }
      *reinterpret_cast<float4*>(y + (p0 + q) * ldy + 4 * lane_c) = make_float4(a0, a1, a2, a3);
    }
  }
}

inline int c1_blocks(long npix, int ppb) { long nb = (npix + ppb - 1) / ppb; return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb)); }
}  // namespace

bool segsde_c1_supported(int C, int ldx) { return (C == 64 || C == 128 || C == 256) && (ldx % 4 == 0); }

size_t segsde_c1_wgrad_workspace(int C) { return (size_t)1024 * 9 * C * sizeof(float); }

inline int c1s_blocks(int B, int H, int W, int groups) {
  const long nb = ((long)B * H * (W / CS) + groups - 1) / groups;
  return (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
}
inline bool c1_strips(int B, int H, int W) { return W % CS == 0 && (long)B * H * W < (1L << 31); }

#define C1_DISPATCH(KERNEL, grid, smem, ...)                                                       \
  do {                                                                                             \
    if (C == 64) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);  \
    else if (C == 128) hipLaunchKernelGGL((KERNEL<32>), grid, dim3(256), smem, ST(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);          \
  } while (0)

int segsde_c1_forward(const float* x, int ldx, int B, int H, int W, int C, const float* wpack, const float* bias, int reflect,
                      int act, float* y, int ldy, void* stream) {
  if (c1_strips(B, H, W)) {
    const dim3 grid(c1s_blocks(B, H, W, 256 / (C / 4)));
    C1_DISPATCH(c1s_fwd_kernel, grid, 0, x, ldx, B, H, W, wpack, bias, reflect, act, y, ldy);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(c1_blocks((long)B * H * W, 256 / (C / 4)));
  C1_DISPATCH(c1_fwd_kernel, grid, 0, x, ldx, B, H, W, wpack, bias, reflect, act, y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_c1_dgrad(const float* dz, int lddz, int B, int H, int W, int C, const float* wdpack, int adjoint, float* dx, int lddx,
                    float* dx2, int lddx2, int nsplit, const float* agy, int agld, int agkind, void* stream) {
  if (!dx2) { dx2 = dx; lddx2 = lddx; nsplit = C; }
  if (c1_strips(B, H, W) && H >= 4 && W >= 2 * CS) {
    const dim3 grid(c1s_blocks(B, H, W, 256 / (C / 4)));
    C1_DISPATCH(c1s_dgrad_kernel, grid, 0, dz, lddz, B, H, W, wdpack, adjoint, dx, lddx, dx2, lddx2, nsplit, agy, agld, agkind);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(c1_blocks((long)B * H * W, 256 / (C / 4)));
  C1_DISPATCH(c1_dgrad_kernel, grid, 0, dz, lddz, B, H, W, wdpack, adjoint, dx, lddx, dx2, lddx2, nsplit, agy, agld, agkind);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_c1_wgrad(const float* x, int ldx, int B, int H, int W, int C, const float* dz, int lddz, int reflect, float* dw,
                    float* workspace, void* stream) {
  const int ppb = 256 / (C / 4);
  long nb = ((long)B * H * W + ppb - 1) / ppb;
  const int nblk = (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
  const size_t smem = (size_t)ppb * 9 * C * sizeof(float);
  if (c1_strips(B, H, W)) {
    long ns = ((long)B * H * (W / CS) + ppb - 1) / ppb;
    const int nbs = (int)(ns < 1 ? 1 : (ns > 1024 ? 1024 : ns));
    C1_DISPATCH(c1s_wgrad_kernel, dim3(nbs), smem, x, ldx, B, H, W, dz, lddz, reflect, workspace);
    SEGSDE_CHECK_LAUNCH();
    hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((9 * C + 15) / 16), dim3(256), 0, ST(stream), (const float*)workspace, nbs,
                       C, dw);
    SEGSDE_CHECK_LAUNCH();
    return 0;
  }
  C1_DISPATCH(c1_wgrad_kernel, dim3(nblk), smem, x, ldx, B, H, W, dz, lddz, reflect, workspace);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((9 * C + 15) / 16), dim3(256), 0, ST(stream), (const float*)workspace, nblk,
                     C, dw);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

bool segsde_skinny_supported(int K, int C) { return K >= 1 && K <= 32 && (C == 64 || C == 128 || C == 256); }

#define SK_DISPATCH(KERNEL, grid, smem, ...)                                                                    \
  do {                                                                                                          \
    if (K <= 20) {                                                                                              \
      if (C == 64) hipLaunchKernelGGL((KERNEL<16, 20>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);         \
      else if (C == 128) hipLaunchKernelGGL((KERNEL<32, 20>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);   \
      else hipLaunchKernelGGL((KERNEL<64, 20>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);                 \
    } else {                                                                                                    \
      if (C == 64) hipLaunchKernelGGL((KERNEL<16, 32>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);         \
      else if (C == 128) hipLaunchKernelGGL((KERNEL<32, 32>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);   \
      else hipLaunchKernelGGL((KERNEL<64, 32>), grid, dim3(256), smem, ST(stream), __VA_ARGS__);                 \
    }                                                                                                           \
  } while (0)

int segsde_skinny_nk(const float* a, int K, const float* w, long M, int C, float* y, int ldy, void* stream) {
  long nb = (M + SKP - 1) / SKP;
  const dim3 grid((unsigned)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb)));
  SK_DISPATCH(skinny_nk_kernel, grid, (size_t)SKP * 32 * sizeof(float), a, K, w, M, y, ldy);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
