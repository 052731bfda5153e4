// Winograd F(2x2,3x3) for the stride-1 3x3 convolutions with many channels (round 4; models/resnet_encoder.py:90-101: conv2 of
// every bottleneck).  A 3x3 / pad 1 convolution over a 2x2 block of output pixels reads a 4x4 input patch d; with
//     V = B^T d B   (per input channel),   U = G g G^T   (per filter),   m = sum_c V_c * U_c   (elementwise in the 4x4 plane),
//     Y = A^T m A   (the 2x2 outputs)
// the 36 multiply-adds per block, channel and filter become 16: 2.25x less work for the fp32 matrix pipe, which is what bounds
// these layers.  Each of the 16 plane positions is an ordinary GEMM over the channels, [T tiles x C] x
// [C x Cout]: they run as ONE grouped launch of conv_igemm_kernel (1x1 convolution over a 16-"image" tensor, weight base
// advancing with the image index).  The transforms are the HBM-side passes in this file (additions only, no multiplies on
// the input / output side):
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// (Lavin & Gray, "Fast algorithms for convolutional neural networks", 2016, the F(2x2,3x3) matrices.)
#include "segsde_common.h"
#include "winograd.h"

namespace {

struct F4 { float x, y, z, w; };
__device__ __forceinline__ F4 ld4(const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); return F4{t.x, t.y, t.z, t.w}; }
__device__ __forceinline__ void st4(float* p, const F4& v) { *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ F4 operator+(const F4& a, const F4& b) { return F4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ F4 operator-(const F4& a, const F4& b) { return F4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

__device__ __forceinline__ int refl_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Tiles of a dilated window (dilation d, padding d: layer4 of the dilated ResNet): the pixels with (h mod d, w mod d) = (a, c)
// form a (H/d) x (W/d) image on which the convolution is an ordinary 3x3 / pad 1 one, so every tile lives on one of the d*d
// sub-lattices: tile t = ((((b * d + a) * d + c) * H2 + i) * W2 + j), H2 = H / (2 d), patch pixel (r, s) = image pixel
// (a + d (2 i - 1 + r), c + d (2 j - 1 + s)).  d = 1: the plain case.
struct TileAt { int b, a, c, i, j; };
__device__ __forceinline__ TileAt tile_at(long t, int d, int H2, int W2) {
  TileAt q;
  q.j = (int)(t % W2); t /= W2;
  q.i = (int)(t % H2); t /= H2;
  q.c = (int)(t % d); t /= d;
  q.a = (int)(t % d); q.b = (int)(t / d);
  return q;
}

// one thread: one tile x four channels.  Consecutive threads take consecutive channel quads of the same tile (16-byte loads /
// stores, whole 64-byte-or-longer runs per pixel); the 4x4 patches of neighbouring tiles overlap by two pixels (L1 / L2).
// Two sources: channels [0, C0) come from x0, [C0, C) from x1 (the decoder's concat, models/depth_decoder.py:93-101).
__global__ __launch_bounds__(256) void wino_in_kernel(const float* x0, int ld0, const float* x1, int ld1, int C0, int B, int H, int W,
                                                      int C, int d, int reflect, long Tp, float* V) {
  // Tp >= T: rows per position plane, T rounded up to the GEMM's 128-row tiles (the rows past T are written as zeros)
  const int CQ = C >> 2, H2 = H / (2 * d), W2 = W / (2 * d), Hs = H / d, Ws = W / d;
  const long T = (long)B * d * d * H2 * W2, total = Tp * CQ;
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= total) return;
  const int cq = (int)(e % CQ);
  const long t = e / CQ;
  if (t >= T) {
    const F4 z{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) st4(V + k * Tp * C + t * C + 4 * cq, z);
    return;
  }
  const TileAt q = tile_at(t, d, H2, W2);
  const bool s0 = 4 * cq < C0;
  const float* src = s0 ? x0 + 4 * cq : x1 + (4 * cq - C0);
  const int ld = s0 ? ld0 : ld1;
  F4 p[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int hh = 2 * q.i - 1 + r;                      // row on the sub-lattice
    const bool hv = reflect || (unsigned)hh < (unsigned)Hs;
    hh = refl_idx(hh, Hs);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int ww = 2 * q.j - 1 + c;
      const bool ok = hv && (reflect || (unsigned)ww < (unsigned)Ws);
      ww = refl_idx(ww, Ws);
      p[r][c] = ok ? ld4(src + ((long)(q.b * H + q.a + d * hh) * W + q.c + d * ww) * ld) : F4{0.f, 0.f, 0.f, 0.f};
    }
  }
  F4 u[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {          // B^T d
    u[0][c] = p[0][c] - p[2][c]; u[1][c] = p[1][c] + p[2][c]; u[2][c] = p[2][c] - p[1][c]; u[3][c] = p[1][c] - p[3][c];
  }
  float* out = V + t * C + 4 * cq;
  const long plane = Tp * C;
#pragma unroll
  for (int r = 0; r < 4; ++r) {          // (.) B
    st4(out + (4 * r + 0) * plane, u[r][0] - u[r][2]);
    st4(out + (4 * r + 1) * plane, u[r][1] + u[r][2]);
    st4(out + (4 * r + 2) * plane, u[r][2] - u[r][1]);
    st4(out + (4 * r + 3) * plane, u[r][1] - u[r][3]);
  }
}

// block: 64 channels (16 lanes of 4) x 16 tile lanes; grid (row blocks, Co / 64).  Each lane walks its tiles, stores the four
// output pixels of each and -- STATS -- keeps double-precision column sums of what it stored; the lanes of a block are folded
// in a fixed order into one partial row per block (same [rows][2][C] format as the implicit-GEMM epilogue's partials).
__device__ __forceinline__ F4 act4(const F4& v, const F4& b, int act) {
  F4 r = v + b;
  if (act == SEGSDE_ACT_ELU) {           // the implicit-GEMM epilogue's expression
    r.x = r.x > 0.f ? r.x : __expf(fminf(r.x, 0.f)) - 1.f; r.y = r.y > 0.f ? r.y : __expf(fminf(r.y, 0.f)) - 1.f;
    r.z = r.z > 0.f ? r.z : __expf(fminf(r.z, 0.f)) - 1.f; r.w = r.w > 0.f ? r.w : __expf(fminf(r.w, 0.f)) - 1.f;
  } else if (act != SEGSDE_ACT_NONE) {
    r.x = segsde_act(r.x, act); r.y = segsde_act(r.y, act); r.z = segsde_act(r.z, act); r.w = segsde_act(r.w, act);
  }
  return r;
}

template <bool STATS>
__global__ __launch_bounds__(256) void wino_out_kernel(const float* M, int B, int H, int W, int Co, int d, const float* bias, int act,
                                                       long Tp, float* y, int ldy, double* part) {
  SEGSDE_SMEM;
  double* sh = reinterpret_cast<double*>(segsde_smem);     // [2][16][64]
  const int H2 = H / (2 * d), W2 = W / (2 * d);
  const long T = (long)B * d * d * H2 * W2, plane = Tp * Co;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + 4 * tx;
  const long per = (T + gridDim.x - 1) / gridDim.x;
  const long t0 = blockIdx.x * per;
  long t1 = t0 + per; if (t1 > T) t1 = T;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < Co) {
    const F4 bv = bias ? ld4(bias + c) : F4{0.f, 0.f, 0.f, 0.f};
    const bool post = bias != nullptr || act != SEGSDE_ACT_NONE;
    for (long t = t0 + ty; t < t1; t += 16) {
      const TileAt tl = tile_at(t, d, H2, W2);
      const float* mp = M + t * Co + c;
      F4 a0[4], a1[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {   // A^T m
        const F4 m0 = ld4(mp + (0 + cc) * plane), m1 = ld4(mp + (4 + cc) * plane), m2 = ld4(mp + (8 + cc) * plane),
                 m3 = ld4(mp + (12 + cc) * plane);
        a0[cc] = (m0 + m1) + m2; a1[cc] = (m1 - m2) - m3;
      }
      F4 o[4];                           // (.) A: (0,0) (0,1) (1,0) (1,1)
      o[0] = (a0[0] + a0[1]) + a0[2]; o[1] = (a0[1] - a0[2]) - a0[3];
      o[2] = (a1[0] + a1[1]) + a1[2]; o[3] = (a1[1] - a1[2]) - a1[3];
      if (post) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = act4(o[k], bv, act);
      }
      float* yp = y + ((long)(tl.b * H + tl.a + d * 2 * tl.i) * W + tl.c + d * 2 * tl.j) * ldy + c;
      const long dn = (long)d * ldy, dw = (long)d * W * ldy;
      st4(yp, o[0]); st4(yp + dn, o[1]); st4(yp + dw, o[2]); st4(yp + dw + dn, o[3]);
      if (STATS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double v0 = (double)o[k].x, v1 = (double)o[k].y, v2 = (double)o[k].z, v3 = (double)o[k].w;
          s[0] += v0; q[0] += v0 * v0; s[1] += v1; q[1] += v1 * v1; s[2] += v2; q[2] += v2 * v2; s[3] += v3; q[3] += v3 * v3;
        }
      }
    }
  }
  if (!STATS) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) { sh[ty * 64 + 4 * tx + k] = s[k]; sh[1024 + ty * 64 + 4 * tx + k] = q[k]; }
  __syncthreads();
  const int cc = threadIdx.x;
  if (cc < 64 && blockIdx.y * 64 + cc < Co) {
    double a = 0.0, b2 = 0.0;
    for (int l = 0; l < 16; ++l) { a += sh[l * 64 + cc]; b2 += sh[1024 + l * 64 + cc]; }
    part[((long)blockIdx.x * 2 + 0) * Co + blockIdx.y * 64 + cc] = a;
    part[((long)blockIdx.x * 2 + 1) * Co + blockIdx.y * 64 + cc] = b2;
  }
}

// The output-side transform of the WEIGHT gradient: dM = A dY A^T, a 2x2 block of the output gradient -> its 4x4 plane
// (A = [1 0; 1 1; 1 -1; 0 -1]); with it  dU_p = sum over tiles of V_p^T dM_p  (sixteen GEMMs over the tiles, the weight-gradient
// kernel's own form) and dW = G^T dU G.  One thread: one tile x four channels, like the input transform.
__global__ __launch_bounds__(256) void wino_grad_kernel(const float* dy, int ld, int B, int H, int W, int C, int d, long Tp, float* dM) {
  const int CQ = C >> 2, H2 = H / (2 * d), W2 = W / (2 * d);
  const long T = (long)B * d * d * H2 * W2, total = Tp * CQ;
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= total) return;
  const int cq = (int)(e % CQ);
  const long t = e / CQ;
  if (t >= T) {
    const F4 z0{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) st4(dM + k * Tp * C + t * C + 4 * cq, z0);
    return;
  }
  const TileAt q = tile_at(t, d, H2, W2);
  const float* src = dy + ((long)(q.b * H + q.a + d * 2 * q.i) * W + q.c + d * 2 * q.j) * ld + 4 * cq;
  const long dn = (long)d * ld, dw = (long)d * W * ld;
  const F4 g00 = ld4(src), g01 = ld4(src + dn), g10 = ld4(src + dw), g11 = ld4(src + dw + dn);
  const F4 z{0.f, 0.f, 0.f, 0.f};
  F4 r[4][2];                            // A dY
  r[0][0] = g00; r[0][1] = g01; r[1][0] = g00 + g10; r[1][1] = g01 + g11; r[2][0] = g00 - g10; r[2][1] = g01 - g11;
  r[3][0] = z - g10; r[3][1] = z - g11;
  float* out = dM + t * C + 4 * cq;
  const long plane = Tp * C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // (.) A^T
    st4(out + (4 * k + 0) * plane, r[k][0]);
    st4(out + (4 * k + 1) * plane, r[k][0] + r[k][1]);
    st4(out + (4 * k + 2) * plane, r[k][0] - r[k][1]);
    st4(out + (4 * k + 3) * plane, z - r[k][1]);
  }
}

// part [16 * s][C][Co] (split slabs of the sixteen position GEMMs, s per position) -> dW [Co][C][3][3] = G^T dU G with
// dU_p = the sum of position p's slabs in slab order (deterministic).  A block = 16 (c, co) pairs (co fastest) x the sixteen
// positions: thread (p, pair) sums the s slabs of ONE position (four independent partial sums: the loads of a thread do not wait
// for each other), the positions meet in LDS, threads p < 3 finish output row p.  (Round 4: 64 pairs x 4 position rows per block,
// each thread 4 s loads -- a 128 x 64 weight was 128 blocks on 256 CUs and 36 us for 67 MB; the one-kernel weight gradient of
// round 5 runs this 77 times per step.)
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* part, int s, int C, int Co, float* dw) {
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);   // [16 p][16 pairs]
  const long total = (long)C * Co;
  const int el = threadIdx.x & 15, pp = threadIdx.x >> 4;
  const long e = blockIdx.x * 16L + el;
  if (e < total) {
    const float* src = part + (long)pp * s * total + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int z = 0;
    for (; z + 3 < s; z += 4) {
      a0 += src[(long)z * total]; a1 += src[(long)(z + 1) * total]; a2 += src[(long)(z + 2) * total]; a3 += src[(long)(z + 3) * total];
    }
    for (; z < s; ++z) a0 += src[(long)z * total];
    sh[pp * 16 + el] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (e >= total || pp >= 3) return;
  const int q = pp;
  const int co = (int)(e % Co), c = (int)(e / Co);
  float t[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {          // row q of G^T dU
    const float u0 = sh[(0 * 4 + v) * 16 + el], u1 = sh[(1 * 4 + v) * 16 + el], u2 = sh[(2 * 4 + v) * 16 + el], u3 = sh[(3 * 4 + v) * 16 + el];
    t[v] = q == 0 ? u0 + 0.5f * (u1 + u2) : (q == 1 ? 0.5f * (u1 - u2) : 0.5f * (u1 + u2) + u3);
  }
  float* out = dw + ((long)co * C + c) * 9 + 3 * q;   // (.) G
  out[0] = t[0] + 0.5f * (t[1] + t[2]);
  out[1] = 0.5f * (t[1] - t[2]);
  out[2] = 0.5f * (t[1] + t[2]) + t[3];
}

// thread (a, b), b fastest: forward a = o, b = i reads w[o][i][3][3] (36 contiguous bytes) and writes U[p][o][i]; the
// data-gradient pack a = i, b = o reads the same nine values with the taps flipped and writes U'[p][i][o] -- coalesced writes
// in both, the strided reads of the second variant hit L2 (the whole weight is 2.4 MB)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* w, int O, int I, int tf, float* U) {
  const int A = tf ? I : O, Bn = tf ? O : I;
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)A * Bn) return;
  const int b = (int)(e % Bn), a = (int)(e / Bn);
  const int o = tf ? b : a, i = tf ? a : b;
  const float* g = w + ((long)o * I + i) * 9;
  float k[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) k[r][c] = tf ? g[(2 - r) * 3 + (2 - c)] : g[r * 3 + c];
  float t[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {          // G g
    t[0][c] = k[0][c];
    t[1][c] = 0.5f * ((k[0][c] + k[1][c]) + k[2][c]);
    t[2][c] = 0.5f * ((k[0][c] - k[1][c]) + k[2][c]);
    t[3][c] = k[2][c];
  }
  const long plane = (long)A * Bn;
  float* out = U + e;
#pragma unroll
  for (int r = 0; r < 4; ++r) {          // (.) G^T
    out[(4 * r + 0) * plane] = t[r][0];
    out[(4 * r + 1) * plane] = 0.5f * ((t[r][0] + t[r][1]) + t[r][2]);
    out[(4 * r + 2) * plane] = 0.5f * ((t[r][0] - t[r][1]) + t[r][2]);
    out[(4 * r + 3) * plane] = t[r][2];
  }
}

// all the Winograd weight packs of a training step in ONE launch (models/layers.weight_pack_scope): job j owns the blocks
// [block0_j, block0_{j+1}); its first half transforms the forward pack, the second half the data-gradient pack
__global__ __launch_bounds__(256) void wino_weight_multi_kernel(const segsde_wino_job* jobs, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block0) ++j;
  const segsde_wino_job jb = jobs[j];
  const long per = (long)jb.O * jb.I;
  const int nblk = (int)((per + 255) / 256);
  const int lb = (int)blockIdx.x - jb.block0, tf = lb >= nblk ? 1 : 0;
  const long e = (long)(lb - tf * nblk) * 256 + threadIdx.x;
  if (e >= per) return;
  // reserved & 1: the [16][K][N] layout of the one-kernel route (winograd_fused.hip): forward [16][I][O], data-gradient [16][O][I]
  const int tr = tf ^ (jb.reserved & 1);
  const int A = tr ? jb.I : jb.O, Bn = tr ? jb.O : jb.I;
  const int b = (int)(e % Bn), a = (int)(e / Bn);
  const int o = tr ? b : a, i = tr ? a : b;
  const float* g = jb.w + ((long)o * jb.I + i) * 9;
  float k[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) k[r][c] = tf ? g[(2 - r) * 3 + (2 - c)] : g[r * 3 + c];
  float t[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[0][c] = k[0][c];
    t[1][c] = 0.5f * ((k[0][c] + k[1][c]) + k[2][c]);
    t[2][c] = 0.5f * ((k[0][c] - k[1][c]) + k[2][c]);
    t[3][c] = k[2][c];
  }
  const long plane = (long)A * Bn;
  float* base = tf ? jb.u_dgrad : jb.u_fwd;
  const bool blocked = (jb.reserved & 2) && (Bn % 64 == 0);      // the one-kernel route's blocked layout (winograd_fused.hip)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v0 = t[r][0], v1 = 0.5f * ((t[r][0] + t[r][1]) + t[r][2]), v2 = 0.5f * ((t[r][0] - t[r][1]) + t[r][2]), v3 = t[r][2];
    if (blocked) {
      float* o = base + (((((long)r * A + a) * (Bn >> 6) + (b >> 6)) * 32 + (b & 31)) * 4) * 2 + ((b >> 5) & 1);
      o[0] = v0; o[2] = v1; o[4] = v2; o[6] = v3;
    } else {
      float* out = base + e;
      out[(4 * r + 0) * plane] = v0; out[(4 * r + 1) * plane] = v1; out[(4 * r + 2) * plane] = v2; out[(4 * r + 3) * plane] = v3;
    }
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

long segsde_wino_stats_rows(long T) { long nb = T / 64; return nb < 1 ? 1 : (nb > 512 ? 512 : nb); }

long segsde_wino_rows(long T) { return (T + 127) / 128 * 128; }

int segsde_wino_input(const float* x0, int ld0, const float* x1, int ld1, int C0, int B, int H, int W, int C, int dil, int reflect,
                      float* V, void* stream) {
  const long Tp = segsde_wino_rows((long)B * (H / 2) * (W / 2)), total = Tp * (C / 4);
  hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), x0, ld0, x1 ? x1 : x0, ld1, C0, B,
                     H, W, C, dil, reflect, Tp, V);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_wino_output(const float* M, int B, int H, int W, int Co, int dil, const float* bias, int act, float* y, int ldy,
                       double* part, void* stream) {
  const long T = (long)B * (H / 2) * (W / 2), Tp = segsde_wino_rows(T);
  const dim3 grid((unsigned)segsde_wino_stats_rows(T), (unsigned)((Co + 63) / 64));
  if (part)
    hipLaunchKernelGGL(wino_out_kernel<true>, grid, dim3(256), 2 * 16 * 64 * sizeof(double), ST(stream), M, B, H, W, Co, dil, bias, act,
                       Tp, y, ldy, part);
  else
    hipLaunchKernelGGL(wino_out_kernel<false>, grid, dim3(256), 0, ST(stream), M, B, H, W, Co, dil, bias, act, Tp, y, ldy, part);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_wino_grad(const float* dy, int ld, int B, int H, int W, int C, int dil, float* dM, void* stream) {
  const long Tp = segsde_wino_rows((long)B * (H / 2) * (W / 2)), total = Tp * (C / 4);
  hipLaunchKernelGGL(wino_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), dy, ld, B, H, W, C, dil, Tp, dM);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_wino_wgrad_finish(const float* part, int s, int C, int Co, float* dw_oihw, void* stream) {
  const long total = (long)C * Co;
  hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 16 * 16 * sizeof(float), ST(stream), part, s, C, Co, dw_oihw);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_wino_weights_multi(const segsde_wino_job* jobs_device, int njobs, int total_blocks, void* stream) {
  hipLaunchKernelGGL(wino_weight_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ST(stream), jobs_device, njobs);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

int segsde_wino_weights(const float* w_oihw, int O, int I, int transpose_flip, float* U, void* stream) {
  const long total = (long)O * I;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), w_oihw, O, I, transpose_flip, U);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
