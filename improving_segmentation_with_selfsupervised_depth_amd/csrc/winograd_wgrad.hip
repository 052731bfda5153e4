// Weight gradient of the 3x3 / stride 1 / padding 1 convolutions as Winograd F(2x2,3x3) with BOTH operand transforms inside the
// GEMM kernel (round 5).  For a 2x2 block of outputs ("tile", 4x4 input patch d, output gradient dY)
//
//     dU_p = sum over tiles  V_p (x) dM_p,     V = B^T d B (per input channel),   dM = A dY A^T (per filter),   dW = G^T dU G
//
// is sixteen GEMMs [Cin x tiles] x [tiles x Cout] with 16 instead of 36 multiply-adds per tile, channel and filter.  The grouped
// route of winograd.hip materialises V and dM (4x the activation each) around one launch of the direct weight-gradient kernel;
// here neither exists: the raw input patch and the raw output gradient of a block of tiles are staged in LDS by the loads
// themselves (buffer_load ... lds, pixel-major as they lie in memory: the lanes of an MFMA operand are CHANNELS here, so the
// natural [pixel][channel] image is conflict-free and needs no transposing writes), and every lane forms its operands from 8 + 8
// LDS reads with 16 vector instructions per 8 MFMAs.  The layers: ResNet layer1 / layer2 conv2 (64 / 128 channels,
// models/resnet_encoder.py:90-101 via torchvision's blocks), layer3 conv2 (256), and the decoders' Conv3x3 incl. the two-source
// [upsample(x) | skip] layers with mirrored padding (models/depth_decoder.py:88-101, models/monodepth_layers.py:127-142) --
// nearest upsampling, concat and reflection are index arithmetic of the patch loader.
//
// Workgroup = 32 input channels x 64 filters x all sixteen positions (wave w owns transform row w: positions 4w .. 4w+3, 128
// accumulators), looping over its share of the tile blocks (split-K); partial sums go to slabs [16 S][Cin][Cout] that
// wino_wgrad_finish_kernel (winograd.hip) folds in slab order and transforms back: deterministic, no atomics.
#include <stdlib.h>
#include "segsde_common.h"
#include "winograd.h"

namespace {
#define ST(s) static_cast<hipStream_t>(s)
constexpr int WT_W = 8;                       // tiles per block row (16 output pixels)
constexpr int PW = 2 * WT_W + 2;              // patch width: 18 pixels
constexpr int KC = 32, NC = 64;               // input channels / filters of a workgroup

struct WgradP {
  const float* x0; const float* x1; int ld0, ld1, C0, up0;   // virtual input [up2x?(x0) | x1]
  const float* dy; int lddy;
  int B, H, W, C, Co, reflect;
  int nbh, nbw, nblk, S, ncol;
  float* part;
};

template <int WT_H>
struct Geo {
  static constexpr int NT = WT_H * WT_W;                  // tiles per block
  static constexpr int PH = 2 * WT_H + 2;                 // patch rows
  static constexpr int XPX = PH * PW;                     // patch pixels
  static constexpr int XI = ((XPX * (KC / 4) + 63) / 64 + 3) / 4;   // patch loads per wave (64 lanes x 16 bytes = 8 pixels each)
  static constexpr int XSLOTS = XI * 4 * 8;               // pixel slots of the patch region in LDS
  static constexpr int YPX = 2 * WT_H * 2 * WT_W;         // output-gradient pixels of the block
  static constexpr int YI = YPX * (NC / 4) / 64 / 4;      // dY loads per wave (4 pixels each)
  static constexpr int XF = XSLOTS * KC;                  // floats of the patch region
  static constexpr int STAGE = XF + YPX * NC;             // floats of one stage
};

// UPSKIP (launch-level): the launch has a nearest-upsampled source that carries at least half of the input channels; workgroups
// whose 32 channels come from it skip the Winograd positions that are identically zero there (see `compute`).  Only these
// launches carry the wave-uniform branches around the two MFMAs -- they cost a workgroup of the OTHER source 2-3 %
// (probe_r06_upskip_wgrad.log), so launches where that source dominates keep the plain kernel.
template <int WT_H, bool DB, int MINB, bool UPSKIP>
__global__ __launch_bounds__(256, MINB) void wino_wgrad_fused_kernel(WgradP p) {
  using G = Geo<WT_H>;
  SEGSDE_SMEM;
  float* lds = reinterpret_cast<float*>(segsde_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int i = lane & 31, kk = lane >> 5;
  int logical = segsde_xcd_remap(blockIdx.x, gridDim.x);
  const int col = logical % p.ncol, split = logical / p.ncol;
  const int ncj = p.Co / NC;
  const int cin0 = (col / ncj) * KC, co0 = (col % ncj) * NC;
  const int q_begin = (int)((long)split * p.nblk / p.S), q_end = (int)((long)(split + 1) * p.nblk / p.S);
  // the source this workgroup's 32 input channels come from
  const bool s0 = cin0 < p.C0;
  const float* xs = s0 ? p.x0 : p.x1;
  const int ldx = s0 ? p.ld0 : p.ld1, cb = s0 ? cin0 : cin0 - p.C0, sh = (s0 && p.up0) ? 1 : 0;
  const int Hs = p.H >> sh, Ws = p.W >> sh;
  const bool skip2 = UPSKIP && sh;                      // this workgroup's input channels are nearest-upsampled (see compute)
  const unsigned lds0 = segsde_lds_addr(lds);

  // rows of the patch that transform row `wave` of B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] combines (winograd_fused.hip)
  const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int a2 = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sgn = wave == 1 ? 1.f : -1.f;
  // rows of dY that row `wave` of A = [1 0; 1 1; 1 -1; 0 -1] combines: r = y[p0] + c1 * y[p1]; the sign of row 3 (and of column 3
  // below) is applied once, to the accumulators, in the epilogue
  const int p0 = wave == 3 ? 1 : 0, p1 = wave == 3 ? 0 : 1;
  const float c1 = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
  const int xl = 2 * kk * KC + i;                        // lane part of the patch address: tile 2s + kk is two pixels to the right
  const int yl = 2 * kk * NC + i;
  const int xo1 = a1 * PW * KC, xo2 = a2 * PW * KC, yo0 = p0 * 2 * WT_W * NC, yo1 = p1 * 2 * WT_W * NC;

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

  // Per-lane byte offsets of a block's tile loads relative to the block's own origin: the same for every INTERIOR block (patch
  // and gradient pixels all inside the image: no padding, no mirroring, no tile past the edge), whose loads then cost no vector
  // instructions at all -- the block's origin goes into the (scalar) resource base.  PMC of the first version: 6.0 vector
  // instructions per MFMA over the kernel against 3.3 inside the loop; the difference was this index arithmetic.
  unsigned relx[G::XI], rely[G::YI];
#pragma unroll
  for (int k = 0; k < G::XI; ++k) {
    const int e = (k * 4 + wv) * 64 + lane;
    const int px = e >> 3, cq = e & 7;
    const int pr = px / PW, pc = px - pr * PW;
    // source pixel of patch pixel (pr, pc) relative to the patch origin's source pixel (origin coordinates are odd: 2 m - 1)
    relx[k] = px < G::XPX ? (unsigned)((((pr + sh) >> sh) * Ws + ((pc + sh) >> sh)) * ldx + 4 * cq) * 4u : SEGSDE_OOB;
  }
#pragma unroll
  for (int k = 0; k < G::YI; ++k) {
    const int e = (k * 4 + wv) * 64 + lane;
    const int px = e >> 4, cq = e & 15;
    rely[k] = (unsigned)(((px >> 4) * p.W + (px & 15)) * p.lddy + 4 * cq) * 4u;
  }

  // tile loads of block q into the stage at byte offset `stage`
  auto issue = [&](int q, unsigned stage) {
    const int bw = q % p.nbw; int t = q / p.nbw;
    const int bh = t % p.nbh, b = t / p.nbh;
    const int h_top = 2 * bh * WT_H - 1, w_left = 2 * bw * WT_W - 1;
    if (h_top >= 0 && w_left >= 0 && h_top + G::PH <= p.H && w_left + PW <= p.W) {
      const segsde_rsrc rxi = segsde_make_rsrc(xs + (size_t)b * Hs * Ws * ldx + cb + ((size_t)(h_top >> sh) * Ws + (w_left >> sh)) * ldx);
#pragma unroll
      for (int k = 0; k < G::XI; ++k) segsde_buffer_load4_lds(rxi, relx[k], 0u, lds0 + stage + (unsigned)(k * 4 + wv) * 1024u);
      const segsde_rsrc ryi = segsde_make_rsrc(p.dy + (size_t)b * p.H * p.W * p.lddy + co0 + ((size_t)(h_top + 1) * p.W + (w_left + 1)) * p.lddy);
#pragma unroll
      for (int k = 0; k < G::YI; ++k)
        segsde_buffer_load4_lds(ryi, rely[k], 0u, lds0 + stage + (unsigned)(G::XF * 4) + (unsigned)(k * 4 + wv) * 1024u);
      return;
    }
    const segsde_rsrc rx = segsde_make_rsrc(xs + (size_t)b * Hs * Ws * ldx + cb);
#pragma unroll
    for (int k = 0; k < G::XI; ++k) {
      const int inst = k * 4 + wv;
      const int e = inst * 64 + lane;
      const int px = e >> 3, cq = e & 7;
      const int pr = px / PW, pc = px - pr * PW;
      int hh = h_top + pr, ww = w_left + pc;
      bool ok = px < G::XPX;
      if (p.reflect) {      // ReflectionPad2d(1); pixels of tiles past the image: any valid address (their dY is zero)
        hh = hh < 0 ? -hh : (hh >= p.H ? 2 * p.H - 2 - hh : hh); ww = ww < 0 ? -ww : (ww >= p.W ? 2 * p.W - 2 - ww : ww);
        hh = hh < 0 ? 0 : hh; ww = ww < 0 ? 0 : ww;
      }
      ok = ok && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)(((hh >> sh) * Ws + (ww >> sh)) * ldx + 4 * cq) * 4u : SEGSDE_OOB;
      segsde_buffer_load4_lds(rx, voff, 0u, lds0 + stage + (unsigned)inst * 1024u);
    }
    const segsde_rsrc ry = segsde_make_rsrc(p.dy + (size_t)b * p.H * p.W * p.lddy + co0);
#pragma unroll
    for (int k = 0; k < G::YI; ++k) {
      const int inst = k * 4 + wv;
      const int e = inst * 64 + lane;
      const int px = e >> 4, cq = e & 15;
      const int hh = 2 * bh * WT_H + (px >> 4), ww = 2 * bw * WT_W + (px & 15);
      const unsigned voff = (hh < p.H && ww < p.W) ? (unsigned)((hh * p.W + ww) * p.lddy + 4 * cq) * 4u : SEGSDE_OOB;
      segsde_buffer_load4_lds(ry, voff, 0u, lds0 + stage + (unsigned)(G::XF * 4) + (unsigned)inst * 1024u);
    }
  };

  // the sixteen position GEMMs over the tiles of one staged block.  The operand transforms stay SCALAR fp32 instructions: this
  // file is compiled with -fno-slp-vectorize (__graft_entry__.py) -- left to itself the compiler packs them into v_pk_fma_f32 /
  // v_pk_add_f32 plus the moves that re-pair the ds_read2 results (26 vector instructions per 8 MFMAs), and a packed fp32
  // instruction beside MFMAs costs ~22 cycles more than the two scalar ones it replaces (MI355X_MICROARCH.md, MFMA microbenchmarks;
  // measured here: hand-packed 8 + 4 per step 104-107 TFLOP/s executed, scalar 16 + 4: see profiles/probe_r05_*)
  auto compute = [&](const float* X) {
    const float* Y = X + G::XF;
    const float* xp = X + xl;
    const float* yp = Y + yl;
    constexpr int NS = G::NT / 2;
    // Software-pipelined step (round 6, like wino_fused_kernel): the operands of step s + 1 are formed during step s, and every
    // LDS read / vector instruction of a step sits in the shadow of one of its eight MFMAs -- slots 0..3 issue the sixteen reads
    // of the next step (four each), slots 4..7 the sixteen additions that turn them into its operands.  Before, the sixteen reads
    // came in one bunch and the twenty additions stood in front of the MFMAs that use them: a wave alone on its SIMD (the sibling
    // workgroup waiting for its stage) could not keep the matrix pipe fed.
    float rv[8], yv[8], An[2][4], Bn[2][2][4], e1k = 0.f;
    auto rd_x = [&](int s, int h) {            // raw pixels of step s: patch columns 2 h, 2 h + 1 of both rows
      const int th = s >> 2, tw = 2 * (s & 3);
      const float* xq = xp + ((2 * th) * PW + 2 * tw) * KC;
#pragma unroll
      for (int bb = 2 * h; bb < 2 * h + 2; ++bb) { rv[bb] = xq[xo1 + bb * KC]; rv[4 + bb] = xq[xo2 + bb * KC]; }
    };
    auto rd_y = [&](int s, int nb) {           // output gradients of step s, filter half nb
      const int th = s >> 2, tw = 2 * (s & 3);
      const float* yq = yp + ((2 * th) * 2 * WT_W + 2 * tw) * NC;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) { yv[4 * nb + bb] = yq[yo0 + bb * NC + 32 * nb]; yv[4 * nb + 2 + bb] = yq[yo1 + bb * NC + 32 * nb]; }
    };
    auto form_a = [&](int q, int h) {
      if (h == 0) {
        const float e0 = __builtin_fmaf(rv[4], sgn, rv[0]), e1 = __builtin_fmaf(rv[5], sgn, rv[1]), e2 = __builtin_fmaf(rv[6], sgn, rv[2]);
        An[q][0] = e0 - e2; An[q][1] = e1 + e2; An[q][2] = e2 - e1; e1k = e1;
      } else {
        const float e3 = __builtin_fmaf(rv[7], sgn, rv[3]);
        An[q][3] = e1k - e3;
      }
    };
    auto form_b = [&](int q, int nb, int h) {
      if (h == 0) {
        Bn[q][nb][0] = __builtin_fmaf(yv[4 * nb + 2], c1, yv[4 * nb]); Bn[q][nb][3] = __builtin_fmaf(yv[4 * nb + 3], c1, yv[4 * nb + 1]);
      } else {
        Bn[q][nb][1] = Bn[q][nb][0] + Bn[q][nb][3]; Bn[q][nb][2] = Bn[q][nb][0] - Bn[q][nb][3];
      }
    };
    // Input channels from the NEAREST-UPSAMPLED source: rows 2 t / 2 t + 1 of a tile's patch are one low-resolution row (columns
    // likewise), so V = B^T d B is identically zero in transform row 2 and column 2 and so is dU there -- wave 2 (row 2) sits the
    // workgroup's blocks out, the others leave out position 2 of their row (6 MFMAs per step instead of 8; winograd_fused.hip)
    if (skip2 && wv == 2) return;
    rd_x(0, 0); rd_x(0, 1); rd_y(0, 0); rd_y(0, 1);
    form_a(0, 0); form_a(0, 1); form_b(0, 0, 0); form_b(0, 0, 1); form_b(0, 1, 0); form_b(0, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int q = s & 1;
      const bool nx = s + 1 < NS;
#define WG_MF_(j, nb) acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(An[q][j], Bn[q][nb][j], acc[j][nb], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
      WG_MF_(0, 0); if (nx) rd_x(s + 1, 0); __builtin_amdgcn_sched_barrier(0);
      WG_MF_(0, 1); if (nx) rd_x(s + 1, 1); __builtin_amdgcn_sched_barrier(0);
      WG_MF_(1, 0); if (nx) rd_y(s + 1, 0); __builtin_amdgcn_sched_barrier(0);
      WG_MF_(1, 1); if (nx) rd_y(s + 1, 1); __builtin_amdgcn_sched_barrier(0);
      if (!skip2) { WG_MF_(2, 0); }
      if (nx) form_a(q ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!skip2) { WG_MF_(2, 1); }
      if (nx) { form_a(q ^ 1, 1); form_b(q ^ 1, 0, 0); }
      __builtin_amdgcn_sched_barrier(0);
      WG_MF_(3, 0); if (nx) { form_b(q ^ 1, 0, 1); form_b(q ^ 1, 1, 0); } __builtin_amdgcn_sched_barrier(0);
      WG_MF_(3, 1); if (nx) form_b(q ^ 1, 1, 1); __builtin_amdgcn_sched_barrier(0);
#undef WG_MF_
    }
  };

  if constexpr (DB) {
    // two stages: the loads of block q + 1 fly while block q is multiplied; one barrier per block
    if (q_begin < q_end) {
      issue(q_begin, 0u);
      segsde_wait_vmcnt0();
      __syncthreads();
      for (int q = q_begin; q < q_end; q += 2) {
        if (q + 1 < q_end) issue(q + 1, (unsigned)(G::STAGE * 4));
        compute(lds);
        segsde_wait_vmcnt0();
        __syncthreads();
        if (q + 1 < q_end) {
          if (q + 2 < q_end) issue(q + 2, 0u);
          compute(lds + G::STAGE);
          segsde_wait_vmcnt0();
          __syncthreads();
        }
      }
    }
  } else {
    for (int q = q_begin; q < q_end; ++q) {
      issue(q, 0u);
      segsde_wait_vmcnt0();
      __syncthreads();
      compute(lds);
      __syncthreads();                               // every wave's reads of the stage are done before the next block's loads land
    }
  }

  // ---- epilogue: slab (position 4 wave + j, this split), rows = input channels, columns = filters.  True dM carries a minus
  // sign in row 3 and in column 3 of the 4x4 position grid (A's last row is [0 -1]): applied here, once.
  const size_t plane = (size_t)p.C * p.Co;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sg = ((wave == 3) != (j == 3)) ? -1.f : 1.f;
    float* dst = p.part + ((size_t)(4 * wave + j) * p.S + split) * plane + (size_t)(cin0 + 4 * kk) * p.Co + co0 + i;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        dst[(size_t)row * p.Co + 32 * nb] = sg * acc[j][nb][r];
      }
  }
}

int target_wgs() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SEGSDE_WGRAD_FUSED_WGS"); v = e ? atoi(e) : 512; if (v <= 0) v = 512; }
  return v;
}
// staging variant: 0 = 32-tile blocks, one stage; 1 = 16-tile blocks, two stages (the next block's loads fly during the
// multiplies); 2 = 32-tile blocks, two stages, one workgroup per CU.  Measured (profiles/probe_r05_winograd_wgrad_*.log): 1 is
// 2-3 % ahead on the large maps (>= 128 x 256 at batch 16: many blocks per workgroup), 0 on the small ones, 2 loses everywhere.
// SEGSDE_WGRAD_FUSED_VAR forces one (read once per process).
int forced_variant() {
  static int v = -2;
  if (v == -2) { const char* e = getenv("SEGSDE_WGRAD_FUSED_VAR"); v = e ? atoi(e) : -1; }
  return v;
}
int variant(const segsde_conv_desc* d) {
  const int f = forced_variant();
  if (f >= 0) return f;
  return (long)d->B * d->H * d->W >= (1L << 19) ? 1 : 0;
}
int block_h(const segsde_conv_desc* d) { return variant(d) == 1 ? 2 : 4; }

bool shape_ok(const segsde_conv_desc* d) {
  if (!d || d->B <= 0 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1 || d->in_div > 1 || d->sum2x2) return false;
  if (d->pad_mode != SEGSDE_PAD_ZERO && d->pad_mode != SEGSDE_PAD_REFLECT) return false;
  if (d->H != d->Ho || d->W != d->Wo || d->H < 4 || d->W < 4 || (d->H & 1) || (d->W & 1)) return false;
  if (d->C0 <= 0 || d->C0 % KC || d->C1 < 0 || d->C1 % KC || d->Cout % NC) return false;
  if (d->ld0 < d->C0 || d->ld0 % 4 || (d->C1 && (d->ld1 < d->C1 || d->ld1 % 4))) return false;
  const long ldm = d->ld0 > d->ld1 ? d->ld0 : d->ld1;
  // 32-bit byte offsets inside one image (the buffer resources are rebased per image)
  if ((long)d->H * d->W * ldm * 4 >= (1L << 31)) return false;
  return true;
}

struct Plan { int nbh, nbw, nblk, ncol, S; };
Plan make_plan(const segsde_conv_desc* d) {
  Plan pl;
  const int bh = block_h(d);
  pl.nbh = ((d->H >> 1) + bh - 1) / bh; pl.nbw = ((d->W >> 1) + WT_W - 1) / WT_W;
  const long nblk = (long)d->B * pl.nbh * pl.nbw;
  pl.nblk = (int)nblk;
  pl.ncol = ((d->C0 + d->C1) / KC) * (d->Cout / NC);
  int S = target_wgs() / pl.ncol;
  if (S < 1) S = 1;
  if (S > nblk) S = (int)nblk;
  pl.S = S;
  return pl;
}
}  // namespace

extern "C" size_t segsde_conv2d_wgrad_winograd_fused_workspace(const segsde_conv_desc* d) {
  if (!shape_ok(d)) return 0;
  const long nblk = (long)d->B * (((d->H >> 1) + 1) / 2) * (((d->W >> 1) + WT_W - 1) / WT_W);
  if (nblk >= (1L << 30)) return 0;
  const Plan pl = make_plan(d);
  return (size_t)16 * pl.S * (d->C0 + d->C1) * d->Cout * sizeof(float);
}

extern "C" int segsde_conv2d_wgrad_winograd_fused(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                                                  float* dw_oihw, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return SEGSDE_ERR_NULL;
  if (!x0 || !dy || !dw_oihw || !workspace || (d->C1 > 0 && !x1)) return SEGSDE_ERR_NULL;
  const size_t need = segsde_conv2d_wgrad_winograd_fused_workspace(d);
  if (!need || lddy < d->Cout || lddy % 4 || (long)d->H * d->W * lddy * 4 >= (1L << 31)) return SEGSDE_ERR_UNSUPPORTED;
  if (((uintptr_t)x0 | (uintptr_t)dy | (uintptr_t)(x1 ? x1 : x0)) & 15) return SEGSDE_ERR_UNSUPPORTED;
  if (workspace_bytes < need) return SEGSDE_ERR_WORKSPACE;
  const Plan pl = make_plan(d);
  WgradP p;
  p.x0 = x0; p.x1 = x1 ? x1 : x0; p.ld0 = d->ld0; p.ld1 = d->C1 ? d->ld1 : d->ld0; p.C0 = d->C0; p.up0 = d->up0 ? 1 : 0;
  p.dy = dy; p.lddy = lddy;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C = d->C0 + d->C1; p.Co = d->Cout; p.reflect = d->pad_mode == SEGSDE_PAD_REFLECT;
  p.nbh = pl.nbh; p.nbw = pl.nbw; p.nblk = pl.nblk; p.S = pl.S; p.ncol = pl.ncol;
  p.part = static_cast<float*>(workspace);
  const dim3 grid((unsigned)(pl.ncol * pl.S));
  const int var = variant(d);
  // zero-position skipping for upsampled-source channel blocks: only where that source carries at least half of the channels
  // (SEGSDE_WINO_UP_SKIP=0: never)
  static int upskip_on = -1;
  if (upskip_on < 0) { const char* e = getenv("SEGSDE_WINO_UP_SKIP"); upskip_on = e ? (atoi(e) != 0) : 1; }
  const bool upskip = upskip_on && p.up0 && 2 * d->C0 >= d->C0 + d->C1;
  auto go = [&](auto k, size_t lb) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
    hipLaunchKernelGGL(k, grid, dim3(256), lb, ST(stream), p);
  };
  if (var == 1) {
    const size_t lb = (size_t)2 * Geo<2>::STAGE * sizeof(float);
    if (upskip) go(wino_wgrad_fused_kernel<2, true, 2, true>, lb); else go(wino_wgrad_fused_kernel<2, true, 2, false>, lb);
  } else if (var == 2) {
    const size_t lb = (size_t)2 * Geo<4>::STAGE * sizeof(float);
    if (upskip) go(wino_wgrad_fused_kernel<4, true, 1, true>, lb); else go(wino_wgrad_fused_kernel<4, true, 1, false>, lb);
  } else {
    const size_t lb = (size_t)Geo<4>::STAGE * sizeof(float);
    if (upskip) go(wino_wgrad_fused_kernel<4, false, 2, true>, lb); else go(wino_wgrad_fused_kernel<4, false, 2, false>, lb);
  }
  SEGSDE_CHECK_LAUNCH();
  return segsde_wino_wgrad_finish(p.part, pl.S, p.C, p.Co, dw_oihw, stream);
}
