// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense fp32; there is no TF32 on CDNA4).
//
// One kernel family serves every convolution of the reference's encoder / decoders
// (SURVEY.md 2.3): 1x1, 3x3 (zero or reflection padded, dilated, strided), 7x7 s2, and --
// run with flipped/transposed weights -- their data gradients.  A second kernel computes
// weight gradients (pixel-reduction GEMM, split over pixels, deterministic two-pass reduce).
//
//   forward / dgrad :  Y[m, n] = sum_k  A(m, k) * Wp[n, k]        m = (b, ho, wo), k = (kh, kw, c)
//   wgrad           :  dW[k, n] = sum_m A(m, k) * dY[m, n]
//
// A(m, k) is never materialised: the tile loader gathers it from NHWC activations and applies, on the
// fly, the padding rule (zero / reflect), dilation, stride, an optional nearest x2 upsample of source 0
// and an optional channel concat [source0 | source1] (depth_decoder.py:93-100), so neither the padded,
// the upsampled nor the concatenated tensor ever exists in HBM.
//
// Layout: activations NHWC fp32 (channel-contiguous => the k-run of one tap is a coalesced 128 B line per
// pixel), weights pre-packed [Cout][KH*KW*C] (k-contiguous) so that A and B tiles stage identically.
// Tiles: BM x BN output tile per 256-thread workgroup (4 waves), BK = 32; LDS rows padded to 36 floats so
// that the per-lane ds_read_b128 fragment reads are bank-conflict free; register-staged double buffering
// (global loads of chunk t+1 are in flight while the MFMAs of chunk t run; one barrier per chunk).
// MFMA operand order inside a 8-wide k group is permuted (lane half h, step s) -> k = 4h + s so that each
// lane fetches its 4 A (and 4 B) values with ONE 16-byte LDS read.
#include <type_traits>
#include "segsde_common.h"
#include "conv_small.h"
#include "winograd.h"
#include <cstdlib>
#include <mutex>
#include <cstring>

namespace {

struct ConvP {
  const float* x0; const float* x1; const float* w; const float* bias;
  float* y; float* y2;
  int B, H, W, C0, C1, ld0, ld1, up0;
  int Ho, Wo, N, ldy, ldy2, nsplit;
  int KH, KW, stride, dil, pad, pad_mode, in_div;
  int Ctot, Ktot, M, act, sum2x2;
  int vecout;          // outputs allow 16-byte stores (pitches, split point and bases multiples of 4 floats / 16 B)
  int nb, ne;          // output-channel range [nb, ne) handled by this launch (tile-shape mixing for Cout % 128 == 64)
  const float* zero;   // 256 bytes of zeros: target of out-of-range tile loads
  // Sub-problem view used by the parity-class decomposition of stride-2 data-gradients (FAST path only):
  // the loop tap (kh', kw') stands for the real tap (kh0 + khs*kh', kw0 + kws*kw') of a KHf x KWf kernel whose packed
  // rows are Kfull long, and output pixel (b, i, j) of the Ho x Wo sub-grid is stored at (b, os*i + oph, os*j + opw) of
  // an OHf x OWf image.  Defaults (0, 1, 0, 1, KW, Ktot, os = 1) describe the plain problem.
  int kh0, khs, kw0, kws, KWf, Kfull, os, oph, opw, OHf, OWf;
  // BatchNorm statistics fused into the epilogue (nullable): per-tile column sums / sums of squares of the stored tile,
  // [ntm * (256 / BN)][2][N] doubles (sum, sum of squares per row slice of a tile) -- the next layer's batch statistics
  // without re-reading the tensor
  double* stats;
  int accum;           // 1: the staged epilogue adds the tile to what the destination already holds
  // Activation backward fused into a data-gradient's epilogue (nullable): the tensor this launch differentiates with
  // respect to (destination y, channels < nsplit) is the saved OUTPUT of an activation of kind agkind; the stored value is
  // multiplied by act'(agy) -- the gradient with respect to the PRE-activation, which is what the producing conv's own
  // backward needs (no separate 12-byte-per-element activation-backward pass).  Same row order / channel order as y, pitch agld.
  const float* agy; int agld, agkind;
  // division of a GEMM row index (< 2^31) by the image size / row length without the ~30-instruction runtime division:
  // q = umulhi(n, magic) >> shift (set_divs).  d1 = Ho*Wo, d2 = Wo; for sum2x2 launches the 2x2-block counts.
  unsigned mg1, mg2; int sf1, sf2, d1, d2;
  unsigned mgC, mgKW; int sfC, sfKW;   // the same for k / Ctot and tap / KW (generic gathers: decode_k)
  int lin;   // 1x1, stride 1, unpadded, one source at output resolution: GEMM row m IS pixel m of the input (no row decode)
  // Upsample-folded sub-problems (segsde_conv2d_*_upfold, FAST / table-driven paths only): padw = the column padding (the
  // parity classes of a folded 3x3 pad rows and columns differently; = pad everywhere else); wtap = floats between two taps
  // of a packed weight row (= Ctot unless the launch reads a channel slice of wider rows); pad_mode SEGSDE_PAD_CLAMP_ =
  // out-of-range taps read the nearest border pixel (what mirrored padding of a nearest-upsampled image amounts to on the
  // low-resolution grid); osfast = a strided sub-grid store (os > 1) whose tiles lie inside one sub-grid row.
  int padw, wtap, osfast;
  int submap;   // output rows are not rows m of the destination: out_row() maps them (strided sub-grids, single border rows / columns)
  // Dilated windows with zero padding (ASPP, rates 6 / 12 / 18 on a 32-row map; layer4 of a dilated ResNet): a tap ROW whose
  // source rows lie outside the image for every pixel of a tile is an all-zero operand -- the tile's K loop runs over its
  // live tap rows only (tile-uniform: a tile is a run of consecutive pixels, its rows an interval).  Weight gradient: a
  // pixel chunk whose rows are dead for the workgroup's tap row is skipped.
  int tapskip;
  // Grouped weights (the sixteen position GEMMs of a Winograd convolution as one launch, FAST / LDS-DMA path only): a tile
  // whose first row lies in image b of the B-image input reads its weight rows from w + b * wbstride floats.  0: one weight.
  long wbstride;
  int nt;       // 1: the staged epilogue's full-tile stores are streaming stores (set by the launcher for outputs beyond the caches)
  int f16;      // 1: half-precision operands on the LDS-DMA loop (segsde_conv_desc.compute)
};
constexpr int SEGSDE_PAD_CLAMP_ = 3;   // internal (never crosses the ABI)

struct KInfo {  // decoded reduction index k -> tap + channel + source
  const float* src; int ld, Hs, Ws, shift, dh, dw, cc; bool valid;
};

// n / d for 0 <= n < 2^31 with the (magic, shift) pair of set_divs; d == 1 is flagged by magic == 0
__device__ __forceinline__ int fast_div(int n, unsigned magic, int shift) {
  return magic ? (int)(__umulhi((unsigned)n, magic) >> shift) : n;
}

__device__ __forceinline__ KInfo decode_k(const ConvP& p, int k) {
  KInfo t;
  t.valid = k < p.Ktot;
  const int kk = t.valid ? k : 0;
  const int tap = fast_div(kk, p.mgC, p.sfC), c = kk - tap * p.Ctot;
  const int kh = fast_div(tap, p.mgKW, p.sfKW), kw = tap - kh * p.KW;
  t.dh = kh * p.dil - p.pad;
  t.dw = kw * p.dil - p.pad;
  if (c < p.C0) { t.src = p.x0; t.ld = p.ld0; t.cc = c; t.shift = p.up0; }
  else { t.src = p.x1; t.ld = p.ld1; t.cc = c - p.C0; t.shift = 0; }
  t.Hs = p.H >> t.shift;
  t.Ws = p.W >> t.shift;
  return t;
}

// element offset of A(m,k) in its source, or -1 when the tap falls on zero padding / a stride hole
__device__ __forceinline__ long a_offset(const ConvP& p, const KInfo& t, int b, int hb, int wb) {
  int hi = hb + t.dh, wi = wb + t.dw;
  if (p.in_div > 1) {  // data-gradient of a strided conv: only every in_div-th position carries a value
    if (hi < 0 || wi < 0 || (hi % p.in_div) != 0 || (wi % p.in_div) != 0) return -1;
    hi /= p.in_div; wi /= p.in_div;
  }
  if (p.pad_mode == SEGSDE_PAD_REFLECT) {
    hi = hi < 0 ? -hi : (hi >= p.H ? 2 * p.H - 2 - hi : hi);
    wi = wi < 0 ? -wi : (wi >= p.W ? 2 * p.W - 2 - wi : wi);
  } else if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.W) {
    return -1;
  }
  hi >>= t.shift; wi >>= t.shift;
  return ((long)(b * t.Hs + hi) * t.Ws + wi) * t.ld + t.cc;
}

__device__ __forceinline__ void decode_m(const ConvP& p, int m, int& b, int& hb, int& wb, bool& ok) {
  ok = m < p.M;
  const int mm = ok ? m : 0;
  if (p.sum2x2) {
    // output pixels enumerated patch-major: m = ((b*Ho/2 + h/2)*Wo/2 + w/2)*4 + (h&1)*2 + (w&1), so the four members of
    // a 2x2 block are four consecutive GEMM rows = the four registers (r&3) of one lane in the MFMA accumulator
    const int q = mm >> 2, sub = mm & 3;
    b = fast_div(q, p.mg1, p.sf1);
    const int rem = q - b * p.d1, h2 = fast_div(rem, p.mg2, p.sf2);
    hb = 2 * h2 + (sub >> 1);
    wb = 2 * (rem - h2 * p.d2) + (sub & 1);
    return;
  }
  b = fast_div(mm, p.mg1, p.sf1);
  const int rem = mm - b * p.d1;
  const int ho = fast_div(rem, p.mg2, p.sf2);
  hb = ho * p.stride;
  wb = (rem - ho * p.d2) * p.stride;
}

// row index of output GEMM row m in the destination tensor (identity unless the launch writes a strided sub-grid)
__device__ __forceinline__ long out_row(const ConvP& p, int m) {
  if (!p.submap) return m;
  const int b = fast_div(m, p.mg1, p.sf1), rem = m - b * p.d1, i = fast_div(rem, p.mg2, p.sf2), j = rem - i * p.d2;
  return ((long)b * p.OHf + p.os * i + p.oph) * p.OWf + p.os * j + p.opw;
}


template <bool VEC>
__device__ __forceinline__ float4 fetch_a4(const ConvP& p, int k, int b, int hb, int wb, bool row_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return v;
  if (VEC) {
    const KInfo t = decode_k(p, k);
    if (!t.valid) return v;
    const long off = a_offset(p, t, b, hb, wb);
    if (off >= 0) v = *reinterpret_cast<const float4*>(t.src + off);
  } else {
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[j] = 0.f;
      const KInfo t = decode_k(p, k + j);
      if (t.valid) {
        const long off = a_offset(p, t, b, hb, wb);
        if (off >= 0) e[j] = t.src[off];
      }
    }
    v = make_float4(e[0], e[1], e[2], e[3]);
  }
  return v;
}

// float4 gather with the chunk's decoded (tap, channel) handed in: decode_k once per chunk and thread, not once per tile row
__device__ __forceinline__ float4 fetch_a4_at(const ConvP& p, const KInfo& t, int b, int hb, int wb, bool row_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok || !t.valid) return v;
  const long off = a_offset(p, t, b, hb, wb);
  if (off >= 0) v = *reinterpret_cast<const float4*>(t.src + off);
  return v;
}
template <bool VEC>
__device__ __forceinline__ float4 fetch_w4(const ConvP& p, int n, int k) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n >= p.ne) return v;
  const float* row = p.w + (long)n * p.Ktot;
  if (VEC) {
    if (k < p.Ktot) v = *reinterpret_cast<const float4*>(row + k);
  } else {
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = (k + j < p.Ktot) ? row[k + j] : 0.f;
    v = make_float4(e[0], e[1], e[2], e[3]);
  }
  return v;
}

// reduction elements per staged chunk: template parameter BK (32, or 64 on the FAST path when channels allow)

// ---------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ---------------------------------------------------------------------------------------------------
// Uniform per-chunk tap state of the FAST path: when Ctot % 32 == 0 (and C0 % 32 == 0 for two sources) a 32-wide
// reduction chunk lies inside ONE tap of ONE source, so tap / source / channel base are wave-uniform scalars that
// advance incrementally (no per-element integer division) and the tile loads become straight-line code:
// clamped (always valid) addresses + a select, instead of the branchy generic gather.
struct ChunkState {
  int c0, kh, kw;   // channel offset inside the concatenated input, tap coordinates
  __device__ __forceinline__ void advance(const ConvP& p, int bk) {
    c0 += bk;
    if (c0 == p.Ctot) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
  }
};

struct SrcSel { const float* src; unsigned ld, Ws, bstride; int shift, cc; };
__device__ __forceinline__ SrcSel select_src(const ConvP& p, int c) {
  SrcSel s;
  if (c < p.C0) { s.src = p.x0; s.ld = p.ld0; s.cc = c; s.shift = p.up0; }
  else { s.src = p.x1; s.ld = p.ld1; s.cc = c - p.C0; s.shift = 0; }
  s.Ws = (unsigned)(p.W >> s.shift);
  s.bstride = (unsigned)(p.H >> s.shift) * s.Ws * s.ld;
  return s;
}

// Loads that fall on zero padding / past the tile edge read this page instead of being zeroed after the fact: a select on
// a loaded value would force an s_waitcnt right behind every load and serialise the eight tile loads of a chunk.
__device__ float segsde_zero_page[64];

__device__ __forceinline__ unsigned off_at(const SrcSel& s, int b, int hi, int wi, int cq) {
  return (unsigned)b * s.bstride + ((unsigned)(hi >> s.shift) * s.Ws + (unsigned)(wi >> s.shift)) * s.ld +
         (unsigned)(s.cc + cq);
}

// (hi, wi) = output pixel + tap offset.  Straight-line on purpose (selects, no branches): the address is always formed
// from clamped coordinates and swapped for the zero page when the tap is out of range.
__device__ __forceinline__ float4 fast_fetch(const ConvP& p, const SrcSel& s, int b, int hi, int wi, bool ok, int cq) {
  // data-gradient of a stride-2 conv (in_div == 2): only even coordinates carry a value.  Branch-free (mask + shift by
  // in_div >> 1) so that the K loop stays one basic block the scheduler can interleave with the MFMA stream.
  const int ds = p.in_div >> 1;
  ok = ok && (((hi | wi) & ds) == 0);
  hi >>= ds; wi >>= ds;
  const bool refl = p.pad_mode == SEGSDE_PAD_REFLECT;
  const int hr = hi < 0 ? -hi : (hi >= p.H ? 2 * p.H - 2 - hi : hi);
  const int wr = wi < 0 ? -wi : (wi >= p.W ? 2 * p.W - 2 - wi : wi);
  const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
  ok = ok && (refl || inb);
  hi = ok ? (refl ? hr : hi) : 0;
  wi = ok ? (refl ? wr : wi) : 0;
  const float* ptr = s.src + off_at(s, b, hi, wi, cq);
  ptr = ok ? ptr : p.zero;
  return *reinterpret_cast<const float4*>(ptr);
}

// MODE 0: generic scalar gather, 1: generic float4 gather, 2: FAST (uniform tap per chunk, branch-free loads),
// 3: FAST + reflection-pad adjoint extras, 4: FAST with the tiles written to LDS by the load itself (LDS-DMA)
// VAR: experiment variants of the LDS-DMA loop (SEGSDE_TUNE="var=N"; 0 = shipped): 1 = all tile loads of a chunk issued
// up front, 2 = no scheduling fences between the MFMA units, 3 = raised wave priority around the MFMA units, 4 = (with BK = 16)
// four LDS stages: the loads of chunk k+3 are issued during chunk k, two chunks of loads stay in flight across barriers
// VARX >= 16 (MODE 4 only): HALF-PRECISION OPERANDS, the arithmetic of the reference's `amp: True` mode (torch autocast runs its
// convolutions on fp16 inputs with fp32 accumulation).  Tiles still travel as fp32 (LDS-DMA cannot convert); a lane's two
// fragment reads of a 16-deep k block (4 + 4 consecutive floats per operand row) are rounded to eight halves
// (v_cvt_pk_f16_f32, round to nearest even) and ONE v_mfma_f32_32x32x16_f16 replaces eight v_mfma_f32_32x32x2_f32 -- which k
// sits in which of the instruction's sixteen slots does not matter as long as both operands agree, and they do (same read
// pattern).  Accumulators, epilogues, statistics: unchanged fp32.
template <int BM, int BN, int WM, int WN, int MODE, int BK, int VARX = 0>
__global__ __launch_bounds__(256, (VARX & 15) == 6 ? 3 : 2) void conv_igemm_kernel(ConvP p) {
  constexpr int VAR = VARX & 15;
  constexpr bool F16 = VARX >= 16;
  static_assert(!F16 || MODE == 4, "half-precision operands: LDS-DMA loop only");
  constexpr bool VEC = MODE >= 1;
  constexpr bool FAST = MODE >= 2;
  constexpr bool ADJ = MODE == 3;
  constexpr bool DMA = MODE == 4 || (MODE == 3 && VAR != 5);   // MODE 3: the waves that own no border pixel run the LDS-DMA loop too
  // LDS rows are unpadded (BK floats); the 16-byte column groups of a row are XOR-swizzled with the row index so that
  // the 16 lanes of every ds_read_b128 lane group hit 16 distinct 16-byte slots of the 256-byte bank row (conflict-free)
  // -- no padding means 48 KB instead of 54 KB for the 128x64 tile, i.e. THREE workgroups per CU instead of two.
  constexpr int LDT = BK;
  constexpr int KQ = BK / 4, RP = 256 / KQ; // float4 columns per tile row, tile rows staged per pass
  constexpr int RPS = 16 / KQ;              // tile rows per 256-byte LDS bank row
  auto swz = [](int row) { return (row / RPS) & (KQ - 1); };
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int AR = BM / RP, BR = BN / RP;
  constexpr int STAGE = (BM + BN) * LDT;
  constexpr int NSTAGES = VAR == 4 ? 4 : 2;
  // floats of LDS in front of the FAST path's tap table: the stages or the staged epilogue's tile, whichever is larger
  constexpr int TAB0 = NSTAGES * STAGE > BM * BN ? NSTAGES * STAGE : BM * BN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  SEGSDE_SMEM;
  float* smem = reinterpret_cast<float*>(segsde_smem);

  const int ntn = (p.ne - p.nb + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int tile = segsde_xcd_remap(blockIdx.x, ntm * ntn);
  const int mt = tile / ntn, nt = tile - mt * ntn;
  const int m0 = mt * BM, n0 = p.nb + nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int kq = tid % KQ, r0 = tid / KQ;

  int rb[AR], rh[AR], rw[AR];
  bool rok[AR];
  const bool lin = FAST && MODE != 3 && p.lin;   // the bottleneck 1x1 convolutions and their data-gradients: rows are consecutive pixels
  if (lin) {
#pragma unroll
    for (int i = 0; i < AR; ++i) { rb[i] = 0; rh[i] = 0; rw[i] = 0; rok[i] = m0 + r0 + RP * i < p.M; }
  } else {
#pragma unroll
    for (int i = 0; i < AR; ++i) decode_m(p, m0 + r0 + RP * i, rb[i], rh[i], rw[i], rok[i]);
  }
  int nchunks = (p.Ktot + BK - 1) / BK;
  int kh_first = 0;
  if constexpr (FAST && MODE != 3) {
    if (p.tapskip) {
      // source row of tap row kh for a pixel with base row hb: hb + kh*dil - pad; the tile's base rows are [hA, hB] of ONE image
      int bA, hA, wA, bB, hB, wB; bool okA, okB;
      decode_m(p, m0, bA, hA, wA, okA);
      decode_m(p, (m0 + BM < p.M ? m0 + BM : p.M) - 1, bB, hB, wB, okB);
      if (bA == bB) {
        const int lo = p.pad - hB, hi = p.H - 1 + p.pad - hA;          // live: lo <= kh*dil <= hi
        const int khA = lo > 0 ? (lo + p.dil - 1) / p.dil : 0;
        int khB = hi / p.dil;
        khB = khB > p.KH - 1 ? p.KH - 1 : khB;
        if (khA <= khB && hi >= 0) {
          kh_first = __builtin_amdgcn_readfirstlane(khA);
          nchunks = __builtin_amdgcn_readfirstlane((khB - khA + 1) * p.KW * (p.Ctot / BK));
        }
      }
    }
  }
  ChunkState cs; cs.c0 = 0; cs.kh = kh_first; cs.kw = 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[AR], rbv[BR], rex[AR];

  auto gload = [&](int kc) {   // generic gather (MODE 0/1)
    // chunks past the end are loaded from clamped (valid) addresses and discarded: the K loop stays branch-free
    const int kcl = kc < nchunks ? kc : nchunks - 1;
    {
      const int k = kcl * BK + 4 * kq;
      if constexpr (VEC) {
        const KInfo t = decode_k(p, k);
#pragma unroll
        for (int i = 0; i < AR; ++i) ra[i] = fetch_a4_at(p, t, rb[i], rh[i], rw[i], rok[i]);
      } else {
#pragma unroll
        for (int i = 0; i < AR; ++i) ra[i] = fetch_a4<VEC>(p, k, rb[i], rh[i], rw[i], rok[i]);
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) rbv[i] = fetch_w4<VEC>(p, n0 + r0 + RP * i, k);
    }
  };
  auto lstore = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * LDT;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      float4 v = ra[i];
      if constexpr (ADJ) { v.x += rex[i].x; v.y += rex[i].y; v.z += rex[i].z; v.w += rex[i].w; }
      const int row = r0 + RP * i;
      *reinterpret_cast<float4*>(As + row * LDT + 4 * (kq ^ swz(row))) = v;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int row = r0 + RP * i;
      *reinterpret_cast<float4*>(Bs + row * LDT + 4 * (kq ^ swz(row))) = rbv[i];
    }
  };
  auto mma_groups = [&](int buf, int g0, int g1) {
    const float* As = smem + buf * STAGE;
    const float* Bs = As + BM * LDT;
    const int arow = wm * TM * 32 + (lane & 31), brow = wn * TN * 32 + (lane & 31), h = lane >> 5;
    const float* Ap = As + arow * LDT;
    const float* Bp = Bs + brow * LDT;
    const int sa = swz(arow), sb = swz(brow);   // rows 32 apart share the swizzle (32 / RPS is a multiple of KQ)
#pragma unroll
    for (int g = g0; g < g1; ++g) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(Ap + i * 32 * LDT + 4 * ((2 * g + h) ^ sa));
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bp + j * 32 * LDT + 4 * ((2 * g + h) ^ sb));
      // k-step outermost: consecutive MFMAs target different accumulators
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    }
  };

  // Software pipeline (one barrier per chunk, nothing but the barrier at the chunk boundary):
  //   registers hold chunk kc+1 (loaded during the previous iteration); first half of chunk kc's MFMAs; the
  //   registers go to the other LDS buffer (free since the barrier that ended iteration kc-1); the loads of chunk
  //   kc+2 are issued; second half of the MFMAs.  LDS writes and the address arithmetic of the loads sit in the
  //   middle of the MFMA stream of the same wave instead of serialising in front of the barrier.
  constexpr int NG = BK / 8;
  if constexpr (FAST) {
    // FAST path.  Two facts measured on gfx950 shape this loop (profiles/ablate_r01_kloop.log,
    // profiles/probe_r01_mfma_valu_overlap.log): (1) fp32 MFMA and VALU instructions do not overlap -- every VALU
    // instruction issued on a SIMD costs ~3 cycles of matrix-pipe time, whichever wave it comes from -- so the ~200
    // VALU instructions of per-chunk address arithmetic cost 15 % of the kernel no matter how they are scheduled;
    // (2) the tile loads themselves (L2 or L1 hits alike) are free.  Hence: tile loads are raw buffer loads whose
    // per-lane byte offset voff[] depends only on (row, tap, source) and is recomputed when the tap or source changes
    // (every Csrc/BK chunks); the per-chunk channel advance is the wave-uniform SGPR offset and padding is the
    // out-of-range offset SEGSDE_OOB (hardware returns zeros).  The K loop proper has no address VALU work left.
    // The chunk's MFMAs are issued as BK/2 k-step units (TM*TN MFMAs each) with the LDS stores of chunk kc+1, the
    // loads of chunk kc+2 and the fragment reads of the next k-group dealt out between them.
    constexpr int U = BK / 2;
    constexpr int LSTEP = (U - 4) / AR;
    int b0, th0, tw0; bool tok0;
    decode_m(p, m0, b0, th0, tw0, tok0);
    b0 = __builtin_amdgcn_readfirstlane(b0);
    const SrcSel s0 = select_src(p, 0);
    const SrcSel s1 = select_src(p, p.C0 < p.Ctot ? p.C0 : 0);
    const float* base0 = s0.src + (size_t)b0 * s0.bstride;   // resources rebased to the first image the tile touches
    const float* base1 = s1.src + (size_t)b0 * s1.bstride;
    const segsde_rsrc rsw = segsde_make_rsrc(p.w + (size_t)b0 * (size_t)p.wbstride);
    unsigned voff[AR], voffB[BR];
    bool wave_bord = false, wave_corner = false;
    int bflag[AR];
    {
      int anyb = 0, anyc = 0;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        rb[i] -= b0;
        const bool br = rok[i] && (rh[i] == 1 || rh[i] == p.H - 2), bc = rok[i] && (rw[i] == 1 || rw[i] == p.W - 2);
        anyb |= (br || bc) ? 1 : 0; anyc |= (br && bc) ? 1 : 0;
        bflag[i] = !rok[i] ? 0 : ((rh[i] == 1 ? 1 : 0) | (rh[i] == p.H - 2 ? 2 : 0) | (rw[i] == 1 ? 4 : 0) | (rw[i] == p.W - 2 ? 8 : 0));
      }
      if constexpr (ADJ) { wave_bord = __any(anyb) != 0; wave_corner = __any(anyc) != 0; }
    }
    // LDS-DMA writes lane l of an instruction to slot l of a 1 KiB block (8 tile rows x 8 sixteen-byte slots for BK = 32):
    // the thread that owns slot kq of row r0 therefore FETCHES the channel group that the swizzled layout keeps there,
    // kq ^ swz(r0) (rows 32 apart share the swizzle, so this is one constant per thread) -- fragment reads are unchanged.
    // A wave that owns a border pixel of a reflection adjoint adds extra pre-images to its rows before they reach LDS and
    // keeps the register-staged path (logical slot, swizzle at the LDS store) -- a wave-uniform choice.
    const unsigned kqs = (DMA && !wave_bord) ? (unsigned)(kq ^ swz(r0)) : (unsigned)kq;
    // reflection-pad adjoint: a pixel in row 1 / H-2 (column 1 / W-2) also collects what flowed into the mirrored
    // padding row -1 / H (column -1 / W), reachable only through the tap with dh = +1 / -1 (dw likewise).  The extra
    // pre-image is one more buffer load per tile row (offset voffX, out of range when there is none); only the four
    // corner-adjacent pixels of an image have up to three extras at once (voffX2/3, loaded when the wave owns one).
    unsigned voffX[AR], voffX2[AR], voffX3[AR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int n = n0 + r0 + RP * i;
      voffB[i] = n < p.ne ? ((unsigned)(n * p.Kfull) + 4u * kqs) * 4u : SEGSDE_OOB;   // rows past Cout read zeros
    }
    // byte offsets of the tile rows' source pixels for tap (kh, kw) of one source (col = the thread's own column part, in
    // floats); with WADJ also the extra pre-images of the reflection adjoint (those need col = 4 * kqs)
    auto compute_voff = [&](bool in0, int kh, int kw, unsigned col, auto wadj_tag, auto&& sink) {   // sink(i, offset, extra offset)
      constexpr bool WADJ = decltype(wadj_tag)::value;
      if (lin) {   // the tile's first image starts b0 * H * W pixels before the resource's base row
#pragma unroll
        for (int i = 0; i < AR; ++i)
          sink(i, rok[i] ? ((unsigned)(m0 + r0 + RP * i - b0 * p.d1) * s0.ld + col) * 4u : SEGSDE_OOB, SEGSDE_OOB);
        return;
      }
      const int dh = kh * p.dil - p.pad, dw = kw * p.dil - p.padw;
      const int sh = in0 ? s0.shift : 0;
      const unsigned ld = in0 ? s0.ld : s1.ld, Ws = in0 ? s0.Ws : s1.Ws, bst = in0 ? s0.bstride : s1.bstride;
      const int ds = p.in_div >> 1;   // data-gradient of a stride-2 conv: only even coordinates carry a value
      // clamp padding (upsample-folded class launches) is a compile-time variant, VAR = 9: as one more run-time flag it cost the
      // K loops their last free SGPRs (the LDS-DMA operands no longer got scalar registers)
      constexpr bool clampm = VAR == 9;
      const bool refl = !clampm && p.pad_mode == SEGSDE_PAD_REFLECT;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        auto boff = [&](int hh, int ww) {
          return ((unsigned)rb[i] * bst + ((unsigned)(hh >> sh) * Ws + (unsigned)(ww >> sh)) * ld + col) * 4u;
        };
        int hi = rh[i] + dh, wi = rw[i] + dw;
        bool ok = rok[i] && (((hi | wi) & ds) == 0);
        hi >>= ds; wi >>= ds;
        const int hr = clampm ? (hi < 0 ? 0 : (hi >= p.H ? p.H - 1 : hi)) : (hi < 0 ? -hi : (hi >= p.H ? 2 * p.H - 2 - hi : hi));
        const int wr = clampm ? (wi < 0 ? 0 : (wi >= p.W ? p.W - 1 : wi)) : (wi < 0 ? -wi : (wi >= p.W ? 2 * p.W - 2 - wi : wi));
        const bool hin = (unsigned)hi < (unsigned)p.H, win = (unsigned)wi < (unsigned)p.W;
        const bool mapped = refl || clampm;
        ok = ok && (mapped || (hin && win));
        const unsigned vmain = ok ? boff(mapped ? hr : hi, mapped ? wr : wi) : SEGSDE_OOB;
        unsigned vext = SEGSDE_OOB;
        if constexpr (WADJ) {
          const int eh = (rh[i] == 1 && dh == 1) ? 0 : ((rh[i] == p.H - 2 && dh == -1) ? p.H - 1 : -1);
          const int ew = (rw[i] == 1 && dw == 1) ? 0 : ((rw[i] == p.W - 2 && dw == -1) ? p.W - 1 : -1);
          const bool t1 = rok[i] && eh >= 0 && win, t2 = rok[i] && ew >= 0 && hin, t3 = rok[i] && eh >= 0 && ew >= 0;
          vext = t1 ? boff(eh, wi) : (t2 ? boff(hi, ew) : SEGSDE_OOB);
          if (wave_corner) {
            voffX2[i] = (t1 && t2) ? boff(hi, ew) : SEGSDE_OOB;
            voffX3[i] = t3 ? boff(eh, ew) : SEGSDE_OOB;
          }
        }
        sink(i, vmain, vext);
      }
    };
    // Tap table.  The ~120 VALU instructions of compute_voff used to run at every tap / source change -- every second
    // chunk of a 64-channel layer, i.e. 3.4 VALU instructions per MFMA over such a tile, and VALU cycles are matrix-pipe
    // cycles on this chip.  Instead every (source, tap) offset of every tile row is computed once per tile, into LDS behind
    // the stages (the eight threads that share a row set split the taps), while the first tile loads are in flight; a tap
    // change in the K loop is then AR ds_read_b32 + AR adds.  Rows of waves that run the adjoint's register loop are
    // left out (that loop recomputes, it needs the extra pre-images anyway).
    unsigned* tab = reinterpret_cast<unsigned*>(smem + TAB0);
    const int ntaps = p.KH * p.KW;
    // XTAB: the adjoint kernel's 128-wide tiles keep a second table bank with the border rows' extra pre-image, so that a
    // bordered wave's tap change is table reads as well (every chunk ends in a barrier: the one wave per tile that
    // recomputed ~250 VALU of offsets per tap set the pace).  The 128x64 tile would lose its third workgroup per CU to
    // the 4.6 KB, and a wave that owns a corner-adjacent pixel (up to three extras) keeps recomputing.
    constexpr bool XTAB = ADJ && BN >= 128 && VAR != 8;   // var=8: A/B knob
    const bool xtab = XTAB && p.W >= 128;   // narrow images: every wave is bordered, building the second bank costs more than it saves
    // The extra pre-image of a border row from its MAIN offset (adjoint: 3x3, pad 1, one source at full resolution): row 1
    // with the tap dh = +1 reads row 2 and also collects row 0, two rows up; row H-2 with dh = -1 two rows down; columns
    // likewise.  A bordered wave of a tile without the second table bank does this per tap: a compare, a select and an
    // add per row and axis instead of the ~250 VALU of the full compute_voff.  bflag[i]: bit 0 row 1, bit 1 row H-2,
    // bit 2 column 1, bit 3 column W-2 (a corner-adjacent pixel has a row and a column bit: those waves recompute).
    auto extra_from_main = [&](int kh, int kw) {
      const int rbit = kh == 2 ? 1 : (kh == 0 ? 2 : 0), cbit = kw == 2 ? 4 : (kw == 0 ? 8 : 0);
      const unsigned rstep = 2u * s0.Ws * s0.ld * 4u, cstep = 2u * s0.ld * 4u;
      const unsigned rdelta = kh == 2 ? 0u - rstep : rstep, cdelta = kw == 2 ? 0u - cstep : cstep;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool live = (int)voff[i] >= 0;           // the main pre-image exists (not the out-of-range marker)
        voffX[i] = (live && (bflag[i] & rbit)) ? voff[i] + rdelta : ((live && (bflag[i] & cbit)) ? voff[i] + cdelta : SEGSDE_OOB);
      }
    };
    const bool tiny = ADJ && (p.H < 3 || p.W < 3);   // row 1 is also row H-2's neighbour: the main pre-image can be missing
    auto tab_build = [&](auto wadj_tag) {
      constexpr bool WADJ = decltype(wadj_tag)::value;
      if (WADJ && (wave_corner || tiny)) return;
      const bool with_x = WADJ && xtab;
      const int T = WADJ ? ntaps : (p.C0 < p.Ctot ? 2 : 1) * ntaps;
      for (int t = 1 + kq; t < T; t += KQ) {   // entry 0 is the first tap, computed directly and never revisited
        const bool in0 = t < ntaps;
        const int tt = in0 ? t : t - ntaps, kh = tt / p.KW, kw = tt - kh * p.KW;
        if (with_x) {
          compute_voff(in0, kh, kw, 0u, wadj_tag, [&](int i, unsigned v, unsigned vx) {
            tab[t * BM + r0 + RP * i] = v;
            if constexpr (XTAB) tab[(ntaps + t) * BM + r0 + RP * i] = vx;
          });
        } else {
          compute_voff(in0, kh, kw, 0u, std::false_type{}, [&](int i, unsigned v, unsigned) { tab[t * BM + r0 + RP * i] = v; });
        }
      }
    };
    // direct: the table is not visible yet (prologue, before the first barrier)
    auto tap_update = [&](auto wadj_tag, bool direct) {
      constexpr bool WADJ = decltype(wadj_tag)::value;
      const bool in0 = cs.c0 < p.C0;
      if (direct || (WADJ && (wave_corner || tiny))) {
        compute_voff(in0, cs.kh, cs.kw, 4u * kqs, wadj_tag, [&](int i, unsigned v, unsigned vx) {
          voff[i] = v;
          if constexpr (WADJ) voffX[i] = vx;
        });
      } else {
        const unsigned* tp = tab + ((in0 ? 0 : ntaps) + cs.kh * p.KW + cs.kw) * BM + r0;
#pragma unroll
        for (int i = 0; i < AR; ++i) voff[i] = tp[RP * i] + 16u * kqs;
        if constexpr (WADJ) {
          if (xtab) {
#pragma unroll
            for (int i = 0; i < AR; ++i) voffX[i] = tp[ntaps * BM + RP * i] + 16u * kqs;
          } else {
            extra_from_main(cs.kh, cs.kw);
          }
        }
      }
    };
    // The whole K loop exists twice in the adjoint kernel: waves that own no pixel next to the border run the plain
    // loop, the others the one with the extra loads.  The choice is wave-uniform and both execute the same barriers.
    auto run = [&](auto wadj_tag) {
      constexpr bool WADJ = decltype(wadj_tag)::value;
      // state of the chunk whose loads are being issued
      bool more; segsde_rsrc rsa; unsigned soffA, soffB;
      auto chunk_begin = [&](int kc) {
        more = kc + 1 < nchunks;   // chunks past the end reload the last one (cs stops advancing there) and are discarded
        soffB = (unsigned)(((p.kh0 + p.khs * cs.kh) * p.KWf + p.kw0 + p.kws * cs.kw) * p.wtap + cs.c0) * 4u;
        const bool in0 = cs.c0 < p.C0;
        rsa = segsde_make_rsrc(in0 ? base0 : base1);
        soffA = (unsigned)(in0 ? cs.c0 : cs.c0 - p.C0) * 4u;
      };
      auto loadA = [&](int i) {
        ra[i] = segsde_buffer_load4(rsa, voff[i], soffA);
        if constexpr (WADJ) {
          rex[i] = segsde_buffer_load4(rsa, voffX[i], soffA);
          if (wave_corner) {   // rare (four pixels per image): summed on the spot, no registers held across the loop
            const float4 t2 = segsde_buffer_load4(rsa, voffX2[i], soffA), t3 = segsde_buffer_load4(rsa, voffX3[i], soffA);
            rex[i].x += t2.x + t3.x; rex[i].y += t2.y + t3.y; rex[i].z += t2.z + t3.z; rex[i].w += t2.w + t3.w;
          }
        }
      };
      auto loadB = [&](int i) { rbv[i] = segsde_buffer_load4(rsw, voffB[i], soffB); };
      auto chunk_end = [&](bool direct) {
        if (more) {
          cs.advance(p, BK);
          if (cs.c0 == 0 || cs.c0 == p.C0) tap_update(wadj_tag, direct);
        }
      };
      auto storeA = [&](float* As) {
  #pragma unroll
        for (int i = 0; i < AR; ++i) {
          float4 v = ra[i];
          if constexpr (WADJ) {
            v.x += rex[i].x; v.y += rex[i].y; v.z += rex[i].z; v.w += rex[i].w;
          }
          const int row = r0 + RP * i;
          *reinterpret_cast<float4*>(As + row * LDT + 4 * (kq ^ swz(row))) = v;
        }
      };
      auto storeB = [&](float* Bs) {
  #pragma unroll
        for (int i = 0; i < BR; ++i) {
          const int row = r0 + RP * i;
          *reinterpret_cast<float4*>(Bs + row * LDT + 4 * (kq ^ swz(row))) = rbv[i];
        }
      };
      auto load_chunk = [&](int kc) {     // prologue only
        chunk_begin(kc);
  #pragma unroll
        for (int i = 0; i < AR; ++i) loadA(i);
  #pragma unroll
        for (int i = 0; i < BR; ++i) loadB(i);
        chunk_end(true);
      };

      float4 fa[2][TM], fb[2][TN];
      if constexpr (DMA && !WADJ) {
        // Two LDS stages, no register staging: during the MFMAs of chunk kc the loads of chunk kc+1 write the other
        // stage directly (free since the barrier that ended iteration kc-1); they are waited for (vmcnt(0)) right before
        // the barrier that ends iteration kc.  No ds_write, no staging VGPRs, the loop body has no vector memory
        // instruction that returns to registers.
        static_assert(KQ == 8 || KQ == 4, "an LDS-DMA piece is 1 KiB: 8 rows x 128 bytes or 16 rows x 64 bytes");
        constexpr int NST = VAR == 4 ? 4 : 2;                  // LDS stages; loads run NST-1 chunks ahead of the MFMAs
        constexpr int INFLIGHT = (NST - 2) * (AR + BR);        // loads allowed to stay outstanding at a chunk's barrier
        // LDS byte address of this wave's 1 KiB piece in pass 0 of the A tile of stage 0 (scalar from here on)
        const unsigned lds0 = segsde_lds_addr(smem) + (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 1024);
        constexpr unsigned PASS = RP * LDT * 4, BOFF = BM * LDT * 4, STG = STAGE * 4;
        // weight-row byte offset of the chunk being fetched: consecutive chunks are consecutive 128-byte pieces of the
        // packed row except across a tap change of a parity-class sub-problem, where it is recomputed from the tap
        unsigned wbyte = (unsigned)(((p.kh0 + p.khs * kh_first) * p.KWf + p.kw0) * p.wtap) * 4u;
        auto dma_begin = [&]() {
          const bool in0 = cs.c0 < p.C0;
          rsa = segsde_make_rsrc(in0 ? base0 : base1);
          soffA = (unsigned)(in0 ? cs.c0 : cs.c0 - p.C0) * 4u;
          soffB = wbyte;
        };
        auto dma_end = [&](bool adv, bool direct) {
          if (adv) {
            cs.advance(p, BK);
            wbyte += BK * 4u;
            if (cs.c0 == 0) wbyte = (unsigned)(((p.kh0 + p.khs * cs.kh) * p.KWf + p.kw0 + p.kws * cs.kw) * p.wtap) * 4u;
            if (cs.c0 == 0 || cs.c0 == p.C0) tap_update(wadj_tag, direct);
          }
        };
        auto dmaA = [&](unsigned stage, int i) { segsde_buffer_load4_lds(rsa, voff[i], soffA, lds0 + stage + PASS * i); };
        auto dmaB = [&](unsigned stage, int i) { segsde_buffer_load4_lds(rsw, voffB[i], soffB, lds0 + stage + BOFF + PASS * i); };
        tap_update(wadj_tag, true);
  #pragma unroll
        for (int j = 0; j < NST - 1; ++j) {                    // chunks 0 .. NST-2
          dma_begin();
  #pragma unroll
          for (int i = 0; i < AR; ++i) dmaA((unsigned)j * STG, i);
  #pragma unroll
          for (int i = 0; i < BR; ++i) dmaB((unsigned)j * STG, i);
          dma_end(j + 1 < nchunks, true);
        }
        tab_build(wadj_tag);                                   // under the latency of the first loads
        segsde_wait_vmcnt<INFLIGHT>();                         // chunk 0 has landed
        __syncthreads();
        const int arow = wm * TM * 32 + (lane & 31), brow = wn * TN * 32 + (lane & 31), h = lane >> 5;
        const int sa = swz(arow), sb = swz(brow);
        auto fread = [&](int buf, int g, int slot) {
          const float* Ap = smem + buf * STAGE + arow * LDT;
          const float* Bp = smem + buf * STAGE + BM * LDT + brow * LDT;
  #pragma unroll
          for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const float4*>(Ap + i * 32 * LDT + 4 * ((2 * g + h) ^ sa));
  #pragma unroll
          for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * LDT + 4 * ((2 * g + h) ^ sb));
        };
        auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
        constexpr int DSTEP = (U - 2) / (AR + BR) > 0 ? (U - 2) / (AR + BR) : 1;
        // one chunk; buf is a compile-time constant in the two-stage loop (unrolled by two below) so that the fragment
        // reads and the LDS-DMA destinations use immediate offsets -- the 8 v_lshl_add_u32 per chunk that rebuilt the
        // stage base were most of the loop's remaining VALU work
        auto chunk = [&](int kc, auto buf_c, auto first_c) {
          const int buf = buf_c;
          constexpr bool FIRST = decltype(first_c)::value;   // the tile's first chunk: its first k-step starts from zero
                                                             // accumulators (inline constant) instead of 64 v_mov up front
          fread(buf, 0, 0);
          dma_begin();                     // past the last chunk: the last one is fetched again (harmless, waited for)
          const unsigned stn = (unsigned)((buf + NST - 1) & (NST - 1)) * STG;
          if constexpr (VAR == 1) {
  #pragma unroll
            for (int i = 0; i < AR; ++i) dmaA(stn, i);
  #pragma unroll
            for (int i = 0; i < BR; ++i) dmaB(stn, i);
          }
          if constexpr (VAR == 3) __builtin_amdgcn_s_setprio(1);
  #pragma unroll
          for (int u = 0; u < U; ++u) {
            const int g = u / 4, st = u % 4;
            if (st == 2 && g + 1 < NG) fread(buf, g + 1, (g + 1) & 1);
            if constexpr (F16) {
              // groups g - 1 (slot 0) and g (slot 1) together are one 16-deep k block: issued once both have been read
              if ((g & 1) == 1 && st == 0) {
                f16x8 ha[TM], hb[TN];
  #pragma unroll
                for (int i = 0; i < TM; ++i) ha[i] = segsde_pack_f16(fa[0][i], fa[1][i]);
  #pragma unroll
                for (int j = 0; j < TN; ++j) hb[j] = segsde_pack_f16(fb[0][j], fb[1][j]);
  #pragma unroll
                for (int i = 0; i < TM; ++i)
  #pragma unroll
                  for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], (FIRST && u == 4) ? f32x16{} : acc[i][j], 0, 0, 0);
              }
            } else {
  #pragma unroll
            for (int i = 0; i < TM; ++i)
  #pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[g & 1][i], st), comp(fb[g & 1][j], st),
                                                                 (FIRST && u == 0) ? f32x16{} : acc[i][j], 0, 0, 0);
            }
            if constexpr (VAR != 1) {
  #pragma unroll
              for (int i = 0; i < AR; ++i)
                if (u == DSTEP * i) dmaA(stn, i);
  #pragma unroll
              for (int i = 0; i < BR; ++i)
                if (u == DSTEP * (AR + i)) dmaB(stn, i);
            }
            if constexpr (VAR != 2) __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (VAR == 3) __builtin_amdgcn_s_setprio(0);
          dma_end(kc + NST < nchunks, false);
          segsde_wait_vmcnt<INFLIGHT>();   // chunk kc+1 has landed (later chunks may still be in flight)
          __syncthreads();
        };
        if constexpr (NST == 2 && VAR != 7) {   // var=7: A/B knob (round-2 loop before this unroll and the kernarg refresh)
          chunk(0, std::integral_constant<int, 0>{}, std::true_type{});
          if (1 < nchunks) chunk(1, std::integral_constant<int, 1>{}, std::false_type{});
          for (int kc = 2; kc < nchunks; kc += 2) {
            chunk(kc, std::integral_constant<int, 0>{}, std::false_type{});
            if (kc + 1 < nchunks) chunk(kc + 1, std::integral_constant<int, 1>{}, std::false_type{});
          }
        } else {
          for (int kc = 0; kc < nchunks; ++kc) chunk(kc, kc & (NST - 1), std::false_type{});
        }
        if constexpr (INFLIGHT > 0) {      // the epilogue reuses the stages: nothing may still be landing
          segsde_wait_vmcnt0();
          __syncthreads();
        }
        return;
      }
      tap_update(wadj_tag, true);
      load_chunk(0);
      storeA(smem); storeB(smem + BM * LDT);
      load_chunk(1);
      tab_build(wadj_tag);
      __syncthreads();

      const int arow = wm * TM * 32 + (lane & 31), brow = wn * TN * 32 + (lane & 31), h = lane >> 5;
      const int sa = swz(arow), sb = swz(brow);
      auto fread = [&](int buf, int g, int slot) {
        const float* Ap = smem + buf * STAGE + arow * LDT;
        const float* Bp = smem + buf * STAGE + BM * LDT + brow * LDT;
  #pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const float4*>(Ap + i * 32 * LDT + 4 * ((2 * g + h) ^ sa));
  #pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * LDT + 4 * ((2 * g + h) ^ sb));
      };
      auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
      auto rchunk = [&](int kc, auto buf_c) {   // stage index as a compile-time constant, as in the LDS-DMA loop
        constexpr int buf = decltype(buf_c)::value;
        fread(buf, 0, 0);
        chunk_begin(kc + 2);
        float* Asn = smem + (buf ^ 1) * STAGE;
  #pragma unroll
        for (int u = 0; u < U; ++u) {
          const int g = u / 4, st = u % 4;
          if (st == 2 && g + 1 < NG) fread(buf, g + 1, (g + 1) & 1);
  #pragma unroll
          for (int i = 0; i < TM; ++i)
  #pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[g & 1][i], st), comp(fb[g & 1][j], st), acc[i][j], 0, 0, 0);
          if (u == 0) storeA(Asn);
          if (u == 1) storeB(Asn + BM * LDT);
  #pragma unroll
          for (int i = 0; i < AR; ++i)
            if (u == 2 + LSTEP * i) loadA(i);
          if (u == U - 2) {
  #pragma unroll
            for (int i = 0; i < BR; ++i) loadB(i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        chunk_end(false);
        __syncthreads();
      };
      for (int kc = 0; kc < nchunks; kc += 2) {
        rchunk(kc, std::integral_constant<int, 0>{});
        if (kc + 1 < nchunks) rchunk(kc + 1, std::integral_constant<int, 1>{});
      }
    };
    if constexpr (ADJ) {
      if (wave_bord) run(std::true_type{}); else run(std::false_type{});
    } else {
      run(std::false_type{});
    }
  } else {
    gload(0);
    lstore(0);
    gload(1);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
      mma_groups(kc & 1, 0, NG / 2);
      lstore((kc + 1) & 1);
      gload(kc + 2);
      mma_groups(kc & 1, NG / 2, NG);
      __syncthreads();
    }
  }

  // The epilogue's parameters (destinations, pitches, split point, statistics / activation-gradient pointers ...) are
  // re-read from the kernel-argument segment HERE: held in SGPRs across the K loop they pushed the adjoint kernel over the
  // SGPR budget -- 16 v_readlane + 2 scratch accesses per chunk inside its loop (and every VALU cycle is a matrix cycle).
  const ConvP pe = VAR == 7 ? p : SEGSDE_REFRESH_KERNARG(ConvP, p);   // var=7: A/B knob, no refresh
  // epilogue, staged variant: the accumulator tile goes through LDS (free after the K loop) so that every output row
  // leaves as 16-byte stores covering whole 512-byte (BN*4) row segments, instead of 4-byte stores per lane -- the
  // short-K layers (1x1 bottleneck convs and their gradients) are bound by exactly this store stream.
  if (pe.vecout && !pe.sum2x2) {
    float* Ct = smem;                      // [BM][BN] floats = 2 stages of (BM+BN)*BK only when BN <= 2*BK... checked on host
    const int col = lane & 31, rhalf = lane >> 5;
    if (!pe.bias && pe.act == SEGSDE_ACT_NONE) {
      // every BatchNorm-followed convolution and every data-gradient: the accumulators go to LDS as they are (one base
      // address per thread, immediate offsets, no VALU work)
      float* cw = Ct + (wm * TM * 32 + 4 * rhalf) * BN + wn * TN * 32 + col;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) cw[(i * 32 + (r & 3) + 8 * (r >> 2)) * BN + j * 32] = acc[i][j][r];
    } else {
      // one copy of the loop per activation kind, branch-free inside: the generic segsde_act(v, pe.act) cost ~45 VALU
      // instructions per accumulator (a libm expf behind a divergent branch, for each of 64 values)
      auto put = [&](auto f) {
        float* cw = Ct + (wm * TM * 32 + 4 * rhalf) * BN + wn * TN * 32 + col;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + (wn * TN + j) * 32 + col;
          const float bias = (pe.bias && n < pe.ne) ? pe.bias[n] : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cw[(i * 32 + (r & 3) + 8 * (r >> 2)) * BN + j * 32] = f(acc[i][j][r] + bias);
        }
      };
      // an accumulating launch (the skip-source half of an upsample-folded convolution) activates AFTER adding what the
      // destination holds: only the bias goes in here
      if (pe.accum) put([](float v) { return v; });
      else if (pe.act == SEGSDE_ACT_ELU) put([](float v) { return v > 0.f ? v : __expf(fminf(v, 0.f)) - 1.f; });
      else if (pe.act == SEGSDE_ACT_RELU) put([](float v) { return fmaxf(v, 0.f); });
      else if (pe.act == SEGSDE_ACT_SIGMOID) put([](float v) { return segsde_act(v, SEGSDE_ACT_SIGMOID); });
      else put([](float v) { return v; });
    }
    __syncthreads();
    if (pe.stats) {
      // each thread: one column x one slice of the tile's rows (conflict-free LDS reads), no cross-thread reduction --
      // the slices are simply more partial rows for the (double precision) reduction that follows
      constexpr int R = 256 / BN, RS = BM / R;
      const int scol = tid % BN, sl = tid / BN, sn = n0 + scol;
      // Double accumulation, eight independent chains.  var = E[x^2] - mean^2 cancels catastrophically in fp32 when a
      // channel's mean dominates its spread; and with exact (double) sums the statistics come out bit-identical to the
      // separate segsde_bn_stats pass, which the golden-vector tests of whole models rely on (tiny feature maps make
      // the gradients sensitive to the last bit of invstd).  A shifted-fp32 variant was no faster end to end.
      double s8[8], q8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s8[u] = 0.0; q8[u] = 0.0; }
      for (int ml = sl * RS; ml < (sl + 1) * RS; ml += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double v = (m0 + ml + u < pe.M) ? (double)Ct[(ml + u) * BN + scol] : 0.0;
          s8[u] += v; q8[u] += v * v;
        }
      }
      if (sn < pe.ne) {
        double* pr = pe.stats + ((long)(mt * R + sl) * 2) * pe.N + sn;
        pr[0] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        pr[pe.N] = ((q8[0] + q8[1]) + (q8[2] + q8[3])) + ((q8[4] + q8[5]) + (q8[6] + q8[7]));
      }
    }
    constexpr int CQ = BN / 4, RPP = 256 / CQ;    // float4 columns per row, rows per pass
    const int cq = tid % CQ, rr = tid / CQ;
    const int n = n0 + 4 * cq;
    {
      // Common case -- a full tile whose rows are consecutive rows of ONE destination tensor: the stores (and the loads of
      // the accumulate / activation-gradient variants) are raw buffer operations on a resource placed at the tile's first
      // row, the per-thread byte offset is computed once and the row advance rides in the scalar offset.  The general
      // code below spends ~45 VALU instructions per row segment on 64-bit addresses, bounds and out_row -- 700 per thread
      // on a 128x128 tile, more than one per MFMA of an 8-chunk 1x1 layer, and VALU cycles are matrix-pipe cycles here.
      const int nend = n0 + BN < pe.ne ? n0 + BN : pe.ne;
      const bool side1 = n0 >= pe.nsplit;
      if ((!pe.submap || pe.osfast) && m0 + BM <= pe.M && (side1 || nend <= pe.nsplit)) {
        constexpr int NR = BM / RPP;
        // strided sub-grid (osfast: the tile lies inside one sub-grid row): consecutive GEMM rows are os destination rows apart
        const long row0 = pe.submap ? out_row(pe, m0) : (long)m0;
        float* dbase = side1 ? pe.y2 + (row0 * pe.ldy2 + (n0 - pe.nsplit)) : pe.y + (row0 * pe.ldy + n0);
        const unsigned ld = (unsigned)(side1 ? pe.ldy2 : pe.ldy) * (unsigned)pe.os;
        const segsde_rsrc rd = segsde_make_rsrc(dbase);
        const unsigned vo = n < pe.ne ? ((unsigned)rr * ld + 4u * cq) * 4u : SEGSDE_OOB;
        const unsigned step = (unsigned)RPP * ld * 4u;
        const bool ag = pe.agy && !side1;
        const unsigned agl = (unsigned)pe.agld * (unsigned)pe.os;
        const segsde_rsrc ra = segsde_make_rsrc(ag ? pe.agy + (row0 * pe.agld + n0) : pe.zero);
        const unsigned voa = (ag && n < pe.ne) ? ((unsigned)rr * agl + 4u * cq) * 4u : SEGSDE_OOB;
        const unsigned stepa = (unsigned)RPP * agl * 4u;
        const float* cp = Ct + rr * BN + 4 * cq;
        float4 o[NR];
        if (pe.accum) {
          unsigned so = 0;
#pragma unroll
          for (int t = 0; t < NR; ++t) { o[t] = segsde_buffer_load4(rd, vo, so); so += step; }
        }
        unsigned so = 0, soa = 0;
        if (pe.accum && pe.act != SEGSDE_ACT_NONE) {
          // the skip-source launch of an upsample-folded forward: add what the class launches left, THEN activate.  A loop of
          // its own: with the activation inside the shared loop below the accumulating 1x1 data-gradients (1024 -> 256,
          // 8 chunks, epilogue-bound) lost 12 % (113 -> 100 TF, measured)
#pragma unroll
          for (int t = 0; t < NR; ++t) {
            float4 v = *reinterpret_cast<const float4*>(cp + t * RPP * BN);
            v.x += o[t].x; v.y += o[t].y; v.z += o[t].z; v.w += o[t].w;
            if (pe.act == SEGSDE_ACT_ELU) {
              v.x = v.x > 0.f ? v.x : __expf(fminf(v.x, 0.f)) - 1.f; v.y = v.y > 0.f ? v.y : __expf(fminf(v.y, 0.f)) - 1.f;
              v.z = v.z > 0.f ? v.z : __expf(fminf(v.z, 0.f)) - 1.f; v.w = v.w > 0.f ? v.w : __expf(fminf(v.w, 0.f)) - 1.f;
            } else {
              v.x = segsde_act(v.x, pe.act); v.y = segsde_act(v.y, pe.act); v.z = segsde_act(v.z, pe.act); v.w = segsde_act(v.w, pe.act);
            }
            segsde_buffer_store4(rd, vo, so, v);
            so += step;
          }
          return;
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) {
          float4 v = *reinterpret_cast<const float4*>(cp + t * RPP * BN);
          if (ag) {
            const float4 yv = segsde_buffer_load4(ra, voa, soa);
            v.x *= segsde_act_grad_from_out(yv.x, pe.agkind); v.y *= segsde_act_grad_from_out(yv.y, pe.agkind);
            v.z *= segsde_act_grad_from_out(yv.z, pe.agkind); v.w *= segsde_act_grad_from_out(yv.w, pe.agkind);
          }
          if (pe.accum) { v.x += o[t].x; v.y += o[t].y; v.z += o[t].z; v.w += o[t].w; }
          if (pe.nt) segsde_buffer_store4_nt(rd, vo, so, v); else segsde_buffer_store4(rd, vo, so, v);
          so += step; soa += stepa;
        }
        return;
      }
    }
    if (n < pe.ne) {
      float* dst; long ld; int nn;
      if (n < pe.nsplit) { dst = pe.y; ld = pe.ldy; nn = n; }
      else { dst = pe.y2; ld = pe.ldy2; nn = n - pe.nsplit; }
      if (pe.accum) {
        // y += tile: every read-modify-write of a row segment depends on a global load; issue all of a thread's loads
        // first (the accumulators are in LDS by now, registers are free) so that their latency is paid once, not
        // BM / RPP times in a row -- on the 8-chunk 1x1 data-gradients this epilogue was a third of the tile's time
        constexpr int NR = BM / RPP;
        float4 o[NR];
        const float* ag = (pe.agy && n < pe.nsplit) ? pe.agy + nn : nullptr;
#pragma unroll
        for (int t = 0; t < NR; ++t) {
          const int m = m0 + rr + t * RPP;
          o[t] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < pe.M) o[t] = *reinterpret_cast<const float4*>(dst + out_row(pe, m) * ld + nn);
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) {
          const int ml = rr + t * RPP, m = m0 + ml;
          if (m < pe.M) {
            float4 v = *reinterpret_cast<const float4*>(Ct + ml * BN + 4 * cq);
            if (ag) {
              const float4 yv = *reinterpret_cast<const float4*>(ag + out_row(pe, m) * pe.agld);
              v.x *= segsde_act_grad_from_out(yv.x, pe.agkind); v.y *= segsde_act_grad_from_out(yv.y, pe.agkind);
              v.z *= segsde_act_grad_from_out(yv.z, pe.agkind); v.w *= segsde_act_grad_from_out(yv.w, pe.agkind);
            }
            v.x += o[t].x; v.y += o[t].y; v.z += o[t].z; v.w += o[t].w;
            if (pe.act == SEGSDE_ACT_ELU) {      // same expression as the staged path (accumulate, then activate)
              v.x = v.x > 0.f ? v.x : __expf(fminf(v.x, 0.f)) - 1.f; v.y = v.y > 0.f ? v.y : __expf(fminf(v.y, 0.f)) - 1.f;
              v.z = v.z > 0.f ? v.z : __expf(fminf(v.z, 0.f)) - 1.f; v.w = v.w > 0.f ? v.w : __expf(fminf(v.w, 0.f)) - 1.f;
            } else if (pe.act != SEGSDE_ACT_NONE) {
              v.x = segsde_act(v.x, pe.act); v.y = segsde_act(v.y, pe.act); v.z = segsde_act(v.z, pe.act); v.w = segsde_act(v.w, pe.act);
            }
            *reinterpret_cast<float4*>(dst + out_row(pe, m) * ld + nn) = v;
          }
        }
      } else if (pe.agy && n < pe.nsplit) {
        const float* ag = pe.agy + nn;
#pragma unroll 4
        for (int ml = rr; ml < BM; ml += RPP) {
          const int m = m0 + ml;
          if (m < pe.M) {
            const long row = out_row(pe, m);
            float4 v = *reinterpret_cast<const float4*>(Ct + ml * BN + 4 * cq);
            const float4 yv = *reinterpret_cast<const float4*>(ag + row * pe.agld);
            v.x *= segsde_act_grad_from_out(yv.x, pe.agkind); v.y *= segsde_act_grad_from_out(yv.y, pe.agkind);
            v.z *= segsde_act_grad_from_out(yv.z, pe.agkind); v.w *= segsde_act_grad_from_out(yv.w, pe.agkind);
            *reinterpret_cast<float4*>(dst + row * ld + nn) = v;
          }
        }
      } else {
#pragma unroll 4
        for (int ml = rr; ml < BM; ml += RPP) {
          const int m = m0 + ml;
          if (m < pe.M) {
            float4* q = reinterpret_cast<float4*>(dst + out_row(pe, m) * ld + nn);
            *q = *reinterpret_cast<const float4*>(Ct + ml * BN + 4 * cq);
          }
        }
      }
    }
    return;
  }

  // Fused 2x2 sum (data-gradient through the nearest x2 upsample), full tile on the summed side of the split: the four
  // members of a block are four registers of one lane (patch-major rows); the BM/4 x BN tile of sums goes through LDS and
  // leaves as 16-byte buffer stores, the activation derivative (saved output at half resolution) is read the same way --
  // the per-value path below pays 64-bit address arithmetic and a scattered 4-byte load + store for every sum
  if (pe.sum2x2 && pe.vecout && m0 + BM <= pe.M && (n0 + BN < pe.ne ? n0 + BN : pe.ne) <= pe.nsplit) {
    float* Ct = smem;                      // [BM / 4][BN] floats
    constexpr int CQ = BN / 4, RPP = 256 / CQ, NR = (BM / 4) / RPP;
    const int cq = tid % CQ, rr = tid / CQ, n = n0 + 4 * cq;
    // the saved activation outputs are requested first: they do not depend on the accumulators, and their latency then
    // overlaps the trip of the sums through LDS
    const bool ag = pe.agy != nullptr;
    const segsde_rsrc ra = segsde_make_rsrc(ag ? pe.agy + ((long)(m0 >> 2) * pe.agld + n0) : pe.zero);
    const unsigned voa = (ag && n < pe.ne) ? ((unsigned)rr * (unsigned)pe.agld + 4u * cq) * 4u : SEGSDE_OOB;
    float4 yv[NR];
    if (ag) {
      unsigned soa = 0;
#pragma unroll
      for (int t = 0; t < NR; ++t) { yv[t] = segsde_buffer_load4(ra, voa, soa); soa += (unsigned)RPP * (unsigned)pe.agld * 4u; }
    }
    {
      const int col = lane & 31, rhalf = lane >> 5;
      float* cw = Ct + (wm * TM * 8 + rhalf) * BN + wn * TN * 32 + col;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            cw[(i * 8 + 2 * g) * BN + j * 32] =
                (acc[i][j][4 * g] + acc[i][j][4 * g + 1]) + (acc[i][j][4 * g + 2] + acc[i][j][4 * g + 3]);
    }
    __syncthreads();
    const unsigned ld = (unsigned)pe.ldy;
    const segsde_rsrc rd = segsde_make_rsrc(pe.y + ((long)(m0 >> 2) * pe.ldy + n0));
    const unsigned vo = n < pe.ne ? ((unsigned)rr * ld + 4u * cq) * 4u : SEGSDE_OOB;
    unsigned so = 0;
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      float4 v = *reinterpret_cast<const float4*>(Ct + (rr + t * RPP) * BN + 4 * cq);
      if (ag) {
        v.x *= segsde_act_grad_from_out(yv[t].x, pe.agkind); v.y *= segsde_act_grad_from_out(yv[t].y, pe.agkind);
        v.z *= segsde_act_grad_from_out(yv[t].z, pe.agkind); v.w *= segsde_act_grad_from_out(yv[t].w, pe.agkind);
      }
      segsde_buffer_store4(rd, vo, so, v);
      so += (unsigned)RPP * ld * 4u;
    }
    return;
  }

  // epilogue: bias + activation, channel-split store (concat data-gradients go to two tensors)
  const int col = lane & 31, rhalf = lane >> 5;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + col;
    if (n >= pe.ne) continue;
    const float bias = pe.bias ? pe.bias[n] : 0.f;
    float* dst; long ld; int nn;
    if (n < pe.nsplit) { dst = pe.y; ld = pe.ldy; nn = n; }
    else { dst = pe.y2; ld = pe.ldy2; nn = n - pe.nsplit; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (pe.sum2x2) {
        // data-gradient through the nearest x2 upsample: channels of source 0 are summed over each 2x2 block in
        // registers and stored at half resolution; skip-connection channels are stored per pixel
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = m0 + (wm * TM + i) * 32 + 8 * g + 4 * rhalf;
          if (m >= pe.M) continue;
          if (n < pe.nsplit) {
            float v = (acc[i][j][4 * g] + acc[i][j][4 * g + 1]) + (acc[i][j][4 * g + 2] + acc[i][j][4 * g + 3]);
            if (pe.agy) v *= segsde_act_grad_from_out(pe.agy[(long)(m >> 2) * pe.agld + nn], pe.agkind);
            dst[(long)(m >> 2) * ld + nn] = v;
          } else {
            int b, hb, wb; bool ok;
            decode_m(pe, m, b, hb, wb, ok);
            float* q = dst + ((long)(b * pe.Ho + hb) * pe.Wo + wb) * ld + nn;   // sub-pixel 0 of the block
            q[0] = acc[i][j][4 * g]; q[ld] = acc[i][j][4 * g + 1];
            q[(long)pe.Wo * ld] = acc[i][j][4 * g + 2]; q[(long)pe.Wo * ld + ld] = acc[i][j][4 * g + 3];
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
          if (m < pe.M) {
            float v = segsde_act(acc[i][j][r] + bias, pe.act);
            if (pe.agy && n < pe.nsplit) v *= segsde_act_grad_from_out(pe.agy[out_row(pe, m) * pe.agld + nn], pe.agkind);
            dst[out_row(pe, m) * ld + nn] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight-gradient kernel: partial[z][k][n] = sum over this split's pixels of A(m,k) * dY[m,n]
// ---------------------------------------------------------------------------------------------------
// k -> (tap, channel): tap-major (tap, [src0|src1]) by default; source-major (all taps of source 0, then all taps of
// source 1) when srcC0 > 0 -- the table-driven weight-gradient kernel orders two-source reductions that way so that a
// 128-column tile does not straddle the concat boundary inside every tap
__device__ __forceinline__ void wgrad_k_decode(int k, int Ctot, int taps, int srcC0, int& tap, int& c) {
  if (srcC0 > 0) {
    const int n0 = taps * srcC0;
    if (k < n0) { tap = k / srcC0; c = k - tap * srcC0; }
    else { const int C1 = Ctot - srcC0, kk = k - n0; tap = kk / C1; c = srcC0 + kk - tap * C1; }
  } else { tap = k / Ctot; c = k - tap * Ctot; }
}

constexpr int BP = 32;  // pixels per staged chunk

// Reduction of the split partials inside the weight-gradient kernel (round-3 EXPERIMENT, off by default: SEGSDE_TUNE="wred=1").
// Every workgroup stores its partial slab, fences, and takes a ticket of its (k-tile, n-tile); the workgroup that draws the
// LAST ticket sums all slabs of the tile in slab order -- the same order whichever workgroup does it, so the result is
// deterministic -- writes OIHW and puts the ticket back to zero.  Correct (119 GPU tests), but MEASURED SLOWER: the
// device-scope release fence every workgroup needs writes its XCD's whole L2 back (8 XCDs, no cross-XCD L2 coherence):
// conv_wgrad 398 -> 648 us per launch, the step 334 -> 420 ms together with the same trick in the column reductions
// (profiles/experiments_r03.md).  A kernel boundary pays that write-back once per launch; the 184 wgrad_reduce launches of
// ~14 us stay.  tickets == nullptr (default): partial slabs only, a separate reduce kernel follows.
struct WRed { unsigned* tickets; float* dw; int CtotDst, cOff, taps, srcC0; };

// MODEX >= 16 (with MODE 5): half-precision operands, like conv_igemm_kernel's VARX >= 16 -- the eight scalars a lane reads for
// eight consecutive fp32 MFMAs of a 16-pixel block (pixels 2 u + lane half of two fragment groups) become ONE operand of
// v_mfma_f32_32x32x16_f16; both operands are read with the same pattern, so the k slots agree
template <int BKT, int BN, int WM, int WN, int MODEX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(ConvP p, const float* dy, int lddy, float* part,
                                                         int chunks_per_split, WRed wr) {
  constexpr int MODE = MODEX & 15;
  constexpr bool F16 = MODEX >= 16;
  static_assert(!F16 || MODE == 5, "half-precision operands: the pipelined LDS-DMA loop only");
  // MODE 0: scalar gather, 1: float4 gather, 2: FAST A side + vector dY + rows at least 32 pixels wide (straight-line
  // loop), 3: FAST A side with the general row walk / scalar dY (odd Cout, tiny feature maps), 4: MODE 2 with the tile
  // loads writing LDS themselves (LDS-DMA, see the forward kernel), 5: MODE 4 with the chunk's barrier moved into the chunk
  // (round 4: tile loads run two chunks ahead, the first fragments of the next chunk are read before the chunk boundary)
  constexpr bool VEC = MODE >= 1;
  constexpr bool FAST = MODE >= 2;
  constexpr bool SIMPLE = MODE == 2 || MODE == 4 || MODE == 5;
  constexpr bool DMA = MODE == 4 || MODE == 5;
  constexpr bool PIPE = MODE == 5;
  constexpr int TM = BKT / (WM * 32), TN = BN / (WN * 32);
  constexpr int AQ = BKT / 4, DQ = BN / 4;             // float4 columns per tile row
  constexpr int AI = (BP * AQ) / 256, DI = (BP * DQ) / 256;
  constexpr int STAGE = BP * (BKT + BN);
  static_assert(WM * WN == 4, "4 waves per workgroup");
  SEGSDE_SMEM;
  float* smem = reinterpret_cast<float*>(segsde_smem);

  // XCD-aware tile order (1-D launch): the hardware deals consecutive workgroups round-robin over the 8 XCDs; after the
  // remap every XCD owns a contiguous range of (split, n-tile, k-tile) triples with the k-tile fastest, so the
  // workgroups that read the same dY pixel range (all k-tiles of one split) share one L2 instead of fetching it
  // through eight (rocprofv3 FETCH_SIZE showed ~8x the algorithmic reads on the fabric side before)
  const int nkt = (p.Ktot + BKT - 1) / BKT, nnt = (p.N + BN - 1) / BN;
  const int tile = segsde_xcd_remap(blockIdx.x, gridDim.x);
  const int kt = tile % nkt, rest = tile / nkt, nt = rest % nnt, zt = rest / nnt;
  const int k0 = kt * BKT, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;

  const int nchunks_total = (p.M + BP - 1) / BP;
  const int c_begin = zt * chunks_per_split;
  const int c_end = min(nchunks_total, c_begin + chunks_per_split);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // FAST: this thread's reduction column (tap, source, channel) is fixed for the whole kernel; its AI pixel rows
  // advance by BP pixels per chunk with carries instead of divisions; loads are clamped + selected (branch-free)
  const int fk = k0 + 4 * (tid % AQ);
  const bool fk_ok = fk < p.Ktot;
  SrcSel fs; int fdh = 0, fdw = 0;
  int fb[AI], fh[AI], fw[AI], fm[AI];
  if (FAST && !SIMPLE) {
    const int kk = fk_ok ? fk : 0;
    const int tap = kk / p.Ctot, c = kk - tap * p.Ctot;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    fs = select_src(p, c);
    fdh = kh * p.dil - p.pad; fdw = kw * p.dil - p.pad;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      bool ok;
      fm[i] = c_begin * BP + (tid + 256 * i) / AQ;
      decode_m(p, fm[i], fb[i], fh[i], fw[i], ok);   // fh/fw are ho*stride, wo*stride
    }
  }
  // MODE 2 (Wo % 32 == 0: a chunk is 32 consecutive pixels of ONE image row): the A-side offsets come from four small
  // LDS tables -- byte offset of padded input row hi / column wi inside one image of source 0 / 1, or TAB_MARK on zero
  // padding -- so that a chunk's tile loads cost ~10 VALU instructions per thread instead of ~180 (fp32 MFMA and VALU
  // share issue cycles on gfx950).  voff = Htab[h*stride + kh*dil] + Wtab[(w0 + row)*stride + kw*dil] + channel; the
  // image base and the dY row base ride in per-chunk buffer resources (scalar work only).
  constexpr unsigned TAB_MARK = 0x40000000u;     // = num_records of the resources: one image is at most 1 GB (host check);
                                                 // two markers + a legitimate offset stay below 2^32 (no wrap-around)
  unsigned* tabs = reinterpret_cast<unsigned*>(smem + 2 * STAGE);
  const int He = (p.Ho - 1) * p.stride + (p.KH - 1) * p.dil + 1, We = (p.Wo - 1) * p.stride + (p.KW - 1) * p.dil + 1;
  unsigned thaddr = 0, twaddr[AI], tcc = 0, voffD[DI];
  bool tsrc0 = true;
  int wave_src = 0;                              // 0 / 1: every column of this wave lies in that source; 2: mixed
  int cb = 0, chh = 0, cw = 0;                   // image / output row / first output column of the next chunk to load
  // per-image geometry of the two sources (plain wave-uniform scalars: they feed SGPR buffer resources)
  const unsigned tWs0 = (unsigned)(p.W >> p.up0), tld0 = (unsigned)p.ld0, tld1 = (unsigned)p.ld1;
  const size_t tbs0 = (size_t)(p.H >> p.up0) * tWs0 * tld0, tbs1 = (size_t)p.H * p.W * tld1;
  if constexpr (SIMPLE) {
    const bool refl = p.pad_mode == SEGSDE_PAD_REFLECT || p.pad_mode == SEGSDE_PAD_CLAMP_, clampm = p.pad_mode == SEGSDE_PAD_CLAMP_;
    for (int j = tid; j < He; j += 256) {
      const int hi = j - p.pad;
      const bool ok = refl || (unsigned)hi < (unsigned)p.H;
      const int hr = clampm ? (hi < 0 ? 0 : (hi >= p.H ? p.H - 1 : hi)) : (hi < 0 ? -hi : (hi >= p.H ? 2 * p.H - 2 - hi : hi));
      tabs[j] = ok ? (unsigned)(hr >> p.up0) * tWs0 * tld0 * 4u : TAB_MARK;
      tabs[He + We + j] = ok ? (unsigned)hr * (unsigned)p.W * tld1 * 4u : TAB_MARK;
    }
    for (int j = tid; j < We; j += 256) {
      const int wi = j - p.padw;
      const bool ok = refl || (unsigned)wi < (unsigned)p.W;
      const int wr = clampm ? (wi < 0 ? 0 : (wi >= p.W ? p.W - 1 : wi)) : (wi < 0 ? -wi : (wi >= p.W ? 2 * p.W - 2 - wi : wi));
      tabs[He + j] = ok ? (unsigned)(wr >> p.up0) * tld0 * 4u : TAB_MARK;
      tabs[2 * He + We + j] = ok ? (unsigned)wr * tld1 * 4u : TAB_MARK;
    }
    const int kk = fk_ok ? fk : 0;
    int tap, c;
    wgrad_k_decode(kk, p.Ctot, p.KH * p.KW, p.C1 > 0 ? p.C0 : 0, tap, c);   // source-major for two-source inputs
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    tsrc0 = c < p.C0;
    const int tb = tsrc0 ? 0 : He + We;
    thaddr = (unsigned)(tb + kh * p.dil) * 4u;
#pragma unroll
    for (int i = 0; i < AI; ++i) twaddr[i] = (unsigned)(tb + He + kw * p.dil + ((tid + 256 * i) / AQ) * p.stride) * 4u;
    tcc = fk_ok ? (unsigned)(tsrc0 ? c : c - p.C0) * 4u : 0x80000000u;
    wave_src = __builtin_amdgcn_readfirstlane(__all(tsrc0 || !fk_ok) ? 0 : (__all(!tsrc0 || !fk_ok) ? 1 : 2));
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int e = tid + 256 * i, row = e / DQ, nq = e - row * DQ;
      voffD[i] = n0 + 4 * nq < p.N ? (unsigned)(row * p.os * lddy + n0 + 4 * nq) * 4u : SEGSDE_OOB;   // os > 1: dY on a strided sub-grid
    }
    int b_, h_, w_; bool ok_;
    decode_m(p, c_begin * BP < p.M ? c_begin * BP : 0, b_, h_, w_, ok_);
    cb = __builtin_amdgcn_readfirstlane(b_);
    chh = __builtin_amdgcn_readfirstlane(h_ / p.stride);
    cw = __builtin_amdgcn_readfirstlane(w_ / p.stride);
    __syncthreads();
  }
  const int wstep = BP * p.stride, wlim = p.Wo * p.stride, hlim = p.Ho * p.stride;
  const bool dyvec = SIMPLE || ((p.N % 4 == 0) && (lddy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0));

  float4 ra[AI], rd[DI];
  const bool wide = SIMPLE || p.Wo >= BP;   // a 32-pixel step wraps at most one image row: carries, not divisions
  // MODE 2 pieces of one chunk's tile loads, issued separately so they can be dealt out between the MFMA units
  unsigned thv = 0, twv[AI];
  auto tload = [&]() {      // table lookups for the chunk at (cb, chh, cw)
    const char* tb = reinterpret_cast<const char*>(tabs);
    thv = *reinterpret_cast<const unsigned*>(tb + thaddr + (unsigned)(chh * p.stride) * 4u);
#pragma unroll
    for (int i = 0; i < AI; ++i) twv[i] = *reinterpret_cast<const unsigned*>(tb + twaddr[i] + (unsigned)(cw * p.stride) * 4u);
  };
  auto aload = [&](int c) {
    const bool live = c < nchunks_total;       // chunks past the last pixel: zero records, every lane reads zeros
    // the 128 reduction columns of a tile usually lie in one source: a wave-uniform resource select; a tile that
    // straddles the concat boundary loads from both with the other source's lanes out of range, and ORs the halves
    if (wave_src != 2) {
      const segsde_rsrc rr = segsde_make_rsrc(wave_src == 0 ? p.x0 + (size_t)cb * tbs0 : p.x1 + (size_t)cb * tbs1, live ? TAB_MARK : 0u);
#pragma unroll
      for (int i = 0; i < AI; ++i) ra[i] = segsde_buffer_load4(rr, thv + twv[i] + tcc, 0u);
    } else {
      const segsde_rsrc r0 = segsde_make_rsrc(p.x0 + (size_t)cb * tbs0, live ? TAB_MARK : 0u);
      const segsde_rsrc r1 = segsde_make_rsrc(p.x1 + (size_t)cb * tbs1, live ? TAB_MARK : 0u);
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const unsigned vo = thv + twv[i] + tcc;
        const float4 a = segsde_buffer_load4(r0, tsrc0 ? vo : SEGSDE_OOB, 0u);
        const float4 b = segsde_buffer_load4(r1, tsrc0 ? SEGSDE_OOB : vo, 0u);
        ra[i] = make_float4(__uint_as_float(__float_as_uint(a.x) | __float_as_uint(b.x)), __uint_as_float(__float_as_uint(a.y) | __float_as_uint(b.y)),
                            __uint_as_float(__float_as_uint(a.z) | __float_as_uint(b.z)), __uint_as_float(__float_as_uint(a.w) | __float_as_uint(b.w)));
      }
    }
  };
  // first dY row of the chunk at (cb, chh, cw): pixel c * BP of the plain problem; for the class launches of an
  // upsample-folded weight gradient (os = 2) the pixels of a chunk are every second pixel of row os * chh + oph of the
  // full-resolution gradient image
  auto dybase = [&](int c) -> const float* {
    if (p.os == 1) return dy + (size_t)c * BP * lddy;
    // 32-bit pixel index (B * H * W < 2^31, host check), ONE widening multiply: stays on the scalar unit (a 64 x 32-bit
    // product would go through the vector ALU and leave the buffer resource in VGPRs)
    const unsigned pix = (unsigned)((cb * p.OHf + p.os * chh + p.oph) * p.OWf + p.os * cw + p.opw);
    return dy + (size_t)pix * (unsigned)lddy;
  };
  auto dload = [&](int c) {
    const segsde_rsrc rd_ = segsde_make_rsrc(dybase(c), c < nchunks_total ? 0x7fffffffu : 0u);
#pragma unroll
    for (int i = 0; i < DI; ++i) rd[i] = segsde_buffer_load4(rd_, voffD[i], 0u);
    cw += BP;
    if (cw == p.Wo) { cw = 0; if (++chh == p.Ho) { chh = 0; ++cb; } }
  };
  auto gload = [&](int c) {
    if constexpr (SIMPLE) {
      tload();
      aload(c);
      dload(c);
      return;
    }
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        ra[i] = fast_fetch(p, fs, fb[i], fh[i] + fdh, fw[i] + fdw, fk_ok && fm[i] < p.M, 0);
        fm[i] += BP;
        if (wide) {
          fw[i] += wstep;
          const bool cw = fw[i] >= wlim;
          fw[i] -= cw ? wlim : 0; fh[i] += cw ? p.stride : 0;
          const bool ch = fh[i] >= hlim;
          fh[i] -= ch ? hlim : 0; fb[i] += ch ? 1 : 0;
        } else {
          bool ok;
          decode_m(p, fm[i], fb[i], fh[i], fw[i], ok);
        }
      }
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        const int e = tid + 256 * i, row = e / DQ, nq = e - row * DQ;
        const int m = c * BP + row, n = n0 + 4 * nq;
        float4 v;
        if (dyvec) {
          const bool ok = m < p.M && n < p.N;
          v = *reinterpret_cast<const float4*>(ok ? dy + (long)m * lddy + n : p.zero);
        } else {   // odd Cout (19 classes, 1 disparity channel): scalar, clamped
          const float* src = dy + (long)(m < p.M ? m : 0) * lddy;
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { const bool ok = m < p.M && n + j < p.N; const float x = src[ok ? n + j : 0]; t[j] = ok ? x : 0.f; }
          v = make_float4(t[0], t[1], t[2], t[3]);
        }
        rd[i] = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + 256 * i, row = e / AQ, kq = e - row * AQ;
      int b, hb, wb; bool ok;
      decode_m(p, c * BP + row, b, hb, wb, ok);
      ra[i] = fetch_a4<VEC>(p, k0 + 4 * kq, b, hb, wb, ok);
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int e = tid + 256 * i, row = e / DQ, nq = e - row * DQ;
      const int m = c * BP + row, n = n0 + 4 * nq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M) {
        const float* src = dy + (long)m * lddy + n;
        if (VEC) { if (n < p.N) v = *reinterpret_cast<const float4*>(src); }
        else {
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = (n + j < p.N) ? src[j] : 0.f;
          v = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
      rd[i] = v;
    }
  };
  auto lstore = [&](int buf) {
    float* At = smem + buf * STAGE;
    float* Dt = At + BP * BKT;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + 256 * i, row = e / AQ, kq = e - row * AQ;
      *reinterpret_cast<float4*>(At + row * BKT + 4 * kq) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int e = tid + 256 * i, row = e / DQ, nq = e - row * DQ;
      *reinterpret_cast<float4*>(Dt + row * BN + 4 * nq) = rd[i];
    }
  };

  auto mma_steps = [&](int buf, int s0, int s1) {
    const float* At = smem + buf * STAGE;
    const float* Dt = At + BP * BKT;
    // one base register per 32-column fragment, opaque to the compiler: reads of consecutive k-steps (a multiple of 256 B
    // apart) then pair into ds_read2st64_b32 with immediate offsets; paired across fragments (128 B apart) every pair
    // needed its own VALU add for the base -- and VALU cycles are MFMA cycles here
    const float* Ap[TM]; const float* Dp[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { int o = (lane >> 5) * BKT + wm * TM * 32 + (lane & 31) + i * 32; SEGSDE_OPAQUE(o); Ap[i] = At + o; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { int o = (lane >> 5) * BN + wn * TN * 32 + (lane & 31) + j * 32; SEGSDE_OPAQUE(o); Dp[j] = Dt + o; }
    // fetch the fragments of 4 k-steps at a time ahead of their MFMAs (counted lgkmcnt waits) instead of read-wait-use
    // per step, which exposed the LDS latency every four MFMAs
#pragma unroll
    for (int sb = s0; sb < s1; sb += 4) {
      float a[4][TM], d[4][TN];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[u][i] = Ap[i][2 * (sb + u) * BKT];
#pragma unroll
        for (int j = 0; j < TN; ++j) d[u][j] = Dp[j][2 * (sb + u) * BN];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the read batch ahead of the MFMAs (the scheduler sinks it otherwise)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], d[u][j], acc[i][j], 0, 0, 0);
    }
  };
  // same software pipeline as the forward kernel: LDS stores + the loads of chunk c+2 sit between the two halves of
  // chunk c's MFMAs; rows past M load the zero page, so running one or two chunks past c_end is harmless
  if constexpr (DMA) {
    // Two LDS stages filled by the loads themselves.  The plain row-major [pixel][channel] image is lane-linear as it is:
    // thread e = tid + 256 i owns float4 column e % Q of tile row e / Q, so the 64 lanes of a wave cover 64 / Q consecutive
    // rows = 1 KiB (Q = 32: two 512-byte rows; Q = 16: four 256-byte rows).  A wave whose 128 reduction columns straddle
    // the concat boundary (wave_src == 2) has to OR two loads and keeps the register path for its A pieces.
    const unsigned lds0 = segsde_lds_addr(smem);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned STG = STAGE * 4;
    // Dead rows (ConvP::tapskip): with zero padding and a dilated window the tap rows [khl, khh] this workgroup's reduction
    // columns belong to read source rows h*stride + kh*dil - pad; an output row h for which that lies outside the image for
    // all of them contributes nothing to this tile -- its chunks (A side AND dY side) are never loaded.  Live rows are the
    // interval [hlo, hhi] of every image; the chunk walk jumps from the end of row hhi to row hlo of the next image.
    const int cpr = p.Wo / BP;                    // chunks per output row (SIMPLE: Wo % BP == 0)
    int hlo = 0, hhi = p.Ho - 1;
    if (p.tapskip && p.C1 == 0) {
      const int kper = p.KW * p.Ctot;
      const int khl = k0 / kper, khh = ((k0 + BKT < p.Ktot ? k0 + BKT : p.Ktot) - 1) / kper;
      const int lo = p.pad - khh * p.dil, hi = p.H - 1 + p.pad - khl * p.dil;     // live: lo <= h*stride <= hi
      hlo = lo > 0 ? (lo + p.stride - 1) / p.stride : 0;
      hhi = hi < 0 ? -1 : (hi / p.stride < p.Ho - 1 ? hi / p.stride : p.Ho - 1);
      hlo = __builtin_amdgcn_readfirstlane(hlo); hhi = __builtin_amdgcn_readfirstlane(hhi);
    }
    const int nlr = hhi - hlo + 1;                // live rows per image (<= 0: none)
    auto live_before = [&](int c) {               // live chunks among the raw chunks [0, c)
      const int per = p.Ho * cpr, q = c / per, r = c - q * per;
      int t = r - hlo * cpr;
      t = t < 0 ? 0 : (t > nlr * cpr ? nlr * cpr : t);
      return q * nlr * cpr + t;
    };
    const int nlive = nlr <= 0 ? 0 : (nlr == p.Ho ? c_end - c_begin : live_before(c_end) - live_before(c_begin));
    int cn = c_begin;                             // raw index of the next chunk to load, = chunk (cb, chh, cw)
    if (nlive > 0) {                              // settle on the first live chunk at or after c_begin
      if (chh < hlo) { cn += (hlo - chh) * cpr - cw / BP; chh = hlo; cw = 0; }
      else if (chh > hhi) { cn += (p.Ho - chh + hlo) * cpr - cw / BP; chh = hlo; cw = 0; ++cb; }
    }
    auto issue = [&](unsigned stage) {            // all tile loads of chunk cn into stage (byte offset); advances to the next live chunk
      const int c = cn;
      const bool live = c < nchunks_total;
      if (wave_src != 2) {
        const segsde_rsrc rr = segsde_make_rsrc(wave_src == 0 ? p.x0 + (size_t)cb * tbs0 : p.x1 + (size_t)cb * tbs1, live ? TAB_MARK : 0u);
#pragma unroll
        for (int i = 0; i < AI; ++i)
          segsde_buffer_load4_lds(rr, thv + twv[i] + tcc, 0u, lds0 + stage + (unsigned)(((256 / AQ) * i + (64 / AQ) * wv) * BKT * 4));
      } else {
        aload(c);
        float* At = smem + (stage / 4);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const int e = tid + 256 * i, row = e / AQ, kq = e - row * AQ;
          *reinterpret_cast<float4*>(At + row * BKT + 4 * kq) = ra[i];
        }
      }
      const segsde_rsrc rd_ = segsde_make_rsrc(dybase(c), live ? 0x7fffffffu : 0u);
#pragma unroll
      for (int i = 0; i < DI; ++i)
        segsde_buffer_load4_lds(rd_, voffD[i], 0u, lds0 + stage + (unsigned)((BP * BKT + ((256 / DQ) * i + (64 / DQ) * wv) * BN) * 4));
      cw += BP; ++cn;
      if (cw == p.Wo) {
        cw = 0;
        if (++chh > hhi) { cn += (p.Ho - 1 - hhi + hlo) * cpr; chh = hlo; ++cb; }
      }
    };
    if (nlive > 0) {
      tload();
      issue(0u);
    }
    int left = nlive - 1;                         // PIPE: live chunks whose tile loads are still to be issued
    if constexpr (PIPE) {
      if (left > 0) { tload(); issue(STG); --left; }
    }
    segsde_wait_vmcnt0();
    __syncthreads();
    float fa[2][4][TM], fd[2][4][TN];
    int ao[TM], dofs[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ao[i] = (lane >> 5) * BKT + wm * TM * 32 + (lane & 31) + i * 32; SEGSDE_OPAQUE(ao[i]); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { dofs[j] = BP * BKT + (lane >> 5) * BN + wn * TN * 32 + (lane & 31) + j * 32; SEGSDE_OPAQUE(dofs[j]); }
    auto fread = [&](int buf, int g, int slot) {
      const float* St = smem + buf * STAGE;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][u][i] = St[ao[i] + 2 * (4 * g + u) * BKT];
#pragma unroll
        for (int j = 0; j < TN; ++j) fd[slot][u][j] = St[dofs[j] + 2 * (4 * g + u) * BN];
      }
    };
    // unrolled by two so that the stage index is a compile-time constant (immediate LDS offsets, no address VALU)
    auto chunk = [&](auto buf_c) {
      constexpr int buf = decltype(buf_c)::value;
      fread(buf, 0, 0);
      tload();
#pragma unroll
      for (int u = 0; u < BP / 2; ++u) {
        const int g = u / 4, st = u % 4;
        if (st == 2 && g + 1 < BP / 8) fread(buf, g + 1, (g + 1) & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][st][i], fd[g & 1][st][j], acc[i][j], 0, 0, 0);
        if (u == 1) issue((unsigned)(buf ^ 1) * STG);
        __builtin_amdgcn_sched_barrier(0);
      }
      segsde_wait_vmcnt0();
      __syncthreads();
    };
    // PIPE (MODE 5).  In the loop above every chunk ends in wait + barrier and the next one starts with the LDS reads of its
    // first fragments: a read-latency chain right behind every barrier, on all four waves at once.  Here the barrier sits
    // INSIDE the chunk, after the wave's last fragment read of this stage (group 3 is read at unit 10): once every wave has
    // passed it, (a) this stage is free -- the tile loads of chunk c + 2 go into it right away (two chunks of lead instead of
    // one), and (b) the other stage (loads issued one chunk ago, waited for before the barrier) is visible -- the first
    // fragments of chunk c + 1 are read at unit 14, while the last MFMAs of chunk c still run.  Nothing waits at the chunk
    // boundary any more; still one barrier per chunk.
    auto pchunk = [&](auto buf_c) {
      constexpr int buf = decltype(buf_c)::value;
#pragma unroll
      for (int u = 0; u < BP / 2; ++u) {
        const int g = u / 4, st = u % 4;
        if (st == 2 && g + 1 < BP / 8) fread(buf, g + 1, (g + 1) & 1);
        if (u == 11 && left > 0) tload();
        if (u == 12) {
          segsde_wait_vmcnt0();
          __syncthreads();
        }
        if constexpr (F16) {
          if ((g & 1) == 1 && st == 0) {           // groups g - 1 (slot 0) and g (slot 1): sixteen pixels
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const f16x8 ha = segsde_pack_f16(make_float4(fa[0][0][i], fa[0][1][i], fa[0][2][i], fa[0][3][i]),
                                               make_float4(fa[1][0][i], fa[1][1][i], fa[1][2][i], fa[1][3][i]));
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const f16x8 hd = segsde_pack_f16(make_float4(fd[0][0][j], fd[0][1][j], fd[0][2][j], fd[0][3][j]),
                                                 make_float4(fd[1][0][j], fd[1][1][j], fd[1][2][j], fd[1][3][j]));
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hd, acc[i][j], 0, 0, 0);
              }
            }
          }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][st][i], fd[g & 1][st][j], acc[i][j], 0, 0, 0);
        }
        if (u == 12 && left > 0) { issue((unsigned)buf * STG); --left; }
        if (u == 14) fread(buf ^ 1, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (PIPE) {
      fread(0, 0, 0);
      for (int i = 0; i < nlive; i += 2) {
        pchunk(std::integral_constant<int, 0>{});
        if (i + 1 < nlive) pchunk(std::integral_constant<int, 1>{});
      }
    } else
    for (int i = 0; i < nlive; i += 2) {
      chunk(std::integral_constant<int, 0>{});
      if (i + 1 < nlive) chunk(std::integral_constant<int, 1>{});
    }
  } else {
  gload(c_begin);
  lstore(0);
  gload(c_begin + 1);
  __syncthreads();
  if constexpr (SIMPLE) {
    // dealt-out schedule (see the forward kernel): 16 k-step units per chunk, fragments of 4 k-steps double-buffered
    // and fetched two units ahead, LDS stores / table lookups / tile loads between the units
    float fa[2][4][TM], fd[2][4][TN];
    int ao[TM], dofs[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ao[i] = (lane >> 5) * BKT + wm * TM * 32 + (lane & 31) + i * 32; SEGSDE_OPAQUE(ao[i]); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { dofs[j] = BP * BKT + (lane >> 5) * BN + wn * TN * 32 + (lane & 31) + j * 32; SEGSDE_OPAQUE(dofs[j]); }
    auto fread = [&](int buf, int g, int slot) {
      const float* St = smem + buf * STAGE;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][u][i] = St[ao[i] + 2 * (4 * g + u) * BKT];
#pragma unroll
        for (int j = 0; j < TN; ++j) fd[slot][u][j] = St[dofs[j] + 2 * (4 * g + u) * BN];
      }
    };
    for (int c = c_begin; c < c_end; ++c) {
      const int buf = (c - c_begin) & 1;
      fread(buf, 0, 0);
      tload();
#pragma unroll
      for (int u = 0; u < BP / 2; ++u) {
        const int g = u / 4, st = u % 4;
        if (st == 2 && g + 1 < BP / 8) fread(buf, g + 1, (g + 1) & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][st][i], fd[g & 1][st][j], acc[i][j], 0, 0, 0);
        if (u == 0) lstore(buf ^ 1);
        if (u == 3) aload(c + 2);
        if (u == 5) dload(c + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  } else
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    mma_steps(buf, 0, BP / 4);
    lstore(buf ^ 1);
    gload(c + 2);
    mma_steps(buf, BP / 4, BP / 2);
    __syncthreads();
  }
  }

  float* out = part + (long)zt * p.Ktot * p.N;
  const int col = lane & 31, rhalf = lane >> 5;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + col;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
        if (k < p.Ktot) out[(long)k * p.N + n] = acc[i][j][r];
      }
  }
  if (!wr.tickets) return;
  // ---- last workgroup of this (k-tile, n-tile) reduces the splits
  __threadfence();                                   // release: the slab is visible device-wide before the ticket is drawn
  __syncthreads();
  unsigned* flag = reinterpret_cast<unsigned*>(smem);
  const int nsplit_z = (int)gridDim.x / (nkt * nnt);
  if (tid == 0) flag[0] = atomicAdd(&wr.tickets[nt * nkt + kt], 1u) == (unsigned)(nsplit_z - 1) ? 1u : 0u;
  __syncthreads();
  if (!flag[0]) return;
  __threadfence();                                   // acquire: the other workgroups' slabs
  const long slab = (long)p.Ktot * p.N;
  for (int e = tid; e < BKT * BN; e += 256) {
    const int kl = e / BN, nl = e - kl * BN, k = k0 + kl, n = n0 + nl;
    if (k >= p.Ktot || n >= p.N) continue;
    const float* src = part + (long)k * p.N + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 < nsplit_z; z += 4) {
      s0 += src[(long)z * slab]; s1 += src[(long)(z + 1) * slab]; s2 += src[(long)(z + 2) * slab]; s3 += src[(long)(z + 3) * slab];
    }
    for (; z < nsplit_z; ++z) s0 += src[(long)z * slab];
    int tap, c;
    wgrad_k_decode(k, p.Ctot, wr.taps, wr.srcC0, tap, c);
    wr.dw[((long)n * wr.CtotDst + wr.cOff + c) * wr.taps + tap] = (s0 + s1) + (s2 + s3);
  }
  if (tid == 0) wr.tickets[nt * nkt + kt] = 0u;      // ready for the next launch that is handed this slice
}

// dW[o][c][kh][kw] (OIHW, the state_dict layout) = sum_z part[z][(kh*KW+kw)*Ctot + c][o], fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, int splits, int Ktot, int N, int Ctot,
                                                           int taps, int srcC0, float* dw, int CtotDst, int cOff) {
  // 32 output elements x 8 split-lanes per block; each lane sums every 8th slab (4 independent chains), fixed order
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);
  const long total = (long)Ktot * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long e = blockIdx.x * 32L + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < total) {
    int z = ty;
    for (; z + 24 < splits; z += 32) {
      s0 += part[(long)z * total + e]; s1 += part[(long)(z + 8) * total + e];
      s2 += part[(long)(z + 16) * total + e]; s3 += part[(long)(z + 24) * total + e];
    }
    for (; z < splits; z += 8) s0 += part[(long)z * total + e];
  }
  sh[ty * 32 + tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && e < total) {
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += sh[j * 32 + tx];
    const int k = (int)(e / N), n = (int)(e - (long)k * N);
    int tap, c;
    wgrad_k_decode(k, Ctot, taps, srcC0, tap, c);
    dw[((long)n * CtotDst + cOff + c) * taps + tap] = s;   // (CtotDst, cOff): the channels of a slice of a wider OIHW tensor
  }
}

// OIHW -> [O][KH][KW][I]  (forward pack)  /  OIHW -> [I][KH][KW][O] with both spatial axes flipped (dgrad pack)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* w, float* out, int O, int I, int KH, int KW,
                                                          int dgrad) {
  const long total = (long)O * I * KH * KW;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    // e indexes the OUTPUT so that writes coalesce
    if (!dgrad) {
      const int i = (int)(e % I); long t = e / I;
      const int kw = (int)(t % KW); t /= KW;
      const int kh = (int)(t % KH); const int o = (int)(t / KH);
      out[e] = w[(((long)o * I + i) * KH + kh) * KW + kw];
    } else {
      const int o = (int)(e % O); long t = e / O;
      const int kw = (int)(t % KW); t /= KW;
      const int kh = (int)(t % KH); const int i = (int)(t / KH);
      out[e] = w[(((long)o * I + i) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
  }
}

// both packs in one launch (one thread per OIHW element): the forward pass of a training step needs the forward pack now and
// the data-gradient pack in its backward pass -- half the (launch-latency bound) pack launches of a step
__global__ __launch_bounds__(256) void pack_weight_both_kernel(const float* w, float* outf, float* outd, int O, int I, int KH,
                                                               int KW) {
  const int total = O * I * KH * KW;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int kw = e % KW; int t = e / KW;
    const int kh = t % KH; t /= KH;
    const int i = t % I, o = t / I;
    const float v = w[e];
    outf[((o * KH + kh) * KW + kw) * I + i] = v;
    outd[((i * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * O + o] = v;
  }
}

// Border correction for the data-gradient of a reflection-padded 3x3 convolution (monodepth_layers.py:127-142).
// The main dgrad GEMM treats the padding as zeros; pixels whose row is 1 or H-2 (col 1 or W-2) additionally
// receive the gradient that flowed into the mirrored padding cells.  dx[b,h,w,c] += sum over extra padded
// pre-images (hp,wp) of (h,w), taps (kh,kw), o:  w[o][c][kh][kw] * dy[b, hp-kh+1, wp-kw+1, o].
__global__ __launch_bounds__(256) void reflect_dgrad_fix_kernel(const float* dy, int lddy, const float* wd /*[Cin][3][3][Cout] flipped*/,
                                                                float* dx, int lddx, float* dx2, int lddx2,
                                                                int nsplit, int B, int H, int W, int Cin, int Cout) {
  // candidate border pixels: 4 lines per image (row 1, row H-2, col 1, col W-2); a pixel that lies on several
  // lines is owned by the first one so that it is corrected exactly once
  const int per_img = 2 * W + 2 * H;
  const long total = (long)B * per_img * Cin;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % Cin); long t = e / Cin;
    const int pi = (int)(t % per_img); const int b = (int)(t / per_img);
    int h, x; bool own;
    if (pi < W) { h = 1; x = pi; own = true; }
    else if (pi < 2 * W) { h = H - 2; x = pi - W; own = (h != 1); }
    else if (pi < 2 * W + H) { h = pi - 2 * W; x = 1; own = (h != 1 && h != H - 2); }
    else { h = pi - 2 * W - H; x = W - 2; own = (h != 1 && h != H - 2 && x != 1); }
    if (!own || h < 0 || h >= H || x < 0 || x >= W) continue;
    // padded pre-images per axis: the pixel itself, -1 (mirror of index 1), size (mirror of index size-2)
    int hp[3], wp[3], nh = 0, nw = 0;
    hp[nh++] = h; if (h == 1) hp[nh++] = -1; if (h == H - 2) hp[nh++] = H;
    wp[nw++] = x; if (x == 1) wp[nw++] = -1; if (x == W - 2) wp[nw++] = W;
    float s = 0.f;
    for (int a = 0; a < nh; ++a)
      for (int bb = 0; bb < nw; ++bb) {
        if (a == 0 && bb == 0) continue;  // the primary pre-image is what the zero-padded GEMM already handled
        for (int kh = 0; kh < 3; ++kh) {
          const int yh = hp[a] - kh + 1;
          if (yh < 0 || yh >= H) continue;
          for (int kw = 0; kw < 3; ++kw) {
            const int yw = wp[bb] - kw + 1;
            if (yw < 0 || yw >= W) continue;
            const float* dyp = dy + ((long)(b * H + yh) * W + yw) * lddy;
            const float* wq = wd + (((long)c * 3 + (2 - kh)) * 3 + (2 - kw)) * Cout;   // = w_oihw[o][c][kh][kw]
            for (int o = 0; o < Cout; ++o) s += dyp[o] * wq[o];
          }
        }
      }
    const long pix = (long)(b * H + h) * W + x;
    if (c < nsplit) dx[pix * lddx + c] += s;
    else dx2[pix * lddx2 + (c - nsplit)] += s;
  }
}

// experiment knob (environment SEGSDE_TUNE="bk64=1"), read once.  Measured on MI355X (profiles/ab_conv_r01.log):
// BK=64 (139 KB LDS => 1 workgroup/CU, half the barriers) loses 15-25 % on the large layers against BK=32 with two
// co-resident workgroups per CU, and start-up staggering of co-resident workgroups changes nothing.
struct Tune { int bk64 = 0; int adjfix = 0; int wplan = 0; int wovh = 4; int nos2 = 0; int dma = 1; int var = 0; int wdma = 2; int adjlds = 1; int wred = 0; int adjb = 1; int tskip = 1; int tsbn = 0; };
const Tune& tune() {
  static Tune t = [] {
    Tune r;
    if (const char* e = getenv("SEGSDE_TUNE")) {
      if (const char* q = strstr(e, "bk64=")) r.bk64 = atoi(q + 5);
      if (const char* q = strstr(e, "nos2=")) r.nos2 = atoi(q + 5);       // 1: stride-2 data-gradients without the parity split
      if (const char* q = strstr(e, "wplan=")) r.wplan = atoi(q + 6);     // 1: previous fixed-target split plan
      if (const char* q = strstr(e, "wovh=")) r.wovh = atoi(q + 5);       // per-workgroup fixed cost in chunk units
      if (const char* q = strstr(e, "adjfix=")) r.adjfix = atoi(q + 7);   // reflection adjoint: plain loop + border fix-up kernel
      if (const char* q = strstr(e, "dma=")) r.dma = atoi(q + 4);         // 0: register-staged tile loads (round-1 loop)
      if (const char* q = strstr(e, "var=")) r.var = atoi(q + 4);         // experiment variants of the LDS-DMA loop
      if (const char* q = strstr(e, "adjl=")) r.adjlds = atoi(q + 5);     // 0: reflection-adjoint loop register-staged in every wave
      if (const char* q = strstr(e, "wlds=")) r.wdma = atoi(q + 5);       // 0: register-staged weight-gradient tile loads, 1: LDS-DMA with the barrier at the chunk end (round 2), 2: barrier inside the chunk (round 4)
      if (const char* q = strstr(e, "adjb=")) r.adjb = atoi(q + 5);       // 0: reflection adjoint always inside the kernel (MODE 3); 1: zero-pad + border launches on the largest maps; 2: everywhere
      if (const char* q = strstr(e, "tsbn=")) r.tsbn = atoi(q + 5);       // 1: 128x64 tiles for every one-round grid with dead tap rows, -1: for none (default 0: where the imbalance exceeds a fifth)
      if (const char* q = strstr(e, "tskip=")) r.tskip = atoi(q + 6);     // 0: dilated zero-padded windows run their dead tap rows too
      if (const char* q = strstr(e, "wred=")) r.wred = atoi(q + 5);       // 1: split partials reduced inside the kernel (measured: slower)
    }
    return r;
  }();
  return t;
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

const float* zero_page() {
  static const float* z = [] {
    void* q = nullptr;
    (void)hipGetSymbolAddress(&q, HIP_SYMBOL(segsde_zero_page));
    return static_cast<const float*>(q);
  }();
  return z;
}

// magic = ceil(2^(31+l) / d), l = ceil(log2 d) >= 1: umulhi(n, magic) >> (l - 1) == n / d for every 0 <= n < 2^31
// (the rounding error of the product is below n / 2^(31+l) < 2^-l <= 1/d); d == 1 -> magic 0 (identity)
void magic_for(int d, unsigned& magic, int& shift) {
  if (d <= 1) { magic = 0; shift = 0; return; }
  int l = 1;
  while ((1L << l) < d) ++l;
  const unsigned long long num = 1ULL << (31 + l);
  magic = (unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d);
  shift = l - 1;
}
void set_divs(ConvP& p) {
  // out_row (strided sub-grid stores) is only used by launches that are not sum2x2
  p.d1 = p.sum2x2 ? (p.Ho >> 1) * (p.Wo >> 1) : p.Ho * p.Wo;
  p.d2 = p.sum2x2 ? (p.Wo >> 1) : p.Wo;
  magic_for(p.d1, p.mg1, p.sf1);
  magic_for(p.d2, p.mg2, p.sf2);
  magic_for(p.Ctot, p.mgC, p.sfC);
  magic_for(p.KW, p.mgKW, p.sfKW);
}

ConvP make_params(const segsde_conv_desc* d, const float* x0, const float* x1, const float* w, const float* bias,
                  float* y, float* y2) {
  ConvP p;
  p.x0 = x0; p.x1 = x1 ? x1 : x0; p.w = w; p.bias = bias; p.y = y; p.y2 = y2 ? y2 : y;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.C1 = d->C1; p.ld0 = d->ld0; p.ld1 = d->C1 ? d->ld1 : d->ld0;
  p.up0 = d->up0; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->Cout; p.ldy = d->ldy;
  p.ldy2 = d->ldy2 ? d->ldy2 : d->ldy; p.nsplit = y2 ? d->nsplit : d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.dil = d->dil; p.pad = d->pad; p.pad_mode = d->pad_mode;
  p.in_div = d->in_div < 1 ? 1 : d->in_div;
  p.Ctot = d->C0 + d->C1; p.Ktot = d->KH * d->KW * p.Ctot; p.M = d->B * d->Ho * d->Wo; p.act = d->act;
  p.sum2x2 = d->sum2x2;
  set_divs(p);
  p.nb = 0; p.ne = p.N;
  p.vecout = (p.N % 4 == 0) && (p.ldy % 4 == 0) && (p.ldy2 % 4 == 0) && (p.nsplit % 4 == 0) && aligned16(p.y) &&
             aligned16(p.y2);
  p.zero = zero_page();
  p.kh0 = 0; p.khs = 1; p.kw0 = 0; p.kws = 1; p.KWf = p.KW; p.Kfull = p.Ktot; p.os = 1; p.oph = 0; p.opw = 0; p.OHf = p.Ho; p.OWf = p.Wo;
  p.stats = nullptr;
  p.accum = d->accumulate ? 1 : 0;
  {
    // streaming stores for outputs far beyond the memory-side cache (SEGSDE_CONV_NT_MB: threshold in MiB, 0 = never; default off until measured)
    static long nt_bytes = -1;
    if (nt_bytes < 0) { const char* e = getenv("SEGSDE_CONV_NT_MB"); nt_bytes = (e ? atol(e) : 0L) << 20; }
    p.nt = (nt_bytes > 0 && (long)p.M * p.N * 4 >= nt_bytes) ? 1 : 0;
  }
  p.f16 = d->compute == 1 ? 1 : 0;
  p.agy = nullptr; p.agld = 0; p.agkind = 0;
  p.lin = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->in_div <= 1 && !d->up0 && !d->sum2x2 && d->C1 == 0 &&
          d->H == d->Ho && d->W == d->Wo;
  p.padw = p.pad; p.wtap = p.Ctot; p.osfast = 0; p.submap = 0;
  p.tapskip = (tune().tskip && d->pad_mode == SEGSDE_PAD_ZERO && d->dil > 1 && d->KH > 1 && d->in_div <= 1 && !d->up0 && !d->sum2x2 &&
               d->stride == 1) ? 1 : 0;
  p.wbstride = 0;
  return p;
}


bool vec_ok(const ConvP& p) {
  return (p.Ctot % 4 == 0) && (p.C0 % 4 == 0) && (p.ld0 % 4 == 0) && (p.ld1 % 4 == 0) && aligned16(p.x0) &&
         aligned16(p.x1) && aligned16(p.w);
}

int validate(const segsde_conv_desc* d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C0 <= 0 || d->C1 < 0 || d->Cout <= 0) return SEGSDE_ERR_SHAPE;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0) return SEGSDE_ERR_SHAPE;
  if (d->up0 && ((d->H & 1) || (d->W & 1))) return SEGSDE_ERR_SHAPE;
  if (d->pad_mode == SEGSDE_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W)) return SEGSDE_ERR_SHAPE;
  if (d->pad_mode == SEGSDE_PAD_REFLECT_ADJOINT &&
      (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1 || d->in_div > 1 || d->C1 != 0 ||
       d->H != d->Ho || d->W != d->Wo || d->H < 2 || d->W < 2))
    return SEGSDE_ERR_SHAPE;
  if (d->ld0 < d->C0 || (d->C1 && d->ld1 < d->C1) || d->ldy <= 0) return SEGSDE_ERR_SHAPE;
  return 0;
}

bool fast_ok(const ConvP& p) {
  // uniform-tap chunks + 32-bit element offsets
  const long e0 = (long)p.B * (p.H >> p.up0) * (p.W >> p.up0) * p.ld0;
  const long e1 = (long)p.B * p.H * p.W * p.ld1;
  return vec_ok(p) && (p.Ctot % 32 == 0) && (p.C1 == 0 || p.C0 % 32 == 0) && p.in_div <= 2 && e0 < (1L << 31) &&
         e1 < (1L << 31);
}
// forward / data-gradient FAST path: 32-bit BYTE offsets inside buffer resources rebased to the tile's first image
bool igemm_fast_ok(const ConvP& p) {
  const long i0 = (long)(p.H >> p.up0) * (p.W >> p.up0) * p.ld0 * 4, i1 = (long)p.H * p.W * p.ld1 * 4;
  const long span = 256 / ((long)p.Ho * p.Wo) + 2;   // images one (up to 256-row) tile can touch
  return fast_ok(p) && p.KH * p.KW <= 16 /* tap table in LDS */ && span * (i0 > i1 ? i0 : i1) < (1L << 31) &&
         (long)p.N * p.Ktot * 4 < (1L << 31);
}
bool bk64_ok(const ConvP& p) { return igemm_fast_ok(p) && (p.Ctot % 64 == 0) && (p.C1 == 0 || p.C0 % 64 == 0); }

template <int BM, int BN, int WM, int WN, int MODE, int BK, int VAR = 0>
int launch_igemm_mode(const ConvP& p, hipStream_t stream) {
  const int nblk = segsde_cdiv(p.M, BM) * segsde_cdiv(p.ne - p.nb, BN);
  size_t smem = (VAR == 4 ? 4 : 2) * (size_t)(BM + BN) * BK * sizeof(float);
  if (smem < (size_t)BM * BN * sizeof(float)) smem = (size_t)BM * BN * sizeof(float);   // the staged epilogue's tile
  if (MODE >= 2) smem += (size_t)(((MODE == 3 && BN >= 128) || p.C0 < p.Ctot) ? 2 : 1) * p.KH * p.KW * BM * sizeof(unsigned);   // tap table
  auto k = conv_igemm_kernel<BM, BN, WM, WN, MODE, BK, VAR>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(k, dim3(nblk), dim3(256), smem, stream, p);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(const ConvP& p, hipStream_t stream) {
  if (igemm_fast_ok(p) && p.pad_mode == SEGSDE_PAD_REFLECT_ADJOINT && !(tune().adjfix && !p.sum2x2) && tune().adjfix < 2)
    return tune().var == 8 ? launch_igemm_mode<BM, BN, WM, WN, 3, 32, 8>(p, stream)
           : (tune().adjlds ? launch_igemm_mode<BM, BN, WM, WN, 3, 32>(p, stream) : launch_igemm_mode<BM, BN, WM, WN, 3, 32, 5>(p, stream));
  if (p.pad_mode == SEGSDE_PAD_CLAMP_) { // upsample-folded class launches (host guarantees the FAST conditions)
    if (!igemm_fast_ok(p)) return SEGSDE_ERR_UNSUPPORTED;
    return p.f16 ? launch_igemm_mode<BM, BN, WM, WN, 4, 32, 16 + 9>(p, stream) : launch_igemm_mode<BM, BN, WM, WN, 4, 32, 9>(p, stream);
  }
  if (tune().bk64 && bk64_ok(p)) return launch_igemm_mode<BM, BN, WM, WN, 2, 64>(p, stream);
  if (igemm_fast_ok(p) && tune().dma) {
    if (tune().var == 1) return launch_igemm_mode<BM, BN, WM, WN, 4, 32, 1>(p, stream);
    if (tune().var == 2) return launch_igemm_mode<BM, BN, WM, WN, 4, 32, 2>(p, stream);
    if (tune().var == 3) return launch_igemm_mode<BM, BN, WM, WN, 4, 32, 3>(p, stream);
    if constexpr (BN >= 64) {
      if (tune().var == 4) return launch_igemm_mode<BM, BN, WM, WN, 4, 16, 4>(p, stream);
    }
    if (tune().var == 7) return launch_igemm_mode<BM, BN, WM, WN, 4, 32, 7>(p, stream);
    if constexpr (BN == 64) {
      if (tune().var == 6) return launch_igemm_mode<BM, BN, WM, WN, 4, 16, 6>(p, stream);
    }
    if (p.f16) return launch_igemm_mode<BM, BN, WM, WN, 4, 32, 16>(p, stream);
    return launch_igemm_mode<BM, BN, WM, WN, 4, 32>(p, stream);
  }
  if (igemm_fast_ok(p)) return launch_igemm_mode<BM, BN, WM, WN, 2, 32>(p, stream);
  if (vec_ok(p)) return launch_igemm_mode<BM, BN, WM, WN, 1, 32>(p, stream);
  return launch_igemm_mode<BM, BN, WM, WN, 0, 32>(p, stream);
}

int launch_reflect_fix(const float* dy, int lddy, const float* wd, float* dx, int lddx, float* dx2, int lddx2, int nsplit,
                       int B, int H, int W, int Cin, int Cout, hipStream_t s) {
  if (!dx2) { dx2 = dx; lddx2 = lddx; nsplit = Cin; }
  const long total = (long)B * (2 * W + 2 * H) * Cin;
  hipLaunchKernelGGL(reflect_dgrad_fix_kernel, dim3(min(4096, segsde_cdiv(total, 256))), dim3(256), 0, s, dy, lddy, wd, dx,
                     lddx, dx2, lddx2, nsplit, B, H, W, Cin, Cout);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
}  // namespace

namespace {
// rows of statistics partials a forward launch of this shape writes (0: the shape does not take the staged epilogue
// with a single tile shape, the statistics then have to come from segsde_bn_stats)
// Dead tap rows make the tiles' K loops unequal (1/3 .. 3/3 of the taps): a grid that fits the chip in ONE round (two
// workgroups per CU) takes as long as its longest tile whatever the others skip.  Narrower tiles = twice the workgroups, the
// second half is handed out as the first ones finish.  (ONE predicate for the launch and for the statistics rows it writes.)
// Round 4: it pays where the one-round grid loses more than a fifth to the imbalance -- mean live tap rows per output row
// below 0.8 of the maximum (rate 12 on the 32-row map: 8 rows with three live tap rows, 24 with two -> 0.75; measured 4.13 ->
// 3.80 ms/step) -- and costs where it does not (rate 6: 0.875, rate 18: 0.94: the 128x64 tile's lower rate is all that is left;
// 4.16 -> 4.54 and 2.78 -> 3.12 ms/step, profiles/experiments_r03.md).  tsbn=1 forces it, tsbn=-1 turns it off.
bool narrow_for_tapskip(const ConvP& q) {
  if (!q.tapskip || tune().tsbn < 0 || q.N <= 64 || (q.ne - q.nb) % 64 != 0 ||
      (long)segsde_cdiv(q.M, 128) * segsde_cdiv(q.ne - q.nb, 128) > 512) return false;
  if (tune().tsbn > 0) return true;
  long live = 0; int most = 0;
  for (int h = 0; h < q.Ho; ++h) {
    int n = 0;
    for (int kh = 0; kh < q.KH; ++kh) { const int hs = h * q.stride + kh * q.dil - q.pad; n += hs >= 0 && hs < q.H; }
    live += n; most = n > most ? n : most;
  }
  return most > 0 && (double)live < 0.8 * (double)most * q.Ho;
}
long stats_rows(const segsde_conv_desc* d, const ConvP& p) {
  if (!p.vecout || d->sum2x2 || d->in_div > 1 || p.y2 != p.y || p.bias || d->act != 0) return 0;
  if (d->Cout == 1 || (d->Cout % 128 > 0 && d->Cout % 128 <= 64 && d->Cout > 64)) return 0;
  const int bn = d->Cout <= 32 ? 32 : ((d->Cout <= 64 || narrow_for_tapskip(p)) ? 64 : 128);
  return (long)segsde_cdiv(p.M, 128) * (256 / bn);
}
}  // namespace

namespace {
int launch_by_n(const ConvP& q, hipStream_t s) {
  if (q.N <= 32) return launch_igemm<128, 32, 4, 1>(q, s);
  if (q.N <= 64) return launch_igemm<128, 64, 2, 2>(q, s);
  if (narrow_for_tapskip(q)) return launch_igemm<128, 64, 2, 2>(q, s);
  if (q.N % 128 > 0 && q.N % 128 <= 64) {
    ConvP a = q, b = q;
    a.ne = q.N - q.N % 128; b.nb = a.ne;
    if (int e = launch_igemm<128, 128, 2, 2>(a, s)) return e;
    return (b.ne - b.nb <= 32) ? launch_igemm<128, 32, 4, 1>(b, s) : launch_igemm<128, 64, 2, 2>(b, s);
  }
  return launch_igemm<128, 128, 2, 2>(q, s);
}

// corner terms of the reflection adjoint: pixel (1,1) / (1,W-2) / (H-2,1) / (H-2,W-2) also collects the gradient of the output
// corner next to it through the tap that reached the doubly mirrored padding cell
__global__ __launch_bounds__(256) void adjoint_corner_kernel(const float* dy, int lddy, const float* wd /*[N][9][C]*/, float* y, int ldy,
                                                             float* y2, int ldy2, int nsplit, int B, int H, int W, int N, int C,
                                                             const float* agy, int agld, int agkind, int accumulate_only) {
  const int total = B * 4 * N;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int n = e % N, t = e / N, k = t & 3, b = t >> 2;
    const int bot = k >> 1, rgt = k & 1;
    const int i = bot ? H - 2 : 1, j = rgt ? W - 2 : 1, r = bot ? H - 1 : 0, sc = rgt ? W - 1 : 0;
    const int tap = (bot ? 0 : 2) * 3 + (rgt ? 0 : 2);
    const float* dp = dy + ((long)(b * H + r) * W + sc) * lddy;
    const float* wp = wd + ((long)n * 9 + tap) * C;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += dp[c] * wp[c];
    const long pix = (long)(b * H + i) * W + j;
    if (agy && n < nsplit) acc *= segsde_act_grad_from_out(agy[pix * agld + n], agkind);
    if (n < nsplit) y[pix * ldy + n] += acc; else y2[pix * ldy2 + (n - nsplit)] += acc;
  }
}

// Reflection-pad data-gradient as the plain zero-padded data-gradient (LDS-DMA loop, no bordered waves) + the mirrored-padding
// contributions as four small launches of the same kernel that ADD onto rows 1 / H-2 and columns 1 / W-2 (a 1x3 / 3x1
// convolution of gradient row 0 / H-1, column 0 / W-1 with the taps that reached the padding) + the corner terms -- the same
// construction as the clamp adjoint of the upsample-folded route.  (The in-kernel variant, MODE 3, runs the one wave per tile
// that owns a border pixel on a register-staged loop with extra loads; every chunk ends in a barrier, so that wave set the
// pace of half the tiles of a 512-pixel-wide image: 108-135 TFLOP/s against 139-145 for the forward of the same layers.)
int launch_adjoint_by_borders(const ConvP& p, hipStream_t s, bool with_main = true) {
  ConvP q = p;
  q.pad_mode = SEGSDE_PAD_ZERO;
  ConvP bl[4];
  for (int k = 0; k < 4; ++k) {
    const bool rowl = k < 2, far = k & 1;           // 0: row 1 <- gradient row 0, 1: row H-2 <- row H-1, 2: column 1 <- column 0, 3: column W-2 <- W-1
    ConvP b = p;
    b.pad_mode = SEGSDE_PAD_ZERO;
    b.KH = rowl ? 1 : 3; b.KW = rowl ? 3 : 1;
    b.Ho = rowl ? 1 : p.H; b.Wo = rowl ? p.W : 1; b.M = p.B * b.Ho * b.Wo;
    b.pad = rowl ? (far ? -(p.H - 1) : 0) : 1;
    b.padw = rowl ? 1 : (far ? -(p.W - 1) : 0);
    b.Ktot = 3 * p.Ctot; b.Kfull = p.Kfull; b.KWf = b.KW;
    b.wtap = rowl ? p.Ctot : 3 * p.Ctot;
    const int tap0 = rowl ? (far ? 0 : 6) : (far ? 0 : 2);      // first of the three taps that reached the padding
    b.w = p.w + (long)tap0 * p.Ctot;
    b.kh0 = 0; b.khs = 1; b.kw0 = 0; b.kws = 1;
    b.submap = 1; b.os = 1; b.OHf = p.H; b.OWf = p.W;
    b.oph = rowl ? (far ? p.H - 2 : 1) : 0; b.opw = rowl ? 0 : (far ? p.W - 2 : 1);
    b.osfast = (rowl && p.W % 128 == 0) ? 1 : 0;
    b.accum = 1; b.lin = 0; b.stats = nullptr; b.bias = nullptr; b.act = 0;
    set_divs(b);
    if (!igemm_fast_ok(b) || !b.vecout) return SEGSDE_ERR_UNSUPPORTED;
    bl[k] = b;
  }
  if (with_main)
    if (int e = launch_by_n(q, s)) return e;
  for (int k = 0; k < 4; ++k)
    if (int e = launch_by_n(bl[k], s)) return e;
  hipLaunchKernelGGL(adjoint_corner_kernel, dim3(segsde_cdiv((long)p.B * 4 * p.N, 256)), dim3(256), 0, s, p.x0, p.ld0, p.w, p.y, p.ldy,
                     p.y2, p.ldy2, p.nsplit, p.B, p.H, p.W, p.N, p.Ctot, p.agy, p.agld, p.agkind, 0);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
bool adjoint_by_borders_ok(const ConvP& p) {
  // measured (profiles/experiments_r03.md): the five extra launches cost ~70 us per call -- a gain only on the largest maps
  // (128 -> 64 @256x512 x 16: 5.63 -> 5.46 ms/step), a loss below (256 -> 256 @32x64: 0.69 -> 1.15); adjb=2 forces it everywhere
  // (half-precision operand mode: always -- the in-kernel adjoint, MODE 3, has no half-precision variant and would run in fp32)
  const bool big = (long)p.B * p.H * p.W >= (1L << 21) || tune().adjb == 2 || p.f16;
  return tune().adjb && big && p.pad_mode == SEGSDE_PAD_REFLECT_ADJOINT && igemm_fast_ok(p) && !p.sum2x2 && p.vecout && p.H >= 4 &&
         p.W >= 4 && p.C1 == 0 && p.KH == 3 && p.KW == 3 && p.nb == 0 && p.ne == p.N;
}
}  // namespace

// The mirrored-padding part of a reflection-padded 3x3 / stride 1 convolution's data-gradient ALONE: four border launches of the
// implicit-GEMM kernel + the corner terms, ADDED onto y (rows 1 / H-2, columns 1 / W-2), each term multiplied by the activation
// derivative when act_out is given.  The zero-padded part is the caller's (the one-kernel Winograd data-gradient,
// segsde_conv2d_winograd_fused_dgrad, which has no bordered tiles).  d: the descriptor of segsde_conv2d_dgrad_actgrad with
// pad_mode = SEGSDE_PAD_REFLECT_ADJOINT.
namespace {
int borders_params(const segsde_conv_desc* d, const float* dy, const float* wdpack, float* y, const float* act_out, int act_ld,
                   int act_kind, ConvP& p) {
  if (int e = validate(d)) return e;
  if (d->pad_mode != SEGSDE_PAD_REFLECT_ADJOINT || d->C1 || d->up0 || d->sum2x2 || d->in_div > 1 || d->act != 0) return SEGSDE_ERR_UNSUPPORTED;
  p = make_params(d, dy, nullptr, wdpack, nullptr, y, nullptr);
  if (act_out) {
    if (act_kind < SEGSDE_ACT_RELU || act_kind > SEGSDE_ACT_SIGMOID || act_ld < p.nsplit || (act_ld % 4) || !aligned16(act_out))
      return SEGSDE_ERR_UNSUPPORTED;
    p.agy = act_out; p.agld = act_ld; p.agkind = act_kind;
  }
  if (!(igemm_fast_ok(p) && p.vecout && p.H >= 4 && p.W >= 4 && p.KH == 3 && p.KW == 3 && p.nb == 0 && p.ne == p.N))
    return SEGSDE_ERR_UNSUPPORTED;
  return 0;
}
}  // namespace

// would segsde_reflect_adjoint_borders take this descriptor (asked BEFORE the zero-padded launch writes dx)
extern "C" int segsde_reflect_adjoint_borders_ok(const segsde_conv_desc* d, int act_ld) {
  ConvP p;
  float* fake = reinterpret_cast<float*>(16);
  return borders_params(d, fake, fake, fake, act_ld ? fake : nullptr, act_ld, SEGSDE_ACT_ELU, p) == 0;
}

extern "C" int segsde_reflect_adjoint_borders(const segsde_conv_desc* d, const float* dy, const float* wdpack, float* y,
                                              const float* act_out, int act_ld, int act_kind, void* stream) {
  if (!dy || !wdpack || !y) return SEGSDE_ERR_NULL;
  ConvP p;
  if (int e = borders_params(d, dy, wdpack, y, act_out, act_ld, act_kind, p)) return e;
  return launch_adjoint_by_borders(p, static_cast<hipStream_t>(stream), false);
}

extern "C" long segsde_conv2d_stats_rows(const segsde_conv_desc* d) {
  if (validate(d)) return 0;
  float dummy[4];
  const ConvP p = make_params(d, dummy, dummy, dummy, nullptr, reinterpret_cast<float*>(16), nullptr);
  return stats_rows(d, p);
}

extern "C" int segsde_conv2d_forward(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                     const float* bias, float* y, float* y2, void* stream) {
  return segsde_conv2d_forward_stats(d, x0, x1, wpack, bias, y, y2, nullptr, stream);
}

extern "C" int segsde_conv2d_forward_stats(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                           const float* bias, float* y, float* y2, double* stats, void* stream) {
  return segsde_conv2d_dgrad_actgrad(d, x0, x1, wpack, bias, y, y2, stats, nullptr, 0, 0, stream);
}

extern "C" int segsde_conv2d_dgrad_actgrad(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                           const float* bias, float* y, float* y2, double* stats, const float* act_out,
                                           int act_ld, int act_kind, void* stream) {
  if (int e = validate(d)) return e;
  if (!x0 || !wpack || !y || (d->C1 && !x1)) return SEGSDE_ERR_NULL;
  ConvP p = make_params(d, x0, x1, wpack, bias, y, y2);
  if (act_out) {
    // fused activation backward: a data-gradient launch (no bias / activation / statistics of its own), 16-byte rows
    if (stats || bias || d->act != 0 || act_kind < SEGSDE_ACT_RELU || act_kind > SEGSDE_ACT_SIGMOID || act_ld < p.nsplit ||
        (act_ld % 4) || !aligned16(act_out) || d->in_div > 1)
      return SEGSDE_ERR_UNSUPPORTED;
    p.agy = act_out; p.agld = act_ld; p.agkind = act_kind;
  }
  if (stats) {
    if (stats_rows(d, p) == 0) return SEGSDE_ERR_UNSUPPORTED;
    p.stats = stats;
  }
  // data-gradient of a stride-2 convolution as four parity classes (below): what an accumulating launch needs from it
  const bool parity_split = d->in_div == 2 && d->stride == 1 && (d->dil & 1) && d->pad_mode == SEGSDE_PAD_ZERO && !d->sum2x2 &&
                            d->C1 == 0 && !d->up0 && !bias && d->act == 0 && !y2 && !tune().nos2 && p.vecout;
  if (d->accumulate) {
    // only the plain staged-epilogue launches add in place: one destination, no activation / bias, no special route
    // (round 4: and the parity classes of a stride-2 data-gradient, whose sub-grid stores add just the same)
    if (y2 || bias || d->act != 0 || d->sum2x2 || (d->in_div > 1 && !parity_split) || !p.vecout || d->Cout == 1 || d->C0 % 4 != 0 ||
        d->C0 == 1 || (d->pad_mode == SEGSDE_PAD_REFLECT_ADJOINT && !igemm_fast_ok(p)))
      return SEGSDE_ERR_UNSUPPORTED;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // disparity heads (Cout = 1) and their data-gradients (one gradient channel in): HBM-bound stencil kernels
  const bool plain3x3 = d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->C1 == 0 && !d->up0 &&
                        d->in_div <= 1 && !d->sum2x2 && d->H == d->Ho && d->W == d->Wo;
  if (plain3x3 && d->Cout == 1 && d->pad_mode != SEGSDE_PAD_REFLECT_ADJOINT && segsde_c1_supported(d->C0, d->ld0) &&
      aligned16(x0) && aligned16(wpack))
    return segsde_c1_forward(x0, d->ld0, d->B, d->H, d->W, d->C0, wpack, bias, d->pad_mode == SEGSDE_PAD_REFLECT, d->act, y,
                             d->ldy, stream);
  if (plain3x3 && d->C0 == 1 && d->pad_mode != SEGSDE_PAD_REFLECT && !bias && d->act == 0 &&
      segsde_c1_supported(d->Cout, p.ldy) && p.vecout)
    return segsde_c1_dgrad(x0, d->ld0, d->B, d->H, d->W, d->Cout, wpack, d->pad_mode == SEGSDE_PAD_REFLECT_ADJOINT, y, p.ldy,
                           y2, p.ldy2, p.nsplit, p.agy, p.agld, p.agkind, stream);
  // 1x1 with a narrow, non-vectorisable input side (data-gradient of the 19-class head): HBM-bound register kernel
  const bool plain1x1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->C1 == 0 && !d->up0 && d->in_div <= 1 &&
                        !d->sum2x2 && d->H == d->Ho && d->W == d->Wo;
  if (plain1x1 && !stats && !p.agy && d->C0 % 4 != 0 && d->ld0 == d->C0 && aligned16(x0) && !bias && d->act == 0 && !y2 &&
      segsde_skinny_supported(d->C0, d->Cout) && p.vecout)
    return segsde_skinny_nk(x0, d->C0, wpack, (long)d->B * d->H * d->W, d->Cout, y, p.ldy, stream);
  // Data-gradient of a stride-2 convolution (in_div == 2): three quarters of the (pixel, tap) pairs hit the holes between
  // the strided outputs.  Split the gradient image into its four (row parity, column parity) classes: each class is a
  // dense stride-1 problem over the taps of matching parity (1 / 2 / 2 / 4 of the 9 taps of a 3x3, 1 / 0 / 0 / 0 of a 1x1)
  // that reads dY without holes and stores into the strided sub-grid -- a quarter of the matrix work, no wasted MFMAs.
  if (parity_split) {
    bool ok = true;
    ConvP sub[4]; int nsub = 0; bool empty = false;
    for (int ph = 0; ph < 2 && ok; ++ph)
      for (int pw = 0; pw < 2 && ok; ++pw) {
        ConvP q = p;
        // first tap of matching parity per axis; taps then alternate (dilation is odd)
        int kh0 = -1, kw0 = -1;
        for (int k = 0; k < d->KH; ++k) if (((ph + k * d->dil - d->pad) & 1) == 0) { kh0 = k; break; }
        for (int k = 0; k < d->KW; ++k) if (((pw + k * d->dil - d->pad) & 1) == 0) { kw0 = k; break; }
        const int nI = (d->Ho - ph + 1) / 2, nJ = (d->Wo - pw + 1) / 2;
        if (nI <= 0 || nJ <= 0) continue;
        if (kh0 < 0 || kw0 < 0) { empty = true; continue; }   // no tap reaches this class: its pixels are zero
        q.kh0 = kh0; q.khs = 2; q.kw0 = kw0; q.kws = 2; q.KWf = d->KW; q.Kfull = p.Ktot;
        q.KH = (d->KH - kh0 + 1) / 2; q.KW = (d->KW - kw0 + 1) / 2;
        q.Ktot = q.KH * q.KW * q.Ctot;
        q.in_div = 1;
        // source row of loop tap kh' for sub-grid row i:  i + (ph + kh0*dil - pad)/2 + kh'*dil   (exact division)
        q.pad = -((ph + kh0 * d->dil - d->pad) / 2);
        q.padw = -((pw + kw0 * d->dil - d->pad) / 2);   // (per-axis padding: round 3)
        q.os = 2; q.submap = 1; q.oph = ph; q.opw = pw; q.OHf = d->Ho; q.OWf = d->Wo;
        q.Ho = nI; q.Wo = nJ; q.M = d->B * nI * nJ;
        set_divs(q);
        if (!igemm_fast_ok(q)) ok = false;
        sub[nsub++] = q;
      }
    if (!ok && d->accumulate) return SEGSDE_ERR_UNSUPPORTED;
    if (ok) {
      if (empty && !d->accumulate) {   // 1x1: only the (even, even) class receives anything (accumulating: the rest keeps what it holds)
        if (hipMemsetAsync(y, 0, (size_t)d->B * d->Ho * d->Wo * p.ldy * sizeof(float), s) != hipSuccess) return SEGSDE_ERR_SHAPE;
      }
      for (int i = 0; i < nsub; ++i) {
        const ConvP& q = sub[i];
        int e;
        if (q.N <= 32) e = launch_igemm<128, 32, 4, 1>(q, s);
        else if (q.N <= 64) e = launch_igemm<128, 64, 2, 2>(q, s);
        else e = launch_igemm<128, 128, 2, 2>(q, s);
        if (e) return e;
      }
      return 0;
    }
  }
  // shapes whose mirrored-padding contributions are added by a second kernel (below) cannot have the activation derivative
  // applied in the first kernel's epilogue: the caller runs the separate pass
  if (p.agy && d->pad_mode == SEGSDE_PAD_REFLECT_ADJOINT && (!igemm_fast_ok(p) || (tune().adjfix && !p.sum2x2)))
    return SEGSDE_ERR_UNSUPPORTED;
  if (d->sum2x2) {
    if ((d->Ho & 1) || (d->Wo & 1) || d->stride != 1 || d->act != 0 || bias) return SEGSDE_ERR_SHAPE;
    if (!igemm_fast_ok(p)) return SEGSDE_ERR_UNSUPPORTED;   // caller falls back to the two-pass path (full-res dgrad + 2x2 sum)
  }
  if (adjoint_by_borders_ok(p)) {
    const int e = launch_adjoint_by_borders(p, s);
    if (e != SEGSDE_ERR_UNSUPPORTED) return e;
  }
  // tile width by N: 32 / 64 for narrow outputs (a 256x64 tile measured 3 % slower), 128-wide tiles for the bulk and 64-wide
  // ones for a 64-channel tail (e.g. the 192-channel concat data-gradient)
  if (int e = launch_by_n(p, s)) return e;
  if (d->pad_mode == SEGSDE_PAD_REFLECT_ADJOINT && (!igemm_fast_ok(p) || (tune().adjfix && !p.sum2x2)))
    // the generic gathers treat the padding as zeros; add the mirrored-padding contributions on the border pixels
    return launch_reflect_fix(x0, p.ld0, wpack, y, p.ldy, y2, p.ldy2, p.nsplit, p.B, p.H, p.W, p.N, p.C0, s);
  return 0;
}

namespace {
template <int BKT, int BN, int WM, int WN, int MODEX>
int launch_wgrad_mode(const ConvP& p, const float* dy, int lddy, float* ws, int splits, int cps, hipStream_t stream, WRed wr) {
  constexpr int MODE = MODEX & 15;
  const dim3 grid(segsde_cdiv(p.Ktot, BKT) * segsde_cdiv(p.N, BN) * splits);
  size_t smem = 2 * (size_t)BP * (BKT + BN) * sizeof(float);
  if (MODE == 2 || MODE == 4 || MODE == 5)   // + the four offset tables (padded rows / columns of the two sources)
    smem += 2 * (size_t)((p.Ho - 1) * p.stride + (p.KH - 1) * p.dil + 1 + (p.Wo - 1) * p.stride + (p.KW - 1) * p.dil + 1) * sizeof(unsigned);
  auto k = conv_wgrad_kernel<BKT, BN, WM, WN, MODEX>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(k, grid, dim3(256), smem, stream, p, dy, lddy, ws, cps, wr);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

// 2: table-driven loader (rows a multiple of 32 pixels wide), 3: general fast gather, 1: float4 gather, 0: scalar gather
int wgrad_mode(const ConvP& p, const float* dy, int lddy) {
  const bool vec = vec_ok(p) && (p.N % 4 == 0) && (lddy % 4 == 0) && aligned16(dy);
  const long e0 = (long)p.B * (p.H >> p.up0) * (p.W >> p.up0) * p.ld0, e1 = (long)p.B * p.H * p.W * p.ld1;
  const bool fast = vec_ok(p) && e0 < (1L << 31) && e1 < (1L << 31);   // the dY side may be scalar (odd Cout)
  const long i0 = (long)(p.H >> p.up0) * (p.W >> p.up0) * p.ld0 * 4, i1 = (long)p.H * p.W * p.ld1 * 4;
  const bool table = p.Wo % BP == 0 && i0 <= (1L << 30) && i1 <= (1L << 30) && (long)BP * p.os * lddy * 4 < (1L << 30);
  if (fast && vec && table) return 2;
  if (fast) return 3;
  return vec ? 1 : 0;
}

template <int BKT, int BN, int WM, int WN>
int launch_wgrad(const ConvP& p, const float* dy, int lddy, float* ws, int splits, int cps, hipStream_t stream, WRed wr = WRed{}) {
  switch (wgrad_mode(p, dy, lddy)) {
    case 2:
      if (tune().wdma == 2) return p.f16 ? launch_wgrad_mode<BKT, BN, WM, WN, 16 + 5>(p, dy, lddy, ws, splits, cps, stream, wr)
                                         : launch_wgrad_mode<BKT, BN, WM, WN, 5>(p, dy, lddy, ws, splits, cps, stream, wr);
      if (tune().wdma) return launch_wgrad_mode<BKT, BN, WM, WN, 4>(p, dy, lddy, ws, splits, cps, stream, wr);
      return launch_wgrad_mode<BKT, BN, WM, WN, 2>(p, dy, lddy, ws, splits, cps, stream, wr);
    case 3: return launch_wgrad_mode<BKT, BN, WM, WN, 3>(p, dy, lddy, ws, splits, cps, stream, wr);
    case 1: return launch_wgrad_mode<BKT, BN, WM, WN, 1>(p, dy, lddy, ws, splits, cps, stream, wr);
    default: return launch_wgrad_mode<BKT, BN, WM, WN, 0>(p, dy, lddy, ws, splits, cps, stream, wr);
  }
}

WRed make_wred(const ConvP& p, int bkt, int bn, float* dw, int CtotDst, int cOff, int taps, int srcC0) {
  WRed wr{};
  if (!tune().wred) return wr;
  wr.tickets = segsde_ticket_slice(segsde_cdiv(p.Ktot, bkt) * segsde_cdiv(p.N, bn));
  wr.dw = dw; wr.CtotDst = CtotDst; wr.cOff = cOff; wr.taps = taps; wr.srcC0 = srcC0;
  return wr;
}

void wgrad_plan(const segsde_conv_desc* d, int& bkt, int& bn, int& splits, int& cps) {
  const int Ktot = d->KH * d->KW * (d->C0 + d->C1);
  const long M = (long)d->B * d->Ho * d->Wo;
  bn = d->Cout <= 32 ? 32 : (d->Cout <= 64 ? 64 : 128);
  bkt = 128;
  const long tiles = (long)segsde_cdiv(Ktot, bkt) * segsde_cdiv(d->Cout, bn);
  const int nchunks = segsde_cdiv(M, BP);
  long want;
  if (tune().wplan == 0) {
    // Equal-sized workgroups run in waves of `slots` (2 per CU for the 128x128 tile, 3 for the narrower ones): a split
    // count that puts a handful of workgroups into one more wave costs a whole extra pass.  Pick the split count that
    // minimises waves x (chunks per split + fixed per-workgroup cost), preferring fewer splits on ties.
    const long slots = 256L * (bn == 128 ? 2 : 3);
    const long ovh = tune().wovh;
    long best = -1, best_cost = 0;
    const long smax = nchunks / 8 > 1 ? (nchunks / 8 < 1024 ? nchunks / 8 : 1024) : 1;
    for (long sp = 1; sp <= smax; ++sp) {
      const long c = (nchunks + sp - 1) / sp, se = (nchunks + c - 1) / c;
      const long waves = (tiles * se + slots - 1) / slots;
      const long cost = waves * (c + ovh);
      if (best < 0 || cost < best_cost) { best = se; best_cost = cost; }
    }
    want = best;
  } else {
    want = (1536 + tiles - 1) / tiles;          // ~6 workgroups per CU overall (2 resident): measured best for long loops
    if (nchunks / want < 64) {                       // short reductions: fewer, longer splits amortise prologue/epilogue
      want = (1024 + tiles - 1) / tiles;
      if (want > nchunks / 8) want = nchunks / 8;
    }
  }
  cps = segsde_cdiv(nchunks, want);
  splits = segsde_cdiv(nchunks, cps);
}
}  // namespace

namespace {
bool c1_wgrad_route(const segsde_conv_desc* d) {
  return d->Cout == 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->C1 == 0 && !d->up0 &&
         d->H == d->Ho && d->W == d->Wo && d->pad_mode != SEGSDE_PAD_REFLECT_ADJOINT && segsde_c1_supported(d->C0, d->ld0);
}
}  // namespace

extern "C" size_t segsde_conv2d_wgrad_workspace(const segsde_conv_desc* d) {
  if (validate(d)) return 0;
  if (c1_wgrad_route(d)) return segsde_c1_wgrad_workspace(d->C0);
  int bkt, bn, splits, cps;
  wgrad_plan(d, bkt, bn, splits, cps);
  return (size_t)splits * d->KH * d->KW * (d->C0 + d->C1) * d->Cout * sizeof(float);
}

extern "C" int segsde_conv2d_wgrad(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy,
                                   int lddy, float* dw_oihw, float* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate(d)) return e;
  if (!x0 || !dy || !dw_oihw || !workspace || (d->C1 && !x1)) return SEGSDE_ERR_NULL;
  if (workspace_bytes < segsde_conv2d_wgrad_workspace(d)) return SEGSDE_ERR_WORKSPACE;
  if (c1_wgrad_route(d) && aligned16(x0))
    return segsde_c1_wgrad(x0, d->ld0, d->B, d->H, d->W, d->C0, dy, lddy, d->pad_mode == SEGSDE_PAD_REFLECT, dw_oihw, workspace,
                           stream);
  ConvP p = make_params(d, x0, x1, dy /*unused as w; keeps alignment test meaningful*/, nullptr, workspace, nullptr);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int bkt, bn, splits, cps;
  wgrad_plan(d, bkt, bn, splits, cps);
  int e;
  const int srcmaj = (wgrad_mode(p, dy, lddy) == 2 && p.C1 > 0) ? p.C0 : 0;
  const WRed wr = make_wred(p, bkt, bn, dw_oihw, p.Ctot, 0, d->KH * d->KW, srcmaj);
  if (bn == 32) e = launch_wgrad<128, 32, 4, 1>(p, dy, lddy, workspace, splits, cps, s, wr);
  else if (bn == 64) e = launch_wgrad<128, 64, 2, 2>(p, dy, lddy, workspace, splits, cps, s, wr);
  else e = launch_wgrad<128, 128, 2, 2>(p, dy, lddy, workspace, splits, cps, s, wr);
  if (e) return e;
  if (wr.tickets) return 0;
  const long total = (long)p.Ktot * p.N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(segsde_cdiv(total, 32)), dim3(256), 1024, s, workspace, splits,
                     p.Ktot, p.N, p.Ctot, d->KH * d->KW, (wgrad_mode(p, dy, lddy) == 2 && p.C1 > 0) ? p.C0 : 0, dw_oihw, p.Ctot, 0);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Upsample-folded 3x3 convolutions (round 3)
// ---------------------------------------------------------------------------------------------------
// models/depth_decoder.py:93-101 runs Conv3x3 (ReflectionPad2d(1) + 3x3, monodepth_layers.py:127-142) on
// [upsample(x) | skip]: nearest x2 upsampling followed by a 3x3 window.  On the upsampled channels the nine taps of an
// output pixel (2i+py, 2j+px) land on only 2x2 DISTINCT low-resolution pixels -- rows {i-1, i} for py = 0 and {i, i+1} for
// py = 1, columns likewise -- and the mirrored padding of the upsampled image is clamping on the low-resolution grid (row -1
// mirrors to row 1 = low-resolution row 0).  So per parity class (py, px) the upsampled half of the convolution is a 2x2
// convolution of the LOW-resolution tensor with pre-summed weights  Wf[class][th][tw] = sum_{kh in S(py,th)} sum_{kw in
// S(px,tw)} W[kh][kw],  S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}:  4 instead of 9 multiply-adds per
// upsampled channel, an exact identity up to the association of the weight sums.  The three directions become
//   forward : 4 class launches (2x2, clamp padding, low-resolution source, strided sub-grid store) + the ordinary 3x3 on the
//             skip channels, which accumulates onto them and applies bias + activation;
//   dgrad   : d/d(low-resolution x) = a 4x4 stride-2 convolution of dY (16 (class, tap) pairs instead of 36 tap visits per
//             2x2 block) + the clamp adjoint on the border pixels (upfold_dgrad_fix_kernel); d/d(skip) = the ordinary
//             reflection-adjoint data-gradient restricted to the skip channels;
//   wgrad   : 4 class launches (K = 2*2*C0 columns each, dY read on the strided sub-grid), unfolded into the 3x3 taps by
//             upfold_wgrad_reduce_kernel (dW[kh][kw] = sum over the four classes of the folded tap it belongs to) + the
//             ordinary weight gradient of the skip channels.
namespace {
__device__ __host__ inline int upfold_th(int py, int kh) { return py == 0 ? (kh == 0 ? 0 : 1) : (kh == 2 ? 1 : 0); }

// OIHW [Cout][Ctot][3][3] -> wf [4][Cout][2][2][C0] (forward / fix-up) and wd [C0][4][4][Cout] (data-gradient: tap a of the
// 4x4 stride-2 kernel stands for (class parity, folded tap) = (1,1), (0,1), (1,0), (0,0) for a = 0..3, rows and columns alike)
__global__ __launch_bounds__(256) void upfold_pack_kernel(const float* w, int Cout, int C0, int Ctot, float* wf, float* wd) {
  const long total = 4L * Cout * 4 * C0;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C0); long t = e / C0;
    const int tw = (int)(t & 1), th = (int)((t >> 1) & 1); t >>= 2;
    const int n = (int)(t % Cout), cls = (int)(t / Cout);
    const int py = cls >> 1, px = cls & 1;
    const float* wp = w + ((long)n * Ctot + c) * 9;
    float sum = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      if (upfold_th(py, kh) != th) continue;
      float r = 0.f;
      for (int kw = 0; kw < 3; ++kw)
        if (upfold_th(px, kw) == tw) r += wp[kh * 3 + kw];
      sum += r;
    }
    wf[e] = sum;
    const int a = 3 - 2 * th - py, b = 3 - 2 * tw - px;
    wd[(((long)c * 4 + a) * 4 + b) * Cout + n] = sum;
  }
}

// clamp adjoint of the folded data-gradient: a low-resolution border pixel (row 0 / H2-1, column 0 / W2-1) also receives
// what its clamped taps read -- row 0 through tap a = 3 from dY row 0 (class py = 0, folded tap th = 0 of output row 0),
// row H2-1 through a = 0 from dY row H-1; columns alike.  One thread per (border pixel, channel); dY values are wave-uniform,
// the forward-folded weights are read channel-contiguous.  agy (nullable): the saved activation output the gradient is
// multiplied with (same rule as the main launch's epilogue).
__global__ __launch_bounds__(256) void upfold_dgrad_fix_kernel(const float* dy, int lddy, const float* wf, float* dx, int lddx,
                                                               int B, int H2, int W2, int C0, int Cout, const float* agy,
                                                               int agld, int agkind) {
  const int nbord = H2 >= 2 ? 2 * W2 + 2 * (H2 - 2) : W2;
  const long total = (long)B * nbord * C0;
  const int H = 2 * H2, W = 2 * W2;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C0); long t = e / C0;
    const int q = (int)(t % nbord), b = (int)(t / nbord);
    int i, j;
    if (q < W2) { i = 0; j = q; }
    else if (q < 2 * W2) { i = H2 - 1; j = q - W2; }
    else { const int r = q - 2 * W2; i = 1 + (r >> 1); j = (r & 1) ? W2 - 1 : 0; }
    float acc = 0.f;
    for (int a = 0; a < 4; ++a) {
      const int rm = 2 * i - 1 + a;
      const int rx = (i == 0 && a == 3) ? 0 : ((i == H2 - 1 && a == 0) ? H - 1 : -1);
      const int py = (a & 1) ? 0 : 1, th = a >= 2 ? 0 : 1;          // a = 3 - 2 th - py
      for (int bb = 0; bb < 4; ++bb) {
        const int cm = 2 * j - 1 + bb;
        const int cx = (j == 0 && bb == 3) ? 0 : ((j == W2 - 1 && bb == 0) ? W - 1 : -1);
        if (rx < 0 && cx < 0) continue;
        const int px = (bb & 1) ? 0 : 1, tw = bb >= 2 ? 0 : 1;
        const float* wrow = wf + (((long)(py * 2 + px) * Cout) * 4 + th * 2 + tw) * C0 + c;
        for (int v = 0; v < 3; ++v) {      // (extra row, main column), (main row, extra column), (extra row, extra column)
          const int r = v == 1 ? rm : rx, sc = v == 0 ? cm : cx;
          if (r < 0 || r >= H || sc < 0 || sc >= W) continue;
          const float* dp = dy + ((long)(b * H + r) * W + sc) * lddy;
          float part = 0.f;
          for (int n = 0; n < Cout; ++n) part += dp[n] * wrow[(long)n * 4 * C0];
          acc += part;
        }
      }
    }
    const long pix = (long)(b * H2 + i) * W2 + j;
    if (agy) acc *= segsde_act_grad_from_out(agy[pix * agld + c], agkind);
    dx[pix * lddx + c] += acc;
  }
}

// the four corner pixels of an image collect (extra row, extra column) as well: dY at the full-resolution corner through the
// corner tap of the 4x4 kernel.  One thread per (image, corner, channel).
__global__ __launch_bounds__(256) void upfold_dgrad_corner_kernel(const float* dy, int lddy, const float* wdfold, float* dx, int lddx,
                                                                  int B, int H2, int W2, int C0, int Cout, const float* agy,
                                                                  int agld, int agkind) {
  const int total = B * 4 * C0;
  const int H = 2 * H2, W = 2 * W2;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c = e % C0, t = e / C0, k = t & 3, b = t >> 2;
    const int bot = k >> 1, rgt = k & 1;
    const int i = bot ? H2 - 1 : 0, j = rgt ? W2 - 1 : 0, r = bot ? H - 1 : 0, sc = rgt ? W - 1 : 0;
    const int a = bot ? 0 : 3, bb = rgt ? 0 : 3;
    const float* dp = dy + ((long)(b * H + r) * W + sc) * lddy;
    const float* wp = wdfold + (((long)c * 4 + a) * 4 + bb) * Cout;
    float acc = 0.f;
    for (int n = 0; n < Cout; ++n) acc += dp[n] * wp[n];
    const long pix = (long)(b * H2 + i) * W2 + j;
    if (agy) acc *= segsde_act_grad_from_out(agy[pix * agld + c], agkind);
    dx[pix * lddx + c] += acc;
  }
}

// dW[n][c][kh][kw] (c < C0, OIHW with CtotDst channels) = sum over the four classes and the splits of the folded tap's partial
// slab part[class][z][(th*2 + tw) * C0 + c][n]: fixed order (class-major, then splits), deterministic
__global__ __launch_bounds__(256) void upfold_wgrad_reduce_kernel(const float* part, int splits, int C0, int N, float* dw,
                                                                  int CtotDst) {
  // 32 consecutive (c, tap, n) elements (n fastest: the slab reads coalesce) x 8 split-lanes per block, like wgrad_reduce_kernel
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);
  const long total = (long)N * C0 * 9;
  const long slab = 4L * C0 * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long e = blockIdx.x * 32L + tx;
  float acc = 0.f;
  int n = 0, tap = 0, c = 0;
  if (e < total) {
    n = (int)(e % N); const long t = e / N;
    tap = (int)(t % 9); c = (int)(t / 9);
    const int kh = tap / 3, kw = tap - kh * 3;
    for (int cls = 0; cls < 4; ++cls) {
      const int th = upfold_th(cls >> 1, kh), tw = upfold_th(cls & 1, kw);
      const float* src = part + (long)cls * splits * slab + ((long)(th * 2 + tw) * C0 + c) * N + n;
      float s0 = 0.f, s1 = 0.f;
      int z = ty;
      for (; z + 8 < splits; z += 16) { s0 += src[(long)z * slab]; s1 += src[(long)(z + 8) * slab]; }
      if (z < splits) s0 += src[(long)z * slab];
      acc += s0 + s1;
    }
  }
  sh[ty * 32 + tx] = acc;
  __syncthreads();
  if (ty == 0 && e < total) {
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += sh[j * 32 + tx];
    dw[((long)n * CtotDst + c) * 9 + tap] = s;
  }
}

bool upfold_shape_ok(const segsde_conv_desc* d) {
  return d->up0 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->pad_mode == SEGSDE_PAD_REFLECT &&
         d->H == d->Ho && d->W == d->Wo && !(d->H & 1) && !(d->W & 1) && d->H >= 4 && d->W >= 4 && d->C0 % 32 == 0 &&
         d->C1 % 32 == 0 && d->Cout % 4 == 0 && d->in_div <= 1 && !d->sum2x2 && !d->accumulate;
}

// class launch of the forward fold: 2x2 clamp-padded convolution of the low-resolution source, stored to sub-grid (py, px)
ConvP upfold_class_fwd(const segsde_conv_desc* d, const float* x0, const float* wf, const float* bias, float* y, int py, int px) {
  segsde_conv_desc c = *d;
  c.H = d->H / 2; c.W = d->W / 2; c.Ho = c.H; c.Wo = c.W; c.up0 = 0; c.C1 = 0; c.ld1 = 0; c.KH = 2; c.KW = 2;
  c.pad = 1 - py; c.pad_mode = SEGSDE_PAD_ZERO; c.nsplit = 0; c.ldy2 = 0;
  if (d->C1) c.act = 0;                               // bias + activation belong to the launch that completes the sum
  ConvP q = make_params(&c, x0, nullptr, wf + (long)(py * 2 + px) * d->Cout * 4 * d->C0, d->C1 ? nullptr : bias, y, nullptr);
  q.pad_mode = SEGSDE_PAD_CLAMP_; q.padw = 1 - px;
  q.os = 2; q.submap = 1; q.oph = py; q.opw = px; q.OHf = d->H; q.OWf = d->W;
  q.osfast = (c.Wo % 128 == 0) ? 1 : 0;
  q.lin = 0;
  return q;
}
}  // namespace

extern "C" int segsde_upfold_pack(const float* w_oihw, int Cout, int C0, int Ctot, float* wfold, float* wdfold, void* stream) {
  if (!w_oihw || !wfold || !wdfold) return SEGSDE_ERR_NULL;
  if (Cout <= 0 || C0 <= 0 || Ctot < C0) return SEGSDE_ERR_SHAPE;
  const long total = 16L * Cout * C0;
  hipLaunchKernelGGL(upfold_pack_kernel, dim3(min(2048, segsde_cdiv(total, 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_oihw, Cout, C0, Ctot, wfold, wdfold);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_conv2d_forward_upfold(const segsde_conv_desc* d, const float* x0, const float* x1, const float* wpack,
                                            const float* wfold, const float* bias, float* y, void* stream) {
  if (int e = validate(d)) return e;
  if (!x0 || !wpack || !wfold || !y || (d->C1 && !x1)) return SEGSDE_ERR_NULL;
  if (!upfold_shape_ok(d) || d->ldy % 4 || !aligned16(y) || !aligned16(wfold)) return SEGSDE_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvP cls[4];
  for (int k = 0; k < 4; ++k) {
    cls[k] = upfold_class_fwd(d, x0, wfold, bias, y, k >> 1, k & 1);
    if (!igemm_fast_ok(cls[k]) || !cls[k].vecout) return SEGSDE_ERR_UNSUPPORTED;
  }
  ConvP r;
  if (d->C1) {
    // the skip channels: the ordinary reflection-padded 3x3 over source 1 alone, reading its channel slice of the packed rows,
    // added onto the class launches' sums; bias and activation are applied here
    segsde_conv_desc c = *d;
    c.C0 = d->C1; c.C1 = 0; c.ld0 = d->ld1; c.ld1 = 0; c.up0 = 0; c.accumulate = 1; c.nsplit = 0; c.ldy2 = 0;
    r = make_params(&c, x1, nullptr, wpack + d->C0, bias, y, nullptr);
    r.wtap = d->C0 + d->C1; r.Kfull = 9 * (d->C0 + d->C1);
    if (!igemm_fast_ok(r) || !r.vecout || (long)r.N * r.Kfull * 4 >= (1L << 31)) return SEGSDE_ERR_UNSUPPORTED;
  }
  for (int k = 0; k < 4; ++k)
    if (int e = launch_by_n(cls[k], s)) return e;
  if (d->C1)
    if (int e = launch_by_n(r, s)) return e;
  return 0;
}

extern "C" int segsde_conv2d_dgrad_upfold(const segsde_conv_desc* d, const float* dy, int lddy, const float* wdpack,
                                          const float* wfold, const float* wdfold, float* dx0, float* dx1, int accumulate_dx1,
                                          const float* act_out, int act_ld, int act_kind, void* stream) {
  // d: the FORWARD geometry (H x W virtual input, C0 upsampled + C1 skip channels, Cout).  dx0 [B,H/2,W/2,C0] dense (nullable),
  // dx1 [B,H,W,C1] dense (nullable); act_out: the saved activation output dx0 is differentiated through (nullable)
  if (int e = validate(d)) return e;
  if (!dy || !wdpack || !wfold || !wdfold || (!dx0 && !dx1)) return SEGSDE_ERR_NULL;
  if (!upfold_shape_ok(d) || lddy % 4 || !aligned16(dy) || d->Cout % 32) return SEGSDE_ERR_UNSUPPORTED;
  if (act_out && (act_kind < SEGSDE_ACT_RELU || act_kind > SEGSDE_ACT_SIGMOID || act_ld < d->C0 || (act_ld % 4) || !aligned16(act_out)))
    return SEGSDE_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int H2 = d->H / 2, W2 = d->W / 2;
  ConvP q, r;
  if (dx0) {
    segsde_conv_desc c = *d;            // 4x4 stride-2 zero-padded convolution of dY -> low-resolution gradient
    c.C0 = d->Cout; c.C1 = 0; c.ld0 = lddy; c.ld1 = 0; c.up0 = 0; c.Ho = H2; c.Wo = W2; c.Cout = d->C0; c.ldy = d->C0; c.ldy2 = 0;
    c.nsplit = 0; c.KH = 4; c.KW = 4; c.stride = 2; c.dil = 1; c.pad = 1; c.pad_mode = SEGSDE_PAD_ZERO; c.act = 0;
    q = make_params(&c, dy, nullptr, wdfold, nullptr, dx0, nullptr);
    if (act_out) { q.agy = act_out; q.agld = act_ld; q.agkind = act_kind; }
    if (!igemm_fast_ok(q) || !q.vecout) return SEGSDE_ERR_UNSUPPORTED;
  }
  if (dx1 && d->C1) {
    segsde_conv_desc c = *d;            // reflection-adjoint data-gradient of the skip channels: rows C0.. of the dgrad pack
    c.C0 = d->Cout; c.C1 = 0; c.ld0 = lddy; c.ld1 = 0; c.up0 = 0; c.Cout = d->C1; c.ldy = d->C1; c.ldy2 = 0; c.nsplit = 0;
    c.pad_mode = SEGSDE_PAD_REFLECT_ADJOINT; c.act = 0;
    c.accumulate = accumulate_dx1 ? 1 : 0;   // dx1 already holds another consumer's gradient of the skip tensor: add onto it
    if (int e = validate(&c)) return e;
    r = make_params(&c, dy, nullptr, wdpack + (long)d->C0 * 9 * d->Cout, nullptr, dx1, nullptr);
    if (!igemm_fast_ok(r) || !r.vecout) return SEGSDE_ERR_UNSUPPORTED;
  }
  if (dx0) {
    if (int e = launch_by_n(q, s)) return e;
    // Clamp adjoint.  The border pixels of the low-resolution gradient also collect what their clamped taps read: row 0 through
    // the taps a = 3 from dY row 0, row H2-1 through a = 0 from dY row H-1, columns alike.  (extra row) x (main columns) is a
    // 1x4 stride-2 convolution of ONE dY row into one low-resolution row, (main rows) x (extra column) a 4x1 one into one
    // column: four small launches of the same matrix-core kernel that ADD onto the main launch's result (same fused activation
    // derivative), reading their taps in place inside the 4x4 pack; (extra row) x (extra column) exists at the four corners.
    // (The first version did all of this with one thread per border pixel and channel: 0.4 ms per layer.)
    bool ok = true;
    ConvP bl[4];
    for (int k = 0; k < 4 && ok; ++k) {
      const bool rowl = k < 2, far = k & 1;        // 0 top row, 1 bottom row, 2 left column, 3 right column
      segsde_conv_desc c = *d;
      c.C0 = d->Cout; c.C1 = 0; c.ld0 = lddy; c.ld1 = 0; c.up0 = 0; c.Cout = d->C0; c.ldy = d->C0; c.ldy2 = 0; c.nsplit = 0;
      c.Ho = rowl ? 1 : H2; c.Wo = rowl ? W2 : 1; c.KH = rowl ? 1 : 4; c.KW = rowl ? 4 : 1; c.stride = 2; c.dil = 1;
      c.pad = rowl ? (far ? -(d->H - 1) : 0) : 1; c.pad_mode = SEGSDE_PAD_ZERO; c.act = 0; c.accumulate = 1;
      const int a = far ? 0 : 3;                   // the clamped tap
      ConvP b = make_params(&c, dy, nullptr, wdfold + (long)(rowl ? a * 4 : a) * d->Cout, nullptr, dx0, nullptr);
      b.padw = rowl ? 1 : (far ? -(d->W - 1) : 0);
      b.Kfull = 16 * d->Cout; b.wtap = rowl ? d->Cout : 4 * d->Cout;
      b.submap = 1; b.os = 1; b.OHf = H2; b.OWf = W2; b.oph = rowl ? (far ? H2 - 1 : 0) : 0; b.opw = rowl ? 0 : (far ? W2 - 1 : 0);
      b.osfast = (rowl && W2 % 128 == 0) ? 1 : 0;
      b.lin = 0;
      if (act_out) { b.agy = act_out; b.agld = act_ld; b.agkind = act_kind; }
      if (!igemm_fast_ok(b) || !b.vecout) ok = false;
      bl[k] = b;
    }
    if (ok) {
      for (int k = 0; k < 4; ++k)
        if (int e = launch_by_n(bl[k], s)) return e;
      hipLaunchKernelGGL(upfold_dgrad_corner_kernel, dim3(segsde_cdiv((long)d->B * 4 * d->C0, 256)), dim3(256), 0, s, dy, lddy, wdfold,
                         dx0, d->C0, d->B, H2, W2, d->C0, d->Cout, act_out, act_ld, act_kind);
      SEGSDE_CHECK_LAUNCH();
    } else {
      const long total = (long)d->B * (2 * W2 + 2 * H2) * d->C0;
      hipLaunchKernelGGL(upfold_dgrad_fix_kernel, dim3(min(8192, segsde_cdiv(total, 256))), dim3(256), 0, s, dy, lddy, wfold, dx0,
                         d->C0, d->B, H2, W2, d->C0, d->Cout, act_out, act_ld, act_kind);
      SEGSDE_CHECK_LAUNCH();
    }
  }
  if (dx1 && d->C1) {
    int e = adjoint_by_borders_ok(r) ? launch_adjoint_by_borders(r, s) : SEGSDE_ERR_UNSUPPORTED;
    if (e == SEGSDE_ERR_UNSUPPORTED) e = launch_by_n(r, s);
    if (e) return e;
  }
  return 0;
}

namespace {
struct UpfoldWgradPlan { segsde_conv_desc cls, skip; int bn, splits, cps, bn1, splits1, cps1; size_t ws_fold, ws_skip; };
bool upfold_wgrad_plan(const segsde_conv_desc* d, UpfoldWgradPlan& pl) {
  if (!upfold_shape_ok(d)) return false;
  pl.cls = *d;
  pl.cls.H = d->H / 2; pl.cls.W = d->W / 2; pl.cls.Ho = pl.cls.H; pl.cls.Wo = pl.cls.W; pl.cls.up0 = 0; pl.cls.C1 = 0; pl.cls.ld1 = 0;
  pl.cls.KH = 2; pl.cls.KW = 2; pl.cls.pad = 1; pl.cls.pad_mode = SEGSDE_PAD_ZERO; pl.cls.act = 0;
  int bkt;
  wgrad_plan(&pl.cls, bkt, pl.bn, pl.splits, pl.cps);
  pl.ws_fold = 4 * (size_t)pl.splits * 4 * d->C0 * d->Cout * sizeof(float);
  pl.ws_skip = 0;
  if (d->C1) {
    pl.skip = *d;
    pl.skip.C0 = d->C1; pl.skip.C1 = 0; pl.skip.ld0 = d->ld1; pl.skip.ld1 = 0; pl.skip.up0 = 0; pl.skip.act = 0;
    wgrad_plan(&pl.skip, bkt, pl.bn1, pl.splits1, pl.cps1);
    pl.ws_skip = (size_t)pl.splits1 * 9 * d->C1 * d->Cout * sizeof(float);
  }
  return true;
}
template <typename... A>
int launch_wgrad_by_bn(int bn, A... a) {
  if (bn == 32) return launch_wgrad<128, 32, 4, 1>(a...);
  if (bn == 64) return launch_wgrad<128, 64, 2, 2>(a...);
  return launch_wgrad<128, 128, 2, 2>(a...);
}
}  // namespace

extern "C" size_t segsde_conv2d_wgrad_upfold_workspace(const segsde_conv_desc* d) {
  UpfoldWgradPlan pl;
  if (validate(d) || !upfold_wgrad_plan(d, pl)) return 0;
  return pl.ws_fold + pl.ws_skip;
}

extern "C" int segsde_conv2d_wgrad_upfold(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                                          float* dw_oihw, float* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate(d)) return e;
  if (!x0 || !dy || !dw_oihw || !workspace || (d->C1 && !x1)) return SEGSDE_ERR_NULL;
  UpfoldWgradPlan pl;
  if (!upfold_wgrad_plan(d, pl)) return SEGSDE_ERR_UNSUPPORTED;
  if (workspace_bytes < pl.ws_fold + pl.ws_skip) return SEGSDE_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Ctot = d->C0 + d->C1;
  ConvP cls[4];
  const size_t slab = (size_t)pl.splits * 4 * d->C0 * d->Cout;
  for (int k = 0; k < 4; ++k) {
    const int py = k >> 1, px = k & 1;
    segsde_conv_desc c = pl.cls;
    c.pad = 1 - py;
    ConvP q = make_params(&c, x0, nullptr, dy, nullptr, workspace + k * slab, nullptr);
    q.pad_mode = SEGSDE_PAD_CLAMP_; q.padw = 1 - px;
    q.os = 2; q.oph = py; q.opw = px; q.OHf = d->H; q.OWf = d->W;
    if (wgrad_mode(q, dy, lddy) != 2) return SEGSDE_ERR_UNSUPPORTED;   // the table-driven loader is the one that knows sub-grids
    cls[k] = q;
  }
  ConvP r;
  if (d->C1) {
    r = make_params(&pl.skip, x1, nullptr, dy, nullptr, workspace + 4 * slab, nullptr);
  }
  for (int k = 0; k < 4; ++k)
    if (int e = launch_wgrad_by_bn(pl.bn, cls[k], dy, lddy, workspace + k * slab, pl.splits, pl.cps, s, WRed{})) return e;
  {
    const long total = (long)d->Cout * d->C0 * 9;
    hipLaunchKernelGGL(upfold_wgrad_reduce_kernel, dim3(segsde_cdiv(total, 32)), dim3(256), 1024, s, workspace, pl.splits,
                       d->C0, d->Cout, dw_oihw, Ctot);
    SEGSDE_CHECK_LAUNCH();
  }
  if (d->C1) {
    float* ws1 = workspace + 4 * slab;
    const WRed wr = make_wred(r, 128, pl.bn1, dw_oihw, Ctot, d->C0, 9, 0);
    if (int e = launch_wgrad_by_bn(pl.bn1, r, dy, lddy, ws1, pl.splits1, pl.cps1, s, wr)) return e;
    if (wr.tickets) return 0;
    const long total = (long)r.Ktot * r.N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(segsde_cdiv(total, 32)), dim3(256), 1024, s, ws1, pl.splits1, r.Ktot, r.N, r.Ctot, 9, 0,
                       dw_oihw, Ctot, d->C0);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Network stems (round 3): 7x7, stride 2, pad 3 on 3 / 6 input planes (resnet_encoder.py:40-52, :90-93)
// ---------------------------------------------------------------------------------------------------
// With 4 / 8 channels per pixel a 32-float reduction chunk is 8 / 4 neighbouring PIXELS of one input row, so a stem is not a
// FAST-path shape (one tap per chunk) and ran on the generic float4 gather (71-87 TFLOP/s, ~40 VALU instructions of tap decode
// per chunk and thread).  Here the layout kernel writes the normalised input with a zero border (3 rows above / below, 3
// columns left, 5 right), and the convolution is described to the LDS-DMA kernel as a 7x1 convolution over VIRTUAL 32-channel
// pixels with the real pixel pitch (4 / 8 floats): "channel" c of virtual pixel (h, w) is float c behind real pixel (h, w), i.e.
// the 8 real pixels from it to the right (the 7 taps of the row + one zero weight) are the 32 / 64 channels of one tap.  No
// padding logic is left (pad = 0 on the bordered tensor), the loop is the ordinary zero-VALU one; K = 7*32 / 7*64 instead of
// 49*4 / 49*8.
namespace {
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* w, int Cout, int C, int cp, float* out) {
  const int G = 8 * cp;
  const long total = (long)Cout * 7 * G;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int g = (int)(e % G); long t = e / G;
    const int kh = (int)(t % 7), n = (int)(t / 7);
    const int px = g / cp, ch = g - px * cp;
    out[e] = (px < 7 && ch < C) ? w[(((long)n * C + ch) * 7 + kh) * 7 + px] : 0.f;
  }
}
// dwp [Cout][8*cp][7] (what the weight-gradient reduce writes: n, virtual channel g = px*cp + ch, kh) -> OIHW [Cout][C][7][7]
__global__ __launch_bounds__(256) void stem_unpack_kernel(const float* dwp, int Cout, int C, int cp, float* dw) {
  const long total = (long)Cout * C * 49;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int px = (int)(e % 7); long t = e / 7;
    const int kh = (int)(t % 7); t /= 7;
    const int ch = (int)(t % C), n = (int)(t / C);
    dw[e] = dwp[((long)n * 8 * cp + px * cp + ch) * 7 + kh];
  }
}
bool stem_desc(int B, int Hp, int Wp, int cp, int Cout, segsde_conv_desc& c) {
  if (B <= 0 || Hp < 8 || Wp < 10 || (cp != 4 && cp != 8) || Cout <= 0 || Cout % 4) return false;
  const int H = Hp - 6, W = Wp - 8;
  memset(&c, 0, sizeof(c));
  c.B = B; c.H = Hp; c.W = Wp; c.C0 = 8 * cp; c.C1 = 0; c.ld0 = cp; c.ld1 = 0; c.up0 = 0;
  c.Ho = (H - 1) / 2 + 1; c.Wo = (W - 1) / 2 + 1; c.Cout = Cout; c.ldy = Cout;
  c.KH = 7; c.KW = 1; c.stride = 2; c.dil = 1; c.pad = 0; c.pad_mode = SEGSDE_PAD_ZERO; c.in_div = 1;
  return true;
}
}  // namespace

extern "C" int segsde_stem_pack(const float* w_oihw, int Cout, int C, int cp, float* wstem, void* stream) {
  if (!w_oihw || !wstem) return SEGSDE_ERR_NULL;
  if (Cout <= 0 || C <= 0 || C > cp || (cp != 4 && cp != 8)) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(stem_pack_kernel, dim3(segsde_cdiv((long)Cout * 56 * cp, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_oihw, Cout, C, cp, wstem);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" long segsde_stem7x7_stats_rows(int B, int Hp, int Wp, int cp, int Cout) {
  segsde_conv_desc c;
  if (!stem_desc(B, Hp, Wp, cp, Cout, c)) return 0;
  float dummy[4];
  const ConvP p = make_params(&c, dummy, dummy, dummy, nullptr, reinterpret_cast<float*>(16), nullptr);
  return stats_rows(&c, p);
}

extern "C" int segsde_stem7x7_forward(const float* xpad, int B, int Hp, int Wp, int cp, const float* wstem, int Cout, float* y,
                                      double* stats, void* stream) {
  if (!xpad || !wstem || !y) return SEGSDE_ERR_NULL;
  segsde_conv_desc c;
  if (!stem_desc(B, Hp, Wp, cp, Cout, c)) return SEGSDE_ERR_SHAPE;
  ConvP p = make_params(&c, xpad, nullptr, wstem, nullptr, y, nullptr);
  if (!igemm_fast_ok(p) || !p.vecout) return SEGSDE_ERR_UNSUPPORTED;
  if (stats) {
    if (stats_rows(&c, p) == 0) return SEGSDE_ERR_UNSUPPORTED;
    p.stats = stats;
  }
  return launch_by_n(p, static_cast<hipStream_t>(stream));
}

extern "C" size_t segsde_stem7x7_wgrad_workspace(int B, int Hp, int Wp, int cp, int Cout) {
  segsde_conv_desc c;
  if (!stem_desc(B, Hp, Wp, cp, Cout, c)) return 0;
  int bkt, bn, splits, cps;
  wgrad_plan(&c, bkt, bn, splits, cps);
  return ((size_t)splits + 1) * 7 * 8 * cp * Cout * sizeof(float);      // split slabs + the [Cout][8 cp][7] gradient before unpacking
}

extern "C" int segsde_stem7x7_wgrad(const float* xpad, int B, int Hp, int Wp, int cp, const float* dy, int lddy, int Cout, int C,
                                    float* dw_oihw, float* workspace, size_t workspace_bytes, void* stream) {
  if (!xpad || !dy || !dw_oihw || !workspace) return SEGSDE_ERR_NULL;
  segsde_conv_desc c;
  if (!stem_desc(B, Hp, Wp, cp, Cout, c) || C <= 0 || C > cp) return SEGSDE_ERR_SHAPE;
  if (workspace_bytes < segsde_stem7x7_wgrad_workspace(B, Hp, Wp, cp, Cout)) return SEGSDE_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvP p = make_params(&c, xpad, nullptr, dy, nullptr, workspace, nullptr);
  const int mode = wgrad_mode(p, dy, lddy);
  if (mode != 2 && mode != 3) return SEGSDE_ERR_UNSUPPORTED;
  int bkt, bn, splits, cps;
  wgrad_plan(&c, bkt, bn, splits, cps);
  if (int e = launch_wgrad_by_bn(bn, p, dy, lddy, workspace, splits, cps, s, WRed{})) return e;
  float* dwp = workspace + (size_t)splits * p.Ktot * p.N;
  const long total = (long)p.Ktot * p.N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(segsde_cdiv(total, 32)), dim3(256), 1024, s, workspace, splits, p.Ktot, p.N, p.Ctot, 7,
                     (mode == 2 && p.C1 > 0) ? p.C0 : 0, dwp, p.Ctot, 0);
  SEGSDE_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_unpack_kernel, dim3(segsde_cdiv((long)Cout * C * 49, 256)), dim3(256), 0, s, dwp, Cout, C, cp, dw_oihw);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_pack_weight(const float* w_oihw, float* out, int O, int I, int KH, int KW, int for_dgrad,
                                  void* stream) {
  if (!w_oihw || !out) return SEGSDE_ERR_NULL;
  if (O <= 0 || I <= 0 || KH <= 0 || KW <= 0) return SEGSDE_ERR_SHAPE;
  const long total = (long)O * I * KH * KW;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(min(2048, segsde_cdiv(total, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w_oihw, out, O, I, KH, KW, for_dgrad);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_pack_weight_both(const float* w_oihw, float* out_fwd, float* out_dgrad, int O, int I, int KH, int KW,
                                       void* stream) {
  if (!w_oihw || !out_fwd || !out_dgrad) return SEGSDE_ERR_NULL;
  if (O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || (long)O * I * KH * KW >= (1L << 31)) return SEGSDE_ERR_SHAPE;
  const long total = (long)O * I * KH * KW;
  hipLaunchKernelGGL(pack_weight_both_kernel, dim3(min(2048, segsde_cdiv(total, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w_oihw, out_fwd, out_dgrad, O, I, KH, KW);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

namespace {
// every convolution weight of a model in ONE launch (a step re-packs ~160 weights: one ~10 us launch each otherwise)
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const segsde_pack_job* jobs, int njobs) {
  SEGSDE_SMEM;
  float* tile = reinterpret_cast<float*>(segsde_smem);   // [o][i][tap] of one 32 x 32 (output, input) channel block
  int lo = 0, hi = njobs;                       // jobs[j].block0 <= blockIdx.x < jobs[j + 1].block0 (sentinel at njobs)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= jobs[mid].block0) lo = mid; else hi = mid;
  }
  const segsde_pack_job j = jobs[lo];
  const int nblk = jobs[lo + 1].block0 - j.block0, lb = (int)blockIdx.x - j.block0;
  const int T = j.KH * j.KW;
  if (T <= 9 && j.reserved == 1) {
    // transposing path (reserved == 1: the host gave this job ceil(O/32) * ceil(I/32) blocks): the block's 32 x 32 x T
    // weights are read as 32 runs of 32*T consecutive floats and leave as 128-byte runs along I (forward pack) and along
    // O (flipped data-gradient pack) -- the element-per-thread path below writes the flipped pack 4 bytes at a time
    const int nit = (j.I + 31) >> 5, o0 = (lb / nit) << 5, i0 = (lb % nit) << 5;
    const int nI = j.I - i0 < 32 ? j.I - i0 : 32, nO = j.O - o0 < 32 ? j.O - o0 : 32, run = nI * T;
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
      const int ol = e / run, r = e - ol * run;
      if (ol < nO) tile[ol * (32 * T) + r] = j.w[((long)(o0 + ol) * j.I + i0) * T + r];     // r = il * T + t
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * T * 32; e += 256) {
      const int il = e & 31, t = (e >> 5) % T, ol = (e >> 5) / T;
      if (il < nI && ol < nO) j.fwd[((long)(o0 + ol) * T + t) * j.I + i0 + il] = tile[ol * (32 * T) + il * T + t];
    }
    for (int e = threadIdx.x; e < 32 * T * 32; e += 256) {
      const int ol = e & 31, t = (e >> 5) % T, il = (e >> 5) / T;
      if (il < nI && ol < nO) j.dgrad[((long)(i0 + il) * T + (T - 1 - t)) * j.O + o0 + ol] = tile[ol * (32 * T) + il * T + t];
    }
    return;
  }
  const int total = j.O * j.I * T;
  for (int e = lb * 256 + threadIdx.x; e < total; e += nblk * 256) {
    const int kw = e % j.KW; int t = e / j.KW;
    const int kh = t % j.KH; t /= j.KH;
    const int i = t % j.I, o = t / j.I;
    const float v = j.w[e];
    j.fwd[((o * j.KH + kh) * j.KW + kw) * j.I + i] = v;
    j.dgrad[((i * j.KH + (j.KH - 1 - kh)) * j.KW + (j.KW - 1 - kw)) * j.O + o] = v;
  }
}
}  // namespace

extern "C" int segsde_pack_weight_both_multi(const segsde_pack_job* jobs_device, int njobs, int total_blocks, void* stream) {
  if (!jobs_device) return SEGSDE_ERR_NULL;
  if (njobs <= 0 || total_blocks <= 0) return SEGSDE_ERR_SHAPE;
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(total_blocks), dim3(256), 32 * 32 * 9 * sizeof(float), static_cast<hipStream_t>(stream), jobs_device,
                     njobs);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int segsde_reflect_dgrad_fix(const float* dy, int lddy, const float* wdpack, float* dx, int lddx, float* dx2,
                                        int lddx2, int nsplit, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!dy || !wdpack || !dx) return SEGSDE_ERR_NULL;
  if (H < 2 || W < 2) return SEGSDE_ERR_SHAPE;
  return launch_reflect_fix(dy, lddy, wdpack, dx, lddx, dx2, lddx2, nsplit, B, H, W, Cin, Cout,
                            static_cast<hipStream_t>(stream));
}


// ---------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) route of the stride-1 3x3 convolutions with many channels (round 4; transforms: winograd.hip)
// ---------------------------------------------------------------------------------------------------
// forward AND data-gradient (the data-gradient of a 3x3 / stride 1 / pad 1 convolution is the same convolution of dY with the
// flipped, transposed kernel: segsde_winograd_pack writes both transformed packs).  One call = input transform -> ONE launch of
// the LDS-DMA implicit-GEMM kernel over the sixteen transform positions (a 1x1 convolution of a 16-"image" tensor whose weight
// base advances with the image index: 16 GEMMs [T x C] x [C x Cout], T = B * H/2 * W/2) -> output transform (+ the BatchNorm
// statistics partials of the output, like the implicit-GEMM epilogue's).  16 instead of 36 multiply-adds per 2x2 output block,
// channel and filter; the transforms add ~4 B/element of cache-resident traffic each way.
namespace {
bool winograd_shape_ok(const segsde_conv_desc* d) {
  if (validate(d)) return false;
  const long T = (long)d->B * (d->H / 2) * (d->W / 2);
  const int C = d->C0 + d->C1;
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil >= 1 && d->pad == d->dil && !d->up0 && d->in_div <= 1 && !d->sum2x2 &&
         !d->accumulate && (d->pad_mode == SEGSDE_PAD_ZERO || (d->pad_mode == SEGSDE_PAD_REFLECT && d->dil == 1)) && d->H == d->Ho &&
         d->W == d->Wo && d->H % (2 * d->dil) == 0 && d->W % (2 * d->dil) == 0 && d->H >= 4 * d->dil && d->W >= 4 * d->dil &&
         C % 32 == 0 && d->C0 % 4 == 0 && d->Cout % 64 == 0 && d->ld0 % 4 == 0 && (!d->C1 || d->ld1 % 4 == 0) && d->ldy % 4 == 0 &&
         16 * segsde_wino_rows(T) * (long)(C > d->Cout ? C : d->Cout) < (1L << 31) && tune().dma;
}
segsde_conv_desc winograd_gemm_desc(const segsde_conv_desc* d) {
  segsde_conv_desc g = *d;
  const long Tp = segsde_wino_rows((long)d->B * (d->H / 2) * (d->W / 2));     // rows per position, whole 128-row tiles
  g.B = 16; g.H = (int)(Tp / 32); g.W = 32; g.C0 = d->C0 + d->C1; g.C1 = 0; g.ld0 = g.C0; g.ld1 = 0; g.Ho = g.H; g.Wo = g.W;
  g.ldy = d->Cout; g.ldy2 = 0; g.nsplit = 0; g.KH = 1; g.KW = 1; g.dil = 1; g.pad = 0; g.pad_mode = SEGSDE_PAD_ZERO; g.act = 0;
  return g;
}
}  // namespace

extern "C" size_t segsde_conv2d_winograd_workspace(const segsde_conv_desc* d) {
  if (!winograd_shape_ok(d)) return 0;
  const size_t T = (size_t)segsde_wino_rows((long)d->B * (d->H / 2) * (d->W / 2));
  return 16 * T * ((size_t)d->C0 + (size_t)d->C1 + (size_t)d->Cout) * sizeof(float) + 256;
}

extern "C" long segsde_conv2d_winograd_stats_rows(const segsde_conv_desc* d) {
  return winograd_shape_ok(d) ? segsde_wino_stats_rows((long)d->B * (d->H / 2) * (d->W / 2)) : 0;
}

extern "C" int segsde_winograd_pack(const float* w_oihw, int Cout, int Cin, float* u_fwd, float* u_dgrad, void* stream) {
  if (!w_oihw || (!u_fwd && !u_dgrad)) return SEGSDE_ERR_NULL;
  if (Cout <= 0 || Cin <= 0) return SEGSDE_ERR_SHAPE;
  if (u_fwd)
    if (int e = segsde_wino_weights(w_oihw, Cout, Cin, 0, u_fwd, stream)) return e;
  if (u_dgrad)
    if (int e = segsde_wino_weights(w_oihw, Cout, Cin, 1, u_dgrad, stream)) return e;
  return 0;
}

extern "C" int segsde_winograd_pack_multi(const segsde_wino_job* jobs_device, int njobs, int total_blocks, void* stream) {
  if (!jobs_device) return SEGSDE_ERR_NULL;
  if (njobs <= 0 || total_blocks <= 0) return SEGSDE_ERR_SHAPE;
  return segsde_wino_weights_multi(jobs_device, njobs, total_blocks, stream);
}

extern "C" int segsde_conv2d_winograd(const segsde_conv_desc* d, const float* x0, const float* x1, const float* u_pack,
                                      const float* bias, float* y, double* stats, float* v_keep, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  // d: the convolution's own geometry (3x3, stride 1, pad = dil; [C0 | C1] -> Cout, activation d->act).  u_pack: [16][Cout][C0 + C1]
  // from segsde_winograd_pack (its forward pack; for a data-gradient call d describes dY -> dX and u_pack is the data-gradient
  // pack).  stats (nullable): [segsde_conv2d_winograd_stats_rows(d)][2][Cout] doubles.
  if (!d || !x0 || !u_pack || !y || !workspace || (d->C1 && !x1)) return SEGSDE_ERR_NULL;
  if (!winograd_shape_ok(d) || !aligned16(x0) || (d->C1 && !aligned16(x1)) || !aligned16(y) || !aligned16(u_pack) ||
      (bias && !aligned16(bias)) || d->act < SEGSDE_ACT_NONE || d->act > SEGSDE_ACT_SIGMOID)
    return SEGSDE_ERR_UNSUPPORTED;
  if (workspace_bytes < segsde_conv2d_winograd_workspace(d)) return SEGSDE_ERR_WORKSPACE;
  const size_t T = (size_t)segsde_wino_rows((long)d->B * (d->H / 2) * (d->W / 2));
  const int C = d->C0 + d->C1;
  // v_keep (nullable, 16 * T * (C0 + C1) floats, 16-byte aligned): the transformed input is written THERE instead of into the
  // workspace -- a training forward keeps it for the weight gradient (segsde_conv2d_wgrad_winograd's v_saved)
  float* W0 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  float* V = v_keep ? v_keep : W0;
  float* Mb = W0 + 16 * T * C;
  if (!aligned16(V)) return SEGSDE_ERR_UNSUPPORTED;
  const segsde_conv_desc g = winograd_gemm_desc(d);
  ConvP q = make_params(&g, V, nullptr, u_pack, nullptr, Mb, nullptr);
  q.wbstride = (long)d->Cout * C;
  if (!igemm_fast_ok(q) || !q.vecout || !q.lin) return SEGSDE_ERR_UNSUPPORTED;
  if (int e = segsde_wino_input(x0, d->ld0, d->C1 ? x1 : nullptr, d->C1 ? d->ld1 : d->ld0, d->C0, d->B, d->H, d->W, C, d->dil,
                                d->pad_mode == SEGSDE_PAD_REFLECT, V, stream))
    return e;
  if (int e = launch_by_n(q, static_cast<hipStream_t>(stream))) return e;
  return segsde_wino_output(Mb, d->B, d->H, d->W, d->Cout, d->dil, bias, d->act, y, d->ldy, stats, stream);
}

// Weight gradient on the same route: dU_p = sum over the tiles of V_p^T dM_p (V = the forward's transformed input, recomputed
// here; dM = A dY A^T) is the weight-gradient kernel's own GEMM form, sixteen times: ONE launch of conv_wgrad_kernel over the
// sixteen-"image" tensors with the split boundaries on the image boundaries (s splits per position), so that the slabs of a
// position are exactly its partial sums; wino_wgrad_finish_kernel folds them in slab order and applies dW = G^T dU G.
namespace {
struct WinoWgradPlan { segsde_conv_desc g; int bn, s, cps; size_t off_dm, off_part, bytes; };
bool winograd_wgrad_plan(const segsde_conv_desc* d, WinoWgradPlan& pl) {
  if (!winograd_shape_ok(d)) return false;
  const long T = segsde_wino_rows((long)d->B * (d->H / 2) * (d->W / 2));
  const int C = d->C0 + d->C1;
  pl.g = winograd_gemm_desc(d);              // rows of 32 tiles: the table-driven loader's chunk
  pl.bn = d->Cout <= 64 ? 64 : 128;
  const long tiles = (long)segsde_cdiv(C, 128) * segsde_cdiv(d->Cout, pl.bn);
  const long cp = T / BP;                          // chunks per position
  int s = 1;
  while (2 * s <= cp && cp % (2 * s) == 0 && cp / (2 * s) >= 8 && tiles * 16 * (2 * s) <= 1024) s *= 2;
  pl.s = s; pl.cps = (int)(cp / s);
  pl.off_dm = 16 * (size_t)T * C * sizeof(float);
  pl.off_part = pl.off_dm + 16 * (size_t)T * d->Cout * sizeof(float);
  pl.bytes = pl.off_part + (size_t)16 * s * C * d->Cout * sizeof(float) + 256;
  return true;
}
}  // namespace

extern "C" size_t segsde_conv2d_wgrad_winograd_workspace(const segsde_conv_desc* d) {
  WinoWgradPlan pl;
  return winograd_wgrad_plan(d, pl) ? pl.bytes : 0;
}

extern "C" int segsde_conv2d_wgrad_winograd(const segsde_conv_desc* d, const float* x0, const float* x1, const float* dy, int lddy,
                                            const float* v_saved, float* dw_oihw, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  // d: the FORWARD geometry ([C0 | C1] -> Cout, 3x3, stride 1, pad = dil, zero or mirrored padding); dy [B,H,W,Cout] (pitch lddy)
  if (!d || !x0 || !dy || !dw_oihw || !workspace || (d->C1 && !x1)) return SEGSDE_ERR_NULL;
  WinoWgradPlan pl;
  if (!winograd_wgrad_plan(d, pl) || !aligned16(x0) || (d->C1 && !aligned16(x1)) || !aligned16(dy) || lddy % 4 || lddy < d->Cout)
    return SEGSDE_ERR_UNSUPPORTED;
  if (workspace_bytes < pl.bytes) return SEGSDE_ERR_WORKSPACE;
  const int C = d->C0 + d->C1;
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  // v_saved (nullable): the transformed input a forward call of segsde_conv2d_winograd left in its v_keep -- no second input transform
  float* V = v_saved ? const_cast<float*>(v_saved) : reinterpret_cast<float*>(base);
  if (!aligned16(V)) return SEGSDE_ERR_UNSUPPORTED;
  float* dM = reinterpret_cast<float*>(base + pl.off_dm);
  float* part = reinterpret_cast<float*>(base + pl.off_part);
  ConvP p = make_params(&pl.g, V, nullptr, dM, nullptr, part, nullptr);
  if (wgrad_mode(p, dM, d->Cout) != 2) return SEGSDE_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!v_saved)
    if (int e = segsde_wino_input(x0, d->ld0, d->C1 ? x1 : nullptr, d->C1 ? d->ld1 : d->ld0, d->C0, d->B, d->H, d->W, C, d->dil,
                                  d->pad_mode == SEGSDE_PAD_REFLECT, V, stream))
      return e;
  if (int e = segsde_wino_grad(dy, lddy, d->B, d->H, d->W, d->Cout, d->dil, dM, stream)) return e;
  int e;
  if (pl.bn == 64) e = launch_wgrad<128, 64, 2, 2>(p, dM, d->Cout, part, 16 * pl.s, pl.cps, s);
  else e = launch_wgrad<128, 128, 2, 2>(p, dM, d->Cout, part, 16 * pl.s, pl.cps, s);
  if (e) return e;
  return segsde_wino_wgrad_finish(part, pl.s, C, d->Cout, dw_oihw, stream);
}
