// Winograd F(2x2,3x3) with the transforms INSIDE the GEMM kernel (round 4 experiment): the 3x3 / stride 1 / pad 1
// convolutions with 64 or 128 channels (ResNet layer1 / layer2 conv2: models/resnet_encoder.py, torchvision Bottleneck.conv2), where
// the route of winograd.hip -- transforms as passes through memory around sixteen grouped position GEMMs -- loses to the direct
// implicit GEMM because V and M (4 x the activation each) cost more than the multiply-adds they save.
//
// One workgroup = 32 tiles (4 x 8: a 8 x 16 pixel block of the output) x 64 filters x ALL sixteen transform positions:
//  * the raw 10 x 18 pixel patch of the block, 64 channels at a time, is staged ONCE in LDS, channel-major (plane pitch 181);
//  * wave w owns the four positions of transform row w (p = 4 w + j): per pair of channels a lane (tile = lane & 31, channel =
//    lane >> 5) reads the 2 x 4 raw pixels row w's combination needs, forms its four A operands with 8 additions -- V never
//    exists in memory -- and takes its eight B operands (4 positions x 2 filter blocks of 32) straight from the transformed
//    weights U[p][c][n] in L2 (n fastest: 128-byte runs per half-wave); 8 x v_mfma_f32_32x32x2_f32 per step, 128 accumulators;
//  * epilogue: the sixteen 32 x 32 position blocks of a filter block meet in LDS (the patch is dead by then), every thread
//    finishes four tiles of one filter: Y = A^T M A, bias / activation, the four output pixels, and -- STATS -- the
//    double-precision column sums of what it stored (the following BatchNorm's batch statistics, same partial-row format as
//    the implicit-GEMM epilogue's).
// 16 instead of 36 multiply-adds per output, and no transform traffic: x is read once (+ halo), y written once.
#include <stdlib.h>
#include <type_traits>
#include "segsde_common.h"
#include "winograd.h"
#ifndef WINO_UP_SKIP
#define WINO_UP_SKIP 1        // 0: build without the zero-position skipping of upsampled-source fills (A/B)
#endif

namespace {
#define ST(s) static_cast<hipStream_t>(s)
constexpr int FT_H = 4, FT_W = 8;                     // tiles of a block: 32 = the rows of the 32x32x2 MFMA
constexpr int FP_H = 2 * FT_H + 2, FP_W = 2 * FT_W + 2;   // its 10 x 18 pixel patch
// LDS image of the patch (round 5: bank-conflict-free on both sides; PMC of the round-4 layout -- plane pitch 181, pixel (pr, pc)
// at pr * 18 + pc -- showed 70 % of the kernel's LDS cycles as bank conflicts: ds_read_b32 serves the 32 lanes of a half-wave, bank =
// dword address mod 32, and the 32 tiles of an A operand sat at 36 th + 2 tw: even banks only, four rows on top of each other).
//  * a channel plane is 10 rows of pitch FROW = 20 with the EVEN pixel columns of a row first (positions 0..8), the odd ones behind
//    (9..17): the 32 tiles of one operand read position 40 th + tw + const -> bank 8 th + tw: all 32 distinct;
//  * plane c starts at c * FPS + 2 * (c >> 2), FPS = 256: the staging store of one float4 component covers planes 4 cq + j of 16
//    channel quads and two neighbouring patch positions per half-wave -> banks 2 cq + position: all 32 distinct (pitch 181: 2-way).
constexpr int FROW = 20, FODD = FP_W / 2;
constexpr int FPS = 256;
__device__ __forceinline__ constexpr int fplane(int c) { return c * FPS + 2 * (c >> 2); }
constexpr int FCH = 64;                               // channels per LDS fill
// the epilogue's half-transformed blocks Z[4 transform rows][2 output columns][64 filters][FZP]: the 32 tiles of a filter are
// contiguous (pitch 36: lanes with consecutive filters cover the 32 banks with their 16-byte accesses), so that a thread moves the
// four tiles its accumulator registers hold side by side, and the eight tiles it finishes, with ds_write_b128 / ds_read_b128 --
// 16 + 16 LDS instructions per thread instead of 64 + 64 (round 6: the epilogue was 5-14 % of the kernel, probe_r06_wino_phases.log)
constexpr int FZP = 36;
constexpr int F_FLOATS = 8 * 64 * FZP > FCH * FPS ? 8 * 64 * FZP : FCH * FPS;
#ifdef WINO_DBG_OCC1           // phase-cost probe: one workgroup per CU (LDS request above half of the 160 KiB)
constexpr size_t F_LDS = 96 * 1024;
#else
constexpr size_t F_LDS = (size_t)F_FLOATS * sizeof(float) + 2 * 4 * 64 * sizeof(double);
#endif

// the sources of the virtual input [up2x?(x0) | x1]: channels [0, C0) from x0 (stored at half resolution when up0: the decoder's
// nearest upsampling, models/depth_decoder.py:91, is index arithmetic of the patch loader), [C0, C) from x1 at full resolution;
// one source: C0 = C, x1 unused.  C0 is a multiple of the 64-channel fill, so a fill reads one source.
struct WinoSrc { const float* x0; const float* x1; int ld0, ld1, C0, up0; };
// epilogue extras of the data-gradient: agy (nullable) = the saved OUTPUT of the activation whose input this gradient is for,
// the result is multiplied by its derivative (monodepth_layers.py:108-125: ConvBlock = conv -> ELU, the gradient of the next
// layer's input is the gradient of this ELU's output)
struct WinoAg { const float* agy; int agld, agkind; };

// UBLK: the transformed weights in the BLOCKED layout U[row w][c][block of 64 filters][32 lanes][4 positions of the row][2 filter
// halves] -- the eight B operands of a lane and step are 32 contiguous bytes (two 16-byte requests, 1 KiB per wave-instruction)
// instead of eight 4-byte requests in eight position planes
// UPSKIP: the launch has a nearest-upsampled source (see the K loop).  A launch-level template parameter: as a run-time flag the
// wave-uniform branches around the two MFMAs cost the single-source launches 2-4 % (probe_r06_upskip_single_source.log).
template <bool STATS, bool UBLK, bool UPSKIP>
__global__ __launch_bounds__(256, 2) void wino_fused_kernel(WinoSrc src, int B, int H, int W, int C, int reflect,
                                                            const float* U, int ldu, int Co, const float* bias, int act, float* y, int ldy,
                                                            double* part, int accumulate, WinoAg ag) {
  SEGSDE_SMEM;
  float* lds = reinterpret_cast<float*>(segsde_smem);
  double* sh = reinterpret_cast<double*>(lds + F_FLOATS);
  const int H2 = H >> 1, W2 = W >> 1;
  const int nbw = (W2 + FT_W - 1) / FT_W, nbh = (H2 + FT_H - 1) / FT_H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 31, kk = lane >> 5, th = t >> 3, tw = t & 7;
  // (round 6, negative twice: PERSISTENT workgroups walking over their share of the items.  First build: 55-112 spilled registers
  // (values of the previous item looked live across the item loop), 1.25x slower; with per-item re-initialisation no spill, but
  // still 3-6 % slower than one workgroup per item, persistent grid or not: the epilogue's cost (5-14 % of this kernel, half of it
  // the store stream, half the vector instructions that finish the tiles) is not a slot-retirement effect.  16-byte stores
  // (thread = four filters): 2-4 % slower.  profiles/experiments_r06.md)
  const int bidx = blockIdx.x;
  int blk = segsde_xcd_remap(bidx, gridDim.x);
  const int bw = blk % nbw; blk /= nbw;
  const int bh = blk % nbh; const int b = blk / nbh;
  const int co0 = blockIdx.y * 64;
  const int h_top = 2 * bh * FT_H - 1, w_left = 2 * bw * FT_W - 1;   // image position of patch pixel (0, 0)
  // rows of the patch that transform row `wave` combines: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int a2 = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sgn = wave == 1 ? 1.f : -1.f;
  const float* pl = lds + kk * FPS + (2 * th) * FROW + tw;     // (the plane skew of channels 2 s + kk does not depend on kk)
  const int o1 = a1 * FROW, o2 = a2 * FROW;

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

  // Per-thread byte offsets of the fill's 12 sixteen-byte requests inside one image of the source being read: they depend on the
  // thread and the source only, not on the fill.  The first fill from a source computes them between its requests (division by
  // 18, column interleave, mirroring, range tests: ~200 of a fill's ~700 vector instructions), the later fills re-issue raw
  // buffer loads with the image base + the fill's channel offset in the scalar resource: no vector instructions in their load
  // issue (+2.5-3 % from 256 channels on, +1-2 % below: profiles/probe_r05_hoist_*.log).  Computing all offsets ahead of the
  // first request instead lost 2-4 % on the single-fill 64-channel layers (the first load left ~200 instructions later).
  // Padding and tiles past the image: SEGSDE_OOB, the load returns zeros.
  constexpr int NL = (FP_H * FP_W * (FCH / 4) + 255) / 256;
  unsigned poff[NL];
  int cur_src = -1, lds_ = 0;
  long img = 0;
  int sh_ = 0, Ws = 0;
  auto patch_offset = [&](int i) -> unsigned {
    const int e = tid + 256 * i;
    const int pix = e >> 4, cq = e & 15;
    const int pr = pix / FP_W, pos = pix - pr * FP_W;                 // pos: position inside the LDS row (even columns first)
    const int pc = pos < FODD ? 2 * pos : 2 * (pos - FODD) + 1;
    int hh = h_top + pr, ww = w_left + pc;
    if (reflect) {       // ReflectionPad2d(1) (monodepth_layers.py:127-142); pixels of tiles past the image: any valid address
      hh = hh < 0 ? -hh : (hh >= H ? 2 * H - 2 - hh : hh); ww = ww < 0 ? -ww : (ww >= W ? 2 * W - 2 - ww : ww);
      hh = hh < 0 ? 0 : hh; ww = ww < 0 ? 0 : ww;
    }
    const bool ok = e < FP_H * FP_W * (FCH / 4) && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
    return ok ? (unsigned)(((hh >> sh_) * Ws + (ww >> sh_)) * lds_ + 4 * cq) * 4u : SEGSDE_OOB;
  };

  for (int c0 = 0; c0 < C; c0 += FCH) {
#ifdef WINO_DBG_LOOP_ONLY      // phase-cost probe: the fill (loads, LDS writes, barriers) only once per workgroup
    if (c0 == 0)
#endif
    {
    __syncthreads();                                   // the previous fill's reads are done
    {
      // all requests of the fill first (12 independent 16-byte loads per thread), then the transposing LDS writes
      float4 v[NL];
      const int sidx = c0 < src.C0 ? 0 : 1;              // wave-uniform: which source this fill reads
      const bool first = sidx != cur_src;                // the first fill from this source: offsets computed between its requests
      if (first) {
        cur_src = sidx;
        sh_ = (sidx == 0 && src.up0) ? 1 : 0;
        Ws = W >> sh_;
        lds_ = sidx == 0 ? src.ld0 : src.ld1;
        img = (long)(H >> sh_) * Ws * lds_;
      }
      const float* xs = sidx == 0 ? src.x0 : src.x1;
      const segsde_rsrc xr = segsde_make_rsrc(xs + (size_t)b * img + (sidx == 0 ? c0 : c0 - src.C0));
#ifdef WINO_DBG_SKIP_LOAD      // phase-cost probe (tools/build_variant.sh): no global patch loads, the rest of the fill unchanged
      (void)xr; (void)first;
#pragma unroll
      for (int i = 0; i < NL; ++i) { poff[i] = patch_offset(i); v[i] = make_float4((float)poff[i], 1.f, 2.f, 3.f); }
#else
      if (first) {
#pragma unroll
        for (int i = 0; i < NL; ++i) { poff[i] = patch_offset(i); v[i] = segsde_buffer_load4(xr, poff[i], 0u); }
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = segsde_buffer_load4(xr, poff[i], 0u);
      }
#endif
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int e = tid + 256 * i;
        if (e < FP_H * FP_W * (FCH / 4)) {
          const int pix = e >> 4, pr = pix / FP_W;
          float* d = lds + fplane(4 * (e & 15)) + pr * FROW + (pix - pr * FP_W);
          d[0] = v[i].x; d[FPS] = v[i].y; d[2 * FPS] = v[i].z; d[3 * FPS] = v[i].w;
        }
      }
    }
    __syncthreads();
    }
    // U[p][c][n]: this lane's column n = co0 + nb * 32 + t of channel c0 + 2 s + kk, positions 4 wave + j.  The address is
    // split into a wave-uniform part (scalar registers, scalar adds) and the lane's constant 32-bit offset: the eight requests
    // of a step cost no vector instructions (with per-lane 64-bit pointers they were two thirds of the loop's VALU work, and
    // VALU issue slots are MFMA issue slots)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const segsde_rsrc ur = segsde_make_rsrc(U);
    // (ldu: row pitch of U -- Co, or the width of the pack this launch takes a column slice of)
    const unsigned nb64 = (unsigned)(ldu >> 6);
    const unsigned lane_off = UBLK ? (unsigned)(kk * nb64 * 256 + t * 8) * 4u : (unsigned)(kk * ldu + t) * 4u;                          // bytes
    const unsigned ubase = UBLK ? (unsigned)((((long)wave_u * C + c0) * nb64 + (co0 >> 6)) * 256 * 4)
                                : (unsigned)((((long)(4 * wave_u) * C + c0) * ldu + co0) * 4);   // bytes; 16 C ldu floats < 2^30
    const unsigned upos = (unsigned)((long)C * ldu * 4), ustep = UBLK ? 2u * nb64 * 256u * 4u : (unsigned)(2 * ldu * 4);
    // B operands come straight from L2 (a few hundred ns): requested PD steps (PD x 512 MFMA cycles) ahead; the raw pixels
    // (LDS) one step ahead
    constexpr int PD = 4, NS = FCH / 2;
    float bv[PD][8], rv[2][8];
    auto fetch_b = [&](int s, int q) {
      const unsigned so = ubase + (unsigned)s * ustep;
      if constexpr (UBLK) {
        const float4 lo = segsde_buffer_load4(ur, lane_off, so), hi = segsde_buffer_load4(ur, lane_off, so + 16u);
        bv[q][0] = lo.x; bv[q][1] = lo.y; bv[q][2] = lo.z; bv[q][3] = lo.w;
        bv[q][4] = hi.x; bv[q][5] = hi.y; bv[q][6] = hi.z; bv[q][7] = hi.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bv[q][2 * j] = segsde_buffer_load1(ur, lane_off, so + j * upos);
          bv[q][2 * j + 1] = segsde_buffer_load1(ur, lane_off, so + j * upos + 128u);
        }
      }
    };
    // (SEGSDE_LDS_READ_IMM: single ds_read_b32 with the plane / column offset in the instruction's 16-bit immediate off the two
    // row bases of the wave.  Left to itself the compiler pairs neighbouring columns into ds_read2_b32, whose 8-bit offsets do not
    // reach past the first plane: 124 v_add_u32 per fill to rebase them -- a third of the loop's vector instructions)
#ifdef WINO_DBG_SKIP_K         // phase-cost probe: one step of the K loop instead of 32
    constexpr int NS_RUN = 1;
#else
    constexpr int NS_RUN = NS;
#endif
    // Software-pipelined step (round 6): the A operands of step s + 1 are formed during step s, and every request of a step sits
    // in the shadow of one of its MFMAs -- slots 0..3 issue the eight raw-pixel reads of the next step (two each), 4 / 5 the two
    // weight requests of step s + PD - 1, 6 / 7 the eight additions that turn the pixels into the next step's operands.  One wave
    // alone on its SIMD (the sibling workgroup in its fill or epilogue) then keeps the matrix pipe fed: with the ten requests in
    // one bunch between two MFMAs and the additions in front of the MFMAs that use them a lone wave reached 0.75 of the pipe.
    float An[2][4];
    auto form_a = [&](int q, int h) {          // h = 0: A0, A1 of buffer q from rv[0]; h = 1: A2, A3
      if (h == 0) {
        const float e0 = __builtin_fmaf(rv[0][4], sgn, rv[0][0]), e1 = __builtin_fmaf(rv[0][5], sgn, rv[0][1]);
        const float e2 = __builtin_fmaf(rv[0][6], sgn, rv[0][2]);
        An[q][0] = e0 - e2; An[q][1] = e1 + e2; An[q][2] = e2 - e1; rv[1][0] = e1;
      } else {
        const float e3 = __builtin_fmaf(rv[0][7], sgn, rv[0][3]);
        An[q][3] = rv[1][0] - e3;
      }
    };
    auto fetch_a2 = [&](int s, int bb) {       // raw pixels of step s, patch column 2 tw + bb: both rows
      const float* src = pl + fplane(2 * s);
      const int cp = (bb & 1) ? FODD + (bb >> 1) : (bb >> 1);
      rv[0][bb] = SEGSDE_LDS_READ_IMM(src + o1, cp); rv[0][4 + bb] = SEGSDE_LDS_READ_IMM(src + o2, cp);
    };
    // A fill from the NEAREST-UPSAMPLED source (the decoders' [upsample(x0) | x1] layers): rows 2 t and 2 t + 1 of an output
    // tile's 4 x 4 patch are the same low-resolution row, columns likewise, so B^T d B is identically zero in transform row 2 and
    // in transform column 2 (d2 - d1 = 0) -- seven of the sixteen positions multiply zeros.  SKIP2: wave 2 (transform row 2)
    // sits such a fill out, the others leave out position 2 of their row: 6 MFMAs per step instead of 8 on three of the four
    // SIMDs.  (Exactly the products that are 0 * w; mirrored / zero padding keeps rows 2 t, 2 t + 1 on one source row.)
    // (ONE loop body with a wave-uniform branch around the two MFMAs: two copies of the unrolled loop behind a branch made the
    // register allocator spill 170 registers at the join of the 128 accumulators)
    const bool skip2 = UPSKIP && c0 < src.C0;            // (UPSKIP launches: src.up0 is set)
    if (!(skip2 && wave_u == 2)) {
#pragma unroll
      for (int s = 0; s < PD - 1; ++s) fetch_b(s, s);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) fetch_a2(0, bb);
      form_a(0, 0); form_a(0, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NS_RUN; ++s) {
        const int q = s & 1, qb = s % PD;
        const bool nx = s + 1 < NS;
#define WINO_MF_(j, nb, bi) acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(An[q][j], bv[qb][bi], acc[j][nb], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
        WINO_MF_(0, 0, 0); if (nx) fetch_a2(s + 1, 0); __builtin_amdgcn_sched_barrier(0);
        WINO_MF_(0, 1, 1); if (nx) fetch_a2(s + 1, 1); __builtin_amdgcn_sched_barrier(0);
        WINO_MF_(1, 0, 2); if (nx) fetch_a2(s + 1, 2); __builtin_amdgcn_sched_barrier(0);
        WINO_MF_(1, 1, 3); if (nx) fetch_a2(s + 1, 3); __builtin_amdgcn_sched_barrier(0);
        if (!skip2) {
          WINO_MF_(2, 0, 4); __builtin_amdgcn_sched_barrier(0);
          WINO_MF_(2, 1, 5); __builtin_amdgcn_sched_barrier(0);
        }
        // (the weight requests of step s + PD - 1 overwrite bv[(s - 1) % PD]: last read by the previous step)
        if (s + PD - 1 < NS) fetch_b(s + PD - 1, (s + PD - 1) % PD);
        __builtin_amdgcn_sched_barrier(0);
        WINO_MF_(3, 0, 6); if (nx) form_a(q ^ 1, 0); __builtin_amdgcn_sched_barrier(0);
        WINO_MF_(3, 1, 7); if (nx) form_a(q ^ 1, 1); __builtin_amdgcn_sched_barrier(0);
#undef WINO_MF_
      }
    }
  }

#ifdef WINO_DBG_SKIP_EPI       // phase-cost probe: no Z exchange, no output
  {
    float sdbg = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sdbg += acc[j][nb][r];
    if (sdbg == 1.2345678f) y[0] = sdbg;
    return;
  }
#endif
  // ---- epilogue.  Y = A^T M A with A^T = [1 1 1 0; 0 1 -1 -1]: wave w holds row w of the 4 x 4 position grid, so the column
  // half (.) A runs on its accumulators (four values -> two); the rows meet in LDS as Z[w][column 0 / 1][tile][filter] (the
  // patch is dead by then), one round for all 64 filters, and every thread finishes eight tiles of one filter
  __syncthreads();                                     // the last fill's patch reads are done
  {
    // accumulator register r of a lane is tile (r & 3) + 8 (r >> 2) + 4 kk of filter nb * 32 + t: four consecutive tiles per r >> 2
    float* zw = lds + ((2 * wave) * 64 + t) * FZP + 4 * kk;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float4 c0v, c1v;
        float* a0 = reinterpret_cast<float*>(&c0v);
        float* a1 = reinterpret_cast<float*>(&c1v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * rg + i;
          const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r];
          a0[i] = (m0 + m1) + m2;
          a1[i] = (m1 - m2) - m3;
        }
        *reinterpret_cast<float4*>(zw + (nb * 32) * FZP + 8 * rg) = c0v;
        *reinterpret_cast<float4*>(zw + (64 + nb * 32) * FZP + 8 * rg) = c1v;
      }
  }
  // thread: filter n, tiles 8 tg .. 8 tg + 7 (tile row tg of the block).  tg is the wave: everything in a pixel's address except the
  // filter is wave-uniform, so the epilogue's loads and stores are raw buffer operations on the image with the lane's filter offset in
  // the vector register and the pixel's offset in a scalar one -- no 64-bit vector address arithmetic (it was about a quarter of the
  // epilogue's vector instructions), and the range tests are scalar branches
  const int n = tid & 63, tg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool post = bias != nullptr || act != SEGSDE_ACT_NONE;
  const int co = co0 + n;
  const int ti = bh * FT_H + tg;
  const unsigned lane_b = (unsigned)co * 4u;
#ifdef WINO_EPI_POINTERS       // A/B: the per-thread pointer epilogue of round 5
  constexpr bool kBuf = false;
#else
  constexpr bool kBuf = true;
#endif
  const segsde_rsrc yr = segsde_make_rsrc(y + (long)b * H * W * ldy);
  const segsde_rsrc agr = segsde_make_rsrc(ag.agy ? ag.agy + (long)b * H * W * ag.agld : y);
  // the data-gradient epilogues read memory (the saved activation output for its derivative, the gradient already in dx for the
  // accumulation): all 8 x 4 requests of a thread are issued HERE, before the barrier of the Z exchange -- inside the tile loop
  // below every tile waited a full memory latency for its own four (round 5, first version: +55 % on a 64-channel layer)
  float agv[FT_W][4], oldv[FT_W][4];
  if (ag.agy) {
#pragma unroll
    for (int q = 0; q < FT_W; ++q) {
      const int tj = bw * FT_W + q;
      const bool ok = ti < H2 && tj < W2;
      if (kBuf) {
        const unsigned s0 = (unsigned)((2 * ti) * W + 2 * tj) * (unsigned)ag.agld * 4u, sr = (unsigned)W * (unsigned)ag.agld * 4u;
        if (ok) {
          agv[q][0] = segsde_buffer_load1(agr, lane_b, s0); agv[q][1] = segsde_buffer_load1(agr, lane_b, s0 + (unsigned)ag.agld * 4u);
          agv[q][2] = segsde_buffer_load1(agr, lane_b, s0 + sr); agv[q][3] = segsde_buffer_load1(agr, lane_b, s0 + sr + (unsigned)ag.agld * 4u);
        } else { agv[q][0] = agv[q][1] = agv[q][2] = agv[q][3] = 0.f; }
      } else {
        const float* ap = ag.agy + ((long)(b * H + 2 * ti) * W + 2 * tj) * ag.agld + co;
        agv[q][0] = ok ? ap[0] : 0.f; agv[q][1] = ok ? ap[ag.agld] : 0.f;
        agv[q][2] = ok ? ap[(long)W * ag.agld] : 0.f; agv[q][3] = ok ? ap[(long)W * ag.agld + ag.agld] : 0.f;
      }
    }
  }
  if (accumulate & 1) {
#pragma unroll
    for (int q = 0; q < FT_W; ++q) {
      const int tj = bw * FT_W + q;
      const bool ok = ti < H2 && tj < W2;
      if (kBuf) {
        const unsigned s0 = (unsigned)((2 * ti) * W + 2 * tj) * (unsigned)ldy * 4u, sr = (unsigned)W * (unsigned)ldy * 4u;
        if (ok) {
          oldv[q][0] = segsde_buffer_load1(yr, lane_b, s0); oldv[q][1] = segsde_buffer_load1(yr, lane_b, s0 + (unsigned)ldy * 4u);
          oldv[q][2] = segsde_buffer_load1(yr, lane_b, s0 + sr); oldv[q][3] = segsde_buffer_load1(yr, lane_b, s0 + sr + (unsigned)ldy * 4u);
        } else { oldv[q][0] = oldv[q][1] = oldv[q][2] = oldv[q][3] = 0.f; }
      } else {
        const float* yp = y + ((long)(b * H + 2 * ti) * W + 2 * tj) * ldy + co;
        oldv[q][0] = ok ? yp[0] : 0.f; oldv[q][1] = ok ? yp[ldy] : 0.f;
        oldv[q][2] = ok ? yp[(long)W * ldy] : 0.f; oldv[q][3] = ok ? yp[(long)W * ldy + ldy] : 0.f;
      }
    }
  }
  __syncthreads();
  const float bsv = bias ? bias[co] : 0.f;
  double ssum = 0.0, ssq = 0.0;
  float zall[4][2][FT_W];                              // [transform row][output column][tile of this thread's tile row]
  {
    const float* zp = lds + n * FZP + tg * FT_W;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < FT_W / 4; ++h) {
          const float4 v = *reinterpret_cast<const float4*>(zp + ((2 * i + c) * 64) * FZP + 4 * h);
          zall[i][c][4 * h] = v.x; zall[i][c][4 * h + 1] = v.y; zall[i][c][4 * h + 2] = v.z; zall[i][c][4 * h + 3] = v.w;
        }
  }
#pragma unroll
  for (int q = 0; q < FT_W; ++q) {
    float z0[4], z1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { z0[i] = zall[i][0][q]; z1[i] = zall[i][1][q]; }
    float o[4];                                        // (0,0) (0,1) (1,0) (1,1)
    o[0] = (z0[0] + z0[1]) + z0[2]; o[1] = (z1[0] + z1[1]) + z1[2];
    o[2] = (z0[1] - z0[2]) - z0[3]; o[3] = (z1[1] - z1[2]) - z1[3];
    if (post) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = segsde_act(o[k] + bsv, act);
    }
    const int tj = bw * FT_W + q;
    if (ti < H2 && tj < W2) {
      float* yp = y + ((long)(b * H + 2 * ti) * W + 2 * tj) * ldy + co;
      if (ag.agy) {         // data-gradient w.r.t. the pre-activation of the producing ConvBlock
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] *= segsde_act_grad_from_out(agv[q][k], ag.agkind);
      }
      if (accumulate & 1) { // a data-gradient added onto the gradient another consumer of the tensor left there
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += oldv[q][k];
      }
#ifdef WINO_DBG_FEW_STORES     // phase-cost probe: one tile of eight is stored
      if (q == 0)
#endif
      if (kBuf) {
        const unsigned s0 = (unsigned)((2 * ti) * W + 2 * tj) * (unsigned)ldy * 4u, sr = (unsigned)W * (unsigned)ldy * 4u, sc = (unsigned)ldy * 4u;
        if (accumulate & 2) {   // streaming stores (the launcher sets the bit for outputs that no cache level can hold)
          segsde_buffer_store1_nt(yr, lane_b, s0, o[0]); segsde_buffer_store1_nt(yr, lane_b, s0 + sc, o[1]);
          segsde_buffer_store1_nt(yr, lane_b, s0 + sr, o[2]); segsde_buffer_store1_nt(yr, lane_b, s0 + sr + sc, o[3]);
        } else {
          segsde_buffer_store1(yr, lane_b, s0, o[0]); segsde_buffer_store1(yr, lane_b, s0 + sc, o[1]);
          segsde_buffer_store1(yr, lane_b, s0 + sr, o[2]); segsde_buffer_store1(yr, lane_b, s0 + sr + sc, o[3]);
        }
      } else if (accumulate & 2) {
        __builtin_nontemporal_store(o[0], yp); __builtin_nontemporal_store(o[1], yp + ldy);
        __builtin_nontemporal_store(o[2], yp + (long)W * ldy); __builtin_nontemporal_store(o[3], yp + (long)W * ldy + ldy);
      } else { yp[0] = o[0]; yp[ldy] = o[1]; yp[(long)W * ldy] = o[2]; yp[(long)W * ldy + ldy] = o[3]; }
      if (STATS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double v = (double)o[k]; ssum += v; ssq += v * v; }
      }
    }
  }
  if (STATS) {
    sh[tg * 64 + n] = ssum; sh[256 + tg * 64 + n] = ssq;
    __syncthreads();
    if (tid < 64) {
      double a = 0.0, c2 = 0.0;
      for (int l = 0; l < 4; ++l) { a += sh[l * 64 + tid]; c2 += sh[256 + l * 64 + tid]; }
      part[((long)bidx * 2 + 0) * Co + co0 + tid] = a;
      part[((long)bidx * 2 + 1) * Co + co0 + tid] = c2;
    }
  }
}

// OIHW 3x3 weight -> U (logical [16][K][N]; stored blocked -- wino_ublk_index -- or as sixteen [K][N] planes): forward (flip = 0) K = I, N = O; data-gradient (flip = 1: the convolution of dY with
// the spatially flipped kernel) K = O, N = I.  thread (k, n), n fastest: coalesced writes, the nine reads of a thread are contiguous.
// blocked layout (wino_fused_kernel<., true>): element (p = 4 r + q, k, n) at ((((r K + k) (N / 64) + n / 64) 32 + n % 32) 4 + q) 2 + (n / 32) % 2
__device__ __forceinline__ long wino_ublk_index(int r, int q, int k, int n, int K, int N) {
  return (((((long)r * K + k) * (N >> 6) + (n >> 6)) * 32 + (n & 31)) * 4 + q) * 2 + ((n >> 5) & 1);
}
__global__ __launch_bounds__(256) void wino_weight_kn_kernel(const float* w, int O, int I, int flip, float* U, int blocked) {
  const int K = flip ? O : I, N = flip ? I : O;
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)K * N) return;
  const int nn = (int)(e % N), kq = (int)(e / N);
  const int o = flip ? kq : nn, i = flip ? nn : kq;
  const float* g = w + ((long)o * I + i) * 9;
  float k[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) k[r][c] = flip ? g[(2 - r) * 3 + (2 - c)] : g[r * 3 + c];
  float tq[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {          // G g (the expressions of wino_weight_kernel)
    tq[0][c] = k[0][c];
    tq[1][c] = 0.5f * ((k[0][c] + k[1][c]) + k[2][c]);
    tq[2][c] = 0.5f * ((k[0][c] - k[1][c]) + k[2][c]);
    tq[3][c] = k[2][c];
  }
  const long plane = (long)K * N;
#pragma unroll
  for (int r = 0; r < 4; ++r) {          // (.) G^T
    const float v0 = tq[r][0], v1 = 0.5f * ((tq[r][0] + tq[r][1]) + tq[r][2]), v2 = 0.5f * ((tq[r][0] - tq[r][1]) + tq[r][2]), v3 = tq[r][2];
    if (blocked) {
      float* o = U + wino_ublk_index(r, 0, kq, nn, K, N);      // the four positions of a row: stride 2 floats
      o[0] = v0; o[2] = v1; o[4] = v2; o[6] = v3;
    } else {
      float* out = U + e;
      out[(4 * r + 0) * plane] = v0; out[(4 * r + 1) * plane] = v1; out[(4 * r + 2) * plane] = v2; out[(4 * r + 3) * plane] = v3;
    }
  }
}

// ---- Mirrored-padding part of the data-gradient of a ReflectionPad2d(1) + 3x3 convolution (models/monodepth_layers.py:127-142), for
// the zero-padded one-kernel launch above: the gradient that entered the padding cells flows back to rows 1 / H-2 and columns
// 1 / W-2.  Per line a 3-tap 1-D convolution of the gradient's border row / column,
//     dx[b, 1, w, c]   += sum_kw sum_o dy[b, 0,   w - kw + 1, o] * w[o][c][0][kw]        (row H-2: dy row H-1, kh = 2)
//     dx[b, h, 1, c]   += sum_kh sum_o dy[b, h - kh + 1, 0,   o] * w[o][c][kh][0]        (column W-2: dy column W-1, kw = 2)
// plus the doubly mirrored corner cells: dx[b, 1, 1, c] += sum_o dy[b, 0, 0, o] * w[o][c][0][0] (and the three other corners).
// Round 4 ran this as four launches of the implicit-GEMM kernel + a corner kernel (~50 us each: fixed per-launch costs of a
// kernel built for whole images); here TWO launches of a 32-pixel x Cin MFMA kernel: mode 0 = both row lines, mode 1 = both
// column lines with the corner terms as two more taps of the workgroup that owns row 1 / H-2 (their source pixels are in its
// staged line already).  Pixels (1,1) ... receive a row-line and a column-line term: from different launches, so no two
// workgroups of a launch ever add to the same address (no atomics, deterministic).
struct BorderP {
  const float* dy; const float* wp /* forward pack [Cout][3][3][Cin] */; float* dx; const float* agy;
  int lddy, lddx, agld, agkind, B, H, W, Cin, Cout, mode, nseg, ldw;
};

__global__ __launch_bounds__(256) void reflect_borders_kernel(BorderP p) {
  SEGSDE_SMEM;
  float* sh = reinterpret_cast<float*>(segsde_smem);                 // [34][Cout + 1]: the source line, positions p0 - 1 .. p0 + 32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, kk = lane >> 5;
  const int L = p.mode == 0 ? p.W : p.H;
  int r = blockIdx.x;
  const int seg = r % p.nseg; r /= p.nseg;
  const int far = r & 1, b = r >> 1;
  const int p0 = seg * 32, pitch = p.Cout + 1, cq4 = p.Cout >> 2;
  const int fixed = p.mode == 0 ? (far ? p.H - 1 : 0) : (far ? p.W - 1 : 0);       // the gradient's border row / column
  for (int idx = tid; idx < 34 * cq4; idx += 256) {
    const int px = idx / cq4, q = idx - px * cq4, pos = p0 - 1 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos >= 0 && pos < L) {
      const long pix = p.mode == 0 ? ((long)(b * p.H + fixed) * p.W + pos) : ((long)(b * p.H + pos) * p.W + fixed);
      v = *reinterpret_cast<const float4*>(p.dy + pix * p.lddy + 4 * q);
    }
    float* d = sh + px * pitch + 4 * q;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  // taps: 0..2 the line's own three (staged index m + 2 - t); 3 / 4 (mode 1 only) the corner terms of the pixels at line position
  // 1 / H - 2, whose source is the staged line at position 0 / H - 1
  const int ktap = far ? 2 : 0;
  const bool top = p.mode == 1 && p0 <= 1 && 1 < p0 + 32, bot = p.mode == 1 && p0 <= p.H - 2 && p.H - 2 < p0 + 32;
  const int steps = p.Cout >> 1;
  for (int nb = wave; nb < (p.Cin >> 5); nb += 4) {
    const int cin = nb * 32 + m;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int t = 0; t < 5; ++t) {
      if (t == 3 && !top) continue;
      if (t == 4 && !bot) continue;
      int kh, kw, src; bool mine = true;
      if (t < 3) {
        kh = p.mode == 0 ? ktap : t; kw = p.mode == 0 ? t : ktap; src = m + 2 - t;
      } else {
        kh = t == 3 ? 0 : 2; kw = ktap;
        const int at = t == 3 ? 1 : p.H - 2;                 // the line position that receives this corner term
        mine = p0 + m == at;
        src = (t == 3 ? 0 : p.H - 1) - (p0 - 1);
      }
      const float* ap = sh + src * pitch + kk;
      const float* bp = p.wp + ((long)kk * 9 + kh * 3 + kw) * p.ldw + cin;      // (ldw: the forward pack's channel count; a source's slice starts at wp + C0)
      const long bstep = 18L * p.ldw;
      // weights straight from L2 (128-byte runs per half-wave): eight steps' requests in flight ahead of the eight being multiplied
      float bv[2][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) bv[0][u] = u < steps ? bp[u * bstep] : 0.f;
      for (int s0 = 0; s0 < steps; s0 += 16) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int sb = s0 + 8 * half;
          if (sb < steps) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[half ^ 1][u] = sb + 8 + u < steps ? bp[(long)(sb + 8 + u) * bstep] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (sb + u < steps) {
                const float a = mine ? ap[2 * (sb + u)] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[half][u], acc, 0, 0, 0);
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * kk, pos = p0 + row;
      if (pos < L) {
        const long pix = p.mode == 0 ? ((long)(b * p.H + (far ? p.H - 2 : 1)) * p.W + pos) : ((long)(b * p.H + pos) * p.W + (far ? p.W - 2 : 1));
        float v = acc[q];
        if (p.agy) v *= segsde_act_grad_from_out(p.agy[pix * p.agld + cin], p.agkind);
        p.dx[pix * p.lddx + cin] += v;
      }
    }
  }
}

}  // namespace
// layout of the one-kernel route's transformed weights: 1 = blocked (default), 0 = sixteen [K][N] planes (SEGSDE_WINO_FUSED_UBLK=0);
// one answer per process -- the pack kernels (here and winograd.hip's multi-pack) and the convolution kernel must agree
int segsde_wino_ublk() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SEGSDE_WINO_FUSED_UBLK"); v = e ? (atoi(e) != 0) : 1; }
  return v;
}
namespace {
inline long fused_blocks(int B, int H, int W) {
  return (long)B * (((H >> 1) + FT_H - 1) / FT_H) * (((W >> 1) + FT_W - 1) / FT_W);
}
}  // namespace

extern "C" int segsde_winograd_fused_ok(int B, int H, int W, int C, int Cout) {
  // (one image of either side below 2^31 bytes: the patch loader's per-thread offsets are 32-bit byte offsets inside an image)
  return B > 0 && H >= 4 && W >= 4 && H % 2 == 0 && W % 2 == 0 && C >= FCH && C % FCH == 0 && Cout >= 64 && Cout % 64 == 0 &&
         (long)H * W * (C > Cout ? C : Cout) * 4 < (1L << 31) &&
         fused_blocks(B, H, W) < (1L << 31) && (long)B * H * W * (C > Cout ? C : Cout) < (1L << 40);
}

extern "C" long segsde_winograd_fused_stats_rows(int B, int H, int W) { return fused_blocks(B, H, W); }

extern "C" int segsde_winograd_fused_pack(const float* w_oihw, int O, int I, int flip, float* U, void* stream) {
  if (!w_oihw || !U) return SEGSDE_ERR_NULL;
  if (O <= 0 || I <= 0) return SEGSDE_ERR_SHAPE;
  const long total = (long)O * I;
  const int blocked = segsde_wino_ublk() && ((flip ? I : O) % 64 == 0);
  hipLaunchKernelGGL(wino_weight_kn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), w_oihw, O, I, flip, U, blocked);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}

namespace {
int launch_fused(const WinoSrc& src, int B, int H, int W, int C, int reflect, const float* u_kn, int ldu, int Cout, const float* bias, int act,
                 float* y, int ldy, int accumulate, double* stats, const WinoAg& ag, void* stream) {
  const dim3 grid((unsigned)fused_blocks(B, H, W), (unsigned)(Cout / 64));
  const bool ublk = segsde_wino_ublk() != 0;
  // Outputs far beyond the 256 MiB of memory-side cache are written with streaming stores: nothing of them would be found in a cache by
  // their reader anyway, and they no longer push the patches' halo rows (re-read by the neighbouring workgroups) out of L2 / MALL
  // (probe_r06_wino_nt.log: 128 -> 128 @128x256, a 268 MB output, -6 %; small maps unchanged).  SEGSDE_WINO_NT_MB: threshold, 0 = never.
  static long nt_bytes = -1;
  if (nt_bytes < 0) { const char* e = getenv("SEGSDE_WINO_NT_MB"); nt_bytes = (e ? atol(e) : 200L) << 20; }
  accumulate = (accumulate ? 1 : 0) | ((nt_bytes > 0 && (long)B * H * W * Cout * 4 >= nt_bytes) ? 2 : 0);
  auto go = [&](auto k, double* st) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS);
    hipLaunchKernelGGL(k, grid, dim3(256), F_LDS, ST(stream), src, B, H, W, C, reflect, u_kn, ldu, Cout, bias, act, y, ldy, st, accumulate, ag);
  };
  static int upskip_on = -1;       // SEGSDE_WINO_UP_SKIP=0: never (A/B; hipops reads the same variable for its executed-flop count)
  if (upskip_on < 0) { const char* e = getenv("SEGSDE_WINO_UP_SKIP"); upskip_on = e ? (atoi(e) != 0) : 1; }
  const bool upskip = WINO_UP_SKIP && upskip_on && src.up0;
  auto pick = [&](auto st_tag, double* st) {
    constexpr bool ST = decltype(st_tag)::value;
    if (upskip) { if (ublk) go(wino_fused_kernel<ST, true, true>, st); else go(wino_fused_kernel<ST, false, true>, st); }
    else { if (ublk) go(wino_fused_kernel<ST, true, false>, st); else go(wino_fused_kernel<ST, false, false>, st); }
  };
  if (stats) pick(std::true_type{}, stats); else pick(std::false_type{}, (double*)nullptr);
  SEGSDE_CHECK_LAUNCH();
  return 0;
}
// the patch loader addresses one image of a source with 32-bit byte offsets
inline bool pitch_ok(int H, int W, int ld) { return (long)H * W * ld * 4 < (1L << 31); }
}  // namespace

extern "C" int segsde_conv2d_winograd_fused(const float* x, int ldx, int B, int H, int W, int C, int reflect, const float* u_kn,
                                            int Cout, const float* bias, int act, float* y, int ldy, int accumulate, double* stats,
                                            void* stream) {
  if (!x || !u_kn || !y) return SEGSDE_ERR_NULL;
  if (!segsde_winograd_fused_ok(B, H, W, C, Cout) || ldx < C || ldx % 4 != 0 || !pitch_ok(H, W, ldx) || ldy < Cout || !pitch_ok(H, W, ldy) || (accumulate && (stats || bias || act)))
    return SEGSDE_ERR_UNSUPPORTED;
  const WinoSrc src{x, nullptr, ldx, 0, C, 0};
  return launch_fused(src, B, H, W, C, reflect, u_kn, Cout, Cout, bias, act, y, ldy, accumulate, stats, WinoAg{nullptr, 0, 0}, stream);
}

// forward on the virtual input [up2x?(x0) | x1] (the decoder's Conv3x3 on the upsampled previous block and the encoder skip,
// models/depth_decoder.py:88-101): x0 [B, H >> up0, W >> up0, C0], x1 [B, H, W, C1] (NULL: one source)
extern "C" int segsde_conv2d_winograd_fused2(const float* x0, int ld0, int C0, int up0, const float* x1, int ld1, int C1, int B, int H,
                                             int W, int reflect, const float* u_kn, int Cout, const float* bias, int act, float* y,
                                             int ldy, double* stats, void* stream) {
  if (!x0 || !u_kn || !y || (C1 > 0 && !x1)) return SEGSDE_ERR_NULL;
  const int C = C0 + (C1 > 0 ? C1 : 0);
  if (!segsde_winograd_fused_ok(B, H, W, C, Cout) || C0 <= 0 || C0 % FCH || ld0 < C0 || ld0 % 4 != 0 || (C1 > 0 && (ld1 < C1 || ld1 % 4 != 0 || !pitch_ok(H, W, ld1))) ||
      !pitch_ok(H >> (up0 ? 1 : 0), W >> (up0 ? 1 : 0), ld0) || ldy < Cout || !pitch_ok(H, W, ldy))
    return SEGSDE_ERR_UNSUPPORTED;
  const WinoSrc src{x0, x1, ld0, ld1, C0, up0 ? 1 : 0};
  return launch_fused(src, B, H, W, C, reflect, u_kn, Cout, Cout, bias, act, y, ldy, 0, stats, WinoAg{nullptr, 0, 0}, stream);
}

// data-gradient of the zero-padded 3x3 / stride 1 convolution (= the convolution of dy with the flipped, transposed pack) with
// the epilogues of the direct route: accumulate onto dx (a gradient another consumer left there), act_out (nullable) = the
// saved activation output whose derivative multiplies the result (before the accumulation, like segsde_conv2d_dgrad_actgrad).
// ldu: row pitch of ud_kn -- Cin, or, for the gradient of ONE source of a two-source convolution (the decoder's skip input,
// channels [C0, C0 + C1) of the weight), the full pack's width with ud_kn pointing at the slice's first column.
// A reflection-padded convolution's data-gradient is this call followed by segsde_reflect_adjoint_borders.
extern "C" int segsde_conv2d_winograd_fused_dgrad(const float* dy, int lddy, int B, int H, int W, int Cout, const float* ud_kn, int ldu,
                                                  int Cin, float* dx, int lddx, int accumulate, const float* act_out, int act_ld,
                                                  int act_kind, void* stream) {
  if (!dy || !ud_kn || !dx) return SEGSDE_ERR_NULL;
  if (!segsde_winograd_fused_ok(B, H, W, Cout, Cin) || ldu < Cin || (long)16 * Cout * ldu >= (1L << 28) || lddy < Cout || lddy % 4 != 0 || !pitch_ok(H, W, lddy) || lddx < Cin || !pitch_ok(H, W, lddx) ||
      (act_out && (act_kind < SEGSDE_ACT_RELU || act_kind > SEGSDE_ACT_SIGMOID || act_ld < Cin || !pitch_ok(H, W, act_ld))))
    return SEGSDE_ERR_UNSUPPORTED;
  const WinoSrc src{dy, nullptr, lddy, 0, Cout, 0};
  return launch_fused(src, B, H, W, Cout, 0, ud_kn, ldu, Cin, nullptr, SEGSDE_ACT_NONE, dx, lddx, accumulate, nullptr,
                      WinoAg{act_out, act_ld, act_kind}, stream);
}

// the mirrored-padding part of a reflection-padded 3x3 convolution's data-gradient, ADDED onto dx (after
// segsde_conv2d_winograd_fused_dgrad wrote the zero-padded part): wpack = the FORWARD pack [Cout][3][3][Cin] (segsde_pack_weight,
// for_dgrad = 0) with channel count ldw (Cin, or the full width when wpack points at one source's channel slice); act_out as in
// segsde_conv2d_winograd_fused_dgrad.  Two launches (row lines; column lines + corners).
extern "C" int segsde_reflect_adjoint_borders2(const float* dy, int lddy, const float* wpack, int ldw, float* dx, int lddx,
                                               const float* act_out, int act_ld, int act_kind, int B, int H, int W, int Cin, int Cout,
                                               void* stream) {
  if (!dy || !wpack || !dx) return SEGSDE_ERR_NULL;
  if (B <= 0 || H < 4 || W < 4 || Cin <= 0 || Cin % 32 || ldw < Cin || Cout <= 0 || Cout % 32 || Cout > 1024 || lddy < Cout || lddy % 4 || lddx < Cin ||
      ((uintptr_t)dy & 15) || (act_out && (act_kind < SEGSDE_ACT_RELU || act_kind > SEGSDE_ACT_SIGMOID || act_ld < Cin)))
    return SEGSDE_ERR_UNSUPPORTED;
  BorderP p;
  p.dy = dy; p.wp = wpack; p.dx = dx; p.agy = act_out; p.lddy = lddy; p.lddx = lddx; p.agld = act_ld; p.agkind = act_kind;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ldw = ldw;
  const size_t lb = (size_t)34 * (Cout + 1) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(reflect_borders_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
  for (int mode = 0; mode < 2; ++mode) {
    p.mode = mode; p.nseg = ((mode == 0 ? W : H) + 31) / 32;
    hipLaunchKernelGGL(reflect_borders_kernel, dim3((unsigned)(B * 2 * p.nseg)), dim3(256), lb, ST(stream), p);
    SEGSDE_CHECK_LAUNCH();
  }
  return 0;
}
