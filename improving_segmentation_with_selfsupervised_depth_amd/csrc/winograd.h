// internal: Winograd F(2x2,3x3) transforms (winograd.hip) around the grouped position GEMMs of conv_igemm.hip
#pragma once
#include <stddef.h>
#include "segsde_common.h"
// x [B,H,W,C] (pixel pitch ld, C % 4 == 0, H and W even), 3x3 / stride 1 / pad 1 with zero or mirrored padding ->
// V [16][T][C], T = B * H/2 * W/2 tiles, position p = 4 * row-transform index + column-transform index
// two sources: channels [0, C0) from x0, [C0, C) from x1 (x1 NULL: one source); dil: the window's dilation (= its padding)
int segsde_wino_input(const float* x0, int ld0, const float* x1, int ld1, int C0, int B, int H, int W, int C, int dil, int reflect,
                      float* V, void* stream);
// M [16][T][Co] -> y [B,H,W,Co] (pitch ldy); part (nullable): [segsde_wino_stats_rows(T)][2][Co] doubles, column sums / sums
// of squares of y (the following BatchNorm's batch statistics, like the implicit-GEMM epilogue's partials)
// y = act(Y + bias) (bias nullable, act a SEGSDE_ACT_* code; the statistics are those of the stored values)
int segsde_wino_output(const float* M, int B, int H, int W, int Co, int dil, const float* bias, int act, float* y, int ldy,
                       double* part, void* stream);
long segsde_wino_stats_rows(long T);
// rows per position plane of V / M / dM: T rounded up to the GEMM's 128-row tiles (the transforms zero-fill / skip the rest)
long segsde_wino_rows(long T);
// OIHW 3x3 weight -> U [16][O][I] (transpose_flip = 0: forward) or U' [16][I][O] of the spatially flipped kernel (1: the
// data-gradient's convolution), U = G g G^T
int segsde_wino_weights(const float* w_oihw, int O, int I, int transpose_flip, float* U, void* stream);
// many weights in one launch: device-resident job table (include/segsde_hip.h: segsde_wino_job), blocks [block0_j, block0_{j+1})
int segsde_wino_weights_multi(const segsde_wino_job* jobs_device, int njobs, int total_blocks, void* stream);
// weight gradient: dM [16][T][C] = A dY A^T of the output gradient dy [B,H,W,C]; part [16 * s][Cin][Co] (the split slabs of the
// sixteen position GEMMs dU_p = V_p^T dM_p, s per position) -> dW [Co][Cin][3][3] = G^T dU G
int segsde_wino_grad(const float* dy, int ld, int B, int H, int W, int C, int dil, float* dM, void* stream);
int segsde_wino_wgrad_finish(const float* part, int s, int Cin, int Co, float* dw_oihw, void* stream);
// 1: the one-kernel route's packs are in the blocked layout (winograd_fused.hip)
int segsde_wino_ublk();
