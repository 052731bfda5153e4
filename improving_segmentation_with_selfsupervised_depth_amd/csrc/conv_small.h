// internal: single-output-channel 3x3 convolution kernels (conv_small.hip), dispatched from the public conv entry points
#pragma once
#include <stddef.h>
bool segsde_c1_supported(int C, int ldx);
size_t segsde_c1_wgrad_workspace(int C);
int segsde_c1_forward(const float* x, int ldx, int B, int H, int W, int C, const float* wpack, const float* bias, int reflect,
                      int act, float* y, int ldy, void* stream);
// agy (nullable): saved OUTPUT of the activation that produced this conv's input (same pixel order as dx, pitch agld): the
// gradient channels < nsplit are multiplied by act'(agy) of kind agkind (SEGSDE_ACT_*) before they are stored
int segsde_c1_dgrad(const float* dz, int lddz, int B, int H, int W, int C, const float* wdpack, int adjoint, float* dx, int lddx,
                    float* dx2, int lddx2, int nsplit, const float* agy, int agld, int agkind, void* stream);
int segsde_c1_wgrad(const float* x, int ldx, int B, int H, int W, int C, const float* dz, int lddz, int reflect, float* dw,
                    float* workspace, void* stream);
// 1x1 convolution with a narrow (<= 32) dense input side (data-gradient of the segmentation head)
bool segsde_skinny_supported(int K, int C);
// the narrow side (a / dz) must be dense: row pitch == K, 16-byte aligned base
int segsde_skinny_nk(const float* a, int K, const float* w, long M, int C, float* y, int ldy, void* stream);
