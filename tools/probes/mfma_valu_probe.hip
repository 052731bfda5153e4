// Micro-probe: does VALU work issued between fp32 MFMAs overlap with the matrix pipe on gfx950, and does it when two
// waves share a SIMD?   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip && ./mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int QUARTER>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  unsigned x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, y = 11;
#define VALU1(r) do { if (QUARTER) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(y)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(y)); } while (0)
#define VALUS()                                      \
  _Pragma("unroll") for (int k = 0; k < K; k += 4) { \
    VALU1(x0); if (k + 1 < K) VALU1(x1); if (k + 2 < K) VALU1(x2); if (k + 3 < K) VALU1(x3); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      VALUS();
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      VALUS();
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
      VALUS();
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
      VALUS();
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(x0 + x1 + x2 + x3);
}

template <int K, int Q>
void run(int blocks_per_cu, float* out) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<K, Q>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<K, Q>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 32 * blocks_per_cu;     // each SIMD holds blocks_per_cu waves
  const double tf = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
  printf("K=%2d %s waves/SIMD=%d  %.3f ms  %.1f ns per MFMA slot  %.1f TFLOP/s\n", K, Q ? "mul_lo" : "add   ", blocks_per_cu, ms,
         ms * 1e6 / mfma_per_simd, tf);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>(w, out); run<4, 0>(w, out); run<8, 0>(w, out); run<12, 0>(w, out); run<16, 0>(w, out); run<24, 0>(w, out); run<32, 0>(w, out);
    run<4, 1>(w, out); run<8, 1>(w, out);
  }
  return 0;
}
