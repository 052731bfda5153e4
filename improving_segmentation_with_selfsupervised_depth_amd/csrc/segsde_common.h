// Shared device helpers for the segsde gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/segsde_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// eight halves = one operand of v_mfma_f32_32x32x16_f16 (the `amp: True` arithmetic of the convolutions, conv_igemm.hip)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 segsde_pack_f16(const float4& a, const float4& b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f16x8 r;
  h2 t;
  t = __builtin_convertvector(f2{a.x, a.y}, h2); r[0] = t[0]; r[1] = t[1];      // v_cvt_pk_f16_f32: round to nearest even
  t = __builtin_convertvector(f2{a.z, a.w}, h2); r[2] = t[0]; r[3] = t[1];
  t = __builtin_convertvector(f2{b.x, b.y}, h2); r[4] = t[0]; r[5] = t[1];
  t = __builtin_convertvector(f2{b.z, b.w}, h2); r[6] = t[0]; r[7] = t[1];
  return r;
}

// all LDS lives in the dynamic region (16-byte aligned base, cdna_hip_programming.md G17)
#ifndef SEGSDE_SMEM
#define SEGSDE_SMEM extern __shared__ __attribute__((aligned(16))) unsigned char segsde_smem[]
#endif

// Raw buffer loads (buffer_load_dwordx4 ... offen): address = resource base + per-lane byte offset (VGPR) + wave-uniform
// byte offset (SGPR).  A lane whose offset is >= num_records reads zeros -- padding taps / rows past the tile edge set
// SEGSDE_OOB instead of selecting a pointer, and the per-chunk channel advance rides in the SGPR, so the K loop of the
// conv kernels issues its tile loads with no vector ALU work at all (on gfx950 fp32 MFMA and VALU share issue cycles:
// every VALU instruction in the loop is ~3 cycles taken from the matrix pipe, profiles/probe_r01_mfma_valu_overlap.log).
#ifndef SEGSDE_BUFFER_OPS   // the host interpreter under tests/hipemu supplies its own
#define SEGSDE_OOB 0x80000000u
typedef __amdgpu_buffer_rsrc_t segsde_rsrc;
__device__ __forceinline__ segsde_rsrc segsde_make_rsrc(const void* base, unsigned num_records = 0x7fffffffu) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)num_records, 0x00020000);
}
__device__ __forceinline__ float4 segsde_buffer_load4(segsde_rsrc r, unsigned voff, unsigned soff) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float segsde_buffer_load1(segsde_rsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void segsde_buffer_store4(segsde_rsrc r, unsigned voff, unsigned soff, float4 v) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t d;
  d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y); d.z = __float_as_uint(v.z); d.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);   // out-of-range voff: the store is dropped
}
// the same store with the non-temporal ("nt": streaming) cache policy -- for outputs no cache level can hold until they are read
__device__ __forceinline__ void segsde_buffer_store4_nt(segsde_rsrc r, unsigned voff, unsigned soff, float4 v) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t d;
  d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y); d.z = __float_as_uint(v.z); d.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 2);
}
// one dword per lane: with the lane's constant offset in the VGPR and everything wave-uniform in the SGPR offset a store / load of a
// row of pixels costs no vector address arithmetic (the Winograd epilogue: 32 stores per thread)
__device__ __forceinline__ void segsde_buffer_store1(segsde_rsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void segsde_buffer_store1_nt(segsde_rsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 2);
}
// LDS-DMA: the same raw buffer load, but the 16 bytes of lane l land in LDS at lds_wave_base + 16*l without passing
// through VGPRs (buffer_load_dwordx4 ... offen lds; destination = M0 + 16*lane, so the LDS image of one instruction is
// 1 KiB lane-linear -- a swizzled layout is obtained by permuting which SOURCE element each lane fetches).  Out-of-range
// lanes store zeros.  Issued through inline asm on purpose: hipcc would order a builtin LDS-DMA before every later LDS
// read with s_waitcnt vmcnt(0) (it cannot prove that the other staging buffer is not the one being read).  hipcc does not
// count these loads: the caller waits with segsde_wait_vmcnt0() before the barrier that publishes the buffer.
// lds_wave_addr: LDS BYTE address (segsde_lds_addr) of the 1 KiB block, an SGPR value.
// LDS byte address of a pointer into the workgroup's LDS (wave-uniform; computed once, outside the loops)
__device__ __forceinline__ unsigned segsde_lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p);
}
__device__ __forceinline__ void segsde_buffer_load4_lds(segsde_rsrc r, unsigned voff, unsigned soff, unsigned lds_wave_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(lds_wave_addr) : "memory");
}
// One LDS dword at p[i] (p: pointer into the workgroup's LDS, i: compile-time index after unrolling) as a single ds_read_b32
// whose immediate carries i (16 bits of byte offset).  The volatile access keeps the load / store optimizer from pairing
// neighbouring reads into ds_read2_b32: that instruction's two 8-bit dword offsets reach 1 KiB, and every pair beyond costs a
// v_add_u32 to rebase -- vector issue slots taken from the matrix pipe in an MFMA loop.  Waits are still the compiler's.
#define SEGSDE_LDS_READ_IMM(p, i) (((const volatile __attribute__((address_space(3))) float*)(p))[i])
__device__ __forceinline__ void segsde_wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// at most N of this wave's vector-memory operations still outstanding (N tile loads of later chunks may stay in flight)
template <int N> __device__ __forceinline__ void segsde_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#endif

// A fresh copy of the kernel's FIRST by-value argument, loaded from the kernel-argument segment at this point of the
// program (scalar loads of the fields that are used): the optimiser cannot merge it with the copy live since kernel entry.
#ifndef SEGSDE_REFRESH_KERNARG
template <typename T>
__device__ __forceinline__ T segsde_kernarg_here() {
  auto k = __builtin_amdgcn_kernarg_segment_ptr();   // constant address space: uniform loads through it are scalar loads
  asm volatile("" : "+s"(k));
  static_assert(sizeof(T) % 4 == 0, "dword copy");
  const __attribute__((address_space(4))) unsigned* src = (const __attribute__((address_space(4))) unsigned*)k;
  T out;
  unsigned* dst = reinterpret_cast<unsigned*>(&out);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];   // only the fields that are used survive
  return out;
}
#define SEGSDE_REFRESH_KERNARG(T, arg) segsde_kernarg_here<T>()
#endif

// hides a VGPR value's provenance from the optimiser (no instruction is emitted)
#ifndef SEGSDE_OPAQUE
#define SEGSDE_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

#define SEGSDE_CHECK_LAUNCH()                       \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return (int)e_ ? (int)e_ : -1; \
  } while (0)

static inline int segsde_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tickets for "the last workgroup finishes the job" reductions (weight-gradient splits, column-sum finalize): a zeroed slice
// of a device-resident ring for one launch (csrc/abi.hip); nullptr if it cannot be provided (the caller then uses its
// two-kernel path).
unsigned* segsde_ticket_slice(int n);

// activation codes shared by conv epilogues, bn_apply and act_backward
__device__ __forceinline__ float segsde_act(float v, int act) {
  if (act == SEGSDE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SEGSDE_ACT_ELU) return v > 0.f ? v : (expf(v) - 1.f);
  if (act == SEGSDE_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}
// derivative expressed through the saved OUTPUT y (what the reference's in-place ELU/ReLU keep too)
__device__ __forceinline__ float segsde_act_grad_from_out(float y, int act) {
  if (act == SEGSDE_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == SEGSDE_ACT_ELU) return y > 0.f ? 1.f : (y + 1.f);
  if (act == SEGSDE_ACT_SIGMOID) return y * (1.f - y);
  return 1.f;
}

// counter-based RNG for dropout masks: the backward pass regenerates the mask from (seed, element index)
__device__ __forceinline__ uint32_t segsde_hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ float segsde_uniform01(uint64_t seed, uint64_t idx) {
  return (segsde_hash32(seed * 0x9E3779B97F4A7C15ULL + idx) >> 8) * (1.0f / 16777216.0f);
}

// XCD-aware, bijective remap of a linear block id: hardware places block b on XCD b%8; give every XCD a
// contiguous range of logical tiles so that neighbouring tiles (shared halo rows / weight panels) share an L2.
__device__ __forceinline__ int segsde_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float segsde_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double segsde_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum (256-thread blocks); `sh` points at >= 4 doubles of LDS; result returned to every thread
__device__ __forceinline__ double segsde_block_sum(double v, double* sh) {
  v = segsde_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}
