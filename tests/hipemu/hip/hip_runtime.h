// TEST INFRASTRUCTURE ONLY -- a tiny *interpreter* for HIP kernels (blocks of a launch are spread over a few host threads).
//
// The build container has no GPU.  To check index math, LDS staging, barriers
// and MFMA/shuffle lane layouts of the real kernel sources (csrc/*.hip) BEFORE
// spending GPU minutes, tests compile those same sources for the host with
// `clang++ -x c++ -I tests/hipemu` so that `#include <hip/hip_runtime.h>`
// resolves to this file.  Every HIP thread of a block becomes a ucontext fiber;
// __syncthreads / wave shuffles / MFMA are rendezvous points between fibers.
// Nothing here is part of the product: the shipped library is built by hipcc
// from the unmodified sources, and the package never loads the emulated .so
// (tests inject it explicitly).
#pragma once
#include <ucontext.h>
#include <sys/mman.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)
#define __restrict__
#define warpSize 64
using std::min;
using std::max;

typedef int hipError_t;
typedef void* hipStream_t;
static const int hipSuccess = 0;
static const int hipErrorInvalidValue = 1;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
// two "CUs": persistent kernels (wino_fused_kernel) then run four workgroups that each walk over several items
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
inline float2 make_float2(float a, float b) { return float2{a, b}; }

namespace emu {
struct Barrier { int count = 0; unsigned gen = 0; int expected = 0; };
struct Fiber {
  ucontext_t ctx; void* stack = nullptr; bool done = false; dim3 tid; int lin = 0;
  Barrier* wait = nullptr; unsigned wait_gen = 0;
};
struct State {
  ucontext_t sched; std::vector<Fiber> fibers; Fiber* cur = nullptr; dim3 bid, bdim, gdim;
  Barrier block_bar; std::vector<Barrier> wave_bar; std::function<void()> body;
  // rendezvous scratch per wave
  std::vector<float> xa, xb, xa8, xb8; std::vector<double> xd; std::vector<long long> xi;
  unsigned char* lds = nullptr;     // this host thread's 160 KB of "LDS" (one block runs per host thread at a time)
};
// one interpreter state per host thread (the blocks of a launch are dealt out to a small pool, see launch()); the TLS slot
// holds a plain pointer, so an access is a load + a predictable branch
inline State*& tls_state() { static thread_local State* p = nullptr; return p; }
inline State& S() {
  State* p = tls_state();
  if (__builtin_expect(!p, 0)) {
    p = new State; p->lds = static_cast<unsigned char*>(aligned_alloc(64, 160 * 1024)); memset(p->lds, 0, 160 * 1024);
    tls_state() = p;
  }
  return *p;
}
static const size_t STACK = 256 * 1024;
inline void yield() { State& s = S(); swapcontext(&s.cur->ctx, &s.sched); }
inline void barrier_wait(Barrier* b) {
  State& s = S(); Fiber* f = s.cur; unsigned g = b->gen;
  if (++b->count == b->expected) { b->count = 0; b->gen++; return; }
  f->wait = b; f->wait_gen = g;
  while (b->gen == g) yield();
  f->wait = nullptr;
}
inline void trampoline() { State& s = S(); s.body(); s.cur->done = true; swapcontext(&s.cur->ctx, &s.sched); }
inline void run_block(int nthreads) {
  State& s = S();
  if ((int)s.fibers.size() < nthreads) {
    size_t old = s.fibers.size(); s.fibers.resize(nthreads);
    for (size_t i = old; i < s.fibers.size(); ++i)
      s.fibers[i].stack = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  }
  int nw = (nthreads + 63) / 64;
  s.wave_bar.assign(nw, Barrier());
  for (int w = 0; w < nw; ++w) s.wave_bar[w].expected = std::min(64, nthreads - 64 * w);
  s.block_bar = Barrier(); s.block_bar.expected = nthreads;
  s.xa.assign(nw * 64, 0.f); s.xb.assign(nw * 64, 0.f); s.xd.assign(nw * 64, 0.0); s.xi.assign(nw * 64, 0);
  for (int i = 0; i < nthreads; ++i) {
    Fiber& f = s.fibers[i]; f.done = false; f.wait = nullptr; f.lin = i;
    f.tid = dim3(i % s.bdim.x, (i / s.bdim.x) % s.bdim.y, i / (s.bdim.x * s.bdim.y));
    getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  int alive = nthreads; long idle_rounds = 0;
  while (alive > 0) {
    bool progressed = false;
    for (int i = 0; i < nthreads; ++i) {
      Fiber& f = s.fibers[i];
      if (f.done) continue;
      if (f.wait && f.wait->gen == f.wait_gen) continue;  // still blocked
      s.cur = &f; swapcontext(&s.sched, &f.ctx); progressed = true;
      if (f.done) --alive;
    }
    if (!progressed && ++idle_rounds > 4) { fprintf(stderr, "hipemu: deadlock (divergent barrier?)\n"); abort(); }
    if (progressed) idle_rounds = 0;
  }
}
// Host-thread pool: the blocks of one launch are independent (what they share goes through integer atomics, fences or the
// next launch), so they are dealt out block by block to SEGSDE_EMU_THREADS host threads (default: the cores, at most 8;
// 1 = the old single-threaded behaviour).  A launch returns when all its blocks are done, like a synchronous stream.
struct Pool {
  std::mutex m; std::condition_variable cv_job, cv_done; std::vector<std::thread> th;
  std::function<void()> body; dim3 grid, block; std::atomic<unsigned> next{0}; unsigned total = 0; int pending = 0;
  unsigned long gen = 0; int nthreads = 0;
  static void work(Pool* p) {
    State& s = S(); s.gdim = p->grid; s.bdim = p->block; s.body = p->body;
    const unsigned gx = p->grid.x, gy = p->grid.y;
    for (unsigned i; (i = p->next.fetch_add(1)) < p->total;) {
      s.bid = dim3(i % gx, (i / gx) % gy, i / (gx * gy)); run_block(p->nthreads);
    }
    s.body = nullptr;
  }
  static void loop(Pool* p) {
    unsigned long seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> l(p->m); p->cv_job.wait(l, [&] { return p->gen != seen; }); seen = p->gen; }
      work(p);
      { std::lock_guard<std::mutex> l(p->m); if (--p->pending == 0) p->cv_done.notify_one(); }
    }
  }
};
inline Pool& pool() {
  static Pool* p = [] {
    Pool* q = new Pool;                      // never destroyed: the workers live until the process exits
    const char* e = getenv("SEGSDE_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    for (int i = 1; i < n; ++i) { q->th.emplace_back(Pool::loop, q); q->th.back().detach(); }
    return q;
  }();
  return *p;
}
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t, A... args) {
  if (smem > 160 * 1024) { fprintf(stderr, "hipemu: %zu B of LDS requested\n", smem); abort(); }
  Pool& p = pool();
  p.grid = grid; p.block = block; p.nthreads = block.x * block.y * block.z;
  p.total = grid.x * grid.y * grid.z; p.next = 0;
  p.body = [=]() { kernel(args...); };
  const bool fan = !p.th.empty() && p.total > 1;
  if (fan) { std::lock_guard<std::mutex> l(p.m); p.pending = (int)p.th.size(); ++p.gen; }
  if (fan) p.cv_job.notify_all();
  Pool::work(&p);
  if (fan) { std::unique_lock<std::mutex> l(p.m); p.cv_done.wait(l, [&] { return p.pending == 0; }); }
  p.body = nullptr;
}
inline int lane() { return S().cur->lin & 63; }
inline int wave() { return S().cur->lin >> 6; }
inline Barrier* wbar() { return &S().wave_bar[wave()]; }
template <class T> inline std::vector<T>& scratch();
template <> inline std::vector<float>& scratch<float>() { return S().xa; }
template <> inline std::vector<double>& scratch<double>() { return S().xd; }
template <> inline std::vector<long long>& scratch<long long>() { return S().xi; }
template <class T, class U> inline T shfl_idx(T v, U srcf) {
  auto& x = scratch<T>(); int base = wave() * 64; x[base + lane()] = v;
  barrier_wait(wbar()); int src = srcf(lane());
  T r = (src >= 0 && src < S().wave_bar[wave()].expected) ? x[base + src] : v;
  barrier_wait(wbar()); return r;
}
}  // namespace emu

#define threadIdx (emu::S().cur->tid)
#define blockIdx (emu::S().bid)
#define blockDim (emu::S().bdim)
#define gridDim (emu::S().gdim)
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) emu::launch(kernel, dim3(grid), dim3(block), smem, stream, ##__VA_ARGS__)

#define SEGSDE_SMEM unsigned char* const segsde_smem = ::emu::S().lds
// raw buffer loads: range-checked against num_records on the per-lane offset, zeros when out of range
#define SEGSDE_BUFFER_OPS 1
#define SEGSDE_OPAQUE(x) ((void)(x))
#define SEGSDE_REFRESH_KERNARG(T, arg) (arg)
#define SEGSDE_OOB 0x80000000u
struct segsde_rsrc { const char* base; unsigned n; };
inline segsde_rsrc segsde_make_rsrc(const void* base, unsigned n = 0x7fffffffu) { return segsde_rsrc{static_cast<const char*>(base), n}; }
inline float4 segsde_buffer_load4(segsde_rsrc r, unsigned voff, unsigned soff) {
  if (voff >= r.n) return make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v;
  memcpy(&v, r.base + (size_t)voff + soff, sizeof(v));
  return v;
}
inline float segsde_buffer_load1(segsde_rsrc r, unsigned voff, unsigned soff) {
  if (voff >= r.n) return 0.f;
  float v;
  memcpy(&v, r.base + (size_t)voff + soff, sizeof(v));
  return v;
}
inline void segsde_buffer_store4(segsde_rsrc r, unsigned voff, unsigned soff, float4 v) {
  if (voff >= r.n) return;
  memcpy(const_cast<char*>(r.base) + (size_t)voff + soff, &v, sizeof(v));
}
inline void segsde_buffer_store4_nt(segsde_rsrc r, unsigned voff, unsigned soff, float4 v) { segsde_buffer_store4(r, voff, soff, v); }
inline void segsde_buffer_store1(segsde_rsrc r, unsigned voff, unsigned soff, float v) {
  if (voff >= r.n) return;
  memcpy(const_cast<char*>(r.base) + (size_t)voff + soff, &v, sizeof(v));
}
inline void segsde_buffer_store1_nt(segsde_rsrc r, unsigned voff, unsigned soff, float v) { segsde_buffer_store1(r, voff, soff, v); }

// LDS-DMA flavour: lane l's 16 bytes land at lds_wave_base + 16*l (executed synchronously here: the interpreter cannot
// model a missing wait, only wrong addresses / wrong buffer hand-over order)
inline unsigned segsde_lds_addr(const void* p) { return (unsigned)(static_cast<const unsigned char*>(p) - ::emu::S().lds); }
inline void segsde_buffer_load4_lds(segsde_rsrc r, unsigned voff, unsigned soff, unsigned lds_wave_addr) {
  const float4 v = segsde_buffer_load4(r, voff, soff);
  memcpy(::emu::S().lds + lds_wave_addr + 16 * emu::lane(), &v, sizeof(v));
}
#define SEGSDE_LDS_READ_IMM(p, i) ((p)[i])
inline void segsde_wait_vmcnt0() {}
template <int N> inline void segsde_wait_vmcnt() {}

inline void __syncthreads() { emu::barrier_wait(&emu::S().block_bar); }
inline float __shfl_xor(float v, int m, int = 64) { return emu::shfl_idx<float>(v, [m](int l) { return l ^ m; }); }
inline float __shfl_down(float v, int d, int = 64) { return emu::shfl_idx<float>(v, [d](int l) { return l + d; }); }
inline float __shfl(float v, int s, int = 64) { return emu::shfl_idx<float>(v, [s](int) { return s; }); }
inline double __shfl_xor(double v, int m, int = 64) { return emu::shfl_idx<double>(v, [m](int l) { return l ^ m; }); }
inline double __shfl_down(double v, int d, int = 64) { return emu::shfl_idx<double>(v, [d](int l) { return l + d; }); }
inline int __shfl_xor(int v, int m, int = 64) { return (int)emu::shfl_idx<long long>(v, [m](int l) { return l ^ m; }); }
inline int __shfl_down(int v, int d, int = 64) { return (int)emu::shfl_idx<long long>(v, [d](int l) { return l + d; }); }
// wave votes: every lane publishes its predicate, then all lanes read the whole wave's
inline int __all(int pred) {
  using namespace emu; auto& x = scratch<long long>(); int base = wave() * 64; x[base + lane()] = pred != 0;
  barrier_wait(wbar()); int n = S().wave_bar[wave()].expected, r = 1;
  for (int l = 0; l < n; ++l) r &= (int)x[base + l];
  barrier_wait(wbar()); return r;
}
inline int __any(int pred) { return !__all(!pred); }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31 (cdna_hip_programming.md section 3).
inline emu_f32x16 emu_mfma_32x32x2(float a, float b, emu_f32x16 c, int, int, int) {
  using namespace emu; State& s = S(); int base = wave() * 64, l = lane();
  s.xa[base + l] = a; s.xb[base + l] = b; barrier_wait(wbar());
  emu_f32x16 d = c; int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(s.xa[base + row + 32 * k], s.xb[base + col + 32 * k], acc);
    d[r] = acc;
  }
  barrier_wait(wbar()); return d;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r: row 4*(l>>4)+r, col l&15.
inline emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c, int, int, int) {
  using namespace emu; State& s = S(); int base = wave() * 64, l = lane();
  s.xa[base + l] = a; s.xb[base + l] = b; barrier_wait(wbar());
  emu_f32x4 d = c; int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r; float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(s.xa[base + row + 16 * k], s.xb[base + col + 16 * k], acc);
    d[r] = acc;
  }
  barrier_wait(wbar()); return d;
}
// v_mfma_f32_32x32x16_f16: lane l supplies A[i=l&31][k=8*(l>>5)..+7], B[k=8*(l>>5)..+7][j=l&31] as halves; fp32 accumulation
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
inline emu_f32x16 emu_mfma_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c, int, int, int) {
  using namespace emu; State& s = S(); int base = wave() * 64, l = lane();
  if (s.xa8.size() < s.xa.size() * 8) { s.xa8.assign(s.xa.size() * 8, 0.f); s.xb8.assign(s.xa.size() * 8, 0.f); }
  for (int q = 0; q < 8; ++q) { s.xa8[(base + l) * 8 + q] = (float)a[q]; s.xb8[(base + l) * 8 + q] = (float)b[q]; }
  barrier_wait(wbar());
  emu_f32x16 d = c; int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int h = 0; h < 2; ++h)
      for (int q = 0; q < 8; ++q) acc = fmaf(s.xa8[(base + row + 32 * h) * 8 + q], s.xb8[(base + col + 32 * h) * 8 + q], acc);
    d[r] = acc;
  }
  barrier_wait(wbar()); return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 emu_mfma_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_16x16x4
#define HIP_SYMBOL(x) (&x)
inline hipError_t hipGetSymbolAddress(void** p, const void* sym) { *p = const_cast<void*>(sym); return 0; }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only applied to wave-uniform values
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }   // v_rcp_f32 (1 ulp on the device)
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
