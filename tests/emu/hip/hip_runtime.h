// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.  A host-side stand-in for <hip/hip_runtime.h> that lets the kernels of
// dgsparse-lib_amd/csrc be compiled as plain C++ (clang++ -x c++) and executed on the CPU, one workgroup at a time, with every
// work-item a fiber and the wave64 / workgroup collectives (shuffles, ballots, readfirstlane, wave and workgroup barriers) as
// rendezvous points (tests/emu/emu_rt.cpp).  It exists because the kernels' control logic - unit tables, hub deals, barrier
// protocols between role-specialised waves, index arithmetic - can then be checked against the oracle by `pytest -m "not gpu"`
// in a container without a GPU.  It says nothing about speed and it is never part of the product: nothing under
// dgsparse-lib_amd/ includes or links it, and the library built from it (tests/emu/_build/libdgs_emu.so) is only ever loaded
// by tests/test_emu_cpu.py.
//
// What the emulation checks that the hardware would not tell: a collective that not every live lane of a wave reaches, or a
// barrier that not every live work-item of a workgroup reaches, is a DEADLOCK REPORT (with the block and the lanes that wait),
// and a lane that reads LDS another lane of its wave wrote without a wave barrier or a collective in between gets stale data
// (fibers do not run in lockstep), i.e. a wrong result instead of an accident that works.
#pragma once
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#define DGS_HOST_EMU 1

// ---- qualifiers -------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static __attribute__((section("emu_lds")))  // one link section: saved / restored when workgroups take turns
#define __align__(n) alignas(n)

// ---- vector types -----------------------------------------------------------------------------------------------------------
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }

// ---- runtime (tests/emu/emu_rt.cpp) -----------------------------------------------------------------------------------------
namespace emu {
struct Item {          // the running work-item
  uint3 tid, bid, bdim, gdim;
};
extern Item *cur;
void wave_collective(const void *in, unsigned bytes, void *all /* 64 x bytes */, unsigned long long *live);
void wave_sync();
void block_sync();
void relax();
void launch_impl(dim3 grid, dim3 block, void (*fn)(void *), void *ctx);
void log_launch(const char *kern, dim3 grid);  // tests ask which kernels a call launched (emu_launch_log)
// relaxed-memory mode (DGS_EMU_MEM=relaxed, emu_rt.cpp): the accesses that go through these see per-wave store queues, per-XCD dirty
// lines and per-CU L1 snapshots inside the watched address ranges instead of "every access completes at once"
bool mem_on();
void mem_load(const void *p, void *out, unsigned bytes, int sc1);
void mem_store(void *p, const void *src, unsigned bytes, int sc1);
void mem_drain();
template <typename F>
inline void launch(dim3 grid, dim3 block, F &&f) {
  auto tramp = [](void *c) { (*static_cast<typename std::remove_reference<F>::type *>(c))(); };
  launch_impl(grid, block, tramp, &f);
}
template <typename T>
inline T shfl_from(T v, int src) {
  T all[64];
  unsigned long long live;
  wave_collective(&v, sizeof(T), all, &live);
  src &= 63;
  return ((live >> src) & 1ull) ? all[src] : v;  // a dead source lane: undefined on hardware, own value here
}
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)

// ---- host API stand-ins (all memory is host memory; streams are in-order by construction) --------------------------------------
typedef int hipError_t;
typedef struct emuStream *hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
  int multiProcessorCount;
  char name[64];
};
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef struct emuEvent *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)(uintptr_t)0x51; return hipSuccess; }  // (launches run at once, in call order)
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)(uintptr_t)0xe1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
// 256 CUs like the MI355X, so that every grid-shape decision that depends on cu_count() (hub block caps, panel workgroups) is taken
// as on the hardware (it used to be 16); DGS_EMU_CUS overrides (read once per process: cu_count() caches it)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  const char *e = getenv("DGS_EMU_CUS");
  const int n = e && *e ? atoi(e) : 256;
  p->multiProcessorCount = n > 0 ? n : 256;
  strcpy(p->name, "emu");
  return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { *p = malloc(n); return *p ? hipSuccess : 1; }
static inline hipError_t hipFreeAsync(void *p, hipStream_t) { free(p); return hipSuccess; }
#define hipLaunchKernelGGL(kern, grid, block, ldsbytes, stream, ...) \
  (::emu::log_launch(#kern, (grid)), ::emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); }))
#define HIP_SYMBOL(x) x
#define hipGetSymbolAddress(pp, sym) ((*(pp) = (void *)&(sym)), hipSuccess)
#define hipMemcpyToSymbol(sym, src, n) (memcpy((void *)&(sym), (src), (n)), hipSuccess)

// ---- device intrinsics ------------------------------------------------------------------------------------------------------
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
template <typename A, typename B>
static inline auto min(A a, B b) -> typename std::common_type<A, B>::type { return a < b ? a : b; }
template <typename A, typename B>
static inline auto max(A a, B b) -> typename std::common_type<A, B>::type { return a > b ? a : b; }

template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  const int lane = (int)(threadIdx.x & 63u);
  return emu::shfl_from(v, width >= 64 ? src : (lane & ~(width - 1)) + (src & (width - 1)));
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return emu::shfl_from(v, (int)(threadIdx.x & 63u) ^ mask);
}
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
  const int lane = (int)(threadIdx.x & 63u);
  const int src = lane - (int)d;
  const T o = emu::shfl_from(v, src < 0 ? lane : src);  // every lane takes part, whatever its source
  return (src < (lane & ~(width - 1))) ? v : o;
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
  const int lane = (int)(threadIdx.x & 63u);
  const int src = lane + (int)d;
  const T o = emu::shfl_from(v, src > 63 ? lane : src);
  return (src >= (lane & ~(width - 1)) + width) ? v : o;
}
static inline unsigned long long __ballot(int pred) {
  int all[64];
  unsigned long long live, m = 0;
  const int p = pred != 0;
  emu::wave_collective(&p, sizeof(int), all, &live);
  for (int l = 0; l < 64; l++)
    if (((live >> l) & 1ull) && all[l]) m |= 1ull << l;
  return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred) { return __ballot(!pred) == 0ull; }
static inline void __syncthreads() { emu::block_sync(); }
static inline long long wall_clock64() { return 0; }

template <typename T>
static inline T emu_readfirstlane(T v) {
  T all[64];
  unsigned long long live;
  emu::wave_collective(&v, sizeof(T), all, &live);
  return live ? all[__builtin_ctzll(live)] : v;
}
#define __builtin_amdgcn_readfirstlane(x) emu_readfirstlane(x)
#define __builtin_amdgcn_wave_barrier() ::emu::wave_sync()
#define __builtin_amdgcn_s_barrier() ::emu::block_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)  // (every access completes at once here)
#define __builtin_amdgcn_s_sleep(x) ::emu::relax()  // the yield point of a spin-wait

// atomics: one fiber runs at a time, so plain read-modify-write is atomic
template <typename T>
static inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }
static inline float unsafeAtomicAdd(float *p, float v) { const float o = *p; *p = o + v; return o; }
template <typename T>
static inline T atomicCAS(T *p, T cmp, T v) { const T o = *p; if (o == cmp) *p = v; return o; }
template <typename T>
static inline T atomicOr(T *p, T v) { const T o = *p; *p = o | v; return o; }
template <typename T>
static inline T atomicMax(T *p, T v) { const T o = *p; *p = o > v ? o : v; return o; }
template <typename T>
static inline T atomicMin(T *p, T v) { const T o = *p; *p = o < v ? o : v; return o; }
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
// __hip_atomic_load / _store at agent scope lower to sc1 accesses on gfx950 (L1 bypass / write-through); a read-modify-write goes
// to memory at once, ahead of the wave's queued stores (relaxed: no ordering) - one fiber runs at a time, so a plain RMW is atomic
namespace emu {
template <typename T>
static inline T atomic_load_(const T *p, int scope) {
  T v;
  if (mem_on()) mem_load(p, &v, sizeof(T), scope >= __HIP_MEMORY_SCOPE_AGENT);
  else v = *p;
  return v;
}
template <typename T>
static inline void atomic_store_(T *p, T v, int scope) {
  if (mem_on()) mem_store(p, &v, sizeof(T), scope >= __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <typename T>
static inline T atomic_fetch_add_(T *p, T v) { const T o = *p; *p = o + v; return o; }
}  // namespace emu
#define __hip_atomic_load(p, order, scope) ::emu::atomic_load_((p), (scope))
#define __hip_atomic_store(p, v, order, scope) ::emu::atomic_store_((p), (v), (scope))
#define __hip_atomic_fetch_add(p, v, order, scope) ::emu::atomic_fetch_add_((p), (v))

// raw buffer loads / stores through a descriptor (the in-kernel fold's 16-byte sc1 accesses): base + byte offset, and - where the
// hardware silently drops an out-of-range access - an abort, because in the product an out-of-range partial-row access is a bug
namespace emu {
struct BufRsrc {
  char *base;
  unsigned bytes;
};
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
static inline void buf_check(const BufRsrc &r, long off, const char *what) {
  if (off < 0 || (unsigned long)off + 16 > r.bytes) {
    fprintf(stderr, "emu: raw buffer %s out of range: byte offset %ld, descriptor covers %u bytes\n", what, off, r.bytes);
    abort();
  }
}
static inline u4_t buf_load128(const BufRsrc &r, int voff, int soff, int aux) {
  buf_check(r, (long)(unsigned)voff + soff, "load");
  u4_t v;
  if (mem_on()) mem_load(r.base + (unsigned)voff + soff, &v, 16, (aux >> 4) & 1);  // aux bit 4 = sc1
  else memcpy(&v, r.base + (unsigned)voff + soff, 16);
  return v;
}
static inline void buf_store128(u4_t v, const BufRsrc &r, int voff, int soff, int aux) {
  buf_check(r, (long)(unsigned)voff + soff, "store");
  if (mem_on()) mem_store(r.base + (unsigned)voff + soff, &v, 16, (aux >> 4) & 1);
  else memcpy(r.base + (unsigned)voff + soff, &v, 16);
}
}  // namespace emu
#define __amdgpu_buffer_rsrc_t ::emu::BufRsrc
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, n, flags) (::emu::BufRsrc{(char *)(p), (unsigned)(n)})
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) ::emu::buf_load128(r, voff, soff, aux)
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, aux) ::emu::buf_store128(v, r, voff, soff, aux)
