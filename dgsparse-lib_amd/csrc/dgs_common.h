// dgs_common.h -- shared device helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "dgsparse_hip.h"

namespace dgs {

// Internal fifth reduce op: masked sum = backward of max/min w.r.t. the dense operand, run on the CSC arrays:
//   out[j,f] = sum_p [Em[idx[p],f] == j] * val[p] * G[idx[p],f]
// (reference csrspmm_seqreduce_rowbalance_with_mask_kernel, include/cuda/spmm_cuda.cuh:400-433; the formula, not that
// kernel's stale-variable behaviour).  Same schedule as the forward; the E pointer carries the saved arg ids (input).
constexpr int kOpMaskSum = 4;

// Dynamic LDS of the column-panel kernels.  (The hooks for tests/emu - the host-side wave64 emulation the CPU test suite runs the
// kernels' control logic on: this declaration, the inline-asm block below - store_vec_hidden, drain_vmem - and DGS_EMU_LD / _ST in
// load_vec / store_vec, through which the emulation's relaxed-memory mode sees the plain vector accesses (tests/emu/emu_rt.cpp:
// per-wave store queues, per-XCD dirty lines, per-CU L1 - what the in-kernel fold's hand-over has to be right about).
// DGS_HOST_EMU is never defined in a product build: the two macros are empty there.)
#ifndef DGS_HOST_EMU
#define DGS_DYN_SHARED(name) extern __shared__ __align__(16) char name[]
#else
#define DGS_DYN_SHARED(name) extern char name[]
#endif

#ifdef DGS_HOST_EMU
#define DGS_EMU_LD(p, o, bytes) do { if (::emu::mem_on()) { ::emu::mem_load((p), (o), (bytes), 0); return; } } while (0)
#define DGS_EMU_ST(p, o, bytes) do { if (::emu::mem_on()) { ::emu::mem_store((p), (o), (bytes), 0); return; } } while (0)
#else
#define DGS_EMU_LD(p, o, bytes) ((void)0)
#define DGS_EMU_ST(p, o, bytes) ((void)0)
#endif

constexpr int kWave = 64;    // CDNA wavefront
constexpr int kBlock = 256;  // 4 waves per workgroup, one per SIMD

// Identities of include/gspmm.h:133-146: (float)INT_MIN / (float)INT_MAX, not +-inf.
template <int OP>
__device__ __forceinline__ float reduce_init() {
  if constexpr (OP == DGS_MAX) return (float)INT_MIN;
  if constexpr (OP == DGS_MIN) return (float)INT_MAX;
  return 0.0f;
}

// One reduction step of algorithm 0 (include/cuda/spmm_cuda.cuh:37-43 + gspmm.h:16-17 macros, taken
// literally so that NaN/tie behaviour is identical): t = w*x is ONE fp32 rounding; E takes the column
// id on a strict improvement, so the first occurrence in CSR order wins ties.
// FMA = false (strict no-contraction mode, DGS_ALG_STRICT_NOFMA): sum/mean as `res + (w * x)` with two roundings - what the
// reference's host loop computes when g++ compiles it without FMA instructions (example/util/sp_util.hpp:73-83).
template <bool FMA>
__device__ __forceinline__ float chain_step(float w, float x, float acc) {
  if constexpr (FMA) {
    return __builtin_fmaf(w, x, acc);
  } else {
    // HIP's __fmul_rn / __fadd_rn are plain operators, which hipcc (-ffp-contract=fast-honor-pragmas) would fuse again
#pragma clang fp contract(off)
    const float t = w * x;
    return acc + t;
  }
}
template <int OP, bool FMA = true>
__device__ __forceinline__ void reduce_step(float &res, int &eidx, float w, float x, int c) {
  if constexpr (OP == DGS_MAX) {
    const float t = w * x;
    if (res < t) eidx = c;
    res = (res < t) ? t : res;
  } else if constexpr (OP == DGS_MIN) {
    const float t = w * x;
    if (res > t) eidx = c;
    res = (res < t) ? res : t;
  } else {
    res = chain_step<FMA>(w, x, res);  // FMA: v_fmac_f32, the contraction nvcc applies to res + val*x
  }
}

// Optional epilogue of the sum / mean SpMM, applied where a finished output row leaves the kernel (saves the read + write of
// the M x N result a separate elementwise pass costs: 0.5 GB of the 2.7 GB a GCN layer moves on the headline graph):
//   out[r, f] = relu(row_scale[r] * acc + bias[f])     every part optional
// with the same three roundings as the unfused torch ops (multiply, add, clamp - no contraction), so fused == unfused bit
// for bit.  relu keeps NaN and -0.0 as torch.relu does.
struct Epi {
  const float *bias;    // [N] or nullptr
  const float *rscale;  // [M] or nullptr
  int relu;
};
template <int V>
__device__ __forceinline__ void epi_apply(float (&o)[V], int64_t row, int f0, const Epi &ep) {
  if (!(ep.bias || ep.rscale || ep.relu)) return;  // uniform
  {
#pragma clang fp contract(off)
    if (ep.rscale) {
      const float s = ep.rscale[row];
#pragma unroll
      for (int v = 0; v < V; v++) o[v] = o[v] * s;
    }
    if (ep.bias) {
#pragma unroll
      for (int v = 0; v < V; v++) o[v] = o[v] + ep.bias[f0 + v];
    }
    if (ep.relu) {
#pragma unroll
      for (int v = 0; v < V; v++) o[v] = (o[v] < 0.0f) ? 0.0f : o[v];
    }
  }
}

// V consecutive floats / ints at p (p is 4*V-byte aligned by construction of the dispatch).
template <int V>
__device__ __forceinline__ void load_vec(const float *p, float (&o)[V]) {
  DGS_EMU_LD(p, o, 4 * V);
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else if constexpr (V == 2) {
    const float2 t = *reinterpret_cast<const float2 *>(p);
    o[0] = t.x; o[1] = t.y;
  } else {
    o[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void load_vec(const int *p, int (&o)[V]) {
  DGS_EMU_LD(p, o, 4 * V);
  if constexpr (V == 4) {
    const int4 t = *reinterpret_cast<const int4 *>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else if constexpr (V == 2) {
    const int2 t = *reinterpret_cast<const int2 *>(p);
    o[0] = t.x; o[1] = t.y;
  } else {
    o[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void store_vec(float *p, const float (&o)[V]) {
  DGS_EMU_ST(p, o, 4 * V);
  if constexpr (V == 4) {
    *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
  } else if constexpr (V == 2) {
    *reinterpret_cast<float2 *>(p) = make_float2(o[0], o[1]);
  } else {
    *p = o[0];
  }
}
template <int V>
__device__ __forceinline__ void store_vec(int *p, const int (&o)[V]) {
  DGS_EMU_ST(p, o, 4 * V);
  if constexpr (V == 4) {
    *reinterpret_cast<int4 *>(p) = make_int4(o[0], o[1], o[2], o[3]);
  } else if constexpr (V == 2) {
    *reinterpret_cast<int2 *>(p) = make_int2(o[0], o[1]);
  } else {
    *p = o[0];
  }
}

// Stores the compiler does not count.  hipcc's s_waitcnt pass treats a wave with BOTH loads and stores pending
// on vmcnt as "out of order" and drains with vmcnt(0), which collapses a rolling window of gathers as soon as a
// row result is stored inside the loop.  An inline-asm store is invisible to that bookkeeping, so the loads keep
// their counted vmcnt(N) waits.  Safe: the hardware counter then only over-counts (loads still retire in order
// among themselves, so "counter <= N" still implies the needed load has landed), nothing ever reads these
// addresses again in the kernel, and outstanding stores complete on their own at s_endpgm.
// The trailing s_nop 1 keeps hipcc's next instruction from overwriting the data registers before the store
// has read them (cdna_hip_programming.md 5.7 item 1).
typedef float dgs_f4 __attribute__((ext_vector_type(4)));
typedef int dgs_i4 __attribute__((ext_vector_type(4)));
#ifndef DGS_NT
#define DGS_NT 1
#endif
#if DGS_NT
#define DGS_NT_SUFFIX " nt"
#else
#define DGS_NT_SUFFIX ""
#endif
// Streaming accesses (col/val/rowptr reads, C/E writes are touched once): non-temporal so that they do not
// displace rows of the dense operand from the 4 MiB XCD L2, which is the only reuse the kernel has.
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
#if DGS_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
// Gather of a dense-operand row slice; DGS_B_NT=1 makes it non-temporal (experiment: scan-resistant insertion?).
#ifndef DGS_B_NT
#define DGS_B_NT 0
#endif
template <int V>
__device__ __forceinline__ void load_vec_gather(const float *p, float (&o)[V]) {
#if DGS_B_NT
  if constexpr (V == 4) {
    const dgs_f4 t = __builtin_nontemporal_load(reinterpret_cast<const dgs_f4 *>(p));
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  } else {
    o[0] = __builtin_nontemporal_load(p);
  }
#else
  load_vec<V>(p, o);
#endif
}
typedef float dgs_f2 __attribute__((ext_vector_type(2)));
// SDDMM's row operand D1: every row is used by its own nnz within one short window and never again, so reading it
// non-temporally (DGS_SD_D1_NT = 1) should keep it from pushing rows of the GATHERED operand out of L2.  Measured and left
// off: the per-nnz re-loads of the row then stop hitting L1 - 1M graph 455 -> 514 us, products-shaped 2227 -> 2306 us.
#ifndef DGS_SD_D1_NT
#define DGS_SD_D1_NT 0
#endif
template <int V>
__device__ __forceinline__ void load_vec_rowop(const float *p, float (&o)[V]) {
#if DGS_SD_D1_NT
  if constexpr (V == 4) {
    const dgs_f4 t = __builtin_nontemporal_load(reinterpret_cast<const dgs_f4 *>(p));
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  } else if constexpr (V == 2) {
    const dgs_f2 t = __builtin_nontemporal_load(reinterpret_cast<const dgs_f2 *>(p));
    o[0] = t[0]; o[1] = t[1];
  } else {
    o[0] = __builtin_nontemporal_load(p);
  }
#else
  load_vec<V>(p, o);
#endif
}
template <int V>
__device__ __forceinline__ void store_vec_stream(float *p, const float (&o)[V]) {
#if DGS_NT
  if constexpr (V == 4) {
    dgs_f4 d = {o[0], o[1], o[2], o[3]};
    __builtin_nontemporal_store(d, reinterpret_cast<dgs_f4 *>(p));
  } else if constexpr (V == 2) {
    dgs_f2 d = {o[0], o[1]};
    __builtin_nontemporal_store(d, reinterpret_cast<dgs_f2 *>(p));
  } else {
    __builtin_nontemporal_store(o[0], p);
  }
#else
  store_vec<V>(p, o);
#endif
}
template <int V>
__device__ __forceinline__ void store_vec_stream(int *p, const int (&o)[V]) {
#if DGS_NT
  if constexpr (V == 4) {
    dgs_i4 d = {o[0], o[1], o[2], o[3]};
    __builtin_nontemporal_store(d, reinterpret_cast<dgs_i4 *>(p));
  } else {
    __builtin_nontemporal_store(o[0], p);
  }
#else
  store_vec<V>(p, o);
#endif
}
#ifndef DGS_HOST_EMU
template <int V>
__device__ __forceinline__ void store_vec_hidden(float *p, const float (&o)[V]) {
  if constexpr (V == 4) {
    dgs_f4 d = {o[0], o[1], o[2], o[3]};
    asm volatile("global_store_dwordx4 %0, %1, off" DGS_NT_SUFFIX "\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  } else {
    asm volatile("global_store_dword %0, %1, off" DGS_NT_SUFFIX "\n\ts_nop 1" ::"v"(p), "v"(o[0]) : "memory");
  }
}
template <int V>
__device__ __forceinline__ void store_vec_hidden(int *p, const int (&o)[V]) {
  if constexpr (V == 4) {
    dgs_i4 d = {o[0], o[1], o[2], o[3]};
    asm volatile("global_store_dwordx4 %0, %1, off" DGS_NT_SUFFIX "\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  } else {
    asm volatile("global_store_dword %0, %1, off" DGS_NT_SUFFIX "\n\ts_nop 1" ::"v"(p), "v"(o[0]) : "memory");
  }
}
// Drain of this wave's vector-memory queue in front of a cross-workgroup publish (the in-kernel fold's arrival counter): written as
// inline asm because hipcc's waitcnt pass never sees - hence never weakens or drops - it (MI355X guide, inter-workgroup visibility,
// "compiler hazard"); the memory clobber pins the stores before it and the atomic behind it.
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else  // host emulation (tests/emu): a store is a store; accesses complete at once unless the relaxed-memory mode is on
__device__ __forceinline__ void drain_vmem() { ::emu::mem_drain(); }
template <int V>
__device__ __forceinline__ void store_vec_hidden(float *p, const float (&o)[V]) { store_vec<V>(p, o); }
template <int V>
__device__ __forceinline__ void store_vec_hidden(int *p, const int (&o)[V]) { store_vec<V>(p, o); }
#endif

// Feature-dimension mapping shared by all row-group kernels: a row is covered by G lanes x V floats.
struct FeatMap {
  int G;      // lanes per row group (power of two, <= 64)
  int V;      // floats per lane (4 when N % 4 == 0 and bases are 16-B aligned, else 1)
  int tiles;  // ceil(N / (G*V)) -> gridDim.y
};
inline FeatMap feat_map(int64_t N, bool aligned16) {
  FeatMap m;
  m.V = (N % 4 == 0 && aligned16) ? 4 : 1;
  int64_t lanes = (N + m.V - 1) / m.V;
  int G = 1;
  while (G < lanes && G < 64) G <<= 1;
  m.G = G;
  m.tiles = (int)((lanes + G - 1) / G);
  return m;
}
inline bool is_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch() { return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ELAUNCH; }

// Tuning overrides (tests and experiments; every one has a compiled-in default at its point of use).  The environment is read
// ONCE per process into an immutable snapshot (misc.hip: std::call_once), so a launch costs no getenv() and host threads can
// launch concurrently; dgs_reload_tuning() publishes a fresh snapshot (tests that flip DGS_PANEL* between calls).
constexpr int kTuneUnset = INT_MIN;
struct Tuning {
  int panel, panel_kb, panel_lead, panel_tlong, min_waves, nbu, strict_mid, strict_hub, strict_nbu, sddmm_fused;
  int plan_tslice, plan_unit, plan_ch, plan_nocut, hub_chain, fold;
};
const Tuning &tuning();
inline int tune(int v, int dflt) { return v == kTuneUnset ? dflt : v; }
int cu_count();  // compute units of the CURRENT device (cached per device)

}  // namespace dgs
