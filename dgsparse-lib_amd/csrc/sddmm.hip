// sddmm.hip -- CSR SDDMM (+ MEAN scaling, + arg mask) for gfx950.
//   out[e] = sum_k D1[row(e),k] * D2[col(e),k]
// Replaces sddmmCSR{2,1}Scale<REDUCE> / sddmmCSR1Scale_with_mask (reference include/cuda/sddmm_cuda.cuh:
// 222-401, 403-507), which are edge-balanced: 4 edges per 32-lane warp slice, a binary search over rowptr per
// edge (findRow, cuda_util.cuh:150-166) and a re-read of the D1 row for every edge.  Here the schedule is
// row-group: a group of G lanes x V features owns a row, keeps its D1 slice (and the E slice for the mask
// variant) in registers for the whole row, streams the row's columns, reduces each dot product across the
// group with a log2(G) xor-butterfly, and writes G results at a time with one coalesced store.
#include "dgs_common.h"

namespace dgs {

// xor-butterfly over the G lanes of a group (G power of two <= 64): every lane ends with the total.
template <int G>
__device__ __forceinline__ float group_allreduce(float x) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

template <int G, int V, bool MEAN, bool MASK>
__global__ __launch_bounds__(kBlock) void sddmm_rowgroup(int M, int F, int tiles, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col,
                                                         const float *__restrict__ D1,
                                                         const float *__restrict__ D2, const int *__restrict__ E,
                                                         float *__restrict__ out) {
  constexpr int ROWS = kBlock / G;
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  int64_t row = (int64_t)blockIdx.x * ROWS + g;
  // Whole groups leave together (row is group-uniform); partial waves keep shuffles inside live groups.
  if (row >= M) return;
  const int s = rowptr[row], e = rowptr[row + 1];
  if (e <= s) return;
  const float scale = MEAN ? 1.0f / 1.0f : 1.0f;
  (void)scale;
  const float deg = (float)(e - s);
  const int f0 = l * V;
  const bool live0 = f0 < F;
  float a0[V];
  int m0[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    a0[v] = 0.0f;
    m0[v] = -2;
  }
  if (live0) {
    load_vec<V>(D1 + row * F + f0, a0);
    if constexpr (MASK) load_vec<V>(E + row * F + f0, m0);
  }
  for (int base = s; base < e; base += G) {
    float keep = 0.0f;
    const int cnt = min(G, e - base);
    for (int j = 0; j < cnt; j++) {
      const int c = col[base + j];
      float part = 0.0f;
      if (live0) {
        float b[V];
        load_vec<V>(D2 + (int64_t)c * F + f0, b);
#pragma unroll
        for (int v = 0; v < V; v++) {
          if constexpr (MASK) {
            if (m0[v] == c) part = __builtin_fmaf(a0[v], b[v], part);
          } else {
            part = __builtin_fmaf(a0[v], b[v], part);
          }
        }
      }
      for (int t = 1; t < tiles; t++) {  // F > G*V: further feature tiles, D1/E slices re-read (L1-hot)
        const int f = (t * G + l) * V;
        if (f < F) {
          float a[V], b[V];
          load_vec<V>(D1 + row * F + f, a);
          load_vec<V>(D2 + (int64_t)c * F + f, b);
          if constexpr (MASK) {
            int m[V];
            load_vec<V>(E + row * F + f, m);
#pragma unroll
            for (int v = 0; v < V; v++)
              if (m[v] == c) part = __builtin_fmaf(a[v], b[v], part);
          } else {
#pragma unroll
            for (int v = 0; v < V; v++) part = __builtin_fmaf(a[v], b[v], part);
          }
        }
      }
      const float tot = group_allreduce<G>(part);
      if (l == j) keep = tot;
    }
    if (l < cnt) {
      if constexpr (MEAN) keep /= deg;  // sddmm_cuda.cuh:266-272: divide by deg(row(e)) when deg > 0
      out[base + l] = keep;
    }
  }
}

template <int G, int V, bool MEAN, bool MASK>
static int launch_sddmm(int64_t M, int64_t F, int tiles, const int *rowptr, const int *col, const float *D1,
                        const float *D2, const int *E, float *out, hipStream_t st) {
  const dim3 grid((unsigned)((M + (kBlock / G) - 1) / (kBlock / G)));
  hipLaunchKernelGGL((sddmm_rowgroup<G, V, MEAN, MASK>), grid, dim3(kBlock), 0, st, (int)M, (int)F, tiles, rowptr,
                     col, D1, D2, E, out);
  return check_launch();
}

template <int V, bool MEAN, bool MASK>
static int dispatch_sddmm(int G, int64_t M, int64_t F, int tiles, const int *rowptr, const int *col, const float *D1,
                          const float *D2, const int *E, float *out, hipStream_t st) {
  switch (G) {
    case 1: return launch_sddmm<1, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
    case 2: return launch_sddmm<2, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
    case 4: return launch_sddmm<4, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
    case 8: return launch_sddmm<8, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
    case 16: return launch_sddmm<16, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
    case 32: return launch_sddmm<32, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
    case 64: return launch_sddmm<64, V, MEAN, MASK>(M, F, tiles, rowptr, col, D1, D2, E, out, st);
  }
  return DGS_EINVAL;
}

template <bool MEAN, bool MASK>
static int run_sddmm(int64_t M, int64_t K, int64_t F, int64_t nnz, const int *rowptr, const int *col, const float *D1,
                     const float *D2, const int *E, float *out, hipStream_t st) {
  if (M < 0 || K < 0 || F < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || F >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (M == 0 || nnz == 0) return DGS_OK;
  if (!rowptr || !col || !out || (F > 0 && (!D1 || !D2)) || (MASK && !E)) return DGS_EINVAL;
  if (F == 0) return hipMemsetAsync(out, 0, (size_t)nnz * sizeof(float), st) == hipSuccess ? DGS_OK : DGS_ELAUNCH;
  const bool al = is_aligned16(D1) && is_aligned16(D2) && (!MASK || is_aligned16(E));
  const FeatMap fm = feat_map(F, al);
  if (fm.V == 4) return dispatch_sddmm<4, MEAN, MASK>(fm.G, M, F, fm.tiles, rowptr, col, D1, D2, E, out, st);
  return dispatch_sddmm<1, MEAN, MASK>(fm.G, M, F, fm.tiles, rowptr, col, D1, D2, E, out, st);
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_sddmm_csr_f32(int reduce_op, int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                                 const int32_t *col, const float *D1, const float *D2, float *out,
                                 dgsStream_t stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (reduce_op == DGS_MEAN) return run_sddmm<true, false>(M, K, F, nnz, rowptr, col, D1, D2, nullptr, out, st);
  if (reduce_op == DGS_SUM) return run_sddmm<false, false>(M, K, F, nnz, rowptr, col, D1, D2, nullptr, out, st);
  return DGS_EINVAL;
}

extern "C" int dgs_sddmm_csr_mask_f32(int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                                      const int32_t *col, const float *D1, const float *D2, const int32_t *E,
                                      float *out, dgsStream_t stream) {
  return run_sddmm<false, true>(M, K, F, nnz, rowptr, col, D1, D2, E, out, static_cast<hipStream_t>(stream));
}
