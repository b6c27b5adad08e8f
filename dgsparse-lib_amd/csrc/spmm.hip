// spmm.hip -- CSR SpMM (sum / max / min / mean + arg ids) for gfx950, row-group schedule.
//
// Replaces csrspmm_seqreduce_rowbalance_kernel (reference include/cuda/spmm_cuda.cuh:10-55), which maps
// ONE THREAD to one (row, feature) and re-reads col/val through L1 once per feature.  Here a row is owned
// by a group of G lanes of a wave64, each lane holding V=4 consecutive features, so one B row is fetched
// by G coalesced dwordx4 loads (N=64: 16 lanes x 16 B = one 256-B row; a wave gathers 4 rows per load
// instruction) and every accumulator still sees its products in CSR order - i.e. the result is the
// algorithm-0 result bit for bit (fmaf chain for sum/mean, single-rounded products for max/min).
//
// Kernel in this file:
//   spmm_rowgroup_seq<G,V,OP,HAS_VAL>   one group per row, sequential over the row's nnz, 4-deep unroll so
//                                       that 4 independent B-row gathers are in flight per group.
#include "dgs_common.h"

namespace dgs {

template <int G, int V, int OP, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void spmm_rowgroup_seq(int M, int N, const int *__restrict__ rowptr,
                                                            const int *__restrict__ col,
                                                            const float *__restrict__ val,
                                                            const float *__restrict__ B, float *__restrict__ C,
                                                            int *__restrict__ E) {
  constexpr int ROWS = kBlock / G;
  constexpr bool ARG = (OP == DGS_MAX || OP == DGS_MIN);
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * ROWS + g;
  const int f0 = (blockIdx.y * G + l) * V;
  if (row >= M || f0 >= N) return;
  const int s = rowptr[row], e = rowptr[row + 1];

  float acc[V];
  int ei[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    acc[v] = reduce_init<OP>();
    ei[v] = -1;
  }
  const float *Bf = B + f0;
  constexpr int U = 4;
  int p = s;
  for (; p + U <= e; p += U) {
    int c[U];
    float w[U];
    float x[U][V];
#pragma unroll
    for (int u = 0; u < U; u++) {
      c[u] = col[p + u];
      w[u] = HAS_VAL ? val[p + u] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) load_vec<V>(Bf + (int64_t)c[u] * N, x[u]);
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int v = 0; v < V; v++) reduce_step<OP>(acc[v], ei[v], w[u], x[u][v], c[u]);
  }
  for (; p < e; p++) {
    const int c = col[p];
    const float w = HAS_VAL ? val[p] : 1.0f;
    float x[V];
    load_vec<V>(Bf + (int64_t)c * N, x);
#pragma unroll
    for (int v = 0; v < V; v++) reduce_step<OP>(acc[v], ei[v], w, x[v], c);
  }
  if (e > s) {
    if constexpr (OP == DGS_MEAN) {
      const float d = (float)(e - s);
#pragma unroll
      for (int v = 0; v < V; v++) acc[v] /= d;
    }
  } else {  // empty row: 0 and E = -1 (spmm_cuda.cuh:49-51)
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = 0.0f;
  }
  store_vec<V>(C + row * N + f0, acc);
  if constexpr (ARG) store_vec<V>(E + row * N + f0, ei);
}

template <int G, int V, int OP>
static int launch_seq(int64_t M, int64_t N, const int *rowptr, const int *col, const float *val, const float *B,
                      float *C, int *E, int tiles, hipStream_t st) {
  const dim3 grid((unsigned)((M + (kBlock / G) - 1) / (kBlock / G)), (unsigned)tiles);
  if (val)
    hipLaunchKernelGGL((spmm_rowgroup_seq<G, V, OP, true>), grid, dim3(kBlock), 0, st, (int)M, (int)N, rowptr, col,
                       val, B, C, E);
  else
    hipLaunchKernelGGL((spmm_rowgroup_seq<G, V, OP, false>), grid, dim3(kBlock), 0, st, (int)M, (int)N, rowptr, col,
                       val, B, C, E);
  return check_launch();
}

template <int G, int V>
static int dispatch_op(int op, int64_t M, int64_t N, const int *rowptr, const int *col, const float *val,
                       const float *B, float *C, int *E, int tiles, hipStream_t st) {
  switch (op) {
    case DGS_SUM:
      return launch_seq<G, V, DGS_SUM>(M, N, rowptr, col, val, B, C, E, tiles, st);
    case DGS_MAX:
      return launch_seq<G, V, DGS_MAX>(M, N, rowptr, col, val, B, C, E, tiles, st);
    case DGS_MIN:
      return launch_seq<G, V, DGS_MIN>(M, N, rowptr, col, val, B, C, E, tiles, st);
    case DGS_MEAN:
      return launch_seq<G, V, DGS_MEAN>(M, N, rowptr, col, val, B, C, E, tiles, st);
  }
  return DGS_EINVAL;
}

template <int V>
static int dispatch_g(int G, int op, int64_t M, int64_t N, const int *rowptr, const int *col, const float *val,
                      const float *B, float *C, int *E, int tiles, hipStream_t st) {
  switch (G) {
    case 1:
      return dispatch_op<1, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
    case 2:
      return dispatch_op<2, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
    case 4:
      return dispatch_op<4, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
    case 8:
      return dispatch_op<8, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
    case 16:
      return dispatch_op<16, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
    case 32:
      return dispatch_op<32, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
    case 64:
      return dispatch_op<64, V>(op, M, N, rowptr, col, val, B, C, E, tiles, st);
  }
  return DGS_EINVAL;
}

}  // namespace dgs

using namespace dgs;

extern "C" size_t dgs_spmm_csr_workspace_bytes(int reduce_op, int64_t M, int64_t N, int64_t nnz) {
  (void)reduce_op;
  (void)M;
  (void)N;
  (void)nnz;
  return 0;
}

extern "C" int dgs_spmm_csr_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                int algorithm, void *workspace, size_t workspace_bytes, dgsStream_t stream) {
  (void)algorithm;  // every algorithm id returns the algorithm-0 result
  (void)workspace;
  (void)workspace_bytes;
  if (reduce_op < DGS_SUM || reduce_op > DGS_MEAN || M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  const bool arg = (reduce_op == DGS_MAX || reduce_op == DGS_MIN);
  if (M == 0 || N == 0) return DGS_OK;
  if (!rowptr || !C || (nnz > 0 && (!col || !B)) || (arg && !E)) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E && !arg) {  // the reference leaves E = -1 for sum/mean (Eidx is never updated)
    if (hipMemsetAsync(E, 0xFF, (size_t)M * N * sizeof(int32_t), st) != hipSuccess) return DGS_ELAUNCH;
  }
  const bool al = is_aligned16(B) && is_aligned16(C) && (!arg || is_aligned16(E));
  const FeatMap fm = feat_map(N, al);
  if (fm.V == 4) return dispatch_g<4>(fm.G, reduce_op, M, N, rowptr, col, val, B, C, E, fm.tiles, st);
  return dispatch_g<1>(fm.G, reduce_op, M, N, rowptr, col, val, B, C, E, fm.tiles, st);
}
