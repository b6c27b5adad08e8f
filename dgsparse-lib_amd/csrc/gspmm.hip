// gspmm.hip -- generalised SpMM  out[r,:] = reduce_p compute(val[p], B[col[p],:])  for the non-multiplicative compute
// ops of the reference's gspmm-fp demo module (src/gspmm-fp/gspmm.{h,cu,cc}: GSpMM_u_e / GSpMM_u with
// COMPUTEOP { ADD a+b, SUB b-a, MUL a*b, DIV b/a } and REDUCEOP { SUM, MAX, MIN, MEAN }).
// compute == MUL is the hot path and goes to the full SpMM schedule (spmm_impl.h); ADD/SUB/DIV take this compact
// kernel: one group of G lanes x V features per row, sequential CSR order (so results equal the reference's simple
// kernel, weightedSimpleSPMMKernel gspmm.cu:212-245, operation for operation), 4 gathers in flight per lane.  Long
// rows are NOT split here (the demo module is not part of the measured path).
#include "dgs_common.h"

namespace dgs {

enum { kAdd = 0, kSub = 1, kMul = 2, kDiv = 3 };  // src/gspmm-fp/gspmm.h:16

template <int COMPUTE>
__device__ __forceinline__ float gcompute(float a, float b) {
  if constexpr (COMPUTE == kAdd) return a + b;
  if constexpr (COMPUTE == kSub) return b - a;
  if constexpr (COMPUTE == kDiv) return b / a;
  return a * b;
}
template <int OP>
__device__ __forceinline__ float greduce(float acc, float t) {
  if constexpr (OP == DGS_MAX) return (acc < t) ? t : acc;
  if constexpr (OP == DGS_MIN) return (acc < t) ? acc : t;
  return acc + t;
}

template <int G, int V, int OP, int COMPUTE>
__global__ __launch_bounds__(kBlock) void gspmm_rowgroup(int M, int N, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col, const float *__restrict__ val,
                                                         const float *__restrict__ B, float *__restrict__ C) {
  constexpr int ROWS = kBlock / G;
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * ROWS + g;
  const int f0 = (blockIdx.y * G + l) * V;
  if (row >= M || f0 >= N) return;
  const int s = rowptr[row], e = rowptr[row + 1];
  float acc[V];
#pragma unroll
  for (int v = 0; v < V; v++) acc[v] = (e > s) ? reduce_init<OP>() : 0.0f;
  constexpr int U = 4;
  int p = s;
  for (; p + U <= e; p += U) {
    int c[U];
    float w[U], x[U][V];
#pragma unroll
    for (int u = 0; u < U; u++) {
      c[u] = col[p + u];
      w[u] = val ? val[p + u] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) load_vec<V>(B + (int64_t)c[u] * N + f0, x[u]);
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int v = 0; v < V; v++) acc[v] = greduce<OP>(acc[v], gcompute<COMPUTE>(w[u], x[u][v]));
  }
  for (; p < e; p++) {
    const int c = col[p];
    const float w = val ? val[p] : 1.0f;
    float x[V];
    load_vec<V>(B + (int64_t)c * N + f0, x);
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = greduce<OP>(acc[v], gcompute<COMPUTE>(w, x[v]));
  }
  if constexpr (OP == DGS_MEAN) {
    if (e > s) {
      const float d = (float)(e - s);
#pragma unroll
      for (int v = 0; v < V; v++) acc[v] /= d;
    }
  }
  store_vec<V>(C + row * N + f0, acc);
}

template <int G, int V, int OP, int COMPUTE>
static int launch_g(int64_t M, int64_t N, int tiles, const int *rowptr, const int *col, const float *val, const float *B,
                    float *C, hipStream_t st) {
  const dim3 grid((unsigned)((M + (kBlock / G) - 1) / (kBlock / G)), (unsigned)tiles);
  hipLaunchKernelGGL((gspmm_rowgroup<G, V, OP, COMPUTE>), grid, dim3(kBlock), 0, st, (int)M, (int)N, rowptr, col, val, B, C);
  return check_launch();
}
template <int G, int V, int OP>
static int disp_c(int cop, int64_t M, int64_t N, int tiles, const int *rowptr, const int *col, const float *val,
                  const float *B, float *C, hipStream_t st) {
  switch (cop) {
    case kAdd: return launch_g<G, V, OP, kAdd>(M, N, tiles, rowptr, col, val, B, C, st);
    case kSub: return launch_g<G, V, OP, kSub>(M, N, tiles, rowptr, col, val, B, C, st);
    case kDiv: return launch_g<G, V, OP, kDiv>(M, N, tiles, rowptr, col, val, B, C, st);
    case kMul: return launch_g<G, V, OP, kMul>(M, N, tiles, rowptr, col, val, B, C, st);
  }
  return DGS_EINVAL;
}
template <int G, int V>
static int disp_r(int rop, int cop, int64_t M, int64_t N, int tiles, const int *rowptr, const int *col, const float *val,
                  const float *B, float *C, hipStream_t st) {
  switch (rop) {
    case DGS_SUM: return disp_c<G, V, DGS_SUM>(cop, M, N, tiles, rowptr, col, val, B, C, st);
    case DGS_MAX: return disp_c<G, V, DGS_MAX>(cop, M, N, tiles, rowptr, col, val, B, C, st);
    case DGS_MIN: return disp_c<G, V, DGS_MIN>(cop, M, N, tiles, rowptr, col, val, B, C, st);
    case DGS_MEAN: return disp_c<G, V, DGS_MEAN>(cop, M, N, tiles, rowptr, col, val, B, C, st);
  }
  return DGS_EINVAL;
}
template <int V>
static int disp_g(int G, int rop, int cop, int64_t M, int64_t N, int tiles, const int *rowptr, const int *col,
                  const float *val, const float *B, float *C, hipStream_t st) {
  switch (G) {
    case 1: return disp_r<1, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
    case 2: return disp_r<2, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
    case 4: return disp_r<4, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
    case 8: return disp_r<8, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
    case 16: return disp_r<16, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
    case 32: return disp_r<32, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
    case 64: return disp_r<64, V>(rop, cop, M, N, tiles, rowptr, col, val, B, C, st);
  }
  return DGS_EINVAL;
}

}  // namespace dgs

using namespace dgs;

extern "C" size_t dgs_gspmm_csr_workspace_bytes(int reduce_op, int compute_op, int64_t M, int64_t N, int64_t nnz) {
  return (compute_op == kMul && (reduce_op == DGS_SUM || reduce_op == DGS_MEAN))
             ? dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz) : 0;
}

extern "C" int dgs_gspmm_csr_f32(int reduce_op, int compute_op, int64_t M, int64_t K, int64_t N, int64_t nnz,
                                 const int32_t *rowptr, const int32_t *col, const float *val, const float *B, float *C,
                                 void *workspace, size_t workspace_bytes, dgsStream_t stream) {
  if (reduce_op < DGS_SUM || reduce_op > DGS_MEAN || compute_op < kAdd || compute_op > kDiv) return DGS_EINVAL;
  if (M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (M == 0 || N == 0) return DGS_OK;
  if (!rowptr || !C || (nnz > 0 && (!col || !B))) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // MUL with sum/mean is the hot path: full schedule.  MUL with max/min would need the arg-id scratch of
  // dgs_spmm_csr_f32, which this entry does not have, so it takes the compact kernel like ADD/SUB/DIV.
  if (compute_op == kMul && (reduce_op == DGS_SUM || reduce_op == DGS_MEAN))
    return dgs_spmm_csr_f32(reduce_op, M, K, N, nnz, rowptr, col, val, B, C, nullptr, 0, workspace, workspace_bytes, stream);
  const FeatMap fm = feat_map(N, is_aligned16(B) && is_aligned16(C));
  if (fm.V == 4) return disp_g<4>(fm.G, reduce_op, compute_op, M, N, fm.tiles, rowptr, col, val, B, C, st);
  return disp_g<1>(fm.G, reduce_op, compute_op, M, N, fm.tiles, rowptr, col, val, B, C, st);
}
