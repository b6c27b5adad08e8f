// dist_merge.hip -- the merge step of the overlapped MIN of the multi-GPU path (dgsparse/dist.py), and the non-finite
// detector that guards it.  No reference counterpart (the reference is single-GPU); the arithmetic being
// reproduced is algorithm 0's sequential chain (reference include/cuda/spmm_cuda.cuh:27-47 with the MIN macro of
// include/gspmm.h:16-17 taken literally).
//
// A shard row with sorted columns reads, in CSR order,  [halo columns of lower ranks | columns this rank owns | halo
// columns of higher ranks].  MIN keeps the LATER operand's bits on a tie (it matters for -0.0 / +0.0) while E names the
// FIRST minimum, so the pair (value, arg) of a segment can only be appended to what precedes it, never inserted in front:
// the halo product is therefore taken over a matrix with every row split in two (row 2r = the lower-rank halo entries of
// shard row rowmap[r], row 2r + 1 = the higher-rank ones) and this kernel folds  lower -> local -> higher  with the very
// reduce step of the kernels.  That is exact as long as no product is NaN: MIN(acc, NaN) = NaN and MIN(NaN, t) = t, so a
// NaN inside a segment makes the chain forget everything before it, which no (value, arg) pair can express.  A product
// can only be NaN if a feature or an edge value is NaN or infinite; dgs_nonfinite_flag_f32 looks for those, and when the
// flag is up the kernel recomputes its rows sequentially over the whole [local | halo] shard instead of merging.
// The schedule dgsparse.dist actually uses folds the two halo halves into (C, E) with the accumulating min kernels instead
// (dgs_spmm_csr_acc_min_f32, lower half then higher half: no (Ch, Eh) round trip through memory) and calls this kernel
// with rowptr2 == NULL: nothing to merge, only the redo when the flag is up.
#include "dgs_common.h"

namespace dgs {

__global__ __launch_bounds__(kBlock) void nonfinite_kernel(int64_t n, const float *__restrict__ x, int *__restrict__ flag,
                                                           bool vec) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned bad = 0;
  if (vec) {
    const int64_t nq = n >> 2;
    for (int64_t q = t; q < nq; q += stride) {
      const dgs_i4 v = __builtin_nontemporal_load(reinterpret_cast<const dgs_i4 *>(x) + q);
#pragma unroll
      for (int k = 0; k < 4; k++) bad |= (unsigned)(((unsigned)v[k] & 0x7f800000u) == 0x7f800000u);
    }
    for (int64_t i = (nq << 2) + t; i < n; i += stride) bad |= (unsigned)((__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u);
  } else {
    for (int64_t i = t; i < n; i += stride) bad |= (unsigned)((__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u);
  }
  if (bad) atomicOr(flag, 1);
}

// One lane per V features of one halo row pair.
template <int V, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void min_merge_kernel(int64_t R, int N, const int *__restrict__ rowmap,
                                                           const int *__restrict__ rowptr2, const float *__restrict__ Ch,
                                                           const int *__restrict__ Eh, int col_off,
                                                           const int *__restrict__ loc_rowptr, float *__restrict__ C,
                                                           int *__restrict__ E, const int *__restrict__ nonfinite,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const float *__restrict__ val, const float *__restrict__ B) {
  const bool redo = nonfinite && *nonfinite;  // uniform
  if (!rowptr2 && !redo) return;              // redo-only call (the accumulating min kernels did the merge) and nothing to redo
  const int lanes = N / V;
  // grid-stride: the launch is capped so that the redo-only call with the flag down costs a few microseconds
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < R * lanes; t += (int64_t)gridDim.x * kBlock) {
  const int64_t r = t / lanes;
  const int f = (int)(t % lanes) * V;
  const int64_t row = rowmap ? rowmap[r] : r;
  float acc[V];
  int e[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    acc[v] = reduce_init<DGS_MIN>();
    e[v] = -1;
  }
  if (redo) {  // the chain itself, over the whole shard row
    for (int p = rowptr[row], pe = rowptr[row + 1]; p < pe; p++) {
      const int c = col[p];
      const float w = HAS_VAL ? val[p] : 1.0f;
      float x[V];
      load_vec<V>(B + (int64_t)c * N + f, x);
#pragma unroll
      for (int v = 0; v < V; v++) reduce_step<DGS_MIN>(acc[v], e[v], w, x[v], c);
    }
  } else {
    float b[V];
    int eb[V];
    if (rowptr2[2 * r] < rowptr2[2 * r + 1]) {  // halo entries that precede the local columns
      load_vec<V>(Ch + (2 * r) * N + f, b);
      load_vec<V>(Eh + (2 * r) * N + f, eb);
#pragma unroll
      for (int v = 0; v < V; v++) reduce_step<DGS_MIN>(acc[v], e[v], 1.0f, b[v], eb[v] + col_off);
    }
    if (loc_rowptr[row] < loc_rowptr[row + 1]) {  // what the local product left in (C, E)
      load_vec<V>(C + row * N + f, b);
      load_vec<V>(E + row * N + f, eb);
#pragma unroll
      for (int v = 0; v < V; v++) reduce_step<DGS_MIN>(acc[v], e[v], 1.0f, b[v], eb[v]);
    }
    if (rowptr2[2 * r + 1] < rowptr2[2 * r + 2]) {  // halo entries that follow them
      load_vec<V>(Ch + (2 * r + 1) * N + f, b);
      load_vec<V>(Eh + (2 * r + 1) * N + f, eb);
#pragma unroll
      for (int v = 0; v < V; v++) reduce_step<DGS_MIN>(acc[v], e[v], 1.0f, b[v], eb[v] + col_off);
    }
  }
  store_vec<V>(C + row * N + f, acc);
  store_vec<V>(E + row * N + f, e);
  }
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_nonfinite_flag_f32(int64_t n, const float *x, int32_t *flag, dgsStream_t stream) {
  if (n < 0) return DGS_EINVAL;
  if (n == 0) return DGS_OK;
  if (!x || !flag) return DGS_EINVAL;
  const bool vec = is_aligned16(x);
  const int64_t threads = vec ? (n + 3) / 4 : n;
  int64_t blocks = (threads + kBlock - 1) / kBlock;
  const int64_t cap = 2048;  // 8 workgroups per CU of a 256-CU part; grid-stride beyond that
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), n, x, flag,
                     vec);
  return check_launch();
}

extern "C" int dgs_spmm_min_merge_f32(int64_t R, int64_t N, const int32_t *rowmap, const int32_t *rowptr2, const float *Ch,
                                      const int32_t *Eh, int32_t col_off, const int32_t *loc_rowptr, float *C, int32_t *E,
                                      const int32_t *nonfinite, const int32_t *rowptr, const int32_t *col,
                                      const float *val, const float *B, dgsStream_t stream) {
  if (R < 0 || N < 0 || N >= INT32_MAX || 2 * R >= INT32_MAX) return DGS_EINVAL;
  if (R == 0 || N == 0) return DGS_OK;
  if (!C || !E) return DGS_EINVAL;
  if (rowptr2 ? (!Ch || !Eh || !loc_rowptr) : !nonfinite) return DGS_EINVAL;  // rowptr2 == NULL: redo-only call
  if (nonfinite && (!rowptr || !col || !B)) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool v4 = (N % 4 == 0) && (!rowptr2 || (is_aligned16(Ch) && is_aligned16(Eh))) && is_aligned16(C) && is_aligned16(E) &&
                  (!nonfinite || is_aligned16(B));
  const int64_t lanes = v4 ? N / 4 : N;
  int64_t blocks = (R * lanes + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;  // 16 workgroups per CU of a 256-CU part; grid-stride beyond that
  const dim3 grid((unsigned)blocks);
#define DGS_MERGE(V, HV)                                                                                                \
  hipLaunchKernelGGL((min_merge_kernel<V, HV>), grid, dim3(kBlock), 0, st, R, (int)N, rowmap, rowptr2, Ch, Eh, col_off, \
                     loc_rowptr, C, E, nonfinite, rowptr, col, val, B)
  if (v4) {
    if (val) DGS_MERGE(4, true);
    else DGS_MERGE(4, false);
  } else {
    if (val) DGS_MERGE(1, true);
    else DGS_MERGE(1, false);
  }
#undef DGS_MERGE
  return check_launch();
}
