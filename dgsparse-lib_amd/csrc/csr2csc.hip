// csr2csc.hip -- stable CSR -> CSC transpose with an exact integer permutation, for gfx950.
// Replaces csr2cscKernel (reference include/cuda/csr2csc.cuh:8-26: cusparseCsr2cscEx2 + a per-call
// cusparseCreate/cudaMalloc) and the float-encoded permutation of dgsparse/storage.py:164-169 (exact only
// for nnz < 2^24).  Pipeline, all on the caller's stream, no allocation (workspace from the caller):
//   1. iota(pos)                         pos[p] = p
//   2. stable LSD radix sort of (col[p], pos[p]) on the column id  -> perm = sorted pos   (rocPRIM)
//      stable => inside a column entries keep CSR order = increasing row, then CSR position
//   3. colptr[j] = lower_bound(sorted_col, j)          (one thread per column)
//   4. row[q] = upper_bound(rowptr, perm[q]) - 1 ; cscval[q] = val[perm[q]]
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "dgs_common.h"

namespace dgs {

__global__ void iota_kernel(int n, int *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

__global__ void colptr_kernel(int Kcols, int nnz, const int *__restrict__ sorted_col, int *__restrict__ colptr) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > Kcols) return;
  int lo = 0, hi = nnz;  // first q with sorted_col[q] >= j
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sorted_col[mid] < j) lo = mid + 1; else hi = mid;
  }
  colptr[j] = lo;
}

__global__ void fill_kernel(int M, int nnz, const int *__restrict__ rowptr, const int *__restrict__ perm,
                            const float *__restrict__ val, int *__restrict__ row, float *__restrict__ cscval) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nnz) return;
  const int p = perm[q];
  if (row) {
    int lo = 0, hi = M;  // last r with rowptr[r] <= p  (empty rows are skipped by the <=)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (rowptr[mid] <= p) lo = mid; else hi = mid - 1;
    }
    row[q] = lo;
  }
  if (cscval) cscval[q] = val[p];
}

static int key_bits(int64_t Kcols) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) < Kcols) b++;
  return b;
}

struct Csr2cscLayout {
  size_t off_pos, off_keys, off_perm, off_sort, sort_bytes, total;
};

static Csr2cscLayout layout(int64_t Kcols, int64_t nnz) {
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  Csr2cscLayout L;
  size_t sb = 0;
  int *np = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sb, np, np, np, np, (size_t)nnz, 0, key_bits(Kcols), nullptr, false);
  L.sort_bytes = sb;
  L.off_pos = 0;
  L.off_keys = up((size_t)nnz * 4);
  L.off_perm = L.off_keys + up((size_t)nnz * 4);
  L.off_sort = L.off_perm + up((size_t)nnz * 4);
  L.total = L.off_sort + up(sb) + 256;
  return L;
}

}  // namespace dgs

using namespace dgs;

extern "C" size_t dgs_csr2csc_workspace_bytes(int64_t M, int64_t Kcols, int64_t nnz) {
  (void)M;
  if (nnz <= 0 || Kcols < 0) return 256;
  return layout(Kcols, nnz).total;
}

extern "C" int dgs_csr2csc_i32(int64_t M, int64_t Kcols, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                               const float *val, int32_t *colptr, int32_t *row, float *cscval, int32_t *perm,
                               void *workspace, size_t workspace_bytes, dgsStream_t stream) {
  if (M < 0 || Kcols < 0 || nnz < 0 || !colptr) return DGS_EINVAL;
  if (M >= INT32_MAX || Kcols >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (cscval && !val) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nnz == 0) {
    return hipMemsetAsync(colptr, 0, (size_t)(Kcols + 1) * 4, st) == hipSuccess ? DGS_OK : DGS_ELAUNCH;
  }
  if (!rowptr || !col || !workspace) return DGS_EINVAL;
  const Csr2cscLayout L = layout(Kcols, nnz);
  if (workspace_bytes < L.total) return DGS_EWORKSPACE;
  char *ws = static_cast<char *>(workspace);
  int *pos = reinterpret_cast<int *>(ws + L.off_pos);
  int *keys = reinterpret_cast<int *>(ws + L.off_keys);
  int *pout = perm ? perm : reinterpret_cast<int *>(ws + L.off_perm);
  void *sort_ws = ws + L.off_sort;
  const int T = 256;
  hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((nnz + T - 1) / T)), dim3(T), 0, st, (int)nnz, pos);
  size_t sb = L.sort_bytes;
  if (rocprim::radix_sort_pairs(sort_ws, sb, col, keys, pos, pout, (size_t)nnz, 0, key_bits(Kcols), st, false) !=
      hipSuccess)
    return DGS_ELAUNCH;
  hipLaunchKernelGGL(colptr_kernel, dim3((unsigned)((Kcols + 1 + T - 1) / T)), dim3(T), 0, st, (int)Kcols, (int)nnz,
                     keys, colptr);
  if (row || cscval)
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nnz + T - 1) / T)), dim3(T), 0, st, (int)M, (int)nnz, rowptr,
                       pout, val, row, cscval);
  return check_launch();
}
