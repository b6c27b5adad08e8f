// ref_shim.cpp -- C-linkage wrappers around the REFERENCE's own CPU loops, compiled from the
// sources where they lie under /root/reference (nothing is copied into this repo):
//   spmm_reference_host<int,float>   example/util/sp_util.hpp:63-84
//   sddmm_reference_host<int,float>  example/util/sp_util.hpp:88-112
//   read_mtx_file                    example/util/sp_util.hpp:171-251
// sp_util.hpp includes <cuda_runtime_api.h> only for an unused cudaEvent timer; the header that
// ships inside this image (triton/backends/nvidia/include) satisfies it - see oracle/Makefile.
// Test infrastructure only: output goes to oracle/_ref/ (git-ignored, travels with gpurun).
#include "sp_util.hpp"

extern "C" {
void ref_spmm_sum(int M, int N, int K, const int *indptr, const int *indices, const float *values,
                  const float *B, float *C) {
  spmm_reference_host<int, float>(M, N, K, indptr, indices, values, B, C);
}
void ref_sddmm(int M, int N, int K, int nnz, const int *indptr, const int *indices, const float *A,
               const float *B, float *C) {
  sddmm_reference_host<int, float>(M, N, K, nnz, indptr, indices, A, B, C);
}
// Two-call protocol: first call with indptr==NULL returns sizes, second fills caller buffers.
int ref_read_mtx(const char *path, int *nrow, int *ncol, int *nnz, int *indptr, int *indices) {
  static std::vector<int> ip, ix;
  static int r, c, z;
  if (!indptr) {
    read_mtx_file(path, r, c, z, ip, ix);
    *nrow = r; *ncol = c; *nnz = z;
    return 0;
  }
  std::copy(ip.begin(), ip.end(), indptr);
  std::copy(ix.begin(), ix.end(), indices);
  return 0;
}
}
