// spmm_mask.hip -- masked SpMM: backward of max/min SpMM w.r.t. the dense operand, on the CSC arrays.
//   out[j,f] = sum_{p in [ptr[j],ptr[j+1])} [E[idx[p],f] == j] * val[p] * G[idx[p],f]
// Replaces csrspmm_seqreduce_rowbalance_with_mask_kernel (reference include/cuda/spmm_cuda.cuh:400-433);
// the formula is implemented, not that kernel's stale-`val_pre_red` / un-reset `res` behaviour.
// Same row-group mapping as spmm.hip: G lanes x V features per output row, sequential CSC order.
#include "dgs_common.h"

namespace dgs {

template <int G, int V, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void spmm_mask_rowgroup(int Mout, int N, const int *__restrict__ ptr,
                                                             const int *__restrict__ idx,
                                                             const float *__restrict__ val,
                                                             const float *__restrict__ Gr,
                                                             const int *__restrict__ E, float *__restrict__ out) {
  constexpr int ROWS = kBlock / G;
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  const int64_t j = (int64_t)blockIdx.x * ROWS + g;
  const int f0 = (blockIdx.y * G + l) * V;
  if (j >= Mout || f0 >= N) return;
  const int s = ptr[j], e = ptr[j + 1];
  float acc[V];
#pragma unroll
  for (int v = 0; v < V; v++) acc[v] = 0.0f;
  constexpr int U = 2;
  int p = s;
  for (; p + U <= e; p += U) {
    int i[U];
    float w[U];
    float x[U][V];
    int m[U][V];
#pragma unroll
    for (int u = 0; u < U; u++) {
      i[u] = idx[p + u];
      w[u] = HAS_VAL ? val[p + u] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      load_vec<V>(Gr + (int64_t)i[u] * N + f0, x[u]);
      load_vec<V>(E + (int64_t)i[u] * N + f0, m[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int v = 0; v < V; v++)
        if (m[u][v] == (int)j) acc[v] = __builtin_fmaf(w[u], x[u][v], acc[v]);
  }
  for (; p < e; p++) {
    const int i = idx[p];
    const float w = HAS_VAL ? val[p] : 1.0f;
    float x[V];
    int m[V];
    load_vec<V>(Gr + (int64_t)i * N + f0, x);
    load_vec<V>(E + (int64_t)i * N + f0, m);
#pragma unroll
    for (int v = 0; v < V; v++)
      if (m[v] == (int)j) acc[v] = __builtin_fmaf(w, x[v], acc[v]);
  }
  store_vec<V>(out + j * N + f0, acc);
}

template <int G, int V>
static int launch_mask(int64_t Mout, int64_t N, const int *ptr, const int *idx, const float *val, const float *Gr,
                       const int *E, float *out, int tiles, hipStream_t st) {
  const dim3 grid((unsigned)((Mout + (kBlock / G) - 1) / (kBlock / G)), (unsigned)tiles);
  if (val)
    hipLaunchKernelGGL((spmm_mask_rowgroup<G, V, true>), grid, dim3(kBlock), 0, st, (int)Mout, (int)N, ptr, idx, val,
                       Gr, E, out);
  else
    hipLaunchKernelGGL((spmm_mask_rowgroup<G, V, false>), grid, dim3(kBlock), 0, st, (int)Mout, (int)N, ptr, idx, val,
                       Gr, E, out);
  return check_launch();
}

template <int V>
static int dispatch_mask(int G, int64_t Mout, int64_t N, const int *ptr, const int *idx, const float *val,
                         const float *Gr, const int *E, float *out, int tiles, hipStream_t st) {
  switch (G) {
    case 1: return launch_mask<1, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
    case 2: return launch_mask<2, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
    case 4: return launch_mask<4, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
    case 8: return launch_mask<8, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
    case 16: return launch_mask<16, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
    case 32: return launch_mask<32, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
    case 64: return launch_mask<64, V>(Mout, N, ptr, idx, val, Gr, E, out, tiles, st);
  }
  return DGS_EINVAL;
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_spmm_csr_mask_f32(int64_t Mout, int64_t Min, int64_t N, int64_t nnz, const int32_t *ptr,
                                     const int32_t *idx, const float *val, const float *G, const int32_t *E,
                                     float *out, dgsStream_t stream) {
  if (Mout < 0 || Min < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (Mout >= INT32_MAX || Min >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (Mout == 0 || N == 0) return DGS_OK;
  if (!ptr || !out || (nnz > 0 && (!idx || !G || !E))) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool al = is_aligned16(G) && is_aligned16(E) && is_aligned16(out);
  const FeatMap fm = feat_map(N, al);
  if (fm.V == 4) return dispatch_mask<4>(fm.G, Mout, N, ptr, idx, val, G, E, out, fm.tiles, st);
  return dispatch_mask<1>(fm.G, Mout, N, ptr, idx, val, G, E, out, fm.tiles, st);
}
