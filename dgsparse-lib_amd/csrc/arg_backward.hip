// arg_backward.hip -- backward of SpMM-max/min in ONE pass over the forward's arg ids.
//
// The reference computes the two gradients with two masked kernels over all nnz x N (spmm_cuda_with_mask on the CSC
// arrays, sddmm_cuda_csr_with_mask; src/cuda/spmm_cuda.cu:255-303,363-382):
//   gX[j,f] = sum_{(i,j) in A} [E[i,f] == j] * A[i,j] * gC[i,f]        gW[e=(i,j)] = sum_f [E[i,f] == j] * gC[i,f] * X[j,f]
// Every output element (i,f) of the forward has exactly ONE arg column E[i,f], so both sums are scatters with one
// source per (i,f): for row i and feature f with j0 = E[i,f] >= 0
//   gX[j0,f] += (sum of val[p] over the edges p of row i with col[p] == j0) * gC[i,f]
//   gW[p]    += gC[i,f] * X[j0,f]          for those same edges p
// That is M*N sources instead of nnz*N gathered (grad row, arg-id row) pairs, no CSC arrays, no permuted values.
// The edges of j0 are found by scanning the row (col/val are wave-broadcast loads; nnz*N compares in total, no
// gathers); duplicates of a column all match, exactly as in the formulas above.
// The scatter uses fp32 atomics: sums are reproducible to rounding, not bit-for-bit from run to run (the masked
// kernels stay available and are used when the caller asks for deterministic algorithms).
#include "dgs_common.h"

namespace dgs {

template <int G, int V>
__global__ __launch_bounds__(kBlock) void arg_backward_rows(int M, int N, const int *__restrict__ rowptr,
                                                            const int *__restrict__ col, const float *__restrict__ val,
                                                            const int *__restrict__ E, const float *__restrict__ gC,
                                                            const float *__restrict__ X, float *__restrict__ gX,
                                                            float *__restrict__ gW) {
  constexpr int NG = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane / G, l = lane % G;
  const int64_t row = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * NG + g;
  const int f0 = (blockIdx.y * G + l) * V;
  const bool on = row < M && f0 < N;
  int e[V];
  float gc[V], gx[V], ws[V];
#pragma unroll
  for (int v = 0; v < V; v++) {
    e[v] = -1;
    gc[v] = 0.f;
    gx[v] = 0.f;
    ws[v] = 0.f;
  }
  int s = 0, t = 0;
  if (row < M) {
    s = rowptr[row];
    t = rowptr[row + 1];
  }
  if (on) {
    load_vec<V>(E + row * N + f0, e);
    load_vec<V>(gC + row * N + f0, gc);
#pragma unroll
    for (int v = 0; v < V; v++) {
      if (f0 + v >= N) e[v] = -1;  // V == 1 never overruns; V == 4 has N % 4 == 0
      if (gW && e[v] >= 0) gx[v] = gc[v] * X[(int64_t)e[v] * N + f0 + v];
    }
  }
  const int gbase = lane - l;
  const uint64_t gmask = (G == 64) ? ~0ull : (((1ull << G) - 1) << gbase);
  // ---- is the row sorted by column (duplicates allowed)?  one coalesced pass, G entries at a time ----
  int len = (row < M) ? t - s : 0;
  int wlen = len;
#pragma unroll
  for (int d = G; d < kWave; d <<= 1) wlen = max(wlen, __shfl_xor(wlen, d, kWave));
  bool bad = false;
  int carry = INT_MIN;
  for (int k0 = 0; k0 < wlen; k0 += G) {
    const int idx = s + k0 + l;
    const bool ok = k0 + l < len;
    const int c = ok ? col[idx] : INT_MAX;
    const int up = __shfl_up(c, 1, kWave);
    const int prev = (l == 0) ? carry : up;
    bad |= ok && c < prev;
    carry = __shfl(c, gbase + G - 1, kWave);
  }
  const bool sorted = (__ballot(bad) & gmask) == 0;
  {
    // ---- sorted row: lower_bound of every arg id, then its run of duplicates (no cross-lane traffic in here) ----
    if (sorted && on) {
      // the V searches of a lane advance together: V independent loads per step instead of V chains back to back
      int lo[V], hi[V];
#pragma unroll
      for (int v = 0; v < V; v++) {
        lo[v] = s;
        hi[v] = (e[v] >= 0) ? t : s;
      }
      bool more = true;
      while (more) {
        more = false;
        int cm[V];
#pragma unroll
        for (int v = 0; v < V; v++) cm[v] = (lo[v] < hi[v]) ? col[(lo[v] + hi[v]) >> 1] : 0;
#pragma unroll
        for (int v = 0; v < V; v++) {
          if (lo[v] < hi[v]) {
            const int mid = (lo[v] + hi[v]) >> 1;
            if (cm[v] < e[v]) lo[v] = mid + 1;
            else hi[v] = mid;
            more |= lo[v] < hi[v];
          }
        }
      }
#pragma unroll
      for (int v = 0; v < V; v++) {
        if (e[v] < 0) continue;
        for (int q = lo[v]; q < t && col[q] == e[v]; q++) {
          ws[v] += val ? val[q] : 1.0f;
          if (gW) unsafeAtomicAdd(gW + q, gx[v]);  // hardware fp32 atomic (device memory), no CAS loop
        }
      }
    }
  }
  {
    // ---- unsorted row: compare every entry with every arg id; entries reach the lanes by shuffle, G at a time ----
    // (wave-uniform loop: the whole wave walks to the longest unsorted row in it, sorted groups idle along)
    int ulen = sorted ? 0 : len;
#pragma unroll
    for (int d = G; d < kWave; d <<= 1) ulen = max(ulen, __shfl_xor(ulen, d, kWave));
    for (int k0 = 0; k0 < ulen; k0 += G) {
      const int idx = s + k0 + l;
      const bool ok = !sorted && k0 + l < len;
      const int cc = ok ? col[idx] : INT_MIN;  // never equals an arg id (>= -1)
      const float ww = (ok && val) ? val[idx] : 1.0f;
      const int nj = min(G, ulen - k0);
      for (int j = 0; j < nj; j++) {
        const int c = __shfl(cc, gbase + j, kWave);
        const float w = __shfl(ww, gbase + j, kWave);
#pragma unroll
        for (int v = 0; v < V; v++) {
          if (on && e[v] == c) {
            ws[v] += w;
            if (gW) unsafeAtomicAdd(gW + s + k0 + j, gx[v]);
          }
        }
      }
    }
  }
  if (on && gX) {
#pragma unroll
    for (int v = 0; v < V; v++)
      if (e[v] >= 0) unsafeAtomicAdd(gX + (int64_t)e[v] * N + f0 + v, ws[v] * gc[v]);
  }
}

template <int G, int V>
static int launch_arg_backward(int64_t M, int64_t N, int tiles, const int *rowptr, const int *col, const float *val,
                               const int *E, const float *gC, const float *X, float *gX, float *gW, hipStream_t st) {
  constexpr int rows_per_block = (kBlock / kWave) * (kWave / G);
  const dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block), (unsigned)tiles);
  hipLaunchKernelGGL((arg_backward_rows<G, V>), grid, dim3(kBlock), 0, st, (int)M, (int)N, rowptr, col, val, E, gC, X,
                     gX, gW);
  return check_launch();
}

template <int V>
static int dispatch_arg_backward(int G, int64_t M, int64_t N, int tiles, const int *rowptr, const int *col,
                                 const float *val, const int *E, const float *gC, const float *X, float *gX,
                                 float *gW, hipStream_t st) {
  switch (G) {
    case 1: return launch_arg_backward<1, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
    case 2: return launch_arg_backward<2, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
    case 4: return launch_arg_backward<4, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
    case 8: return launch_arg_backward<8, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
    case 16: return launch_arg_backward<16, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
    case 32: return launch_arg_backward<32, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
    case 64: return launch_arg_backward<64, V>(M, N, tiles, rowptr, col, val, E, gC, X, gX, gW, st);
  }
  return DGS_EINVAL;
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_spmm_arg_backward_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                         const int32_t *col, const float *val, const int32_t *E, const float *gC,
                                         const float *X, float *gX, float *gW, dgsStream_t stream) {
  if (M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gX && K > 0 && N > 0 && hipMemsetAsync(gX, 0, (size_t)K * N * sizeof(float), st) != hipSuccess) return DGS_ELAUNCH;
  if (gW && nnz > 0 && hipMemsetAsync(gW, 0, (size_t)nnz * sizeof(float), st) != hipSuccess) return DGS_ELAUNCH;
  if (M == 0 || N == 0 || nnz == 0 || (!gX && !gW)) return DGS_OK;
  if (!rowptr || !col || !E || !gC || (gW && !X)) return DGS_EINVAL;
  const bool al = is_aligned16(E) && is_aligned16(gC);
  const FeatMap fm = feat_map(N, al);
  if (fm.V == 4) return dispatch_arg_backward<4>(fm.G, M, N, fm.tiles, rowptr, col, val, E, gC, X, gX, gW, st);
  return dispatch_arg_backward<1>(fm.G, M, N, fm.tiles, rowptr, col, val, E, gC, X, gX, gW, st);
}
