// sddmm.hip -- CSR SDDMM (+ MEAN scaling, + arg mask) for gfx950, nnz-balanced.
//   out[e] = sum_k D1[row(e),k] * D2[col(e),k]
// Replaces sddmmCSR{2,1}Scale<REDUCE> / sddmmCSR1Scale_with_mask (reference include/cuda/sddmm_cuda.cuh:222-401,
// 403-507): edge-balanced, 4 edges per 32-lane warp slice, one binary search over rowptr PER EDGE (findRow,
// cuda_util.cuh:150-166).
//
// Schedule here: one wave per chunk of 256 consecutive nnz (grid-stride), so a 50k-nnz row costs the same per nnz
// as a 3-nnz row.  Per 64-nnz tile:
//   * ONE binary search per chunk finds the row of the first nnz; inside the tile the row of every nnz comes from
//     a merge of the (sorted) next 64 row boundaries with the (sorted) nnz positions: boundary lanes drop a +1 at
//     their position in an LDS histogram, a wave prefix sum turns it into "rows advanced" - no per-edge search;
//   * (col,row) pairs are staged in LDS; the NG=64/G groups take interleaved nnz; a group (G lanes x V floats) reads
//     the D1 row slice and the D2 row slice with coalesced dwordx4 loads (the D1 slice is an L1/L2 hit while the
//     row lasts), U nnz in flight per group, xor-butterfly per dot product;
//   * the 64 results go through LDS and leave with one coalesced store.
#include <stdlib.h>

#include <atomic>

#include "dgs_common.h"
#include "sddmm_panel.h"
#include "sddmm_fused.h"

namespace dgs {

constexpr int kSdChunk = 256;  // nnz per wave-chunk
constexpr int kSdU = 4;        // nnz in flight per group

// xor-butterfly over the G lanes of a group (G power of two <= 64): every lane ends with the total.
template <int G>
__device__ __forceinline__ float group_allreduce(float x) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

__device__ __forceinline__ int wave_incl_scan(int x, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int t = __shfl_up(x, d, kWave);
    if (lane >= d) x += t;
  }
  return x;
}

// Dot products of one staged tile: tile[j] = {col, row} of nnz t0 + j (j < cntn); group g takes nnz g, g+NG, ...,
// kSdU of them in flight; the results go through `cnt` (LDS) and leave with one coalesced store.
// ONE_ROW: every entry of the tile belongs to the same row (sddmm_longrows): its D1 slice (and arg ids) are loaded once per
// feature tile instead of once per nnz - half the vector-memory instructions of the batch.
template <int G, int V, bool MEAN, bool MASK, bool ONE_ROW = false>
__device__ __forceinline__ void sd_tile_dots(const int2 *tile, int *cnt, int cntn, int t0, int lane, int F, int tiles,
                                             const int *__restrict__ rowptr, const float *__restrict__ D1,
                                             const float *__restrict__ D2, const int *__restrict__ E,
                                             float *__restrict__ out) {
  constexpr int NG = kWave / G;
  const int g = lane / G, l = lane % G;
  // ---- dot products: group g takes nnz g, g+NG, ...; kSdU of them in flight
  for (int j0 = g; j0 < cntn; j0 += NG * kSdU) {
    float part[kSdU];
    int2 cr[kSdU];
#pragma unroll
    for (int q = 0; q < kSdU; q++) {
      part[q] = 0.0f;
      cr[q] = tile[min(j0 + q * NG, cntn - 1)];
    }
    for (int t = 0; t < tiles; t++) {
      const int f = (t * G + l) * V;
      if (f < F) {
        float a[ONE_ROW ? 1 : kSdU][V], b[kSdU][V];
        int m[ONE_ROW ? 1 : kSdU][V];
        if constexpr (ONE_ROW) {
          load_vec_rowop<V>(D1 + (int64_t)cr[0].y * F + f, a[0]);
          if constexpr (MASK) load_vec<V>(E + (int64_t)cr[0].y * F + f, m[0]);
        }
#pragma unroll
        for (int q = 0; q < kSdU; q++) {
          if constexpr (!ONE_ROW) load_vec_rowop<V>(D1 + (int64_t)cr[q].y * F + f, a[q]);
          load_vec<V>(D2 + (int64_t)cr[q].x * F + f, b[q]);
          if constexpr (MASK && !ONE_ROW) load_vec<V>(E + (int64_t)cr[q].y * F + f, m[q]);
        }
#pragma unroll
        for (int q = 0; q < kSdU; q++)
#pragma unroll
          for (int v = 0; v < V; v++) {
            constexpr int qa = ONE_ROW ? 0 : 1;
            if constexpr (MASK) {
              if (m[q * qa][v] == cr[q].x) part[q] = __builtin_fmaf(a[q * qa][v], b[q][v], part[q]);
            } else {
              part[q] = __builtin_fmaf(a[q * qa][v], b[q][v], part[q]);
            }
          }
      }
    }
#pragma unroll
    for (int q = 0; q < kSdU; q++) {
      float tot = group_allreduce<G>(part[q]);
      const int j = j0 + q * NG;
      if (j < cntn && l == 0) {
        if constexpr (MEAN) {  // sddmm_cuda.cuh:266-272: divide by deg(row(e)) (always > 0 for a stored entry)
          const int rw = cr[q].y;
          tot /= (float)(rowptr[rw + 1] - rowptr[rw]);
        }
        cnt[j] = __float_as_int(tot);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < cntn) out[t0 + lane] = __int_as_float(cnt[lane]);
}

// compiled for 5 waves per SIMD (96 VGPRs, no spills; the natural 108 give 4): 554 -> 494 us on the 1M power-law graph,
// 2.34 -> 2.23 ms products-shaped; 6 waves spill 21 registers and lose (643 us)
#ifndef DGS_SD_WAVES
#define DGS_SD_WAVES 5
#endif
template <int G, int V, bool MEAN, bool MASK>
__global__ __launch_bounds__(kBlock, DGS_SD_WAVES) void sddmm_nnzbal(int M, int F, int tiles, int nnz,
                                                       const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       const float *__restrict__ D1, const float *__restrict__ D2,
                                                       const int *__restrict__ E, float *__restrict__ out) {
  __shared__ int2 s_tile[kBlock / kWave][kWave];  // {col, row}
  __shared__ int s_cnt[kBlock / kWave][kWave];    // boundary histogram, then the 64 results (as float bits)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int2 *tile = s_tile[wave];
  int *cnt = s_cnt[wave];
  const int nchunks = (nnz + kSdChunk - 1) / kSdChunk;
  // XCD-aware chunk mapping (speed hint): block b runs on XCD b % 8 (observed); each XCD walks one contiguous eighth
  // of the nnz range = neighbouring rows, whose D1 rows and - in a locality-preserving order - D2 rows share its L2.
  int c, cend, wstride;
  if ((gridDim.x & 7) == 0) {
    const int x = blockIdx.x & 7;
    c = (int)(((long long)nchunks * x) >> 3) + (blockIdx.x >> 3) * (kBlock / kWave) + wave;
    cend = (int)(((long long)nchunks * (x + 1)) >> 3);
    wstride = (gridDim.x >> 3) * (kBlock / kWave);
  } else {
    c = blockIdx.x * (kBlock / kWave) + wave;
    cend = nchunks;
    wstride = gridDim.x * (kBlock / kWave);
  }
  for (; c < cend; c += wstride) {
    const int p0 = c * kSdChunk, p1 = min(nnz, p0 + kSdChunk);
    // row of p0: last r with rowptr[r] <= p0 (skips empty rows); wave-uniform scalar search
    int r;
    {
      int lo = 0, hi = M;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= p0) lo = mid; else hi = mid - 1;
      }
      r = lo;
    }
    for (int t0 = p0; t0 < p1; t0 += kWave) {
      const int cntn = min(kWave, p1 - t0);
      // ---- rows of the tile's nnz: histogram of row starts falling inside (t0, t0+63], prefix-summed
      __builtin_amdgcn_wave_barrier();
      cnt[lane] = 0;
      __builtin_amdgcn_wave_barrier();
      int rbase = r;
      while (true) {
        const int rr = rbase + 1 + lane;
        const int bnd = (rr <= M) ? rowptr[rr] : INT_MAX;  // start of row rr (rowptr[M] = nnz ends the last row)
        const int d = bnd - t0;
        if (rr < M && d >= 0 && d < kWave) atomicAdd(&cnt[d], 1);
        const int last_bnd = __shfl(bnd, kWave - 1, kWave);
        if (last_bnd > t0 + kWave - 1 || rbase + kWave >= M) break;
        rbase += kWave;  // more than 64 row starts inside this tile (runs of empty rows): keep going
      }
      __builtin_amdgcn_wave_barrier();
      const int adv = wave_incl_scan(cnt[lane], lane);
      const int myrow = r + adv;
      int mycol = 0;
      if (lane < cntn) mycol = ld_stream(col + t0 + lane);
      __builtin_amdgcn_wave_barrier();
      tile[lane] = make_int2(mycol, myrow);
      __builtin_amdgcn_wave_barrier();
      r = __shfl(myrow, cntn - 1, kWave);  // row of the tile's last nnz: where the next tile starts

      sd_tile_dots<G, V, MEAN, MASK>(tile, cnt, cntn, t0, lane, F, tiles, rowptr, D1, D2, E, out);
    }
  }
}

// Rows longer than `minlen` only (the column-panel schedule sweeps the others): a block of 16 waves looks at 64
// consecutive rows (lane = row), and the waves share the 64-nnz tiles of every long row among them round-robin
// (a tile is a chain of dependent loads of ~6 us: a 21 k-nnz row is 330 tiles, 21 per wave).
constexpr int kLongBlock = 1024;
template <int G, int V, bool MEAN, bool MASK>
__global__ __launch_bounds__(kLongBlock) void sddmm_longrows(int M, int F, int tiles, int minlen,
                                                             const int *__restrict__ rowptr,
                                                             const int *__restrict__ col, const float *__restrict__ D1,
                                                             const float *__restrict__ D2, const int *__restrict__ E,
                                                             float *__restrict__ out) {
  __shared__ int2 s_tile[kLongBlock / kWave][kWave];
  __shared__ int s_cnt[kLongBlock / kWave][kWave];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int2 *tile = s_tile[wave];
  int *cnt = s_cnt[wave];
  const int r0 = blockIdx.x * kWave;
  int s = 0, e = 0;
  if (r0 + lane < M) {
    s = rowptr[r0 + lane];
    e = rowptr[r0 + lane + 1];
  }
  unsigned long long todo = __ballot(e - s > minlen);
  while (todo) {
    const int rl = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int rs = __shfl(s, rl, kWave), re = __shfl(e, rl, kWave);
    for (int t0 = rs + wave * kWave; t0 < re; t0 += kLongBlock) {
      const int cntn = min(kWave, re - t0);
      __builtin_amdgcn_wave_barrier();
      tile[lane] = make_int2(lane < cntn ? ld_stream(col + t0 + lane) : 0, r0 + rl);
      __builtin_amdgcn_wave_barrier();
      sd_tile_dots<G, V, MEAN, MASK, true>(tile, cnt, cntn, t0, lane, F, tiles, rowptr, D1, D2, E, out);
    }
  }
}

// COO SDDMM: out[e] = <D1[rowind[e]], D2[colind[e]]> (reference sddmm_cuda_coo, src/cuda/spmm_cuda.cu:305-329 and the
// standalone src/sddmm/sddmm.h:7).  Same group mapping as the CSR kernel, the row id simply comes from the array.
template <int G, int V>
__global__ __launch_bounds__(kBlock) void sddmm_coo(int F, int tiles, int nnz, const int *__restrict__ rowind,
                                                    const int *__restrict__ colind, const float *__restrict__ D1,
                                                    const float *__restrict__ D2, float *__restrict__ out) {
  constexpr int NG = kWave / G;
  __shared__ int2 s_tile[kBlock / kWave][kWave];
  __shared__ int s_res[kBlock / kWave][kWave];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane / G, l = lane % G;
  int2 *tile = s_tile[wave];
  int *res = s_res[wave];
  const int ntiles = (nnz + kWave - 1) / kWave;
  const int wstride = gridDim.x * (kBlock / kWave);
  for (int t = blockIdx.x * (kBlock / kWave) + wave; t < ntiles; t += wstride) {
    const int t0 = t * kWave, cntn = min(kWave, nnz - t0);
    __builtin_amdgcn_wave_barrier();
    if (lane < cntn) tile[lane] = make_int2(ld_stream(colind + t0 + lane), ld_stream(rowind + t0 + lane));
    __builtin_amdgcn_wave_barrier();
    for (int j0 = g; j0 < cntn; j0 += NG * kSdU) {
      float part[kSdU];
      int2 cr[kSdU];
#pragma unroll
      for (int q = 0; q < kSdU; q++) {
        part[q] = 0.0f;
        cr[q] = tile[min(j0 + q * NG, cntn - 1)];
      }
      for (int tt = 0; tt < tiles; tt++) {
        const int f = (tt * G + l) * V;
        if (f < F) {
          float a[kSdU][V], b[kSdU][V];
#pragma unroll
          for (int q = 0; q < kSdU; q++) {
            load_vec_rowop<V>(D1 + (int64_t)cr[q].y * F + f, a[q]);
            load_vec<V>(D2 + (int64_t)cr[q].x * F + f, b[q]);
          }
#pragma unroll
          for (int q = 0; q < kSdU; q++)
#pragma unroll
            for (int v = 0; v < V; v++) part[q] = __builtin_fmaf(a[q][v], b[q][v], part[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < kSdU; q++) {
        const float tot = group_allreduce<G>(part[q]);
        const int j = j0 + q * NG;
        if (j < cntn && l == 0) res[j] = __float_as_int(tot);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < cntn) out[t0 + lane] = __int_as_float(res[lane]);
  }
}

template <int G, int V>
static int launch_coo(int64_t F, int tiles, int64_t nnz, const int *rowind, const int *colind, const float *D1,
                      const float *D2, float *out, hipStream_t st) {
  const int64_t ntiles = (nnz + kWave - 1) / kWave;
  int64_t blocks = (ntiles + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL((sddmm_coo<G, V>), dim3((unsigned)blocks), dim3(kBlock), 0, st, (int)F, tiles, (int)nnz, rowind,
                     colind, D1, D2, out);
  return check_launch();
}

template <int V>
static int dispatch_coo(int G, int64_t F, int tiles, int64_t nnz, const int *rowind, const int *colind,
                        const float *D1, const float *D2, float *out, hipStream_t st) {
  switch (G) {
    case 1: return launch_coo<1, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
    case 2: return launch_coo<2, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
    case 4: return launch_coo<4, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
    case 8: return launch_coo<8, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
    case 16: return launch_coo<16, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
    case 32: return launch_coo<32, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
    case 64: return launch_coo<64, V>(F, tiles, nnz, rowind, colind, D1, D2, out, st);
  }
  return DGS_EINVAL;
}

template <int G, int V, bool MEAN, bool MASK>
static int launch_sddmm(int64_t M, int64_t F, int tiles, int64_t nnz, const int *rowptr, const int *col,
                        const float *D1, const float *D2, const int *E, float *out, hipStream_t st) {
  const int64_t nchunks = (nnz + kSdChunk - 1) / kSdChunk;
  int64_t blocks = (nchunks + (kBlock / kWave) - 1) / (kBlock / kWave);
  if (blocks > 8192) blocks = 8192;  // persistent beyond this: waves stride over the chunks
  if (blocks >= 64) blocks = (blocks + 7) & ~int64_t(7);  // multiple of 8: enables the per-XCD contiguous mapping
  hipLaunchKernelGGL((sddmm_nnzbal<G, V, MEAN, MASK>), dim3((unsigned)blocks), dim3(kBlock), 0, st, (int)M, (int)F,
                     tiles, (int)nnz, rowptr, col, D1, D2, E, out);
  return check_launch();
}

template <int V, bool MEAN, bool MASK>
static int dispatch_sddmm(int G, int64_t M, int64_t F, int tiles, int64_t nnz, const int *rowptr, const int *col,
                          const float *D1, const float *D2, const int *E, float *out, hipStream_t st) {
  switch (G) {
    case 1: return launch_sddmm<1, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
    case 2: return launch_sddmm<2, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
    case 4: return launch_sddmm<4, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
    case 8: return launch_sddmm<8, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
    case 16: return launch_sddmm<16, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
    case 32: return launch_sddmm<32, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
    case 64: return launch_sddmm<64, V, MEAN, MASK>(M, F, tiles, nnz, rowptr, col, D1, D2, E, out, st);
  }
  return DGS_EINVAL;
}

// ---- column-panel schedule for dense graphs (sddmm_panel.h) ----

struct SdPanelPlan {
  bool use;
  int R, tlong, pcols, npanels, nsb, nwg, lead;
  size_t lds;
};

static SdPanelPlan sd_panel_plan(int64_t M, int64_t K, int64_t F, int64_t nnz, int tiles, int G, int V, bool mask) {
  SdPanelPlan P{};
  const Tuning &T = tuning();
  const int force = tune(T.panel, -1);
  if (force == 0 || (tiles != 1 && G != 64) || V != 4 || G < 8 || M <= 0 || K <= 0) return P;
  P.nwg = cu_count();
  const int64_t W = F < 256 ? F : 256;  // feature tile per launch
  int slots = (int)(kSdPanelBytes / (W * (mask ? 8 : 4)));
  if (slots > kPanelRMax) slots = kPanelRMax;
  if (slots < 8) return P;
  // same rule as the SpMM twin (spmm_impl.h panel_plan): D2 must overflow the L2s, every panel row must be reused
  // several times per XCD and a row visit must still hold a handful of nnz
  const double deg = (double)nnz / (double)M;
  const double reuse = (P.nwg / 8.0) * slots * deg / (double)K;
  int pkb = tune(T.panel_kb, 6144);
  if (pkb < 1) pkb = 1;
  int64_t pc = (int64_t)pkb * 1024 / (W * 4);
  if (pc < 64) pc = 64;
  const double visit = deg * (double)pc / (double)K;
  const bool pays = reuse >= 5.5 || (reuse >= 4.0 && visit >= 10.0);
  if (force != 1 && !((double)K * W * 4.0 >= 16e6 && pays && M >= 4096)) return P;
  P.nsb = (int)((M + (int64_t)P.nwg * slots - 1) / ((int64_t)P.nwg * slots));
  P.R = (int)((M + (int64_t)P.nwg * P.nsb - 1) / ((int64_t)P.nwg * P.nsb));
  P.pcols = (int)pc;
  P.npanels = (int)((K + pc - 1) / pc);
  P.lead = tune(T.panel_lead, 1);
  P.tlong = tune(T.panel_tlong, 2048);
  if (P.tlong < 1) P.tlong = 1;
  P.lds = (size_t)P.R * W * (mask ? 8 : 4);
  P.use = true;
  return P;
}

template <int G, bool MEAN, bool MASK>
static int launch_sddmm_panel(const SdPanelPlan &P, int64_t M, int64_t F, const int *rowptr, const int *col,
                              const float *D1, const float *D2, const int *E, float *out, hipStream_t st) {
  auto kern = sddmm_panel<G, MEAN, MASK>;
  static std::atomic<bool> attr_set[64];  // per instantiation and device: allow the large dynamic LDS
  int dev_id = 0;
  if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return DGS_ELAUNCH;
  if (!attr_set[dev_id]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kSdPanelBytes) != hipSuccess)
      return DGS_ELAUNCH;
    attr_set[dev_id] = true;
  }
  int *arrivals = nullptr;
  if (hipGetSymbolAddress(reinterpret_cast<void **>(&arrivals), HIP_SYMBOL(g_sddmm_arrivals)) != hipSuccess)
    return DGS_ELAUNCH;
  for (int64_t fb = 0; fb < F; fb += 256) {  // one sweep per 256-feature tile; later tiles add to out[]
    const int W = (int)(F - fb < 256 ? F - fb : 256);
    const int pass = (fb ? 1 : 0) | (fb + 256 >= F ? 2 : 0);
    if (hipMemsetAsync(arrivals, 0, sizeof(int), st) != hipSuccess) return DGS_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)P.nwg), dim3(kPanelBlock), P.lds, st, (int)M, W, (int)F, pass, P.R, P.tlong,
                       P.pcols, P.npanels, P.nsb, P.lead, rowptr, col, D1 + fb, D2 + fb, E ? E + fb : nullptr, out,
                       arrivals);
  }
  // rows longer than tlong: row-driven kernel, the 16 waves of a block share the tiles of each long row
  hipLaunchKernelGGL((sddmm_longrows<G, 4, MEAN, MASK>), dim3((unsigned)((M + kWave - 1) / kWave)), dim3(kLongBlock), 0, st,
                     (int)M, (int)F, (int)((F + 255) / 256), P.tlong, rowptr, col, D1, D2, E, out);
  return check_launch();
}

template <bool MEAN, bool MASK>
static int dispatch_sddmm_panel(int G, const SdPanelPlan &P, int64_t M, int64_t F, const int *rowptr, const int *col,
                                const float *D1, const float *D2, const int *E, float *out, hipStream_t st) {
  switch (G) {
    case 8: return launch_sddmm_panel<8, MEAN, MASK>(P, M, F, rowptr, col, D1, D2, E, out, st);
    case 16: return launch_sddmm_panel<16, MEAN, MASK>(P, M, F, rowptr, col, D1, D2, E, out, st);
    case 32: return launch_sddmm_panel<32, MEAN, MASK>(P, M, F, rowptr, col, D1, D2, E, out, st);
    case 64: return launch_sddmm_panel<64, MEAN, MASK>(P, M, F, rowptr, col, D1, D2, E, out, st);
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
  const SdPanelPlan P = sd_panel_plan(M, K, F, nnz, fm.tiles, fm.G, fm.V, MASK);
  if (P.use) return dispatch_sddmm_panel<MEAN, MASK>(fm.G, P, M, F, rowptr, col, D1, D2, E, out, st);
  if (fm.V == 4) return dispatch_sddmm<4, MEAN, MASK>(fm.G, M, F, fm.tiles, nnz, rowptr, col, D1, D2, E, out, st);
  return dispatch_sddmm<1, MEAN, MASK>(fm.G, M, F, fm.tiles, nnz, rowptr, col, D1, D2, E, out, st);
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_sddmm_csr_schedule(int64_t M, int64_t K, int64_t F, int64_t nnz, int masked) {
  if (M <= 0 || K <= 0 || F <= 0 || nnz <= 0) return DGS_SCHED_ROWS;
  const FeatMap fm = feat_map(F, true);
  return sd_panel_plan(M, K, F, nnz, fm.tiles, fm.G, fm.V, masked != 0).use ? DGS_SCHED_PANEL : DGS_SCHED_ROWS;
}

extern "C" int dgs_sddmm_csr_f32(int reduce_op, int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                                 const int32_t *col, const float *D1, const float *D2, float *out,
                                 dgsStream_t stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (reduce_op == DGS_MEAN) return run_sddmm<true, false>(M, K, F, nnz, rowptr, col, D1, D2, nullptr, out, st);
  if (reduce_op == DGS_SUM) return run_sddmm<false, false>(M, K, F, nnz, rowptr, col, D1, D2, nullptr, out, st);
  return DGS_EINVAL;
}

// SDDMM over the cached locality plan of (rowptr, col) (sddmm_fused.h).  Shapes the fused kernel does not cover - dense
// graphs (column-panel schedule), feature widths that need several tiles or scalar lanes, tiny inputs - take the plan-free path.
extern "C" int dgs_sddmm_csr_plan_f32(int reduce_op, int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                                      const int32_t *col, const float *D1, const float *D2, float *out, const void *plan,
                                      const dgsSpmmPlanInfo *info, dgsStream_t stream) {
  if (reduce_op != DGS_SUM && reduce_op != DGS_MEAN) return DGS_EINVAL;
  if (M < 0 || K < 0 || F < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || F >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool fused = plan && info && M > 0 && nnz > 0 && F >= 32 && F <= 256 && F % 4 == 0 && rowptr && col && D1 && D2 && out &&
               is_aligned16(plan) && is_aligned16(D1) && is_aligned16(D2) && !tiny_problem(M, nnz);
  if (fused) {
    const FeatMap fm = feat_map(F, true);
    // worth it on hub-heavy graphs, where the units of the column-cut rows (n_pslots of them, ~64..256 nnz each) hold about a
    // third of the nnz or more: 1M-row power-law graph (alpha 2.1) 493 -> 465 us; products-shaped (alpha 2.4, 0.0031 cut
    // units per nnz) 2232 -> 2280 us, so that one stays on the nnz-balanced kernel.  DGS_SDDMM_FUSED=0/1 overrides.
    const int force = tune(tuning().sddmm_fused, -1);
    fused = fm.tiles == 1 && fm.V == 4 && fm.G >= 8 && !sd_panel_plan(M, K, F, nnz, fm.tiles, fm.G, fm.V, false).use &&
            force != 0 && (force == 1 || ((int64_t)info->n_pslots * 256 >= nnz && info->xcd_start[8] > 0));
    // (xcd_start[8] = the real unit count; a PROVISIONAL info (dgs_spmm_plan_provisional_info: upper bounds, zeros here) must
    // not decide - its n_pslots bounds every unit, the rule would pick this kernel where the real counts reject it, and the
    // value-gradient bits of a caller would change once more when the compact plan arrives.  ADVICE r3.)
    if (fused) {
      const PlanHdr *ph = static_cast<const PlanHdr *>(plan);
      const bool mean = reduce_op == DGS_MEAN;
      switch (fm.G) {
#define DGS_SF_CASE(g) case g: return mean ? launch_sddmm_fused<g, true>(M, F, nnz, rowptr, col, D1, D2, out, ph, info, st) \
                                           : launch_sddmm_fused<g, false>(M, F, nnz, rowptr, col, D1, D2, out, ph, info, st);
        DGS_SF_CASE(8) DGS_SF_CASE(16) DGS_SF_CASE(32) DGS_SF_CASE(64)
#undef DGS_SF_CASE
      }
    }
  }
  return dgs_sddmm_csr_f32(reduce_op, M, K, F, nnz, rowptr, col, D1, D2, out, stream);
}

extern "C" int dgs_sddmm_csr_mask_f32(int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                                      const int32_t *col, const float *D1, const float *D2, const int32_t *E,
                                      float *out, dgsStream_t stream) {
  return run_sddmm<false, true>(M, K, F, nnz, rowptr, col, D1, D2, E, out, static_cast<hipStream_t>(stream));
}

extern "C" int dgs_sddmm_coo_f32(int64_t F, int64_t nnz, const int32_t *rowind, const int32_t *colind, const float *D1,
                                 const float *D2, float *out, dgsStream_t stream) {
  if (F < 0 || nnz < 0) return DGS_EINVAL;
  if (F >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (nnz == 0) return DGS_OK;
  if (!rowind || !colind || !out || (F > 0 && (!D1 || !D2))) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (F == 0) return hipMemsetAsync(out, 0, (size_t)nnz * sizeof(float), st) == hipSuccess ? DGS_OK : DGS_ELAUNCH;
  const FeatMap fm = feat_map(F, is_aligned16(D1) && is_aligned16(D2));
  if (fm.V == 4) return dispatch_coo<4>(fm.G, F, fm.tiles, nnz, rowind, colind, D1, D2, out, st);
  return dispatch_coo<1>(fm.G, F, fm.tiles, nnz, rowind, colind, D1, D2, out, st);
}
