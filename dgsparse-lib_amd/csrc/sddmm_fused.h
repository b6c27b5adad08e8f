// sddmm_fused.h -- CSR SDDMM over the cached locality plan of the SpMM schedule (spmm_plan.hip): the same split of the
// matrix as spmm_fused - rows up to T1 nnz in ROW BLOCKS, longer rows as the plan's UNITS (<= 256 nnz, hub rows cut on the
// column grid, the table sorted by (column slice, first column), one slice per XCD) - with the SDDMM inner step:
//   out[e] = < D1[row(e), :], D2[col(e), :] >          (reference include/cuda/sddmm_cuda.cuh:222-401)
//
// Why a second SDDMM kernel: the nnz-balanced one (sddmm.hip) loads the D1 slice AND the D2 slice for every nnz - half of
// its loads are re-reads of a row it already holds - and walks the nnz in CSR order, so the gathers of a hub row are spread
// over the whole column space on every XCD.  Round 2 tried the plan's unit table inside that kernel twice and lost: long
// and short rows interleave, so most 256-nnz chunks were MIXED and the entries a unit block had to skip still took their
// tile slots.  Here nothing is mixed:
//   unit blocks  one wave per unit; the D1 slice (and, masked, the arg-id slice) of the unit's row sits in REGISTERS, so
//                all 8 loads a lane has in flight are D2 gathers; 8 partial dot products are reduced over the G lanes by a
//                transposing butterfly (8 + log2 G - 3 shuffles instead of 8 log2 G); results leave through LDS, one
//                coalesced store per 64 nnz.  XCD x walks column slice x of the table.
//   row blocks   one wave per 64 rows: the (col, row) pairs of runs of short rows are staged compactly into LDS (long rows
//                are skipped - their nnz belong to units), every group takes a CONTIGUOUS range of the run, so consecutive
//                nnz share their row and the D1 loads of a batch mostly coincide (L1 hits).
// Nothing is reduced across nnz, so there are no partial rows, no combine launch and no workspace.
// Summation order inside a dot product: V-wide fma chain per lane, then the butterfly - fixed, run-to-run identical; the
// parity bar is 1e-5 relative against the sequential host loop (sddmm_reference_host, example/util/sp_util.hpp:88-112).
#pragma once
#include "spmm_impl.h"

namespace dgs {

constexpr int kSfU = 8;  // D2 gathers in flight per lane = entries of one butterfly batch

// entry of a batch that lane `lig` of a group holds after the transposing butterfly
__device__ __forceinline__ int tb8_entry(int lig) { return ((lig & 1) ? 4 : 0) + ((lig & 2) ? 2 : 0) + ((lig & 4) ? 1 : 0); }

// 8 partial sums x G lanes -> every lane ends with the group total of entry tb8_entry(lig)
template <int G>
__device__ __forceinline__ float tb8(const float (&pt)[8], int lig) {
  static_assert(G >= 8, "the 8-way transposing butterfly needs 8 lanes");
  float q4[4], q2[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float send = (lig & 1) ? pt[i] : pt[i + 4];
    const float keep = (lig & 1) ? pt[i + 4] : pt[i];
    q4[i] = keep + __shfl_xor(send, 1, 64);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const float send = (lig & 2) ? q4[i] : q4[i + 2];
    const float keep = (lig & 2) ? q4[i + 2] : q4[i];
    q2[i] = keep + __shfl_xor(send, 2, 64);
  }
  const float send = (lig & 4) ? q2[0] : q2[1];
  const float keep = (lig & 4) ? q2[1] : q2[0];
  float tot = keep + __shfl_xor(send, 4, 64);
#pragma unroll
  for (int m = 8; m < G; m <<= 1) tot += __shfl_xor(tot, m, 64);
  return tot;
}

struct SfLds {
  int2 tile[kBlock / kWave][kCap];   // {col, local row} of a staged run (row blocks) / {col, -} of a unit's 64-nnz tile
  float res[kBlock / kWave][kCap];   // results of the run, stored coalesced at its end
  int4 rows[kBlock / kWave][kRowsPerWave + 1];  // {start, end, -, -} of the wave's rows
  int mark[kBlock / kWave][kWave];   // row starts inside one 64-nnz pass (scatter + prefix max -> row of every nnz)
};

// ---------------------------------------------------------------------------------------------------------------------
template <int G, int V, bool MEAN, bool MASK>
__device__ __forceinline__ void sddmm_units_body(int bid, int nblocks, SfLds &lds, int F, const int *__restrict__ rowptr,
                                                 const int *__restrict__ col, const float *__restrict__ D1,
                                                 const float *__restrict__ D2, const int *__restrict__ E,
                                                 float *__restrict__ out, const UnitTab &ut) {
  constexpr int NG = kWave / G;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane / G, l = lane % G;
  int2 *tile = lds.tile[wave];
  float *res = lds.res[wave];
  const int f0 = l * V;
  const bool fl = f0 < F;
  const int fo = fl ? f0 : 0;
  const int n_units = *ut.n_units;
  int u, uend, wstride;
  if ((nblocks & 7) == 0) {  // XCD x walks slice x of the (sorted) unit table: block b runs on XCD b % 8 (speed hint)
    const int x = bid & 7;
    const int lo = ut.xcd_start ? ut.xcd_start[x] : (int)(((long long)n_units * x) >> 3);
    uend = ut.xcd_start ? ut.xcd_start[x + 1] : (int)(((long long)n_units * (x + 1)) >> 3);
    u = lo + (bid >> 3) * (kBlock / kWave) + wave;
    wstride = (nblocks >> 3) * (kBlock / kWave);
  } else {
    u = bid * (kBlock / kWave) + wave;
    uend = n_units;
    wstride = nblocks * (kBlock / kWave);
  }
  const int ent = tb8_entry(l);
  for (; u < uend; u += wstride) {
    const int4 d = ut.units[u];  // {row, first nnz, nnz in the unit, -}
    float a[V];
    int m[V];
    load_vec_rowop<V>(D1 + (int64_t)d.x * F + fo, a);
    if constexpr (MASK) load_vec<V>(E + (int64_t)d.x * F + fo, m);
    float scale = 1.0f;
    if constexpr (MEAN) scale = (float)(rowptr[d.x + 1] - rowptr[d.x]);
    if (!fl) {
#pragma unroll
      for (int v = 0; v < V; v++) a[v] = 0.0f;
    }
    for (int t0 = d.y; t0 < d.y + d.z; t0 += kWave) {
      const int cnt = min(kWave, d.y + d.z - t0);
      __builtin_amdgcn_wave_barrier();
      if (lane < cnt) tile[lane].x = ld_stream(col + t0 + lane);
      __builtin_amdgcn_wave_barrier();
      for (int j0 = 0; j0 < cnt; j0 += NG * kSfU) {
        int c[kSfU];
        float b[kSfU][V], pt[kSfU];
#pragma unroll
        for (int q = 0; q < kSfU; q++) c[q] = tile[min(j0 + q * NG + g, cnt - 1)].x;
#pragma unroll
        for (int q = 0; q < kSfU; q++) load_vec_gather<V>(D2 + (int64_t)c[q] * F + fo, b[q]);
#pragma unroll
        for (int q = 0; q < kSfU; q++) {
          float s = 0.0f;
#pragma unroll
          for (int v = 0; v < V; v++) {
            if constexpr (MASK) {
              if (m[v] == c[q]) s = __builtin_fmaf(a[v], b[q][v], s);
            } else {
              s = __builtin_fmaf(a[v], b[q][v], s);
            }
          }
          pt[q] = s;
        }
        float tot = tb8<G>(pt, l);
        if constexpr (MEAN) tot /= scale;
        const int j = j0 + ent * NG + g;
        if (l < 8 && j < cnt) res[j] = tot;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < cnt) out[t0 + lane] = res[lane];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int G, int V, bool MEAN, bool MASK>
__device__ __forceinline__ void sddmm_rows_body(int bid, int rpw, SfLds &lds, int M, int F, const int *__restrict__ rowptr,
                                                const int *__restrict__ col, const float *__restrict__ D1,
                                                const float *__restrict__ D2, const int *__restrict__ E,
                                                float *__restrict__ out) {
  constexpr int NG = kWave / G;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane / G, l = lane % G;
  const int r0 = (bid * (kBlock / kWave) + wave) * rpw;
  if (r0 >= M) return;  // wave-uniform
  int2 *tile = lds.tile[wave];
  float *res = lds.res[wave];
  int4 *rows = lds.rows[wave];
  int *mark = lds.mark[wave];
  const int nrows = min(rpw, M - r0);
  const int f0 = l * V;
  const bool fl = f0 < F;
  const int fo = fl ? f0 : 0;
  const int ent = tb8_entry(l);

  int s_i = 0, e_i = 0;
  if (lane < nrows) {
    s_i = rowptr[r0 + lane];
    e_i = rowptr[r0 + lane + 1];
  }
  const int len_i = e_i - s_i;
  const bool long_i = len_i > kT1;  // belongs to the unit table
  rows[lane] = make_int4(s_i, e_i, 0, 0);

  int a = 0;
  while (a < nrows) {
    const int s_a = __shfl(s_i, a, 64);
    // first row >= a that cannot join the run: long, or it would overflow the LDS tile, or past the end
    const unsigned long long brk = __ballot(lane >= a && (long_i || (e_i - s_a) > kCap || lane >= nrows));
    const int b = brk ? (__ffsll((long long)brk) - 1) : kRowsPerWave;
    if (b == a) {
      a++;
      continue;
    }
    const int e_b = __shfl(e_i, b - 1, 64);
    const int cnt = e_b - s_a;
    if (cnt == 0) {
      a = b;
      continue;
    }
    // ---- stage {col, local row} of every nnz of the run: per 64-nnz pass the row owners drop their row number at their
    //      first nnz, an inclusive prefix max turns the marks into the row of every nnz
    int carry = a;
    for (int t0 = 0; t0 < cnt; t0 += kWave) {
      __builtin_amdgcn_wave_barrier();
      mark[lane] = -1;
      __builtin_amdgcn_wave_barrier();
      const int d = s_i - s_a - t0;
      if (lane >= a && lane < b && len_i > 0 && d >= 0 && d < kWave) mark[d] = lane;
      __builtin_amdgcn_wave_barrier();
      int rv = mark[lane];
#pragma unroll
      for (int sh = 1; sh < kWave; sh <<= 1) {
        const int o = __shfl_up(rv, sh, kWave);
        if (lane >= sh) rv = max(rv, o);
      }
      rv = max(rv, carry);
      carry = __shfl(rv, kWave - 1, kWave);
      const int t = t0 + lane;
      if (t < cnt) tile[t] = make_int2(ld_stream(col + s_a + t), rv);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- group g takes the contiguous range [ps, pe) of the run, 8 nnz per batch
    const int ps = (int)(((long long)g * cnt) / NG), pe = (int)(((long long)(g + 1) * cnt) / NG);
    for (int p = ps; p < pe; p += kSfU) {
      int2 cr[kSfU];
      float av[kSfU][V], b[kSfU][V], pt[kSfU];
      int mv[kSfU][V];
#pragma unroll
      for (int q = 0; q < kSfU; q++) cr[q] = tile[min(p + q, pe - 1)];
#pragma unroll
      for (int q = 0; q < kSfU; q++) load_vec_gather<V>(D2 + (int64_t)cr[q].x * F + fo, b[q]);
      // D1 slices: consecutive nnz share their row, so most of these are the same address and hit L1 (rows are consecutive:
      // sequential, L2-friendly reads).  Loading only when the row changes inside the batch and copying otherwise was
      // measured no faster (1M graph 462 vs 468 us, products-shaped 2347 vs 2279) and hipcc 7.2 miscompiled the copies for G = 64
#pragma unroll
      for (int q = 0; q < kSfU; q++) {
        load_vec_rowop<V>(D1 + (int64_t)(r0 + cr[q].y) * F + fo, av[q]);
        if constexpr (MASK) load_vec<V>(E + (int64_t)(r0 + cr[q].y) * F + fo, mv[q]);
      }
#pragma unroll
      for (int q = 0; q < kSfU; q++) {
        float s = 0.0f;
#pragma unroll
        for (int v = 0; v < V; v++) {
          if constexpr (MASK) {
            if (fl && mv[q][v] == cr[q].x) s = __builtin_fmaf(av[q][v], b[q][v], s);
          } else {
            if (fl) s = __builtin_fmaf(av[q][v], b[q][v], s);
          }
        }
        pt[q] = s;
      }
      float tot = tb8<G>(pt, l);
      // the lane that holds entry `ent` needs that entry's row for the MEAN scale
      if constexpr (MEAN) {
        int rsel = cr[0].y;
#pragma unroll
        for (int q = 1; q < kSfU; q++) rsel = (ent == q) ? cr[q].y : rsel;
        const int4 rw = rows[rsel];
        tot /= (float)(rw.y - rw.x);
      }
      if (l < 8 && p + ent < pe) res[p + ent] = tot;
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < cnt; t += kWave) out[s_a + t] = res[t];
    __builtin_amdgcn_wave_barrier();
    a = b;
  }
}

template <int G, int V, bool MEAN, bool MASK>
__global__ __launch_bounds__(kBlock, 5) void sddmm_fused(int M, int F, int nbu, int rpw, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col, const float *__restrict__ D1,
                                                         const float *__restrict__ D2, const int *__restrict__ E,
                                                         float *__restrict__ out, const UnitTab ut) {
  __shared__ SfLds lds;
  if ((int)blockIdx.x < nbu) {
    sddmm_units_body<G, V, MEAN, MASK>(blockIdx.x, nbu, lds, F, rowptr, col, D1, D2, E, out, ut);
  } else {
    int rb = blockIdx.x - nbu;
    const int nbr = gridDim.x - nbu;
    const int per = nbr / 8;
    if (rb < per * 8) rb = (rb % 8) * per + rb / 8;  // a contiguous eighth of the row blocks per XCD (spmm_fused)
    sddmm_rows_body<G, V, MEAN, MASK>(rb, rpw, lds, M, F, rowptr, col, D1, D2, E, out);
  }
}

template <int G, bool MEAN>
static int launch_sddmm_fused(int64_t M, int64_t F, int64_t nnz, const int *rowptr, const int *col, const float *D1,
                              const float *D2, float *out, const PlanHdr *plan, const dgsSpmmPlanInfo *info,
                              hipStream_t st) {
  const char *pb = reinterpret_cast<const char *>(plan);
  const PlanLayout PL = plan_layout(nnz);
  const UnitTab ut{&plan->n_units, &plan->n_long, plan->xcd_start, reinterpret_cast<const int4 *>(pb + PL.off_units), nullptr};
  int rpw = kRowsPerWave;
  while (rpw > 8 && M / rpw < 8192) rpw >>= 1;
  const int rows_per_block = (kBlock / kWave) * rpw;
  const int64_t nbr = (M + rows_per_block - 1) / rows_per_block;
  int64_t ub = ((int64_t)info->n_units + 3) / 4;
  ub = (ub + 7) & ~int64_t(7);
  const int nbu = (int)(ub < DGS_NBU ? (ub < 8 ? 8 : ub) : DGS_NBU);
  hipLaunchKernelGGL((sddmm_fused<G, 4, MEAN, false>), dim3((unsigned)(nbr + nbu)), dim3(kBlock), 0, st, (int)M, (int)F, nbu,
                     rpw, rowptr, col, D1, D2, (const int *)nullptr, out, ut);
  return check_launch();
}

}  // namespace dgs
