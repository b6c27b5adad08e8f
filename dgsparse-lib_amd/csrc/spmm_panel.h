// spmm_panel.h -- column-panel SpMM for DENSE graphs (hundreds of nnz per row, dense operand far larger than L2).
//
// The row-stream schedule of spmm_impl.h gathers every B row from the fabric: on a Reddit-shaped input (233 k rows,
// ~490 nnz per row, N = 128) that is 58 GB of gathers per call at the ~7 TB/s the L2-miss path delivers.  This
// kernel blocks the COLUMN space instead: B is walked in panels of a few thousand rows (a few MB, resident in every
// XCD's 4 MiB L2) and all workgroups of the chip sweep the panels together, so a panel is fetched from memory once
// per XCD and then gathered from L2.  What makes that possible without re-reading / re-writing C for every panel is
// the LDS: 160 KiB per CU x 256 CUs hold the fp32 accumulators of 64 k rows (N = 128) at a time -- a "super-block"
// of rows -- and the panel loop runs inside the kernel with the accumulators resident.
//
//   workgroup = 1024 threads = 16 waves, one workgroup per CU, owns R = 32768/N consecutive rows of the super-block:
//     acc[R][N] fp32 in LDS (128 KiB; max: value + 16-bit arg position, R = 21845/N rows; min: + 32-bit arg id),
//     cur[r] = cursor into the row's CSR segment, nextc[r] = column at the cursor
//   for each panel p (columns < pend = (p+1)*pcols) the lane groups (G lanes x 4 floats = one row, as in the
//   row-stream kernels) take row VISITS from an LDS counter, longest rows first:
//     a row with nextc[r] >= pend is skipped (two LDS reads); otherwise the group loads the G (col,val) pairs at the
//     cursor -- one visit AHEAD, while it still works on the previous row -- counts how many columns are < pend,
//     gathers those B rows (kPU in flight) into the row's accumulator and advances the cursor.
//   then the R rows are written to C once (mean: divided by the row length).
//
// Every nnz of a row is consumed exactly once and in CSR order whatever the column order is (the cursor only moves
// forward over a prefix, the last panel has pend = INT_MAX): unsorted rows cost cache hits, never correctness, and a
// row's accumulator is ONE fma chain in CSR order -- the same chain as the reference's sequential fold.
//
// Balance: visits are handed out dynamically, so a panel step costs a workgroup (sum of its visits)/groups plus one
// visit of tail; the rows are ranked by length first (LDS counting sort) because the NG groups of a wave run in
// lockstep and take neighbouring ranks.  Rows longer than `tlong` are left out (never visited, not stored): the
// caller runs them through the unit path of spmm_impl.h.
//
// Panels are swept in step by a SOFT barrier: a workgroup signals a panel step when it hands out its last visit and
// starts step g when `lead` fewer steps have been signalled by everybody (bounded spin).  It is a speed hint only --
// results never depend on it, and the spin is bounded, so a workgroup that is not co-resident cannot deadlock.
#pragma once
#include "dgs_common.h"

namespace dgs {

constexpr int kPanelBlock = 1024;
constexpr int kPW = kPanelBlock / kWave;  // waves per workgroup
constexpr int kPanelAccBytes = 128 * 1024;  // accumulator budget the DISPATCH rule was calibrated with (reuse estimate)
constexpr int kPanelLdsBytes = 160 * 1024 - 512;  // what a workgroup may really take: accumulators + per-row state
constexpr int kPanelRowState = 6 * 4;     // deg, order, cur, rend, nextc, rbeg: 24 bytes per row, carved from the same LDS
constexpr int kPanelRMax = 1024;          // R <= workgroup size (one thread per row in the set-up phases)
#ifndef DGS_PU
#define DGS_PU 8
#endif
constexpr int kPU = DGS_PU;  // B-row gathers in flight per lane inside a visit


__device__ __forceinline__ int dev_load_relaxed(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PanelLds {  // the small arrays; the accumulators follow in dynamic LDS
  int *deg, *order, *cur, *rend, *nextc, *rbeg;
};

// N is the width of the feature tile this launch covers (<= 256, the LDS row length); ld is the row stride of B, C
// and E in memory (the full feature count: wider operands are swept one 256-feature tile per launch, with the
// pointers advanced by the caller).
// One visit: fold the in-panel prefix of the chunk (c,w) loaded at cur[r] into acc[r].  Masked sum (kOpMaskSum: Em =
// saved arg ids, orow = this output row) gathers the arg-id row next to the grad row and gates the fma.  Groups without a row pass
// r = -1 and c = INT_MAX everywhere (cnt = 0).  All 64 lanes must call it together.
template <int G, int OP, bool HAS_VAL>
__device__ __forceinline__ void panel_visit(int r, int c, float w, int pend, int N, int ld, int lig, int gbase, int f0,
                                            bool active, uint64_t gmask, const PanelLds &L, float *acc, unsigned short *acce,
                                            const int *__restrict__ col, const float *__restrict__ val,
                                            const float *__restrict__ B, const int *__restrict__ Em, int orow) {
  constexpr int V = 4;
  constexpr bool ARG = (OP == DGS_MAX || OP == DGS_MIN);
  constexpr int PU = kPU;
  const bool have = r >= 0;
  float a[V] = {0.f, 0.f, 0.f, 0.f};
  // max/min: the arg is kept as the POSITION of the winning nnz inside its row (16 bits: rows swept here have at most
  // tlong < 65535 nnz; 0xFFFF = none yet), 6 bytes per element instead of 8, and becomes a column id at write-out.
  // (Round 1 kept 32-bit ids for min because the packing spilled ~50 VGPRs; with the round-2 visit loop it spills 13 and
  // the extra rows per workgroup win: Reddit-shaped N = 128, 5.9 -> 5.4 ms.)
  constexpr bool E16 = ARG;
  int ae[V];
#pragma unroll
  for (int v = 0; v < V; v++) ae[v] = E16 ? 0xFFFF : -1;
  float4 *ap = reinterpret_cast<float4 *>(acc + (size_t)(have ? r : 0) * N + f0);
  uint2 *ep = reinterpret_cast<uint2 *>(acce + (size_t)(have ? r : 0) * N + f0);
  int4 *ep32 = reinterpret_cast<int4 *>(reinterpret_cast<int *>(acce) + (size_t)(have ? r : 0) * N + f0);
  const int rb = (E16 && have) ? L.rbeg[r] : 0;
  if (have && active) {
    const float4 t = *ap;
    a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
    if constexpr (ARG) {
      if constexpr (E16) {
        const uint2 te = *ep;
        ae[0] = te.x & 0xFFFF; ae[1] = te.x >> 16; ae[2] = te.y & 0xFFFF; ae[3] = te.y >> 16;
      } else {
        const int4 te = *ep32;
        ae[0] = te.x; ae[1] = te.y; ae[2] = te.z; ae[3] = te.w;
      }
    }
  }
  int pos = have ? L.cur[r] : 0;
  const int e = have ? L.rend[r] : 0;
  int cnt;
  bool again;
  do {
    const uint64_t bal = __ballot(c < pend);
    cnt = __popcll((bal >> gbase) & gmask);
    // a full chunk means the row may have more columns in this panel: fetch the next chunk NOW, so that its (HBM)
    // latency runs under the gathers of this one instead of stalling the wave between two chunks of a long row
    again = __any(cnt == G);
    int cn = INT_MAX;
    float wn = 0.f;
    if (again) {
      const int idx = pos + cnt + lig;
      const bool ok = idx < e;
      cn = ok ? ld_stream(col + idx) : INT_MAX;
      if constexpr (HAS_VAL) wn = ok ? ld_stream(val + idx) : 0.f;
    }
    for (int j = 0; __any(j < cnt); j += PU) {
      float x[PU][V];
      float wj[PU];
      int cj[PU];
      int mk[PU][V];
      // all the (column, value) shuffles first, then the gathers: in one loop every gather waited for its own ds_bpermute
      // (an lgkmcnt(0) per entry: 8 serial LDS-crossbar latencies per batch; ISA read, late round 3)
#pragma unroll
      for (int u = 0; u < PU; u++) {
        const int src = gbase + ((j + u) & (G - 1));
        cj[u] = __shfl(c, src);
        if constexpr (HAS_VAL) wj[u] = __shfl(w, src);
        else wj[u] = 1.f;
      }
#pragma unroll
      for (int u = 0; u < PU; u++) {
        if (j + u < cnt && active) {
          load_vec<V>(B + (int64_t)cj[u] * ld + f0, x[u]);
          if constexpr (OP == kOpMaskSum) load_vec<V>(Em + (int64_t)cj[u] * ld + f0, mk[u]);
        } else if constexpr (!ARG) {  // sum: a padded step is fma(0, 0, a) = a
          wj[u] = 0.f;
#pragma unroll
          for (int v = 0; v < V; v++) x[u][v] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < PU; u++) {
        if constexpr (ARG) {
          if (j + u < cnt && active) {
#pragma unroll
            for (int v = 0; v < V; v++)
              reduce_step<OP>(a[v], ae[v], wj[u], x[u][v], E16 ? pos + j + u - rb : cj[u]);
          }
        } else if constexpr (OP == kOpMaskSum) {
          if (j + u < cnt && active) {
#pragma unroll
            for (int v = 0; v < V; v++)
              if (mk[u][v] == orow) a[v] = __builtin_fmaf(wj[u], x[u][v], a[v]);
          }
        } else {
#pragma unroll
          for (int v = 0; v < V; v++) a[v] = __builtin_fmaf(wj[u], x[u][v], a[v]);
        }
      }
    }
    pos += cnt;
    if (again) {
      c = cn;
      if constexpr (HAS_VAL) w = wn;
    }
  } while (again);
  // here cnt < G for every group: lane cnt holds the column at the new cursor (INT_MAX past the row end)
  const int nc = __shfl(c, gbase + cnt);
  if (have) {
    if (lig == 0) {
      L.cur[r] = pos;
      L.nextc[r] = nc;
    }
    if (active) {
      *ap = make_float4(a[0], a[1], a[2], a[3]);
      if constexpr (E16) *ep = make_uint2((unsigned)ae[0] | ((unsigned)ae[1] << 16), (unsigned)ae[2] | ((unsigned)ae[3] << 16));
      else if constexpr (ARG) *ep32 = make_int4(ae[0], ae[1], ae[2], ae[3]);
    }
  }
}

template <int G, int OP, bool HAS_VAL>
__global__ __launch_bounds__(kPanelBlock) void spmm_panel(int M, int N, int ld, int R, int tlong, int pcols, int npanels,
                                                          int nsb, int lead, const int *__restrict__ rowptr,
                                                          const int *__restrict__ col, const float *__restrict__ val,
                                                          const float *__restrict__ B, float *__restrict__ C,
                                                          int *__restrict__ E, int *arrivals, const Epi epi) {
  constexpr int V = 4;
  constexpr bool ARG = (OP == DGS_MAX || OP == DGS_MIN);
  __shared__ int s_ctr;
  DGS_DYN_SHARED(panel_dyn);
  float *acc = reinterpret_cast<float *>(panel_dyn);        // [R][N]
  unsigned short *acce = reinterpret_cast<unsigned short *>(acc + (size_t)R * N);  // [R][N] arg positions (max/min)
  // per-row state behind the accumulators (the whole 160 KiB is one budget: every row slot more is a row less to sweep
  // B for again - Reddit-shaped, N = 128: 304 rows per workgroup = 3 super-blocks instead of 4 with 256)
  constexpr int kEB = ARG ? 6 : 4;
  int *s_deg = reinterpret_cast<int *>(panel_dyn + (((size_t)R * N * kEB + 15) & ~size_t(15)));
  int *s_order = s_deg + R, *s_cur = s_order + R, *s_rend = s_cur + R, *s_nextc = s_rend + R, *s_rbeg = s_nextc + R;
  const PanelLds L{s_deg, s_order, s_cur, s_rend, s_nextc, s_rbeg};

  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int lig = lane & (G - 1), gbase = lane & ~(G - 1);
  const int f0 = lig * V;
  const bool active = f0 < N;
  const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1);
  const int n4 = N / V;

  bool giveup = false;  // thread 0 only: soft barrier abandoned after a timed-out wait
  for (int sb = 0; sb < nsb; ++sb) {
    const int64_t row0 = ((int64_t)sb * gridDim.x + blockIdx.x) * R;
    __syncthreads();
    for (int i = tid; i < R * n4; i += kPanelBlock) {
      const float z = reduce_init<OP>();
      reinterpret_cast<float4 *>(acc)[i] = make_float4(z, z, z, z);
      if constexpr (ARG) reinterpret_cast<uint2 *>(acce)[i] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    // ---- per-row state; rank the workgroup's rows by length (longest first; absent / too-long rows last) ----
    if (tid < R) {
      int d = -1, s = 0, e = 0, c = INT_MAX;
      const int64_t row = row0 + tid;
      if (row < M) {
        s = rowptr[row];
        e = rowptr[row + 1];
        d = e - s;
        if (d > tlong) {
          d = -1;
          e = s;
        } else if (d > 0) {
          c = col[s];
        }
      }
      s_deg[tid] = d;
      s_cur[tid] = s;
      s_rbeg[tid] = s;
      s_rend[tid] = e;
      s_nextc[tid] = c;
    }
    if (tid == 0) s_ctr = 0;
    __syncthreads();
    if (tid < R) {
      const int my = s_deg[tid];
      int rank = 0;
      for (int j = 0; j < R; j++) {
        const int o = s_deg[j];
        rank += (o > my) || (o == my && j < tid);
      }
      s_order[rank] = tid;
    }
    __syncthreads();

    for (int p = 0; p < npanels; ++p) {
      const int pend = (p == npanels - 1) ? INT_MAX : (p + 1) * pcols;
      // hand out visits; chunk loads run one visit ahead (two named buffers, no register rotation)
      auto grab = [&](int &r, int &c, float &w) {
        bool need = true;
        r = -1;
        while (__any(need)) {
          int t = R;
          if (need && lig == 0) {
            t = atomicAdd(&s_ctr, 1);
            if (t == R - 1) atomicAdd(arrivals, 1);  // last visit handed out: signal this panel step early
          }
          t = __shfl(t, gbase);
          if (need) {
            if (t >= R) {
              need = false;
            } else {
              const int i = s_order[t];
              if (s_nextc[i] < pend) {
                r = i;
                need = false;
              }
            }
          }
        }
        c = INT_MAX;
        w = HAS_VAL ? 0.f : 1.f;
        if (r >= 0) {
          const int idx = s_cur[r] + lig;
          if (idx < s_rend[r]) {
            c = ld_stream(col + idx);
            if constexpr (HAS_VAL) w = ld_stream(val + idx);
          }
        }
      };
      int r0, c0, r1, c1;
      float w0, w1;
      grab(r0, c0, w0);
      for (;;) {
        grab(r1, c1, w1);
        panel_visit<G, OP, HAS_VAL>(r0, c0, w0, pend, N, ld, lig, gbase, f0, active, gmask, L, acc, acce, col, val, B, E,
                                    (int)row0 + r0);
        if (!__any(r1 >= 0)) break;
        grab(r0, c0, w0);
        panel_visit<G, OP, HAS_VAL>(r1, c1, w1, pend, N, ld, lig, gbase, f0, active, gmask, L, acc, acce, col, val, B, E,
                                    (int)row0 + r1);
        if (!__any(r0 >= 0)) break;
      }
      __syncthreads();
      if (tid == 0) {
        s_ctr = 0;
        const int64_t target = ((int64_t)sb * npanels + p + 2 - lead) * gridDim.x;
        if (target > 0 && !giveup && !(sb == nsb - 1 && p == npanels - 1)) {
          // every poll is a device-scope load (~1-2 us): 256 of them bound a wait to ~0.4 ms.  A wait that runs out means
          // the other workgroups are not co-resident (CUs taken by another stream, e.g. an overlapped collective): stop
          // waiting for the rest of the launch instead of paying the timeout at every step.
          int spins = 0;
          while (dev_load_relaxed(arrivals) < target && spins < 256) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
          }
          giveup = spins >= 256;
        }
      }
      __syncthreads();
    }

    for (int i = tid; i < R * n4; i += kPanelBlock) {
      const int r = i / n4;
      const int64_t row = row0 + r;
      if (row < M && s_deg[r] >= 0) {
        const float4 t = reinterpret_cast<float4 *>(acc)[i];
        float o[V] = {t.x, t.y, t.z, t.w};
        if constexpr (ARG) {
          int oe[V];
          if constexpr (ARG) {
            const uint2 te = reinterpret_cast<uint2 *>(acce)[i];
            const unsigned pe[V] = {te.x & 0xFFFF, te.x >> 16, te.y & 0xFFFF, te.y >> 16};
#pragma unroll
            for (int v = 0; v < V; v++) oe[v] = (pe[v] == 0xFFFF) ? -1 : col[s_rbeg[r] + (int)pe[v]];
          } else {
            const int4 te = reinterpret_cast<int4 *>(acce)[i];
            oe[0] = te.x; oe[1] = te.y; oe[2] = te.z; oe[3] = te.w;
          }
          if (s_deg[r] == 0) {  // empty row: 0, not the identity (include/cuda/spmm_cuda.cuh:32,49-51)
#pragma unroll
            for (int v = 0; v < V; v++) o[v] = 0.f;
          }
          store_vec_stream<V>(E + row * ld + (int64_t)(i - r * n4) * V, oe);
        }
        if constexpr (OP == DGS_MEAN) {
          const int dg = s_deg[r];
          if (dg > 0) {
            const float d = (float)dg;
#pragma unroll
            for (int v = 0; v < V; v++) o[v] /= d;
          }
        }
        if constexpr (OP == DGS_SUM || OP == DGS_MEAN) epi_apply<V>(o, row, (i - r * n4) * V, epi);
        store_vec_stream<V>(C + row * ld + (int64_t)(i - r * n4) * V, o);
      }
    }
  }
}

struct PanelPlan {
  bool use;
  int R, tlong, pcols, npanels, nsb, nwg, lead;
  size_t lds;
};

}  // namespace dgs
