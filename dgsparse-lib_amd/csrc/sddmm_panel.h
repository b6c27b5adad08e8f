// sddmm_panel.h -- column-panel SDDMM for DENSE graphs: the twin of spmm_panel.h (read that header first).
//
//   out[p] = < D1[row(p),:], D2[col[p],:] >      (MEAN: / deg(row);  MASK: only features f with E[row,f] == col[p])
//
// The nnz-balanced kernel of sddmm.hip gathers every D2 row from the fabric (Reddit-shaped, F = 64: 29 GB at
// 6.8 TB/s = 4.2 ms).  Here D2 is swept in column panels that stay in the XCD L2s while the D1 rows (and, masked, the
// arg-id rows) of the workgroup's R rows sit in LDS for the whole sweep, so both operands of every dot product are
// on-chip reads.  Nothing is accumulated across panels -- each nnz is written exactly once.  Rows longer than
// `tlong` are skipped here and computed by the nnz-balanced kernel run with a row-length filter (sddmm.hip).
//
//   workgroup = 1024 threads, one per CU; owns R consecutive rows of the super-block: d1[R][F] (+ em[R][F]) in LDS
//   row table (non-empty rows up to tlong nnz): cursor, end, column at the cursor, parent row; ranked by length
//   per panel: lane groups take row visits from an LDS counter (chunk of G columns loaded one visit ahead),
//   gather the D2 rows of the columns below the panel end 8 at a time, dot them with the LDS-resident D1 slice,
//   reduce the 8 partial sums over the G lanes with a transposing butterfly (8+log2(G)-3 shuffles instead of
//   8*log2(G)) and store the 8 results; then advance the cursor.
//   soft barrier between panels exactly as in spmm_panel.h.
// F is the width of the feature tile of this launch (<= 256, the LDS row length), ld the row stride of D1, D2 and E.
#pragma once
#include "dgs_common.h"
#include "spmm_panel.h"

namespace dgs {

constexpr int kSdVMax = kPanelRMax;            // row slots per workgroup
constexpr int kSdPanelBytes = 128 * 1024;      // LDS for the D1 (+E) rows (+ 22 KB of row tables = 150 KB)
constexpr int kSdPU = 8;                       // D2-row gathers in flight per lane (fixed by the 8-way butterfly)

__device__ int g_sddmm_arrivals;  // soft-barrier counter (zeroed by a memset node before every launch).  The one piece of
// device-global state in the library (the SDDMM entry points have no workspace argument, like the reference's sddmm_cuda_csr):
// two panel sweeps running CONCURRENTLY on different streams share it.  It is a speed hint only - the spin on it is bounded and
// no result depends on it - and the sweep assumes the whole GPU to itself anyway (spmm_panel.h), so concurrent sweeps lose
// lockstep, never correctness.

template <int G, bool MEAN, bool MASK>
__global__ __launch_bounds__(kPanelBlock) void sddmm_panel(int M, int F, int ld, int pass, int R, int tlong, int pcols,
                                                           int npanels,
                                                           int nsb, int lead, const int *__restrict__ rowptr,
                                                           const int *__restrict__ col, const float *__restrict__ D1,
                                                           const float *__restrict__ D2, const int *__restrict__ E,
                                                           float *__restrict__ out, int *arrivals) {
  constexpr int V = 4;
  static_assert(G >= 8, "the 8-way transposing butterfly needs 8 lanes");
  __shared__ int s_len[kSdVMax], s_order[kSdVMax], s_cur[kSdVMax], s_end[kSdVMax], s_nextc[kSdVMax];
  __shared__ unsigned short s_par[kSdVMax];
  __shared__ int s_wsum[kPW];
  __shared__ int s_ctr, s_nv;
  DGS_DYN_SHARED(sd_dyn);
  float *d1 = reinterpret_cast<float *>(sd_dyn);           // [R][F]
  int *em = reinterpret_cast<int *>(d1 + (size_t)R * F);    // [R][F] (MASK only)

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  const int lig = lane & (G - 1), gbase = lane & ~(G - 1);
  const int f0 = lig * V;
  const bool active = f0 < F;
  const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1);
  const int n4 = F / V;
  // entry of a batch this lane ends up holding after the transposing butterfly
  const int bidx = ((lig & 1) ? 4 : 0) + ((lig & 2) ? 2 : 0) + ((lig & 4) ? 1 : 0);

  bool giveup = false;  // thread 0 only: soft barrier abandoned after a timed-out wait
  for (int sb = 0; sb < nsb; ++sb) {
    const int64_t row0 = ((int64_t)sb * gridDim.x + blockIdx.x) * R;
    __syncthreads();
    // ---- D1 (and arg-id) rows of this workgroup into LDS ----
    for (int i = tid; i < R * n4; i += kPanelBlock) {
      const int r = i / n4;
      const int64_t row = row0 + r;
      float4 t = make_float4(0, 0, 0, 0);
      int4 te = make_int4(-1, -1, -1, -1);
      if (row < M) {
        t = *reinterpret_cast<const float4 *>(D1 + row * ld + (int64_t)(i - r * n4) * V);
        if constexpr (MASK) te = *reinterpret_cast<const int4 *>(E + row * ld + (int64_t)(i - r * n4) * V);
      }
      reinterpret_cast<float4 *>(d1)[i] = t;
      if constexpr (MASK) reinterpret_cast<int4 *>(em)[i] = te;
    }
    // ---- row table: rows of 1..tlong nnz are swept here; longer rows are left to the nnz-balanced kernel (one visit
    //      of such a row would outlast a whole panel step of everybody else) ----
    int s = 0, e = 0, nseg = 0;
    if (tid < R && row0 + tid < M) {
      s = rowptr[row0 + tid];
      e = rowptr[row0 + tid + 1];
      nseg = (e > s && e - s <= tlong) ? 1 : 0;
    }
    int incl = nseg;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int t = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += t;
    }
    if (lane == kWave - 1) s_wsum[wv] = incl;
    if (tid == 0) s_ctr = 0;
    __syncthreads();
    int base = incl - nseg;
    int total = 0;
#pragma unroll
    for (int w = 0; w < kPW; w++) {
      if (w < wv) base += s_wsum[w];
      total += s_wsum[w];
    }
    if (nseg) {
      s_cur[base] = s;
      s_end[base] = e;
      s_len[base] = e - s;
      s_par[base] = (unsigned short)tid;
      s_nextc[base] = col[s];
    }
    if (tid == 0) s_nv = total;
    __syncthreads();
    const int nv = s_nv;
    for (int i = tid; i < nv; i += kPanelBlock) {
      const int my = s_len[i];
      int rank = 0;
      for (int j = 0; j < nv; j++) {
        const int o = s_len[j];
        rank += (o > my) || (o == my && j < i);
      }
      s_order[rank] = i;
    }
    __syncthreads();

    for (int p = 0; p < npanels; ++p) {
      const int pend = (p == npanels - 1) ? INT_MAX : (p + 1) * pcols;
      auto grab = [&](int &r, int &c) {
        bool need = true;
        r = -1;
        while (__any(need)) {
          int t = nv;
          if (need && lig == 0) {
            t = atomicAdd(&s_ctr, 1);
            if (t == nv - 1) atomicAdd(arrivals, 1);  // last visit handed out: signal this panel step early
          }
          t = __shfl(t, gbase);
          if (need) {
            if (t >= nv) {
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
        if (r >= 0) {
          const int idx = s_cur[r] + lig;
          if (idx < s_end[r]) c = ld_stream(col + idx);
        }
      };
      auto visit = [&](int r, int c) {
        const bool have = r >= 0;
        const int par = have ? (int)s_par[r] : 0;
        float a[V] = {0.f, 0.f, 0.f, 0.f};
        int am[V] = {-1, -1, -1, -1};
        if (have && active) {
          const float4 t = *reinterpret_cast<const float4 *>(d1 + (size_t)par * F + f0);
          a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
          if constexpr (MASK) {
            const int4 te = *reinterpret_cast<const int4 *>(em + (size_t)par * F + f0);
            am[0] = te.x; am[1] = te.y; am[2] = te.z; am[3] = te.w;
          }
        }
        int dg = 1;  // MEAN: divide by deg(row) (sddmm_cuda.cuh:266-272), as the nnz-balanced kernel does
        if constexpr (MEAN) {
          if (have) dg = rowptr[row0 + par + 1] - rowptr[row0 + par];
        }
        int pos = have ? s_cur[r] : 0;
        const int e2 = have ? s_end[r] : 0;
        int cnt;
        bool again;
        do {
          const uint64_t bal = __ballot(c < pend);
          cnt = __popcll((bal >> gbase) & gmask);
          for (int j = 0; __any(j < cnt); j += kSdPU) {
            float pt[kSdPU];
            float x[kSdPU][V];
            int cj[kSdPU];
            // all the column shuffles first, then the gathers: in one loop every gather waited for its own ds_bpermute
            // (an lgkmcnt(0) per entry: 8 serial LDS-crossbar latencies per batch; ISA read, late round 3)
#pragma unroll
            for (int u = 0; u < kSdPU; u++) cj[u] = __shfl(c, gbase + ((j + u) & (G - 1)));
#pragma unroll
            for (int u = 0; u < kSdPU; u++) {
              if (j + u < cnt && active) {
                load_vec<V>(D2 + (int64_t)cj[u] * ld + f0, x[u]);
              } else {
#pragma unroll
                for (int v = 0; v < V; v++) x[u][v] = 0.f;
              }
            }
#pragma unroll
            for (int u = 0; u < kSdPU; u++) {
              pt[u] = 0.f;
#pragma unroll
              for (int v = 0; v < V; v++) {
                if constexpr (MASK) {
                  if (am[v] == cj[u]) pt[u] = __builtin_fmaf(a[v], x[u][v], pt[u]);
                } else {
                  pt[u] = __builtin_fmaf(a[v], x[u][v], pt[u]);
                }
              }
            }
            // transposing butterfly: 8 partial sums x G lanes -> lane l holds the total of entry bidx(l)
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
            float tot;
            {
              const float send = (lig & 4) ? q2[0] : q2[1];
              const float keep = (lig & 4) ? q2[1] : q2[0];
              tot = keep + __shfl_xor(send, 4, 64);
            }
#pragma unroll
            for (int m = 8; m < G; m <<= 1) tot += __shfl_xor(tot, m, 64);
            if (lig < 8 && j + bidx < cnt) {
              // operands wider than 256 features are swept one tile per launch: pass bit 0 = add to the partial sum
              // of the earlier tiles, bit 1 = last tile (the MEAN division happens once, on the full dot product)
              if (pass & 1) tot += out[pos + j + bidx];
              if constexpr (MEAN) {
                if (pass & 2) tot /= (float)dg;
              }
              // uncounted store (dgs_common.h): a counted one would make hipcc drain vmcnt(0) - and with it the chunk
              // prefetch and the next batch of gathers - at every batch
              const float o1[1] = {tot};
#if defined(SD_NO_STORE)
              if (tot == 123.456f) store_vec_hidden<1>(out + pos + j + bidx, o1);
#elif defined(SD_PLAIN_STORE)
              out[pos + j + bidx] = tot;
#else
              store_vec_hidden<1>(out + pos + j + bidx, o1);
#endif
            }
          }
          pos += cnt;
          again = __any(cnt == G);
          if (again) {
            const int idx = pos + lig;
            c = (idx < e2) ? ld_stream(col + idx) : INT_MAX;
          }
        } while (again);
        const int nc = __shfl(c, gbase + cnt);
        if (have && lig == 0) {
          s_cur[r] = pos;
          s_nextc[r] = nc;
        }
      };
      int r0, c0, r1, c1;
      grab(r0, c0);
      for (;;) {
        grab(r1, c1);
        visit(r0, c0);
        if (!__any(r1 >= 0)) break;
        grab(r0, c0);
        visit(r1, c1);
        if (!__any(r0 >= 0)) break;
      }
      __syncthreads();
      if (tid == 0) {
        s_ctr = 0;
        if (nv == 0) atomicAdd(arrivals, 1);  // nothing to hand out: signal here
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
  }
}

}  // namespace dgs
