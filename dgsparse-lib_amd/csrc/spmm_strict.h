// spmm_strict.h -- strict-order sum / mean: every (row, feature) is ONE sequential chain in CSR order, whatever the row
// length, i.e. literally algorithm 0 (reference include/cuda/spmm_cuda.cuh:27-47; host twin example/util/sp_util.hpp:73-83).
// Opt-in through the `algorithm` argument (DGS_ALG_STRICT_SUM: fmaf chain = what nvcc makes of the reference kernel;
// DGS_ALG_STRICT_NOFMA: separately rounded product and add = what g++ makes of the reference's host loop).  The default
// schedule keeps that order only for rows up to T1 nnz and folds longer rows with a fixed tree, which is closer to the
// exact sum but up to ~1e-5 away from the reference's own chain on rows of 10^4 nnz.
//
// A chain of L dependent adds cannot be cut, so the parallelism of a long row is (a) its FEATURES - independent chains -
// and (b) everything that is not the chain: the gathers.  Schedule (included by spmm_impl.h, shares its row blocks):
//
//   spmm_classify_strict   rows > T1 become units {row, first nnz, nnz, slice}: rows up to 256 nnz one unit (the whole
//                          feature tile), longer rows 4 units - a unit owns a SLICE of the row's features, so the waves that
//                          work a long row gather narrow pieces of MANY dense rows at once.  Hub rows (> 2048 nnz) are kept in
//                          six length classes, each in its own region of the table, and are taken longest first.
//   unit waves             (the persistent unit blocks of spmm_fused_strict) rows up to 2048 nnz: one WAVE per unit, no
//                          block-level sync: rounds of kUS gathers per lane in the usual lane mapping (GP lanes x V floats per
//                          nnz, 64/GP nnz per load instruction) -> transposed through the wave's LDS region in two halves ->
//                          the CHAIN lanes (one per feature of the slice) walk the half in nnz order: one ds_read (x, and w
//                          beside it for narrow slices) + one fma per nnz; the next round's gathers are issued as soon as a
//                          half has left the registers.  Hub rows: one WORKGROUP per unit (strict_hub_coop): four waves
//                          gather (128 KB in flight per row), the tile sits feature-major in LDS, wave 0 chains with one
//                          ds_read_b128 of x and one of w per four steps.  (Tiles narrower than 64 floats or scalar lanes: up
//                          to 16 wave-level slices per hub row instead.)  All slices of a row run on one XCD.
//   row blocks             unchanged: rows <= T1 are sequential chains already.
//
// Nothing is combined afterwards: no partial rows, no combine launch, bit-identical results from run to run and for any
// grid shape.
#pragma once

namespace dgs {

#ifndef DGS_STRICT_PRIO
#define DGS_STRICT_PRIO 1
#endif
#ifndef DGS_STRICT_DBG
#define DGS_STRICT_DBG 0  // experiment builds: 1 = no chain, 2 = no gathers, 3 = no LDS writes and no chain
#endif
constexpr int kUS = 8;                 // gathers in flight per lane of a strict unit wave
constexpr int kStrictXFloats = 2048;   // per-wave LDS: one half round of gathered rows (8 KB; with w interleaved when narrow)
constexpr int kStrictWFloats = 128;    // ... + the weights of a half round when they are not interleaved
constexpr int kStrictMid = 256;        // rows longer than this are cut into (up to) 4 feature slices
constexpr int kStrictHub = 2048;       // ... longer than this into (up to) 16
constexpr int kStrictWaveFloats = kStrictXFloats + kStrictWFloats;  // per-wave region of the wave-level units
constexpr int kStrictBlockFloats = 8800;  // the block-cooperative hub rounds need 16 x 516 + 512 floats (35.2 KB: 4 workgroups per CU)
static_assert(kStrictBlockFloats >= (kBlock / kWave) * kStrictWaveFloats, "strict LDS");
struct StrictLds {
  alignas(16) float f[kStrictBlockFloats];
  __device__ float *wave_region(int wave) { return f + wave * kStrictWaveFloats; }
};
// Hub rows are taken longest first (a 50 k-nnz row is the critical path of the whole call): class c holds the rows with
// kStrictHub << c < nnz <= kStrictHub << (c + 1) (the last class: everything longer), each class has its own region of the
// unit table behind the front entries, sized for the worst case, and the unit waves walk class 5, 4, ... 0, then the front.
constexpr int kHubClasses = 6;
struct HubTab {
  int base[kHubClasses];  // first table entry of each class region
  int mid_top;            // the 4-slice units grow down from here (the whole-tile units grow up from entry 0)
};
static inline HubTab hub_tab(int64_t nnz, int shub, int thub) {
  HubTab t;
  int64_t o = nnz / kT1 + 2;
  t.mid_top = (int)o;
  for (int c = 0; c < kHubClasses; c++) {
    t.base[c] = (int)o;
    o += (int64_t)shub * (nnz / ((int64_t)thub << c)) + 16;
  }
  return t;
}
__host__ __device__ __forceinline__ int hub_class(int len, int thub) {
  int c = 0;
  while (c < kHubClasses - 1 && len > (thub << (c + 1))) c++;
  return c;
}

constexpr int strict_smid(int G) { return G < 4 ? G : 4; }
// hub rows: feature tiles of >= 32 floats with 16-byte lanes are cut into 4 slices (2 for a 32-float tile), each worked by a whole workgroup
// (strict_hub_coop: four waves gather, one chains); narrower tiles into up to 16 wave-level slices
constexpr bool strict_coop(int G, int V) { return V == 4 && G >= 8; }
constexpr int strict_shub(int G, int V) { return strict_coop(G, V) ? (G >= 16 ? 4 : 2) : (G < 16 ? G : 16); }

// One wave, one feature slice [fbase, fbase + GP*V) of one row [p0, p0+len): returns the chain results in the CHAIN
// layout: lane c < CL holds features fbase + c*VP .. + VP-1 (VP = 1 unless the slice is wider than 64 floats).
template <int V, int GP, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void strict_row(const int p0, const int len, const int fbase, const int lane, const int N,
                                           const int *__restrict__ col, const float *__restrict__ val,
                                           const float *__restrict__ B, float *xb,
                                           float (&acc)[(GP * V > 64) ? GP * V / 64 : 1]) {
  constexpr int NGP = kWave / GP;              // nnz per load instruction
  constexpr int W = GP * V;                    // floats of the slice
  constexpr int VP = W > 64 ? W / 64 : 1;      // floats per chain lane
  constexpr int CL = W / VP;                   // chain lanes
  constexpr bool WI = HAS_VAL && W <= 16;      // narrow slice: w sits behind x in the LDS row of every nnz (one ds_read2)
  constexpr int RS = WI ? 2 * W : W;           // floats per nnz in LDS
  constexpr int H = kUS / 2, NH = H * NGP, NR = kUS * NGP;
  static_assert(NH * RS <= kStrictXFloats && (WI || !HAS_VAL || NH <= kStrictWFloats), "strict LDS region too small");
  float *wb = xb + kStrictXFloats;
  const int gp = lane / GP, lp = lane % GP;
  const int f0 = fbase + lp * V;
  const float *Bl = B + (f0 < N ? f0 : 0);
#pragma unroll
  for (int v = 0; v < VP; v++) acc[v] = 0.0f;

  int c[kUS];
  float w[kUS], x[kUS][V];
  // (col, val) and gathers of round 0; slots past the end of the row repeat its last nnz (never chained)
#pragma unroll
  for (int q = 0; q < kUS; q++) {
    const int i = p0 + min(q * NGP + gp, len - 1);
    c[q] = ld_stream(col + i);
    w[q] = HAS_VAL ? ld_stream(val + i) : 1.0f;
  }
#pragma unroll
  for (int q = 0; q < kUS; q++) load_vec_gather<V>(Bl + (int64_t)c[q] * N, x[q]);

  for (int r0 = 0; r0 < len; r0 += NR) {
    const int cnt = min(NR, len - r0);
    // (col, val) of the next round: in flight behind this round's gathers
    int cn[kUS];
    float wn[kUS];
#pragma unroll
    for (int q = 0; q < kUS; q++) {
      const int i = p0 + min(r0 + NR + q * NGP + gp, len - 1);
      cn[q] = ld_stream(col + i);
      wn[q] = HAS_VAL ? ld_stream(val + i) : 1.0f;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int qq = 0; qq < H; qq++) {
        const int q = h * H + qq, i = qq * NGP + gp;
        if (DGS_STRICT_DBG == 3) { if (x[q][0] == 1234.5f) xb[0] = x[q][0]; continue; }
        store_vec<V>(xb + i * RS + lp * V, x[q]);
        if constexpr (WI) {
          float ww[V];
#pragma unroll
          for (int v = 0; v < V; v++) ww[v] = w[q];
          store_vec<V>(xb + i * RS + W + lp * V, ww);
        } else if constexpr (HAS_VAL) {
          if (lp == 0) wb[i] = w[q];
        }
      }
      // this half has left the registers: its slots take the next round's gathers, which fly under the chain below
#pragma unroll
      for (int qq = 0; qq < H; qq++) {
        const int q = h * H + qq;
        if (DGS_STRICT_DBG != 2) load_vec_gather<V>(Bl + (int64_t)cn[q] * N, x[q]);
        w[q] = wn[q];
      }
      __builtin_amdgcn_wave_barrier();
      const int nh = (DGS_STRICT_DBG == 1 || DGS_STRICT_DBG == 3) ? 0 : min(NH, cnt - h * NH);
      if (CL == kWave || lane < CL) {
        // the chain: one LDS read (x, and w beside it) + one fma per nnz.  Batches of CB steps, the reads of the next batch
        // issued before the fmas of the current one, so the LDS latency is paid once per half round, not once per batch
        constexpr int CB = 8;
        const float *xr = xb + lane * VP;
        auto rd = [&](int i, float (&xv)[VP], float &wv) {
          load_vec<VP>(xr + i * RS, xv);
          wv = WI ? xr[i * RS + W] : (HAS_VAL ? wb[i] : 1.0f);
        };
        float xa[CB][VP], wa[CB], xn[CB][VP], wn2[CB];
        auto rdb = [&](int i0, float (&xx)[CB][VP], float (&ww)[CB]) {
#pragma unroll
          for (int u = 0; u < CB; u++) rd(i0 + u, xx[u], ww[u]);
        };
        auto fmab = [&](const float (&xx)[CB][VP], const float (&ww)[CB]) {
#pragma unroll
          for (int u = 0; u < CB; u++) {
#pragma unroll
            for (int v = 0; v < VP; v++) acc[v] = chain_step<FMA>(ww[u], xx[u][v], acc[v]);
          }
        };
        int i = 0;
        if (nh >= CB) {
          // ping-pong between two register batches (no copies): per nnz one ds_read(2) and one fma
          rdb(0, xa, wa);
          while (true) {
            if (i + 2 * CB <= nh) rdb(i + CB, xn, wn2);
            fmab(xa, wa);
            i += CB;
            if (i + CB > nh) break;
            if (i + 2 * CB <= nh) rdb(i + CB, xa, wa);
            fmab(xn, wn2);
            i += CB;
            if (i + CB > nh) break;
          }
        }
        for (; i < nh; i++) {
          float xv[VP], wv;
          rd(i, xv, wv);
#pragma unroll
          for (int v = 0; v < VP; v++) acc[v] = chain_step<FMA>(wv, xv[v], acc[v]);
        }
      }
    }
  }
}

// One strict unit: the slice `sl` of `S` of the feature tile starting at tbase.
template <int V, int GP, bool MEAN, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void strict_unit(const int row, const int p0, const int len, const int tbase, const int sl,
                                            const int lane, const int N, const int *__restrict__ col,
                                            const float *__restrict__ val, const float *__restrict__ B,
                                            float *__restrict__ C, float *xb) {
  constexpr int W = GP * V, VP = W > 64 ? W / 64 : 1, CL = W / VP;
  const int fbase = tbase + sl * W;
  float acc[VP];
  strict_row<V, GP, HAS_VAL, FMA>(p0, len, fbase, lane, N, col, val, B, xb, acc);
  const int f = fbase + lane * VP;
  if (lane < CL && f < N) {
    if constexpr (MEAN) {
      const float d = (float)len;
#pragma unroll
      for (int v = 0; v < VP; v++) acc[v] /= d;
    }
    store_vec_stream<VP>(C + (int64_t)row * N + f, acc);
  }
}

// Hub rows, block-cooperative: ONE workgroup per (row, slice of the feature tile).  A chain of L steps is a chain of L LDS
// reads as well, and an LDS instruction costs its cycles whatever the number of active lanes: sixteen 4-feature slices per
// row, each chained by its own wave (the first version), made the LDS pipeline the bound (8 chaining waves per CU = 32
// cycles per step) and the call 3.5x slower than the default schedule.  Here the four waves of the workgroup gather, the
// gathered rows are laid out FEATURE-major in LDS, and wave 0 alone chains W = 16 .. 64 features with one ds_read_b128 of x
// and one of w per FOUR steps.
//
// Round 4 (experiments/lds_dma_gather.cpp part 2, profiles/r04_lds_dma_gather.txt): with the fabric saturated by the rest of
// the launch a gather takes ~2.5 - 3 us to come back whatever is done about it (an L2 prefetcher on the same XCD changes
// nothing: hits queue behind everybody's misses), so a row is fed at (bytes in flight for it) / 3 us and nothing else
// matters - Little's law.  Two changes follow.  (a) TWO register sets of kUS gathers per lane, used by alternating phases of
// NRB nnz: a set has two phases to land instead of one, i.e. twice the bytes in flight per row (256 KB over the four slices
// at N = 64).  (b) The chain itself: the ISA of the first version waited for the NEXT batch's LDS reads before the first fma
// of the current one (the prefetch sat in a conditional block, so the merged wait count was the worst case of both paths):
// ~11 clocks per link.  The window below is straight-line - every read is issued, its index clamped into the tile - and
// rolls four b128 pairs (16 links) ahead of the fmas.
template <int GP, bool MEAN, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void strict_hub_coop(const int row, const int p0, const int len, const int tbase, const int sl,
                                                const int N, const int *__restrict__ col, const float *__restrict__ val,
                                                const float *__restrict__ B, float *__restrict__ C, float *lds) {
  constexpr int V = 4, NWG = kBlock / kWave - 1;  // gather waves (wave 0 chains and does nothing else)
  constexpr int NGP = kWave / GP, W = GP * V;   // nnz per load instruction; floats (= chain lanes) of the slice
  constexpr int NWV = kUS * NGP;                // nnz per gather wave and phase
  constexpr int NRB = NWG * NWV;                // nnz per workgroup phase
  constexpr int LD = NRB + 4;                   // row pitch of the feature-major tile: conflict-free ds_read_b128 across lanes
  static_assert(W * LD + NRB <= kStrictBlockFloats && W <= kWave && NRB % 16 == 0, "strict_hub_coop LDS");
  float *xt = lds, *wt = lds + W * LD;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int fbase = tbase + sl * W;
  const int nph = (len + NRB - 1) / NRB;
  // Roles: the two kinds of waves share nothing but the tile and two barriers per phase (A: tile written, B: tile chained), so
  // the kernel's register budget is the larger of the two roles, not their sum (the first round-4 version, every wave with two
  // gather sets AND the chain window, spilled the chain's addresses to scratch and drained vmcnt inside the chain loop).
  if (wave == 0) {
    float acc = 0.0f;
    const float *xr = xt + (lane < W ? lane : 0) * LD;
    const float4 *xr4 = reinterpret_cast<const float4 *>(__builtin_assume_aligned(xr, 16));
    const float4 *wt4 = reinterpret_cast<const float4 *>(__builtin_assume_aligned(wt, 16));
    // the chain is the critical path of the whole call and a dependent sequence: give it the SIMD's issue slots ahead of the
    // waves that share them (DGS_STRICT_PRIO=0 builds measure the difference)
    if (DGS_STRICT_PRIO) __builtin_amdgcn_s_setprio(3);
    for (int ph = 0; ph < nph; ph++) {
      const int cnt = min(NRB, len - ph * NRB);
      __syncthreads();  // A
      // Rolling window of P b128 pairs (x of this lane's feature, w broadcast) = 4 P links ahead of the fmas.  Every read is
      // issued (the window over-reads up to 4 P floats past the phase: still inside the block's LDS, never chained) and the
      // sched_barriers pin reads and fmas where they are written - left alone, hipcc gathers the reads at the loop top and
      // waits for all of them before the first fma
      constexpr int P = 4;
      static_assert(W * LD + NRB + 4 * P <= kStrictBlockFloats, "chain window over-read");
      const int nq = cnt >> 2;  // whole quads of this phase
      float4 xq[P], wq[P];
#pragma unroll
      for (int p = 0; p < P; p++) {  // same issue order as the loop body: its first wait is then lgkmcnt(2 P - 2) on both edges
        xq[p] = xr4[p];
        wq[p] = HAS_VAL ? wt4[p] : make_float4(1.f, 1.f, 1.f, 1.f);
        __builtin_amdgcn_sched_barrier(0);
      }
      int i = 0;
      for (; i + P <= nq; i += P) {
#pragma unroll
        for (int p = 0; p < P; p++) {
          acc = chain_step<FMA>(wq[p].x, xq[p].x, acc);
          acc = chain_step<FMA>(wq[p].y, xq[p].y, acc);
          acc = chain_step<FMA>(wq[p].z, xq[p].z, acc);
          acc = chain_step<FMA>(wq[p].w, xq[p].w, acc);
          __builtin_amdgcn_sched_barrier(0);
          xq[p] = xr4[i + P + p];
          if constexpr (HAS_VAL) wq[p] = wt4[i + P + p];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      for (int k = i * 4; k < cnt; k++) acc = chain_step<FMA>(HAS_VAL ? wt[k] : 1.0f, xr[k], acc);
      __syncthreads();  // B
    }
    if (DGS_STRICT_PRIO) __builtin_amdgcn_s_setprio(0);
    if (lane < W && fbase + lane < N) {
      if constexpr (MEAN) acc /= (float)len;
      float o[1] = {acc};
      store_vec_stream<1>(C + (int64_t)row * N + fbase + lane, o);
    }
    return;
  }
  const int gp = lane / GP, lp = lane % GP;
  const int f0 = fbase + lp * V;
  const float *Bl = B + (f0 < N ? f0 : 0);
  const int mine = (wave - 1) * NWV + gp;  // this lane's nnz of gather q inside a phase: mine + q * NGP
  float xa[kUS][V], xb[kUS][V];  // the two gather sets (even / odd phases)
  int ce[kUS], co[kUS];          // columns of the next gathers of the even / odd set
  float wn[kUS];                 // weights of the phase written NEXT
  // slots past the end of the row repeat its last nnz (never chained)
  auto load_cols = [&](int ph, int (&c)[kUS]) {
#pragma unroll
    for (int q = 0; q < kUS; q++) c[q] = ld_stream(col + p0 + min(ph * NRB + mine + q * NGP, len - 1));
  };
  auto load_w = [&](int ph) {
#pragma unroll
    for (int q = 0; q < kUS; q++) wn[q] = HAS_VAL ? ld_stream(val + p0 + min(ph * NRB + mine + q * NGP, len - 1)) : 1.0f;
  };
  auto issue = [&](float (&x)[kUS][V], const int (&c)[kUS]) {
#pragma unroll
    for (int q = 0; q < kUS; q++) load_vec_gather<V>(Bl + (int64_t)c[q] * N, x[q]);
  };
  // Loads return in order, so whatever a phase waits for must be OLDER than the gathers it wants to keep in flight: the
  // columns (and weights) a phase needs were requested one phase earlier, BEFORE that phase's gathers.
  auto phase = [&](const int ph, float (&x)[kUS][V], const int (&cuse)[kUS], int (&cload)[kUS]) {
#pragma unroll
    for (int q = 0; q < kUS; q++) {
      const int i = mine + q * NGP;
#pragma unroll
      for (int v = 0; v < V; v++) xt[(lp * V + v) * LD + i] = x[q][v];
      if (HAS_VAL && lp == 0) wt[i] = wn[q];
    }
    __syncthreads();  // A: the tile of phase ph is complete
    load_cols(ph + 3, cload);
    load_w(ph + 1);
    __builtin_amdgcn_sched_barrier(0);
    issue(x, cuse);   // this set's next gathers (phase ph + 2) fly under two chains
    __syncthreads();  // B: the chain has left the tile
  };
  load_cols(0, ce);
  load_cols(1, co);
  load_w(0);
  __builtin_amdgcn_sched_barrier(0);
  issue(xa, ce);
  __builtin_amdgcn_sched_barrier(0);  // (hipcc issued set b first: set a's writes then waited for everything)
  issue(xb, co);
  __builtin_amdgcn_sched_barrier(0);
  load_cols(2, ce);
  // (an odd last phase is peeled: with `if (ph + 1 < nph)` inside the loop the CFG has a path from one even phase straight into
  // the next, on which set a's gathers are the youngest loads, and hipcc's merged wait at the loop header drains everything)
  int ph = 0;
  for (; ph + 1 < nph; ph += 2) {
    phase(ph, xa, ce, co);
    phase(ph + 1, xb, co, ce);
  }
  if (ph < nph) phase(ph, xa, ce, co);
}

// Unit blocks of the strict fused launch.  Order of work: hub classes longest first, then the 4-slice units, then the
// whole-tile ones.  All slices of a row read the SAME rows of the dense operand, so they must share an L2: a row's group
// of slices goes to ONE XCD (block b runs on XCD b % 8 - observed placement, a speed hint only) and to neighbouring waves
// of it, which run in step.  (First version: slices dealt round-robin over all waves = over all 8 XCDs - every line of a hub
// row was fetched from memory 8 times, 7.8 GB per call on the headline graph, and the call took 1.4 ms.)
template <int G, int V, bool MEAN, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void spmm_units_strict_body(int bid, int nblocks, StrictLds &lds, int N,
                                                       const int *__restrict__ col, const float *__restrict__ val,
                                                       const float *__restrict__ B, float *__restrict__ C,
                                                       const SpmmWs *__restrict__ hdr, const int4 *__restrict__ units,
                                                       const HubTab ht) {
  constexpr int SM = strict_smid(G), SH = strict_shub(G, V);
  constexpr bool COOP = strict_coop(G, V);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *xb = lds.wave_region(wave);
  const int tbase = blockIdx.y * G * V;
  const int nx = (nblocks & 7) == 0 ? 8 : 1;  // XCDs the mapping distinguishes
  const int x = bid % nx;
  auto run = [&](const int4 d) {  // wave-level unit {row, first nnz, nnz, slice | slices << 8}
    const int S = d.w >> 8, sl = d.w & 255;
    if (S == 1) strict_unit<V, G, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, sl, lane, N, col, val, B, C, xb);
    if constexpr (SM > 1) {
      if (S == SM && S != 1) strict_unit<V, G / SM, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, sl, lane, N, col, val, B, C, xb);
    }
    if constexpr (!COOP && SH > SM) {
      if (S == SH) strict_unit<V, G / SH, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, sl, lane, N, col, val, B, C, xb);
    }
  };
  // one segment: `cnt` units in groups of gs (a row's slices), group g at table entries first(g) .. first(g) + gs - 1; the
  // worker (a wave, or a whole workgroup for cooperative hub units) is slot `slot` of `nslots` on its XCD
  auto segment = [&](int cnt, int gs, int base, bool down, int slot, int nslots, int &rot, auto &&fn) {
    const int ngroups = cnt / gs;
    const int gx = ngroups > x ? (ngroups - x + nx - 1) / nx : 0;  // groups of this XCD: x, x + nx, ...
    const int ux = gx * gs;
    int u = slot - rot;
    if (u < 0) u += nslots;
    for (; u < ux; u += nslots) {
      const int g = x + nx * (u / gs), j = u % gs;
      fn(units[(down ? base - (g + 1) * gs : base + g * gs) + j]);
    }
    rot = (rot + ux) % nslots;
  };
  const int s = (bid / nx) * (kBlock / kWave) + wave, SP = (nblocks / nx) * (kBlock / kWave);  // wave slots of this XCD
  int rot = 0;  // slots used up by earlier segments (keeps the deal round-robin across segments)
  if constexpr (COOP) {
    const int sb = bid / nx, SPB = nblocks / nx;  // workgroup slots of this XCD
    int rotb = 0;
    auto coop = [&](const int4 d) {
      strict_hub_coop<G / SH, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, d.w & 255, N, col, val, B, C, lds.f);
    };
#pragma unroll
    for (int c = kHubClasses - 1; c >= 0; c--) segment(hdr->hub[c], SH, ht.base[c], false, sb, SPB, rotb, coop);
    __syncthreads();  // the last chain has left the LDS before the waves reuse it one by one
    rot = (rotb * (kBlock / kWave)) % SP;
  } else {
#pragma unroll
    for (int c = kHubClasses - 1; c >= 0; c--) segment(hdr->hub[c], SH, ht.base[c], false, s, SP, rot, run);
  }
  segment(hdr->n_pslots, SM, ht.mid_top, true, s, SP, rot, run);
  segment(hdr->n_units, 1, 0, false, s, SP, rot, run);
}

// Unit table of the strict schedule.  Whole-tile units grow up from entry 0 and 4-slice units down from mid_top (together
// at most nnz / T1 entries); same structure as spmm_classify (a block owns 4096 consecutive rows, per-thread counts -> block
// scan -> ONE atomicAdd per block and kind).  Hub rows are rare (hundreds in a million rows): one atomicAdd per row on the
// counter of its length class.
static __global__ __launch_bounds__(kBlock) void spmm_classify_strict(int M, int t1, int tmid, int thub, int smid, int shub,
                                                                      const HubTab ht, const int *__restrict__ rowptr,
                                                                      SpmmWs *__restrict__ hdr, int4 *__restrict__ units) {
  __shared__ int s_f[kBlock / kWave], s_m[kBlock / kWave];
  __shared__ int s_fbase, s_mbase;
  const int tid = blockIdx.x * kBlock * kK0Rows + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int fm = 0, mm = 0;
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < kK0Rows; i++) {
    const int r = i * kBlock + tid;
    if (r < M) {
      const int len = rowptr[r + 1] - rowptr[r];
      if (len > t1) {
        if (len <= tmid) fm++;
        else if (len <= thub) mm += smid;
        mask |= 1u << i;
      }
    }
  }
  int fi = fm, mi = mm;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int tf = __shfl_up(fi, d, kWave), tm = __shfl_up(mi, d, kWave);
    if (lane >= d) {
      fi += tf;
      mi += tm;
    }
  }
  if (lane == kWave - 1) {
    s_f[wave] = fi;
    s_m[wave] = mi;
  }
  __syncthreads();
  int fo = 0, ft = 0, mo = 0, mt = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; w++) {
    if (w < wave) {
      fo += s_f[w];
      mo += s_m[w];
    }
    ft += s_f[w];
    mt += s_m[w];
  }
  if (threadIdx.x == 0) {
    s_fbase = ft ? atomicAdd(&hdr->n_units, ft) : 0;
    s_mbase = mt ? atomicAdd(&hdr->n_pslots, mt) : 0;
  }
  __syncthreads();
  if (!mask) return;
  int foff = s_fbase + fo + fi - fm;
  int moff = s_mbase + mo + mi - mm;
  while (mask) {
    const int i = __ffs((int)mask) - 1;
    mask &= mask - 1;
    const int r = i * kBlock + tid;
    const int rs = rowptr[r], len = rowptr[r + 1] - rs;
    if (len > thub) {
      const int c = hub_class(len, thub);
      const int o = ht.base[c] + atomicAdd(&hdr->hub[c], shub);
      for (int s = 0; s < shub; s++) units[o + s] = make_int4(r, rs, len, s | (shub << 8));
    } else if (len > tmid) {
      const int o = ht.mid_top - moff - smid;
      for (int s = 0; s < smid; s++) units[o + s] = make_int4(r, rs, len, s | (smid << 8));
      moff += smid;
    } else {
      units[foff++] = make_int4(r, rs, len, 1 << 8);
    }
  }
}

}  // namespace dgs
