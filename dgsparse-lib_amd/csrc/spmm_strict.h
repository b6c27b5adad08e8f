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
#ifndef DGS_HUB_COOP_V2
#define DGS_HUB_COOP_V2 1  // 0: the round-3 hub workgroup (one gather set per wave, wave 0 gathers and chains)
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
// (HubTab / hub_tab / hub_class / HubArg: spmm_impl.h, next to the classify pass that fills the tables)
constexpr int strict_smid(int G) { return G < 4 ? G : 4; }
// hub rows: feature tiles of >= 16 floats with 16-byte lanes are cut into slices of >= 16 floats (4 slices from 64 floats on), each worked by a whole workgroup
// (strict_hub_coop: four waves gather, one chains); narrower tiles into up to 16 wave-level slices
constexpr bool strict_coop(int G, int V) { return DGS_HUB_COOP_V2 ? true : (V == 4 && G >= 4); }  // (the round-3 workgroup: 16-byte lanes only)
// lanes of a feature slice: 16 floats per slice where the tile has them (one ds_read_b128 of x feeds four links of 16 chains),
// a quarter of the tile for the wide ones
constexpr int strict_hub_gp(int G, int V) { return V == 4 ? (G >= 16 ? G / 4 : (G >= 4 ? 4 : G)) : (G >= 16 ? 16 : G); }
constexpr int strict_shub(int G, int V) { return strict_coop(G, V) ? G / strict_hub_gp(G, V) : (G < 16 ? G : 16); }

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
        // the chain: one LDS read (x, and w beside it) + one fma per nnz, through a rolling window of CB links: every read is
        // issued (index clamped into the half round) and pinned where it is written, so the waits are counted.  (Round 3 kept
        // two batches and prefetched the next one conditionally: the wait in front of the first fma, merged over both paths,
        // then covered the prefetch as well - no overlap, ~20 clocks per link by the ISA.)
        constexpr int CB = 8;
        const float *xr = xb + lane * VP;
        auto rd = [&](int i, float (&xv)[VP], float &wv) {
          load_vec<VP>(xr + i * RS, xv);
          wv = WI ? xr[i * RS + W] : (HAS_VAL ? wb[i] : 1.0f);
        };
        float xw[CB][VP], ww[CB];
#pragma unroll
        for (int u = 0; u < CB; u++) {
          rd(min(u, NH - 1), xw[u], ww[u]);
          __builtin_amdgcn_sched_barrier(0);
        }
        int i = 0;
        for (; i + CB <= nh; i += CB) {
#pragma unroll
          for (int u = 0; u < CB; u++) {
#pragma unroll
            for (int v = 0; v < VP; v++) acc[v] = chain_step<FMA>(ww[u], xw[u][v], acc[v]);
            __builtin_amdgcn_sched_barrier(0);
            rd(min(i + CB + u, NH - 1), xw[u], ww[u]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // the window holds links i .. i + CB - 1 already
#pragma unroll
        for (int u = 0; u < CB; u++)
          if (i + u < nh) {
#pragma unroll
            for (int v = 0; v < VP; v++) acc[v] = chain_step<FMA>(ww[u], xw[u][v], acc[v]);
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
                                            float *__restrict__ C, float *xb, const Epi &epi = Epi{}) {
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
    epi_apply<VP>(acc, row, f, epi);
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
constexpr int kHubBlockFloats = 7008;  // what strict_hub_coop needs at most (16 x 388 + 2 x 384 + 16): 28 KB, 5 workgroups per CU
constexpr int strict_hub_lds(int V, int GP, int us) {  // floats of LDS of one workgroup phase with `us` gathers per lane and set
  return GP * V * (3 * us * (kWave / GP) + 4) + 2 * 3 * us * (kWave / GP) + 16;
}
template <int V, int GP, bool MEAN, bool HAS_VAL, bool FMA, int LDSF = kStrictBlockFloats>
__device__ __forceinline__ void strict_hub_coop(const int row, const int p0, const int len, const int tbase, const int sl,
                                                const int N, const int *__restrict__ col, const float *__restrict__ val,
                                                const float *__restrict__ B, float *__restrict__ C, float *lds,
                                                const Epi &epi = Epi{}) {
  constexpr int NWG = kBlock / kWave - 1;       // gather waves (wave 0 chains and does nothing else)
  constexpr int NGP = kWave / GP, W = GP * V;   // nnz per load instruction; floats (= chain lanes) of the slice
  // gathers per lane and set: kUS where the tile then fits the launch's LDS (every shape the headline runs), half of it for the
  // narrow tiles whose 64 / GP rows per instruction make the phase long (N = 4, 8)
  constexpr int US = strict_hub_lds(V, GP, kUS) <= LDSF ? kUS : kUS / 2;
  constexpr int NWV = US * NGP;                 // nnz per gather wave and phase
  constexpr int NRB = NWG * NWV;                // nnz per workgroup phase
  constexpr int LD = NRB + 4;                   // row pitch of the feature-major tile: conflict-free ds_read_b128 across lanes
  static_assert(W * LD + 2 * NRB + 16 <= LDSF && W <= kWave && NRB % 16 == 0, "strict_hub_coop LDS");
  float *xt = lds, *wt = lds + W * LD;             // gathered rows (feature-major), weights of the phase being chained
  int *ct = reinterpret_cast<int *>(wt + NRB);     // columns of the phase whose gathers are issued next
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
    __syncthreads();  // P1 .. P4: the gather waves' prologue (column tiles of phases 0 and 1)
    __syncthreads();
    __syncthreads();
    __syncthreads();
    if (DGS_STRICT_PRIO) __builtin_amdgcn_s_setprio(3);
    for (int ph = 0; ph < nph; ph++) {
      const int cnt = min(NRB, len - ph * NRB);
      __syncthreads();  // A
      // Rolling window of P b128 pairs (x of this lane's feature, w broadcast) = 4 P links ahead of the fmas.  Every read is
      // issued (the window over-reads up to 4 P floats past the phase: still inside the block's LDS, never chained) and the
      // sched_barriers pin reads and fmas where they are written - left alone, hipcc gathers the reads at the loop top and
      // waits for all of them before the first fma
      constexpr int P = 4;
      static_assert(4 * P <= NRB, "chain window over-read");  // (past the weights: the column tile; past a feature row: the next one)
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
      epi_apply<1>(o, row, fbase + lane, epi);
      store_vec_stream<1>(C + (int64_t)row * N + fbase + lane, o);
    }
    return;
  }
  const int gp = lane / GP, lp = lane % GP;
  const int f0 = fbase + lp * V;
  const float *Bl = B + (f0 < N ? f0 : 0);
  const int mine = (wave - 1) * NWV + gp;  // this lane's nnz of gather q inside a phase: mine + q * NGP
  const int gt = (wave - 1) * kWave + lane;  // gather-thread index: the (col, val) stream of a phase is loaded coalesced,
  constexpr int NGT = NWG * kWave;           // CPT entries per gather thread, and handed round through LDS (ct / wt)
  constexpr int CPT = (NRB + NGT - 1) / NGT;
  constexpr bool CFULL = NRB % NGT == 0;     // (256-float tiles: 96 nnz per phase, half of the gather threads carry an entry)
  float xa[US][V], xb[US][V];  // the two gather sets (even / odd phases)
  int cv[CPT];                   // columns of the phase whose gathers are issued NEXT (two phases ahead of the chain)
  float wv[CPT];                 // weights of the phase written NEXT
  // slots past the end of the row repeat its last nnz (never chained)
  auto load_cols = [&](int ph) {
#pragma unroll
    for (int k = 0; k < CPT; k++) cv[k] = ld_stream(col + p0 + min(ph * NRB + min(k * NGT + gt, NRB - 1), len - 1));
  };
  auto load_w = [&](int ph) {
#pragma unroll
    for (int k = 0; k < CPT; k++) wv[k] = HAS_VAL ? ld_stream(val + p0 + min(ph * NRB + min(k * NGT + gt, NRB - 1), len - 1)) : 1.0f;
  };
  auto put_cols = [&]() {
#pragma unroll
    for (int k = 0; k < CPT; k++)
      if (CFULL || k * NGT + gt < NRB) ct[k * NGT + gt] = cv[k];
  };
  auto issue = [&](float (&x)[US][V]) {
    int c[US];
#pragma unroll
    for (int q = 0; q < US; q++) c[q] = ct[mine + q * NGP];
#pragma unroll
    for (int q = 0; q < US; q++) load_vec_gather<V>(Bl + (int64_t)c[q] * N, x[q]);
  };
  // Loads return in order, so whatever a phase waits for must be OLDER than the gathers it wants to keep in flight: the
  // (col, val) entries a phase needs were requested one phase earlier, BEFORE that phase's gathers.
  auto phase = [&](const int ph, float (&x)[US][V]) {
#pragma unroll
    for (int q = 0; q < US; q++) {
      const int i = mine + q * NGP;
#pragma unroll
      for (int v = 0; v < V; v++) xt[(lp * V + v) * LD + i] = x[q][v];
    }
    if constexpr (HAS_VAL) {
#pragma unroll
      for (int k = 0; k < CPT; k++)
        if (CFULL || k * NGT + gt < NRB) wt[k * NGT + gt] = wv[k];
    }
    put_cols();       // columns of phase ph + 2
    __syncthreads();  // A: the tile of phase ph is complete (and so is the column tile the gathers below read)
    load_cols(ph + 3);
    load_w(ph + 1);
    __builtin_amdgcn_sched_barrier(0);
    issue(x);         // this set's next gathers (phase ph + 2) fly under two chains
    __syncthreads();  // B: the chain has left the tile, every wave has read its columns
  };
  // prologue: gathers of phases 0 and 1 (their columns go through the tile like everybody's), (col, val) of what follows
  // (four extra barriers per row, matched by the chain wave: the column tile is written by all gather threads and read by all)
  load_cols(0);
  load_w(0);
  put_cols();
  __syncthreads();  // P1: columns of phase 0 in the tile
  issue(xa);
  __builtin_amdgcn_sched_barrier(0);
  load_cols(1);
  __syncthreads();  // P2: everybody has read them
  put_cols();
  __syncthreads();  // P3: columns of phase 1 in the tile
  load_cols(2);     // (before the gathers: the wait for them at phase 0 must not cover set b - loads return in order)
  __builtin_amdgcn_sched_barrier(0);
  issue(xb);
  __syncthreads();  // P4: ... and read (phase 0 overwrites them with those of phase 2)
  // (an odd last phase is peeled: with `if (ph + 1 < nph)` inside the loop the CFG has a path from one even phase straight into
  // the next, on which set a's gathers are the youngest loads, and hipcc's merged wait at the loop header drains everything)
  int ph = 0;
  for (; ph + 1 < nph; ph += 2) {
    phase(ph, xa);
    phase(ph + 1, xb);
  }
  if (ph < nph) phase(ph, xa);
}

// The round-3 form of the hub workgroup (every wave gathers ONE set of kUS rows per lane, wave 0 chains between two barriers;
// ~8 ns per link under load): kept selectable (-DDGS_HUB_COOP_V2=0) as the measured baseline of the round-4 version above
// and as its fallback.  It needs 16 x 516 + 512 floats of LDS, so only the strict launch (kStrictBlockFloats) can host it.
template <int GP, bool MEAN, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void strict_hub_coop_v1(const int row, const int p0, const int len, const int tbase, const int sl,
                                                const int N, const int *__restrict__ col, const float *__restrict__ val,
                                                const float *__restrict__ B, float *__restrict__ C, float *lds,
                                                   const Epi &epi = Epi{}) {
  constexpr int V = 4, NW = kBlock / kWave;
  constexpr int NGP = kWave / GP, W = GP * V;   // nnz per load instruction; floats (= chain lanes) of the slice
  constexpr int NWV = kUS * NGP;                // nnz per wave and round
  constexpr int NRB = NW * NWV;                 // nnz per workgroup round
  constexpr int LD = NRB + 4;                   // row pitch of the feature-major tile: conflict-free ds_read_b128 across lanes
  static_assert(W * LD + NRB <= kStrictBlockFloats && W <= kWave, "strict_hub_coop LDS");
  float *xt = lds, *wt = lds + W * LD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gp = lane / GP, lp = lane % GP;
  const int fbase = tbase + sl * W;
  const int f0 = fbase + lp * V;
  const float *Bl = B + (f0 < N ? f0 : 0);
  const int mine = wave * NWV + gp;  // this lane's nnz of gather q inside a round: mine + q * NGP
  float acc = 0.0f;
  int c[kUS];
  float w[kUS], x[kUS][V];
#pragma unroll
  for (int q = 0; q < kUS; q++) {
    const int i = p0 + min(mine + q * NGP, len - 1);
    c[q] = ld_stream(col + i);
    w[q] = HAS_VAL ? ld_stream(val + i) : 1.0f;
  }
#pragma unroll
  for (int q = 0; q < kUS; q++) load_vec_gather<V>(Bl + (int64_t)c[q] * N, x[q]);
  for (int r0 = 0; r0 < len; r0 += NRB) {
    const int cnt = min(NRB, len - r0);
    int cn[kUS];
    float wn[kUS];
#pragma unroll
    for (int q = 0; q < kUS; q++) {
      const int i = p0 + min(r0 + NRB + mine + q * NGP, len - 1);
      cn[q] = ld_stream(col + i);
      wn[q] = HAS_VAL ? ld_stream(val + i) : 1.0f;
    }
    __syncthreads();  // the chain of the previous round has left the tile
#pragma unroll
    for (int q = 0; q < kUS; q++) {
      const int i = mine + q * NGP;
#pragma unroll
      for (int v = 0; v < V; v++) xt[(lp * V + v) * LD + i] = x[q][v];
      if (HAS_VAL && lp == 0) wt[i] = w[q];
    }
#pragma unroll
    for (int q = 0; q < kUS; q++) {  // the next round's gathers fly under the chain
      load_vec_gather<V>(Bl + (int64_t)cn[q] * N, x[q]);
      w[q] = wn[q];
    }
    __syncthreads();
    if (wave == 0 && lane < W) {
      // the chain is the critical path of the whole call and a dependent sequence: give it the SIMD's issue slots ahead of
      // the three other waves that share them (DGS_STRICT_PRIO=0 builds measure the difference)
      if (DGS_STRICT_PRIO) __builtin_amdgcn_s_setprio(3);
      const float *xr = xt + lane * LD;
      constexpr int CB = 4;  // b128 pairs per batch = 16 steps
      float4 xa[CB], wa[CB], xn[CB], wn4[CB];
      auto rdb = [&](int i0, float4 (&xx)[CB], float4 (&ww)[CB]) {
#pragma unroll
        for (int u = 0; u < CB; u++) {
          xx[u] = *reinterpret_cast<const float4 *>(xr + i0 + 4 * u);
          if constexpr (HAS_VAL) ww[u] = *reinterpret_cast<const float4 *>(wt + i0 + 4 * u);
          else ww[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        }
      };
      auto fmab = [&](const float4 (&xx)[CB], const float4 (&ww)[CB]) {
#pragma unroll
        for (int u = 0; u < CB; u++) {
          acc = chain_step<FMA>(ww[u].x, xx[u].x, acc);
          acc = chain_step<FMA>(ww[u].y, xx[u].y, acc);
          acc = chain_step<FMA>(ww[u].z, xx[u].z, acc);
          acc = chain_step<FMA>(ww[u].w, xx[u].w, acc);
        }
      };
      constexpr int ST = 4 * CB;
      int i = 0;
      if (cnt >= ST) {
        rdb(0, xa, wa);
        while (true) {
          if (i + 2 * ST <= cnt) rdb(i + ST, xn, wn4);
          fmab(xa, wa);
          i += ST;
          if (i + ST > cnt) break;
          if (i + 2 * ST <= cnt) rdb(i + ST, xa, wa);
          fmab(xn, wn4);
          i += ST;
          if (i + ST > cnt) break;
        }
      }
      for (; i < cnt; i++) acc = chain_step<FMA>(HAS_VAL ? wt[i] : 1.0f, xr[i], acc);
      if (DGS_STRICT_PRIO) __builtin_amdgcn_s_setprio(0);
    }
  }
  if (wave == 0 && lane < W && fbase + lane < N) {
    if constexpr (MEAN) acc /= (float)len;
    float o[1] = {acc};
    epi_apply<1>(o, row, fbase + lane, epi);
    store_vec_stream<1>(C + (int64_t)row * N + fbase + lane, o);
  }
}

// Work deal shared by the strict unit blocks and the hub blocks of the default launch.  One segment = `ngroups` groups of gs
// tasks (a row's feature slices); group g belongs to XCD g % nx, and the worker (a wave, or a whole workgroup for cooperative
// hub tasks) that is slot `slot` of `nslots` on XCD x takes every nslots-th task of that XCD's groups.  `rot` carries the deal
// round-robin from one segment to the next.  All slices of a row read the SAME rows of the dense operand, so they must share
// an L2: a row's group of slices goes to ONE XCD (block b runs on XCD b % 8 - observed placement, a speed hint only) and to
// neighbouring workers of it, which run in step.  (First version: slices dealt round-robin over all waves = over all 8 XCDs -
// every line of a hub row was fetched from memory 8 times, 7.8 GB per call on the headline graph, and the call took 1.4 ms.)
template <typename F>
__host__ __device__ __forceinline__ void strict_deal(int ngroups, int gs, int x, int nx, int slot, int nslots, int &rot, F &&fn) {
  const int gx = ngroups > x ? (ngroups - x + nx - 1) / nx : 0;  // groups of this XCD: x, x + nx, ...
  const int ux = gx * gs;
  int s0 = slot - rot;
  if (s0 < 0) s0 += nslots;
  // rounds of nslots tasks, dealt boustrophedon: tasks come longest first, so the worker that got the longest task of one
  // round gets the shortest of the next (a plain stride gave the slots of the 50 k-nnz row the 9th-longest row as well)
  for (int k = 0; k * nslots < ux; k++) {
    const int u = k * nslots + ((k & 1) ? nslots - 1 - s0 : s0);
    if (u < ux) fn(x + nx * (u / gs), u % gs);
  }
  rot = (rot + ux) % nslots;
}

// Hub rows (longer than the hub threshold): every one a sequential chain per feature, longest class first.  Used by the strict
// launch (all rows > kStrictHub) and by the DEFAULT sum / mean launches for the rows above the hub-chain threshold - the rows
// whose fixed-tree sum can be more than 1e-5 away from the reference's own sequential result.  Returns the
// wave-slot rotation the caller's following segments continue from.
template <int G, int V, bool MEAN, bool HAS_VAL, bool FMA, int LDSF = kStrictBlockFloats>
__device__ __forceinline__ int spmm_hub_body(int bid, int nblocks, float *ldsf, int N, const int *__restrict__ col,
                                             const float *__restrict__ val, const float *__restrict__ B,
                                             float *__restrict__ C, const HubArg &ha, const Epi &epi = Epi{}) {
  constexpr int SH = strict_shub(G, V);
  constexpr bool COOP = strict_coop(G, V);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tbase = blockIdx.y * G * V;
  const int nx = (nblocks & 7) == 0 ? 8 : 1;  // XCDs the mapping distinguishes
  const int x = bid % nx;
  const int SP = (nblocks / nx) * (kBlock / kWave);  // wave slots of this XCD
  int rot = 0;
  if constexpr (COOP) {
    const int sb = bid / nx, SPB = nblocks / nx;  // workgroup slots of this XCD
    int rotb = 0;
    for (int c = ha.ncls - 1; c >= 0; c--)
      strict_deal(ha.cnt[c], SH, x, nx, sb, SPB, rotb, [&](int g, int j) {
        const int4 d = ha.rows[ha.ht.base[c] + g];
#if DGS_HUB_COOP_V2
        strict_hub_coop<V, G / SH, MEAN, HAS_VAL, FMA, LDSF>(d.x, d.y, d.z, tbase, j, N, col, val, B, C, ldsf, epi);
#else
        static_assert(LDSF >= kStrictBlockFloats, "the round-3 hub workgroup needs the strict launch's LDS");
        strict_hub_coop_v1<G / SH, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, j, N, col, val, B, C, ldsf, epi);
#endif
      });
    __syncthreads();  // the last chain has left the LDS before the waves reuse it one by one
    rot = (rotb * (kBlock / kWave)) % SP;
  } else {
    static_assert(LDSF >= (kBlock / kWave) * kStrictWaveFloats, "wave-level hub slices need the strict kernel's LDS");
    const int s = (bid / nx) * (kBlock / kWave) + wave;
    float *xb = ldsf + wave * kStrictWaveFloats;
    for (int c = ha.ncls - 1; c >= 0; c--)
      strict_deal(ha.cnt[c], SH, x, nx, s, SP, rot, [&](int g, int j) {
        const int4 d = ha.rows[ha.ht.base[c] + g];
        strict_unit<V, G / SH, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, j, lane, N, col, val, B, C, xb, epi);
      });
  }
  return rot;
}

// Unit blocks of the strict fused launch.  Order of work: hub classes longest first, then the 4-slice units, then the
// whole-tile ones.
template <int G, int V, bool MEAN, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void spmm_units_strict_body(int bid, int nblocks, StrictLds &lds, int N,
                                                       const int *__restrict__ col, const float *__restrict__ val,
                                                       const float *__restrict__ B, float *__restrict__ C,
                                                       const SpmmWs *__restrict__ hdr, const int4 *__restrict__ units,
                                                       const HubTab ht, const StrictPlanHdr *__restrict__ sp) {
  constexpr int SM = strict_smid(G);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *xb = lds.wave_region(wave);
  const int tbase = blockIdx.y * G * V;
  const int nx = (nblocks & 7) == 0 ? 8 : 1;
  const int x = bid % nx;
  const int s = (bid / nx) * (kBlock / kWave) + wave, SP = (nblocks / nx) * (kBlock / kWave);  // wave slots of this XCD
  if (sp) {
    // over a cached plan (round 5): `units` is the plan's strict table - every row > T1, longest first: [hub rows | rows worked as
    // 4 feature slices | whole-tile rows], sizes in the header in front of it; nothing was classified for this call
    const int nh = sp->n_hub, nm = sp->n_mid, nw = sp->n_whole;
    int rot = spmm_hub_body<G, V, MEAN, HAS_VAL, FMA>(bid, nblocks, lds.f, N, col, val, B, C, HubArg{&sp->n_hub, units, HubTab{}, 1});
    strict_deal(nm, SM, x, nx, s, SP, rot, [&](int g, int j) {
      const int4 d = units[nh + g];
      strict_unit<V, G / SM, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, j, lane, N, col, val, B, C, xb);
    });
    strict_deal(nw, 1, x, nx, s, SP, rot, [&](int g, int) {
      const int4 d = units[nh + nm + g];
      strict_unit<V, G, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, 0, lane, N, col, val, B, C, xb);
    });
    return;
  }
  int rot = spmm_hub_body<G, V, MEAN, HAS_VAL, FMA>(bid, nblocks, lds.f, N, col, val, B, C, HubArg{hdr->hub, units, ht, kHubClasses});
  // 4-slice units {row, first nnz, nnz, -}: SM entries' worth of work per row, table grows down from mid_top
  strict_deal(hdr->n_pslots, SM, x, nx, s, SP, rot, [&](int g, int j) {  // (SM = 1 for one-lane tiles: the whole tile again)
    const int4 d = units[ht.mid_top - 1 - g];
    strict_unit<V, G / SM, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, j, lane, N, col, val, B, C, xb);
  });
  strict_deal(hdr->n_units, 1, x, nx, s, SP, rot, [&](int g, int) {
    const int4 d = units[g];
    strict_unit<V, G, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, 0, lane, N, col, val, B, C, xb);
  });
}

// Unit table of the strict schedule: one entry {row, first nnz, nnz, -} per row longer than T1.  Whole-tile rows grow up from
// entry 0 and the rows worked as 4 feature slices down from mid_top (together at most nnz / T1 entries); same structure as spmm_classify (a block owns 4096 consecutive rows, per-thread counts -> block
// scan -> ONE atomicAdd per block and kind).  Hub rows are rare (hundreds in a million rows): one atomicAdd per row on the
// counter of its length class.
static __global__ __launch_bounds__(kBlock) void spmm_classify_strict(int M, int t1, int tmid, int thub,
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
        else if (len <= thub) mm++;
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
      units[ht.base[c] + atomicAdd(&hdr->hub[c], 1)] = make_int4(r, rs, len, 0);
    } else if (len > tmid) {
      units[ht.mid_top - 1 - moff++] = make_int4(r, rs, len, 0);
    } else {
      units[foff++] = make_int4(r, rs, len, 0);
    }
  }
}

}  // namespace dgs
