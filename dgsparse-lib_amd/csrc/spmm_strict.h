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
//                          feature tile), up to 2048 nnz 4 units, longer ("hub") rows 16 units - a unit owns a SLICE of
//                          the row's features, so its wave gathers narrow pieces of MANY dense rows at once: a 50 k-nnz
//                          row has 16 waves x 8 KB of gathers in flight instead of one wave's 8 KB.  Hub units are
//                          written from the back of the table and taken first.
//   unit waves             (the persistent unit blocks of spmm_fused) one wave per unit, no block-level sync: rounds of
//                          kUS gathers per lane in the usual lane mapping (GP lanes x V floats per nnz, 64/GP nnz per
//                          load instruction) -> transposed through the wave's LDS region in two halves -> the CHAIN
//                          lanes (one per feature of the slice) walk the half in nnz order: one ds_read (x, and w beside
//                          it for narrow slices) + one fma per nnz.  The gathers of the next round are issued as soon as
//                          a half has left the registers, so they fly under the chain.
//   row blocks             unchanged: rows <= T1 are sequential chains already.
//
// Nothing is combined afterwards: no partial rows, no combine launch, bit-identical results from run to run and for any
// grid shape.
#pragma once

namespace dgs {

constexpr int kUS = 8;                 // gathers in flight per lane of a strict unit wave
constexpr int kStrictXFloats = 2048;   // per-wave LDS: one half round of gathered rows (8 KB; with w interleaved when narrow)
constexpr int kStrictWFloats = 128;    // ... + the weights of a half round when they are not interleaved
constexpr int kStrictMid = 256;        // rows longer than this are cut into (up to) 4 feature slices
constexpr int kStrictHub = 2048;       // ... longer than this into (up to) 16
struct StrictLds {
  float x[kBlock / kWave][kStrictXFloats + kStrictWFloats];
};

constexpr int strict_smid(int G) { return G < 4 ? G : 4; }
constexpr int strict_shub(int G) { return G < 16 ? G : 16; }

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
        load_vec_gather<V>(Bl + (int64_t)cn[q] * N, x[q]);
        w[q] = wn[q];
      }
      __builtin_amdgcn_wave_barrier();
      const int nh = min(NH, cnt - h * NH);
      if (CL == kWave || lane < CL) {
#pragma unroll 8
        for (int i = 0; i < nh; i++) {
          float xv[VP];
          load_vec<VP>(xb + i * RS + lane * VP, xv);
          const float wv = WI ? xb[i * RS + W + lane] : (HAS_VAL ? wb[i] : 1.0f);
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

// Unit blocks of the strict fused launch.  The table: [0, n_front) whole-tile and 4-slice units in row order, and
// [cap - n_back, cap) the 16-slice units of the hub rows, which every wave takes first (they are the longest chains).
template <int G, int V, bool MEAN, bool HAS_VAL, bool FMA>
__device__ __forceinline__ void spmm_units_strict_body(int bid, int nblocks, StrictLds &lds, int N,
                                                       const int *__restrict__ col, const float *__restrict__ val,
                                                       const float *__restrict__ B, float *__restrict__ C,
                                                       const SpmmWs *__restrict__ hdr, const int4 *__restrict__ units,
                                                       int cap) {
  constexpr int SM = strict_smid(G), SH = strict_shub(G);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *xb = lds.x[wave];
  const int n_front = hdr->n_units, n_back = hdr->n_long;
  const int nw = nblocks * (kBlock / kWave), w0 = bid * (kBlock / kWave) + wave;
  const int tbase = blockIdx.y * G * V;
  for (int k = w0; k < n_back + n_front; k += nw) {
    const int4 d = units[k < n_back ? cap - 1 - k : k - n_back];  // {row, first nnz, nnz, slice | slices << 8}
    const int S = d.w >> 8, sl = d.w & 255;
    if (S == 1) strict_unit<V, G, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, sl, lane, N, col, val, B, C, xb);
    if constexpr (SM > 1) {
      if (S == SM && S != 1) strict_unit<V, G / SM, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, sl, lane, N, col, val, B, C, xb);
    }
    if constexpr (SH > SM) {
      if (S == SH) strict_unit<V, G / SH, MEAN, HAS_VAL, FMA>(d.x, d.y, d.z, tbase, sl, lane, N, col, val, B, C, xb);
    }
  }
}

// Unit table of the strict schedule.  Same structure as spmm_classify (a block owns 4096 consecutive rows, per-thread
// counts -> block scan -> one atomicAdd per block and table end).
static __global__ __launch_bounds__(kBlock) void spmm_classify_strict(int M, int t1, int tmid, int thub, int smid, int shub,
                                                                      int cap, const int *__restrict__ rowptr,
                                                                      SpmmWs *__restrict__ hdr, int4 *__restrict__ units) {
  __shared__ int s_f[kBlock / kWave], s_b[kBlock / kWave];
  __shared__ int s_fbase, s_bbase;
  const int tid = blockIdx.x * kBlock * kK0Rows + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto slices = [&](int len) { return len > thub ? shub : (len > tmid ? smid : 1); };
  int fm = 0, bm = 0;
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < kK0Rows; i++) {
    const int r = i * kBlock + tid;
    if (r < M) {
      const int len = rowptr[r + 1] - rowptr[r];
      if (len > t1) {
        if (len > thub) bm += shub; else fm += slices(len);
        mask |= 1u << i;
      }
    }
  }
  int fi = fm, bi = bm;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int tf = __shfl_up(fi, d, kWave), tb = __shfl_up(bi, d, kWave);
    if (lane >= d) {
      fi += tf;
      bi += tb;
    }
  }
  if (lane == kWave - 1) {
    s_f[wave] = fi;
    s_b[wave] = bi;
  }
  __syncthreads();
  int fo = 0, bo = 0, ft = 0, bt = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; w++) {
    if (w < wave) {
      fo += s_f[w];
      bo += s_b[w];
    }
    ft += s_f[w];
    bt += s_b[w];
  }
  if (threadIdx.x == 0) {
    s_fbase = ft ? atomicAdd(&hdr->n_units, ft) : 0;
    s_bbase = bt ? atomicAdd(&hdr->n_long, bt) : 0;
  }
  __syncthreads();
  if (!mask) return;
  int foff = s_fbase + fo + fi - fm;
  int boff = s_bbase + bo + bi - bm;
  while (mask) {
    const int i = __ffs((int)mask) - 1;
    mask &= mask - 1;
    const int r = i * kBlock + tid;
    const int rs = rowptr[r], len = rowptr[r + 1] - rs;
    if (len > thub) {
      for (int s = 0; s < shub; s++) units[cap - 1 - (boff + s)] = make_int4(r, rs, len, s | (shub << 8));
      boff += shub;
    } else {
      const int S = slices(len);
      for (int s = 0; s < S; s++) units[foff + s] = make_int4(r, rs, len, s | (S << 8));
      foff += S;
    }
  }
}

}  // namespace dgs
