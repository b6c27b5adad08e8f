// spmm_plan.hip -- the cached locality plan of the row-stream SpMM schedule (built once per matrix, same lifetime as
// the CSC view a Storage keeps; new design, no reference counterpart: the reference has no per-matrix state at all).
//
// Why: on a 1M x 1M power-law graph the fused kernel moves 4.8x the algorithmic bytes over the L2-miss path, which is
// the bound (7.3 TB/s of 128-B fabric reads whatever the table size, experiments/gather_sizes.cpp); the only lever is
// fewer misses.  58 % of that graph's nnz sit in rows longer than 64 nnz, which the schedule already cuts into units
// whose partial rows are folded by spmm_combine.  Since the partials are paid for anyway, the units can be cut and
// ordered for the memory system instead of in row order:
//   * rows longer than `tslice` nnz are cut at COLUMN-SLICE boundaries (8 slices with equal reference counts, found
//     from a column histogram), then into chunks of <= 256 nnz; every unit is keyed (slice, first column);
//   * the unit table is sorted by that key and XCD x walks slice x front to back: its 4 MiB L2 only ever sees an
//     eighth of the dense operand, and the units in flight on it at any time cover a narrow column window.
// Values never depend on the plan beyond the (deterministic, fixed-tree) split of long rows, which the plan-free
// schedule has as well; max/min (value, E) stay bit-exact because partials are still folded in position order.
// Rows whose columns are not sorted are left whole-chunked (the cut needs sorted columns; any cut is CORRECT).
//
// Build = a handful of launches + three rocPRIM primitives on the caller's stream, in caller-provided memory.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "spmm_impl.h"

namespace dgs {

// the plan lists rows longer than kT1 as units while the fused kernel hands rows to the unit table from kT2: one threshold
static_assert(kT1 == kT2, "spmm_plan.hip assumes DGS_T2 == DGS_T1 (rows in (T1, T2] would be computed twice)");

struct PlanWs {  // build-time counters (zeroed by the first memset)
  int n_longlist;
  int n_hub;
  int pad[2];
};

struct PlanWsLayout {
  size_t off_cnt, off_cum, off_list, off_info, off_scan, off_keys_in, off_keys_out, off_units_in, off_hkeys_in, off_hkeys_out,
      off_hub_in, off_skeys_in, off_skeys_out, off_srows_in, off_tmp, tmp_bytes, total;
  int64_t cap_long;
};

struct I4Plus {
  __host__ __device__ int4 operator()(const int4 &a, const int4 &b) const {
    return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
};

static PlanWsLayout plan_ws_layout(int64_t K, int64_t nnz) {
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  const PlanLayout PL = plan_layout(nnz);
  PlanWsLayout L;
  L.cap_long = nnz / kT1 + 2;
  size_t t1 = 0, t2 = 0, t3 = 0;
  int *ip = nullptr;
  int4 *i4 = nullptr;
  unsigned long long *kp = nullptr;
  (void)rocprim::exclusive_scan(nullptr, t1, ip, ip, 0, (size_t)(K + 1), rocprim::plus<int>(), nullptr, false);
  (void)rocprim::exclusive_scan(nullptr, t2, i4, i4, make_int4(0, 0, 0, 0), (size_t)L.cap_long, I4Plus(), nullptr, false);
  (void)rocprim::radix_sort_pairs(nullptr, t3, kp, kp, i4, i4, (size_t)PL.max_units, 0, 37, nullptr, false);
  size_t t4 = 0;
  unsigned *hp = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t4, hp, hp, i4, i4, (size_t)PL.max_hub, 0, 32, nullptr, false);
  size_t t5 = 0;  // strict table: the rows longer than T1 sorted by length
  (void)rocprim::radix_sort_pairs(nullptr, t5, hp, hp, i4, i4, (size_t)L.cap_long, 0, 32, nullptr, false);
  L.tmp_bytes = t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3);
  if (t4 > L.tmp_bytes) L.tmp_bytes = t4;
  if (t5 > L.tmp_bytes) L.tmp_bytes = t5;
  size_t o = up(sizeof(PlanWs));
  L.off_cnt = o;          o += up((size_t)(K + 1) * 4);
  L.off_cum = o;          o += up((size_t)(K + 1) * 4);
  L.off_list = o;         o += up((size_t)L.cap_long * 4);
  L.off_info = o;         o += up((size_t)L.cap_long * 16);
  L.off_scan = o;         o += up((size_t)L.cap_long * 16);
  L.off_keys_in = o;      o += up((size_t)PL.max_units * 8);
  L.off_keys_out = o;     o += up((size_t)PL.max_units * 8);
  L.off_units_in = o;     o += up((size_t)PL.max_units * 16);
  L.off_hkeys_in = o;     o += up((size_t)PL.max_hub * 4);
  L.off_hkeys_out = o;    o += up((size_t)PL.max_hub * 4);
  L.off_hub_in = o;       o += up((size_t)PL.max_hub * 16);
  L.off_skeys_in = o;     o += up((size_t)L.cap_long * 4);
  L.off_skeys_out = o;    o += up((size_t)L.cap_long * 4);
  L.off_srows_in = o;     o += up((size_t)L.cap_long * 16);
  L.off_tmp = o;          o += up(L.tmp_bytes);
  L.total = o + 256;
  return L;
}

// ---- kernels ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void plan_hist(int nnz, int K, const int *__restrict__ col, int *__restrict__ cnt) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  for (int p = i; p < nnz; p += gridDim.x * kBlock) {
    const int c = col[p];
    if ((unsigned)c < (unsigned)K) atomicAdd(&cnt[c], 1);  // a bad column id must not become an out-of-bounds WRITE
  }
}

// Column grid: kPlanCells cells with equal reference counts; bounds[c] = first column with (references to columns
// below it) >= c * nnz / kPlanCells.  8 slices (one per XCD) of kPlanCells/8 cells each; a row is cut on the grid at a
// level that leaves it about `unit` nnz per cell: level j = (8 << j) cells, each the union of (16 >> j) finest cells.
__global__ void plan_bounds(int K, int nnz, const int *__restrict__ cum, PlanHdr *__restrict__ hdr, int *__restrict__ bounds,
                            int M, int ch, int t1, int tslice, int unit, int thub) {
  const int x = threadIdx.x;
  if (x == 0) {
    hdr->magic = kPlanMagic;
    hdr->version = 3;  // 3: slot -> long-row map and strict table behind the hub table (round 5)
    hdr->M = M;
    hdr->nnz = nnz;
    hdr->K = K;
    hdr->ch = ch;
    hdr->t1 = t1;
    hdr->tslice = tslice;
    hdr->unit = unit;
    hdr->thub = thub;
  }
  if (x > kPlanCells) return;
  int b;
  if (x == 0) b = 0;
  else if (x == kPlanCells) b = INT_MAX;
  else {
    const long long tgt = (long long)nnz * x / kPlanCells;
    int lo = 0, hi = K;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] < tgt) lo = mid + 1; else hi = mid;
    }
    b = lo;
  }
  bounds[x] = b;
  if ((x & (kPlanCells / 8 - 1)) == 0) hdr->slice_bound[x / (kPlanCells / 8)] = b;
}

// rows longer than t1: appended to a list (order irrelevant: the unit table is sorted later)
__global__ __launch_bounds__(kBlock) void plan_longlist(int M, int t1, const int *__restrict__ rowptr,
                                                        PlanWs *__restrict__ pw, int *__restrict__ list) {
  __shared__ int s_wsum[kBlock / kWave];
  __shared__ int s_base;
  const int r = blockIdx.x * kBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_long = r < M && (rowptr[r + 1] - rowptr[r]) > t1;
  const unsigned long long m = __ballot(is_long);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_wsum[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kBlock / kWave; w++) tot += s_wsum[w];
    s_base = tot ? atomicAdd(&pw->n_longlist, tot) : 0;
  }
  __syncthreads();
  if (!is_long) return;
  int off = s_base + before;
  for (int w = 0; w < wave; w++) off += s_wsum[w];
  list[off] = r;
}

// position of the first entry of sorted row segment [rs,re) with column >= bound
__device__ __forceinline__ int seg_lower_bound(const int *__restrict__ col, int rs, int re, int bound) {
  int lo = rs, hi = re;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < bound) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// grid level of a sliced row: the finest one that still leaves about `unit` nnz per cell
__device__ __forceinline__ int row_level(int len, int unit) {
  int j = 0;
  while (j < 4 && (len >> (j + 1)) >= 8 * unit) j++;
  return j;
}

// The cut of a sorted row [rs,re) on level j, spread over the wave: lane l owns cells 2l and 2l+1 (of nc = 8 << j).
// p0/p1/p2 = first nnz of cell 2l, of cell 2l+1, of cell 2l+2; n0/n1 = units of the two cells; excl = units of the row
// before this lane's cells.  Returns the row's unit count.  All 64 lanes must be active.
struct RowCut {
  int p0, p1, p2, n0, n1, excl;
};
__device__ __forceinline__ int cut_row(const int *__restrict__ col, const int *__restrict__ bounds, int rs, int re, int j,
                                       int ch, int lane, RowCut &rc) {
  const int nc = 8 << j, sh = 4 - j;
  const int c0 = 2 * lane;
  auto pos = [&](int c) { return c <= 0 ? rs : (c >= nc ? re : seg_lower_bound(col, rs, re, bounds[c << sh])); };
  rc.p0 = pos(c0);
  rc.p1 = pos(c0 + 1);
  const int nxt = __shfl_down(rc.p0, 1, 64);
  rc.p2 = (c0 + 2 >= nc) ? re : nxt;
  rc.n0 = (rc.p1 - rc.p0 + ch - 1) / ch;
  rc.n1 = (rc.p2 - rc.p1 + ch - 1) / ch;
  int incl = rc.n0 + rc.n1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  rc.excl = incl - rc.n0 - rc.n1;
  return __shfl(incl, 63, 64);
}

// Strict table (the strict-order schedule over a plan, spmm_strict.h): every listed row as {row, first nnz, nnz, -} with key
// ~nnz (ascending keys = longest first; unused slots keep 0xFFFFFFFF and sort last) ...
__global__ __launch_bounds__(kBlock) void plan_strict_rows(const int *__restrict__ rowptr, const PlanWs *__restrict__ pw,
                                                           const int *__restrict__ list, unsigned *__restrict__ keys,
                                                           int4 *__restrict__ rows) {
  const int n = pw->n_longlist;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int r = list[i];
    const int rs = rowptr[r], len = rowptr[r + 1] - rs;
    keys[i] = ~(unsigned)len;
    rows[i] = make_int4(r, rs, len, 0);
  }
}
// ... and, once sorted, the sizes of its three length classes (len > T  <=>  key < ~T)
__global__ void plan_strict_hdr(const PlanWs *__restrict__ pw, const unsigned *__restrict__ keys, StrictPlanHdr *__restrict__ sh) {
  const int n = pw->n_longlist;
  auto above = [&](int T) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] < ~(unsigned)T) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int nh = above(kStrictHub), nm = above(kStrictMid);
  sh->n_hub = nh;
  sh->n_mid = nm - nh;
  sh->n_whole = n - nm;
  for (int i = 0; i < 5; i++) sh->pad[i] = 0;
}

// One wave per long row: is it cut on the column grid (long enough AND sorted)?  how many units?
// info[i] = {units, units that need a partial slot (0 for a single-unit row), 1 if multi-unit, 1 + level if cut | 0}
// Rows longer than thub are ALSO appended to the hub staging list {row, first nnz, nnz, -} with key ~nnz (sorted longest first
// afterwards): the sum / mean launches chain them whole and skip their units, every other reduce folds them as before.
__global__ __launch_bounds__(kBlock) void plan_rowunits(int ch, int tslice, int nocut, int unit, int thub, int max_hub,
                                                        const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        PlanWs *__restrict__ pw, const int *__restrict__ bounds,
                                                        const int *__restrict__ list, int4 *__restrict__ info,
                                                        unsigned *__restrict__ hkeys, int4 *__restrict__ hub_in) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = pw->n_longlist;
  for (int i = blockIdx.x * (kBlock / kWave) + wave; i < n; i += gridDim.x * (kBlock / kWave)) {
    const int r = list[i];
    const int rs = rowptr[r], re = rowptr[r + 1];
    bool sliced = (re - rs) > tslice && (re - rs) <= nocut;
    if (sliced) {
      bool ok = true;
      for (int p = rs + lane; p + 1 < re; p += kWave) ok &= col[p] <= col[p + 1];
      sliced = __ballot(!ok) == 0ull;
    }
    int nu, lvl = 0;
    if (sliced) {
      lvl = row_level(re - rs, unit);
      RowCut rc;
      nu = cut_row(col, bounds, rs, re, lvl, ch, lane, rc);
    } else {
      nu = (re - rs + ch - 1) / ch;
    }
    const bool hub = (re - rs) > thub;
    if (lane == 0) {
      info[i] = make_int4(nu, nu > 1 ? nu : 0, nu > 1 ? 1 : 0, (sliced ? 1 + lvl : 0) | (hub ? 64 : 0));
      if (hub) {
        const int h = atomicAdd(&pw->n_hub, 1);
        if (h < max_hub) {  // (always: a hub row has more than kHubChainMin nnz and the capacity is nnz / kHubChainMin + 16)
          hkeys[h] = ~(unsigned)(re - rs);
          hub_in[h] = make_int4(r, rs, re - rs, 0);
        }
      }
    }
  }
}

__global__ void plan_totals(const PlanWs *__restrict__ pw, const int4 *__restrict__ info, const int4 *__restrict__ scan,
                            PlanHdr *__restrict__ hdr) {
  const int n = pw->n_longlist;
  int4 t = make_int4(0, 0, 0, 0);
  if (n > 0) t = I4Plus()(scan[n - 1], info[n - 1]);
  hdr->n_units = t.x;
  hdr->n_pslots = t.y;
  hdr->n_long = t.z;
  hdr->n_hub = pw->n_hub;
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// One wave per long row: write its units (+ sort keys) and, for a multi-unit row, its long-row entry.
__global__ __launch_bounds__(kBlock) void plan_emit(int ch, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const PlanWs *__restrict__ pw, const int *__restrict__ bounds,
                                                    const int *__restrict__ list, const int4 *__restrict__ info,
                                                    const int4 *__restrict__ scan, int4 *__restrict__ units,
                                                    unsigned long long *__restrict__ keys, int4 *__restrict__ longrows,
                                                    int *__restrict__ slot_long) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = pw->n_longlist;
  for (int i = blockIdx.x * (kBlock / kWave) + wave; i < n; i += gridDim.x * (kBlock / kWave)) {
    const int r = list[i];
    const int4 inf = info[i], sc = scan[i];  // sc = {first unit, first partial slot, long-row index, -}
    const int rs = rowptr[r], re = rowptr[r + 1];
    const int nu = inf.x;
    const unsigned long long hub = (inf.w & 64) ? 1ull : 0ull;  // hub rows: units behind the others of their slice, long row flagged
    if (lane == 0 && nu > 1) longrows[sc.z] = make_int4(r, sc.y, nu, (int)hub);
    if (inf.w & 63) {
      const int j = (inf.w & 63) - 1;
      RowCut rc;
      cut_row(col, bounds, rs, re, j, ch, lane, rc);
      // units are numbered in position order (= column order of the sorted row): cell 2l first, then cell 2l+1
      int k = rc.excl;
      for (int q = 0; q < 2; q++) {
        const int a = q ? rc.p1 : rc.p0, b = q ? rc.p2 : rc.p1;
        const unsigned long long slice = (unsigned long long)((2 * lane + q) >> j);
        for (int p0 = a; p0 < b; p0 += ch, k++) {
          units[sc.x + k] = make_int4(r, p0, min(ch, b - p0), nu > 1 ? sc.y + k : -1);
          if (nu > 1) slot_long[sc.y + k] = sc.z;  // (in-kernel fold: the long row a partial slot belongs to)
          keys[sc.x + k] = (slice << 33) | (hub << 32) | (unsigned)col[p0];
        }
      }
    } else {
      for (int k = lane; k < nu; k += kWave) {
        const int p0 = rs + k * ch;
        units[sc.x + k] = make_int4(r, p0, min(ch, re - p0), nu > 1 ? sc.y + k : -1);
        if (nu > 1) slot_long[sc.y + k] = sc.z;
        keys[sc.x + k] = ((unsigned long long)(hash32((unsigned)r * 31u + (unsigned)k) & 7u) << 33) | (hub << 32) | (unsigned)col[p0];
      }
    }
  }
}

__global__ void plan_xcd(const unsigned long long *__restrict__ keys, PlanHdr *__restrict__ hdr) {
  const int t = threadIdx.x;
  if (t > 16) return;
  // t even: first unit of slice t / 2; t odd: first HUB-row unit of slice t / 2 (sorted behind the slice's other units)
  const int n = hdr->n_units;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((keys[mid] >> 32) < (unsigned long long)t) lo = mid + 1; else hi = mid;
  }
  if (t & 1) hdr->xcd_hub[t >> 1] = lo;
  else hdr->xcd_start[t >> 1] = lo;
}

}  // namespace dgs

using namespace dgs;

static int plan_tslice() {
  int t = tune(tuning().plan_tslice, 256);
  if (t < kPlanSliceMin) t = kPlanSliceMin;
  return t;
}
static int plan_unit() {  // nnz per cell a cut row should keep (decides how fine long rows are cut on the column grid)
  int u = tune(tuning().plan_unit, 64);
  if (u < kPlanUnitMin) u = kPlanUnitMin;
  return u;
}

extern "C" size_t dgs_spmm_plan_bytes(int64_t M, int64_t K, int64_t nnz) {
  (void)M;
  (void)K;
  if (nnz <= 0) return 256;
  return plan_layout(nnz).total;
}

extern "C" size_t dgs_spmm_plan_workspace_bytes(int64_t M, int64_t K, int64_t nnz) {
  (void)M;
  if (nnz <= 0 || K <= 0) return 256;
  return plan_ws_layout(K, nnz).total;
}

extern "C" int dgs_spmm_plan_build(int64_t M, int64_t K, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                   void *plan, size_t plan_bytes, void *workspace, size_t workspace_bytes,
                                   dgsSpmmPlanInfo *info, dgsStream_t stream) {
  return dgs_spmm_plan_build2(M, K, nnz, rowptr, col, nullptr, plan, plan_bytes, workspace, workspace_bytes, info, stream);
}

// col_prefix (nullable): col_prefix[c] = number of entries with column < c, c = 0 .. K - i.e. the colptr of the matrix's CSC
// view, which a caller that keeps one (dgsparse.Storage) already has.  It replaces the column histogram (nnz atomicAdds) and
// its scan: 0.87 + 0.03 of the build's 1.5 ms on the headline graph.
extern "C" int dgs_spmm_plan_build2(int64_t M, int64_t K, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                    const int32_t *col_prefix, void *plan, size_t plan_bytes, void *workspace,
                                    size_t workspace_bytes, dgsSpmmPlanInfo *info, dgsStream_t stream) {
  if (M <= 0 || K <= 0 || nnz <= 0 || !rowptr || !col || !plan || !workspace) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  const PlanLayout PL = plan_layout(nnz);
  const PlanWsLayout WL = plan_ws_layout(K, nnz);
  if (plan_bytes < PL.total || workspace_bytes < WL.total) return DGS_EWORKSPACE;
  if (!is_aligned16(plan) || !is_aligned16(workspace)) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *pb = static_cast<char *>(plan), *ws = static_cast<char *>(workspace);
  PlanHdr *hdr = reinterpret_cast<PlanHdr *>(pb);
  int *bounds = reinterpret_cast<int *>(pb + PL.off_bounds);
  int4 *units = reinterpret_cast<int4 *>(pb + PL.off_units);
  int4 *longrows = reinterpret_cast<int4 *>(pb + PL.off_long);
  PlanWs *pw = reinterpret_cast<PlanWs *>(ws);
  int *cnt = reinterpret_cast<int *>(ws + WL.off_cnt);
  int *cum = reinterpret_cast<int *>(ws + WL.off_cum);
  int *list = reinterpret_cast<int *>(ws + WL.off_list);
  int4 *rinfo = reinterpret_cast<int4 *>(ws + WL.off_info);
  int4 *rscan = reinterpret_cast<int4 *>(ws + WL.off_scan);
  unsigned long long *keys_in = reinterpret_cast<unsigned long long *>(ws + WL.off_keys_in);
  unsigned long long *keys_out = reinterpret_cast<unsigned long long *>(ws + WL.off_keys_out);
  int4 *units_in = reinterpret_cast<int4 *>(ws + WL.off_units_in);
  unsigned *hkeys_in = reinterpret_cast<unsigned *>(ws + WL.off_hkeys_in);
  unsigned *hkeys_out = reinterpret_cast<unsigned *>(ws + WL.off_hkeys_out);
  int4 *hub_in = reinterpret_cast<int4 *>(ws + WL.off_hub_in);
  unsigned *skeys_in = reinterpret_cast<unsigned *>(ws + WL.off_skeys_in);
  unsigned *skeys_out = reinterpret_cast<unsigned *>(ws + WL.off_skeys_out);
  int4 *srows_in = reinterpret_cast<int4 *>(ws + WL.off_srows_in);
  StrictPlanHdr *shdr = reinterpret_cast<StrictPlanHdr *>(pb + PL.off_strict);
  int4 *hub_rows = reinterpret_cast<int4 *>(pb + PL.off_hub);
  void *tmp = ws + WL.off_tmp;
  // unit length: 256 nnz when there is plenty of work; smaller inputs get shorter units (a unit is a chain of up to
  // ch/64 dependent tiles, and a mid-size graph has too few units to hide it)
  int ch = tune(tuning().plan_ch, nnz >= (8 << 20) ? kPlanCh : (nnz >= (2 << 20) ? 128 : 64));
  ch = ch < kPlanChMin ? kPlanChMin : (ch > (1 << 20) ? (1 << 20) : ch);  // the table capacities assume ch >= kPlanChMin
  const int tslice = plan_tslice(), unit = plan_unit();
  // hub rows (dgs_common.h Tuning::hub_chain; spmm_impl.h hub_threshold): listed longest first for the sum / mean launches
  const int thub = plan_hub_threshold();

  if (hipMemsetAsync(hdr, 0, PL.off_units, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemsetAsync(ws, 0, WL.off_list, st) != hipSuccess) return DGS_ELAUNCH;           // counters, cnt, cum
  if (hipMemsetAsync(rinfo, 0, (size_t)WL.cap_long * 16, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemsetAsync(keys_in, 0xFF, (size_t)PL.max_units * 8, st) != hipSuccess) return DGS_ELAUNCH;  // unused = last
  if (hipMemsetAsync(units_in, 0, (size_t)PL.max_units * 16, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemsetAsync(hkeys_in, 0xFF, (size_t)PL.max_hub * 4, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemsetAsync(hub_in, 0, (size_t)PL.max_hub * 16, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemsetAsync(skeys_in, 0xFF, (size_t)WL.cap_long * 4, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemsetAsync(srows_in, 0, (size_t)WL.cap_long * 16, st) != hipSuccess) return DGS_ELAUNCH;

  size_t tb = WL.tmp_bytes;
  const int *prefix = col_prefix;
  if (!prefix) {
    hipLaunchKernelGGL(plan_hist, dim3(2048), dim3(kBlock), 0, st, (int)nnz, (int)K, col, cnt);
    if (rocprim::exclusive_scan(tmp, tb, cnt, cum, 0, (size_t)(K + 1), rocprim::plus<int>(), st, false) != hipSuccess)
      return DGS_ELAUNCH;
    prefix = cum;
  }
  hipLaunchKernelGGL(plan_bounds, dim3(1), dim3(192), 0, st, (int)K, (int)nnz, prefix, hdr, bounds, (int)M, ch, kT1, tslice,
                     unit, thub);
  hipLaunchKernelGGL(plan_longlist, dim3((unsigned)((M + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int)M, kT1, rowptr,
                     pw, list);
  // strict table: the listed rows sorted longest first + the sizes of the strict schedule's length classes
  hipLaunchKernelGGL(plan_strict_rows, dim3(512), dim3(kBlock), 0, st, rowptr, pw, list, skeys_in, srows_in);
  tb = WL.tmp_bytes;
  if (rocprim::radix_sort_pairs(tmp, tb, skeys_in, skeys_out, srows_in, reinterpret_cast<int4 *>(shdr + 1), (size_t)WL.cap_long, 0, 32,
                                st, false) != hipSuccess)
    return DGS_ELAUNCH;
  hipLaunchKernelGGL(plan_strict_hdr, dim3(1), dim3(1), 0, st, pw, skeys_out, shdr);
  // DGS_PLAN_NOCUT (experiment): rows longer than this are chunked without column cuts (what a hub row costs when it leaves
  // the column-slice order)
  hipLaunchKernelGGL(plan_rowunits, dim3(1024), dim3(kBlock), 0, st, ch, tslice, tune(tuning().plan_nocut, INT_MAX), unit, thub,
                     (int)PL.max_hub, rowptr, col, pw, bounds, list, rinfo, hkeys_in, hub_in);
  tb = WL.tmp_bytes;
  if (rocprim::radix_sort_pairs(tmp, tb, hkeys_in, hkeys_out, hub_in, hub_rows, (size_t)PL.max_hub, 0, 32, st, false) !=
      hipSuccess)
    return DGS_ELAUNCH;
  tb = WL.tmp_bytes;
  if (rocprim::exclusive_scan(tmp, tb, rinfo, rscan, make_int4(0, 0, 0, 0), (size_t)WL.cap_long, I4Plus(), st, false) !=
      hipSuccess)
    return DGS_ELAUNCH;
  hipLaunchKernelGGL(plan_totals, dim3(1), dim3(1), 0, st, pw, rinfo, rscan, hdr);
  hipLaunchKernelGGL(plan_emit, dim3(1024), dim3(kBlock), 0, st, ch, rowptr, col, pw, bounds, list, rinfo, rscan, units_in,
                     keys_in, longrows, reinterpret_cast<int *>(pb + PL.off_slot));
  tb = WL.tmp_bytes;
  if (rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, units_in, units, (size_t)PL.max_units, 0, 37, st, false) !=
      hipSuccess)
    return DGS_ELAUNCH;
  hipLaunchKernelGGL(plan_xcd, dim3(1), dim3(64), 0, st, keys_out, hdr);
  if (check_launch() != DGS_OK) return DGS_ELAUNCH;
  if (info) {  // the counts the host needs to size grids and the partial-row workspace: one blocking copy, once per plan
    PlanHdr h;
    if (hipMemcpyAsync(&h, hdr, sizeof(PlanHdr), hipMemcpyDeviceToHost, st) != hipSuccess) return DGS_ELAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return DGS_ELAUNCH;
    info->n_units = h.n_units;
    info->n_long = h.n_long;
    info->n_pslots = h.n_pslots;
    info->n_hub = h.n_hub;
    info->tslice = h.tslice;
    info->off_long = 0;
    info->off_hub = 0;
    for (int x = 0; x < 9; x++) info->xcd_start[x] = h.xcd_start[x];
  }
  return DGS_OK;
}

// The counts of a finished build from a HOST copy of the plan buffer's first 256 bytes (the device-resident header).  Lets a
// caller build without the blocking copy: dgs_spmm_plan_build(..., info = NULL, stream), an async copy of the header into
// pinned memory behind it, an event - and this call once the event has completed.
extern "C" int dgs_spmm_plan_info_from_header(const void *host_header, size_t bytes, dgsSpmmPlanInfo *info) {
  if (!host_header || !info || bytes < sizeof(PlanHdr)) return DGS_EINVAL;
  PlanHdr h;
  memcpy(&h, host_header, sizeof(PlanHdr));
  if (h.magic != kPlanMagic) return DGS_EINVAL;
  info->n_units = h.n_units;
  info->n_long = h.n_long;
  info->n_pslots = h.n_pslots;
  info->n_hub = h.n_hub;
  info->tslice = h.tslice;
  info->off_long = 0;
  info->off_hub = 0;
  for (int x = 0; x < 9; x++) info->xcd_start[x] = h.xcd_start[x];
  return DGS_OK;
}

// Upper bounds of a plan's counts from four sums over the row lengths (rows longer than t1 / than tslice: how many, how many
// nnz), for callers that queue the build and the first planned calls on one stream WITHOUT waiting for the build's counts:
// the kernels read the real counts from the plan's device header, the host only needs grid and workspace sizes that are
// large enough.  A row of (t1, tslice] nnz makes ceil(len / ch) units; a longer one is cut on at most max(8, len / unit)
// <= 128 column cells, each with one ragged unit, plus len / ch full ones (plan_rowunits / cut_row above).
extern "C" void dgs_spmm_plan_thresholds(int32_t *t1, int32_t *tslice) {
  if (t1) *t1 = kT1;
  if (tslice) *tslice = plan_tslice();
}
extern "C" int dgs_spmm_plan_provisional_info(int64_t nnz, int64_t rows_gt_t1, int64_t nnz_gt_t1, int64_t rows_gt_tslice,
                                              int64_t nnz_gt_tslice, dgsSpmmPlanInfo *info) {
  if (!info || nnz <= 0 || rows_gt_t1 < rows_gt_tslice || nnz_gt_t1 < nnz_gt_tslice || rows_gt_tslice < 0) return DGS_EINVAL;
  int ch = tune(tuning().plan_ch, nnz >= (8 << 20) ? kPlanCh : (nnz >= (2 << 20) ? 128 : 64));
  ch = ch < kPlanChMin ? kPlanChMin : (ch > (1 << 20) ? (1 << 20) : ch);
  const int unit = plan_unit();
  const int64_t mid_rows = rows_gt_t1 - rows_gt_tslice, mid_nnz = nnz_gt_t1 - nnz_gt_tslice;
  int64_t units = mid_rows + mid_nnz / ch + rows_gt_tslice * 9 + nnz_gt_tslice / unit + nnz_gt_tslice / ch;
  const PlanLayout PL = plan_layout(nnz);
  if (units > PL.max_units) units = PL.max_units;
  if (units >= INT32_MAX) return DGS_ERANGE;
  memset(info, 0, sizeof(*info));
  info->n_units = (int32_t)units;
  info->n_long = (int32_t)(rows_gt_t1 < PL.max_long ? rows_gt_t1 : PL.max_long);
  info->n_pslots = (int32_t)units;
  info->tslice = plan_tslice();
  info->off_long = 0;  // the build-time layout
  {  // hub rows, an UPPER bound (it sizes the hub grid, and 0 must mean "no hub row"): every row longer than the threshold is
    // longer than tslice when thub >= tslice, and in any case longer than t1 (thub >= kHubChainMin > t1); no more than fit
    // the nnz of those rows (with DGS_PLAN_TSLICE above DGS_HUB_CHAIN the tslice sums under-counted)
    const int thub = plan_hub_threshold();
    const bool above = thub >= info->tslice;
    const int64_t rows = above ? rows_gt_tslice : rows_gt_t1, nz = above ? nnz_gt_tslice : nnz_gt_t1;
    const int64_t hb = thub == INT_MAX ? 0 : (rows < nz / thub ? rows : nz / thub);
    info->n_hub = (int32_t)(hb < PL.max_hub ? hb : PL.max_hub);
  }
  info->off_hub = 0;
  return DGS_OK;
}

extern "C" size_t dgs_spmm_csr_plan_workspace_bytes(int reduce_op, int64_t M, int64_t N, int64_t nnz,
                                                    const dgsSpmmPlanInfo *info) {
  if (M <= 0 || N <= 0 || nnz <= 0 || !info) return 0;
  return ws_layout_plan(reduce_op, N, info->n_pslots, info->n_long).total;
}

// rows longer than T1 = single-unit rows + multi-unit rows (the units of multi-unit rows are exactly the partial slots)
static inline int64_t plan_srows(const dgsSpmmPlanInfo *info) { return (int64_t)info->n_units - info->n_pslots + info->n_long; }

extern "C" size_t dgs_spmm_plan_compact_bytes(const dgsSpmmPlanInfo *info) {
  if (!info) return 0;
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  return 256 + 768 + up((size_t)info->n_units * sizeof(int4)) + up((size_t)info->n_long * sizeof(int4)) +
         up((size_t)info->n_hub * sizeof(int4)) + up((size_t)info->n_pslots * sizeof(int)) +
         up(sizeof(StrictPlanHdr) + (size_t)plan_srows(info) * sizeof(int4)) + 256;
}

extern "C" int dgs_spmm_plan_compact(const void *plan, dgsSpmmPlanInfo *info, void *compact, size_t compact_bytes,
                                     int64_t nnz, dgsStream_t stream) {
  if (!plan || !info || !compact || nnz <= 0 || info->off_long != 0 || info->off_hub != 0 || info->n_hub < 0) return DGS_EINVAL;
  if (compact_bytes < dgs_spmm_plan_compact_bytes(info)) return DGS_EWORKSPACE;
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  const PlanLayout PL = plan_layout(nnz);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const char *src = static_cast<const char *>(plan);
  char *dst = static_cast<char *>(compact);
  const size_t ub = (size_t)info->n_units * sizeof(int4), lb = (size_t)info->n_long * sizeof(int4);
  const size_t hb = (size_t)info->n_hub * sizeof(int4);
  const size_t off_long = PL.off_units + up(ub), off_hub = off_long + up(lb);
  const size_t sb = (size_t)info->n_pslots * sizeof(int), off_slot = plan_off_slot(off_hub, info->n_hub);
  const size_t tb = sizeof(StrictPlanHdr) + (size_t)plan_srows(info) * sizeof(int4), off_strict = plan_off_strict(off_slot, info->n_pslots);
  if (off_strict + up(tb) > (size_t)INT32_MAX) return DGS_ERANGE;  // the offsets are 32-bit (2^27 units: beyond any int32 nnz / 64)
  if (hipMemcpyAsync(dst, src, PL.off_units + ub, hipMemcpyDeviceToDevice, st) != hipSuccess) return DGS_ELAUNCH;
  if (lb && hipMemcpyAsync(dst + off_long, src + PL.off_long, lb, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return DGS_ELAUNCH;
  if (hb && hipMemcpyAsync(dst + off_hub, src + PL.off_hub, hb, hipMemcpyDeviceToDevice, st) != hipSuccess) return DGS_ELAUNCH;
  if (sb && hipMemcpyAsync(dst + off_slot, src + PL.off_slot, sb, hipMemcpyDeviceToDevice, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipMemcpyAsync(dst + off_strict, src + PL.off_strict, tb, hipMemcpyDeviceToDevice, st) != hipSuccess) return DGS_ELAUNCH;
  info->off_long = (int32_t)off_long;
  info->off_hub = (int32_t)off_hub;
  return DGS_OK;
}
