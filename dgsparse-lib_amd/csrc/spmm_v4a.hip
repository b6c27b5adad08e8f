// spmm_v4a.hip -- instantiates the V=4 sum / mean / masked-sum SpMM kernels and holds the C entry points.
#define DGS_TU_SUM_ONLY
#include "spmm_impl.h"

namespace dgs {
int spmm_run_v4_sum(int G, const SpmmArgs &a) { return dispatch_g<4>(G, a); }
}  // namespace dgs

using namespace dgs;
#if DGS_TRACE
extern "C" void dgs_debug_trace(unsigned long long *buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dgs_trace), &buf, sizeof(buf)); }
#endif

// Feature tiles as PASSES.  The row-stream schedule maps a row's N floats to one group of up to 64 lanes; wider rows
// take gridDim.y tiles, and workgroups are dispatched x-fastest, so the tiles of one launch run one after the other.
// On operands that overflow the L2s a NARROWER tile is the better trade: a pass over 64-float (256-byte) slices keeps
// twice as many rows of the dense operand per XCD as one over 128-float slices, for one more read of (col, val).
// Crossover at ~250 k columns.  Measured on the 1M-row power-law graph (plan, sum / max): N = 128: 830 -> 773 us / 1028 -> 973, N = 192: 1327 -> 1153,
// N = 256: 1765 -> 1567 / 2066 -> 1903, N = 512: 3568 -> 3185; two 32-float passes at N = 64 lose (394 -> 434).
// Mid-size graphs are latency- and issue-bound, more passes cost them: arxiv-shaped N = 128 76 -> 81 us with 64-float
// tiles, but 128-float tiles win from N = 256 (142 -> 126 us; N = 512: 263 -> 254).  The column-panel schedule keeps
// its own 256-feature sweeps (panel_plan rejects narrowed maps, and narrowing is skipped where it applies).
static FeatMap narrow_tiles(FeatMap fm, const SpmmArgs &a) {
  if (fm.V != 4 || a.N < 128 || !a.ws) return fm;
  if (!a.accumulate && fm.G >= 8 && panel_plan(a, fm.tiles, fm.G).use) return fm;
  int G = 0;
  if (a.N % 64 == 0 && a.K >= (1 << 18)) G = 16;  // crossover measured between 200 k (-2 %) and 300 k (+4 %) columns
  else if (a.N % 128 == 0 && a.N >= 256) G = 32;
  if (!G || G >= fm.G) return fm;
  fm.G = G;
  fm.tiles = (int)((a.N / 4 + G - 1) / G);
  return fm;
}

static int run(FeatMap fm, SpmmArgs a) {
  fm = narrow_tiles(fm, a);
  a.tiles = fm.tiles;
  if ((a.hints & (DGS_ALG_STRICT_SUM | DGS_ALG_STRICT_NOFMA)) && (a.reduce_op == DGS_SUM || a.reduce_op == DGS_MEAN) &&
      !a.accumulate && !a.plan)
    return spmm_run_strict(fm.G, fm.V, a);
  if (fm.V != 4) return spmm_run_v1(fm.G, a);
  return (a.reduce_op == DGS_MAX || a.reduce_op == DGS_MIN) ? spmm_run_v4_arg(fm.G, a) : spmm_run_v4_sum(fm.G, a);
}

extern "C" size_t dgs_spmm_csr_workspace_bytes(int reduce_op, int64_t M, int64_t N, int64_t nnz) {
  if (M <= 0 || N <= 0 || nnz <= 0 || tiny_problem(M, nnz)) return 0;
  return ws_layout(reduce_op, N, nnz).total;
}

// Rows longer than this are chained whole by the default sum / mean launches (0 = hub chains are off): spmm_impl.h hub_threshold
extern "C" int dgs_spmm_hub_threshold(void) {
  const int t = hub_threshold();
  return t == INT_MAX ? 0 : t;
}

extern "C" int dgs_spmm_csr_schedule(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz) {
  if (M <= 0 || N <= 0 || nnz <= 0 || tiny_problem(M, nnz)) return DGS_SCHED_SMALL;
  const FeatMap fm = feat_map(N, true);
  SpmmArgs a{M, K, N, nnz, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, fm.tiles, &a, nullptr, reduce_op};
  return (fm.V == 4 && reduce_op >= DGS_SUM && reduce_op <= DGS_MEAN && panel_plan(a, fm.tiles, fm.G).use)
             ? DGS_SCHED_PANEL
             : DGS_SCHED_ROWS;
}

extern "C" int dgs_spmm_csr_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                int algorithm, void *workspace, size_t workspace_bytes, dgsStream_t stream) {
  // every algorithm id returns the algorithm-0 result; bits from 8 up are scheduling hints
  if (reduce_op < DGS_SUM || reduce_op > DGS_MEAN || M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  const bool arg = (reduce_op == DGS_MAX || reduce_op == DGS_MIN);
  if (M == 0 || N == 0) return DGS_OK;
  if (!rowptr || !C || (nnz > 0 && (!col || !B)) || (arg && !E)) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E && !arg) {  // the reference leaves E = -1 for sum/mean (Eidx is never updated)
    if (hipMemsetAsync(E, 0xFF, (size_t)M * N * sizeof(int32_t), st) != hipSuccess) return DGS_ELAUNCH;
  }
  const size_t need = dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(B) && is_aligned16(C) && (!arg || is_aligned16(E)) &&
                  (need == 0 || is_aligned16(workspace));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, arg ? E : nullptr, fm.tiles, need ? workspace : nullptr, st, reduce_op};
  a.hints = algorithm & ~0xff;
  return run(fm, a);
}

// The general entry: dgs_spmm_csr_f32 / dgs_spmm_csr_plan_f32 plus the fused epilogue (sum / mean only).
extern "C" int dgs_spmm_csr_ex_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                   const int32_t *col, const float *val, const float *B, float *C, int32_t *E, int algorithm,
                                   const float *bias, const float *row_scale, int relu, const void *plan,
                                   const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes, dgsStream_t stream) {
  if (reduce_op < DGS_SUM || reduce_op > DGS_MEAN || M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  const bool arg = (reduce_op == DGS_MAX || reduce_op == DGS_MIN);
  const bool epi = bias || row_scale || relu;
  const bool strict = (algorithm & (DGS_ALG_STRICT_SUM | DGS_ALG_STRICT_NOFMA)) && !arg;
  if (epi && (arg || strict)) return DGS_EINVAL;  // the epilogue exists for the default sum / mean schedules
  if (M == 0 || N == 0) return DGS_OK;
  if (!rowptr || !C || (nnz > 0 && (!col || !B)) || (arg && !E)) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E && !arg) {
    if (hipMemsetAsync(E, 0xFF, (size_t)M * N * sizeof(int32_t), st) != hipSuccess) return DGS_ELAUNCH;
  }
  const bool planned = plan && info && !strict && nnz > 0 && !tiny_problem(M, nnz) &&
                       dgs_spmm_csr_schedule(reduce_op, M, K, N, nnz) == DGS_SCHED_ROWS;
  if (planned && !is_aligned16(plan)) return DGS_EINVAL;
  const size_t need = planned ? dgs_spmm_csr_plan_workspace_bytes(reduce_op, M, N, nnz, info)
                              : dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(B) && is_aligned16(C) && (!arg || is_aligned16(E)) && (need == 0 || is_aligned16(workspace)) &&
                  (!bias || is_aligned16(bias));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, arg ? E : nullptr, fm.tiles, need ? workspace : nullptr, st, reduce_op};
  a.hints = algorithm & ~0xff;
  a.acc.epi = Epi{bias, row_scale, relu ? 1 : 0};
  if (planned) {
    a.plan = static_cast<const PlanHdr *>(plan);
    a.plan_units = info->n_units;
    a.plan_long = info->n_long;
    a.plan_pslots = info->n_pslots;
    a.plan_off_long = info->off_long;
    a.plan_hub = info->n_hub;
    a.plan_off_hub = info->off_hub;
  }
  return run(fm, a);
}

// SpMM over a cached plan (spmm_plan.hip).  Shapes that do not take the row-stream schedule ignore the plan.
extern "C" int dgs_spmm_csr_plan_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                     const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                     const void *plan, const dgsSpmmPlanInfo *info, void *workspace,
                                     size_t workspace_bytes, dgsStream_t stream) {
  if (reduce_op < DGS_SUM || reduce_op > DGS_MEAN || M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (!plan || !info || M == 0 || N == 0 || nnz == 0 || tiny_problem(M, nnz) ||
      dgs_spmm_csr_schedule(reduce_op, M, K, N, nnz) != DGS_SCHED_ROWS)
    return DGS_EINVAL;  // callers route such shapes to dgs_spmm_csr_f32
  const bool arg = (reduce_op == DGS_MAX || reduce_op == DGS_MIN);
  if (!rowptr || !C || !col || !B || (arg && !E)) return DGS_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E && !arg) {
    if (hipMemsetAsync(E, 0xFF, (size_t)M * N * sizeof(int32_t), st) != hipSuccess) return DGS_ELAUNCH;
  }
  const size_t need = dgs_spmm_csr_plan_workspace_bytes(reduce_op, M, N, nnz, info);
  if (!workspace || workspace_bytes < need) return DGS_EWORKSPACE;
  if (!is_aligned16(plan)) return DGS_EINVAL;  // the kernels read the tables as int4
  const bool al = is_aligned16(B) && is_aligned16(C) && (!arg || is_aligned16(E)) && is_aligned16(workspace);
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, arg ? E : nullptr, fm.tiles, workspace, st, reduce_op};
  a.plan = static_cast<const PlanHdr *>(plan);
  a.plan_units = info->n_units;
  a.plan_long = info->n_long;
  a.plan_pslots = info->n_pslots;
  a.plan_off_long = info->off_long;
  a.plan_hub = info->n_hub;
  a.plan_off_hub = info->off_hub;
  return run(fm, a);
}

// C[rowmap[r], :] += sum_p val[p] * B[col[p], :]  (rows of A without entries leave C alone).  Optional plan as above.
extern "C" int dgs_spmm_csr_acc_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                    const float *val, const float *B, float *C, const int32_t *rowmap, const void *plan,
                                    const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes,
                                    dgsStream_t stream) {
  if (M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (M == 0 || N == 0 || nnz == 0) return DGS_OK;
  if (!rowptr || !C || !col || !B) return DGS_EINVAL;
  const bool planned = plan && info && !tiny_problem(M, nnz);
  if (planned && !is_aligned16(plan)) return DGS_EINVAL;
  const size_t need = planned ? dgs_spmm_csr_plan_workspace_bytes(DGS_SUM, M, N, nnz, info)
                              : dgs_spmm_csr_workspace_bytes(DGS_SUM, M, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(B) && is_aligned16(C) && (need == 0 || is_aligned16(workspace));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, nullptr, fm.tiles, need ? workspace : nullptr,
             static_cast<hipStream_t>(stream), DGS_SUM};
  a.accumulate = true;
  a.acc.rowmap = rowmap;
  if (planned) {
    a.plan = static_cast<const PlanHdr *>(plan);
    a.plan_units = info->n_units;
    a.plan_long = info->n_long;
    a.plan_pslots = info->n_pslots;
    a.plan_off_long = info->off_long;
    a.plan_hub = info->n_hub;
    a.plan_off_hub = info->off_hub;
  }
  return run(fm, a);
}

// (C, E)[rowmap[r], :] = the better of what they hold and the max over row r of A (arg ids shifted by col_off), the
// earlier column winning ties, columns ordered by acc_key (spmm_impl.h AccArg).
extern "C" int dgs_spmm_csr_acc_max_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                        const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                        const int32_t *rowmap, int32_t col_off, int32_t n_local, int32_t h_lo,
                                        const void *plan, const dgsSpmmPlanInfo *info, void *workspace,
                                        size_t workspace_bytes, dgsStream_t stream) {
  if (M < 0 || K < 0 || N < 0 || nnz < 0 || n_local < 0 || h_lo < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (M == 0 || N == 0 || nnz == 0) return DGS_OK;
  if (!rowptr || !C || !E || !col || !B) return DGS_EINVAL;
  const bool planned = plan && info && !tiny_problem(M, nnz);
  if (planned && !is_aligned16(plan)) return DGS_EINVAL;
  const size_t need = planned ? dgs_spmm_csr_plan_workspace_bytes(DGS_MAX, M, N, nnz, info)
                              : dgs_spmm_csr_workspace_bytes(DGS_MAX, M, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(B) && is_aligned16(C) && is_aligned16(E) && (need == 0 || is_aligned16(workspace));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, E, fm.tiles, need ? workspace : nullptr,
             static_cast<hipStream_t>(stream), DGS_MAX};
  a.accumulate = true;
  a.acc = AccArg{rowmap, col_off, n_local, h_lo};
  if (planned) {
    a.plan = static_cast<const PlanHdr *>(plan);
    a.plan_units = info->n_units;
    a.plan_long = info->n_long;
    a.plan_pslots = info->n_pslots;
    a.plan_off_long = info->off_long;
    a.plan_hub = info->n_hub;
    a.plan_off_hub = info->off_hub;
  }
  return run(fm, a);
}

// (C, E)[rowmap[r], :] = algorithm 0's MIN step applied to what they hold and the min over row r of A, in row order:
// precedes != 0 = this product's columns all come BEFORE the ones (C, E) cover, 0 = all AFTER (spmm_impl.h AccArg).
extern "C" int dgs_spmm_csr_acc_min_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                        const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                        const int32_t *rowmap, int32_t col_off, int32_t precedes, const void *plan,
                                        const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes,
                                        dgsStream_t stream) {
  if (M < 0 || K < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (M >= INT32_MAX || K >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (M == 0 || N == 0 || nnz == 0) return DGS_OK;
  if (!rowptr || !C || !E || !col || !B) return DGS_EINVAL;
  const bool planned = plan && info && !tiny_problem(M, nnz);
  if (planned && !is_aligned16(plan)) return DGS_EINVAL;
  const size_t need = planned ? dgs_spmm_csr_plan_workspace_bytes(DGS_MIN, M, N, nnz, info)
                              : dgs_spmm_csr_workspace_bytes(DGS_MIN, M, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(B) && is_aligned16(C) && is_aligned16(E) && (need == 0 || is_aligned16(workspace));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, E, fm.tiles, need ? workspace : nullptr,
             static_cast<hipStream_t>(stream), DGS_MIN};
  a.accumulate = true;
  a.acc = AccArg{rowmap, col_off, 0, precedes ? 1 : 0};
  if (planned) {
    a.plan = static_cast<const PlanHdr *>(plan);
    a.plan_units = info->n_units;
    a.plan_long = info->n_long;
    a.plan_pslots = info->n_pslots;
    a.plan_off_long = info->off_long;
    a.plan_hub = info->n_hub;
    a.plan_off_hub = info->off_hub;
  }
  return run(fm, a);
}

// Masked SpMM (max/min backward w.r.t. the dense operand) on the CSC arrays: same launcher, internal op kOpMaskSum.
extern "C" size_t dgs_spmm_csr_mask_workspace_bytes(int64_t Mout, int64_t N, int64_t nnz) {
  return dgs_spmm_csr_workspace_bytes(DGS_SUM, Mout, N, nnz);
}

extern "C" int dgs_spmm_csr_mask_f32(int64_t Mout, int64_t Min, int64_t N, int64_t nnz, const int32_t *ptr,
                                     const int32_t *idx, const float *val, const float *G, const int32_t *E,
                                     float *out, void *workspace, size_t workspace_bytes, dgsStream_t stream) {
  if (Mout < 0 || Min < 0 || N < 0 || nnz < 0) return DGS_EINVAL;
  if (Mout >= INT32_MAX || Min >= INT32_MAX || N >= INT32_MAX || nnz >= INT32_MAX) return DGS_ERANGE;
  if (Mout == 0 || N == 0) return DGS_OK;
  if (!ptr || !out || (nnz > 0 && (!idx || !G || !E))) return DGS_EINVAL;
  const size_t need = dgs_spmm_csr_mask_workspace_bytes(Mout, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(G) && is_aligned16(E) && is_aligned16(out) && (need == 0 || is_aligned16(workspace));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{Mout, Min, N, nnz, ptr, idx, val, G, out, const_cast<int32_t *>(E), fm.tiles, need ? workspace : nullptr,
             static_cast<hipStream_t>(stream), kOpMaskSum};
  return run(fm, a);
}
