/*
 * dgsparse_hip.h -- C ABI of the MI355X (gfx950) CSR SpMM / SDDMM / csr2csc hot path.
 *
 * This is the drop-in boundary: plain pointers + sizes + a hipStream_t, no torch types, no allocation,
 * no host synchronisation, no global state.  All pointers are DEVICE pointers on the device that owns
 * `stream`.  Every entry point returns 0 on success or a negative DGS_E* code (never exit()s, unlike
 * the reference's checkCudaError, include/cuda/cuda_util.cuh:116-134).  Launches are asynchronous on
 * `stream` (the reference launches on the legacy default stream, src/cuda/spmm_cuda.cu:57).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the dgSPARSE-Lib tree).
 * Index type int32, value type float32, dense operands row-major - as the reference.
 */
#ifndef DGSPARSE_HIP_H
#define DGSPARSE_HIP_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging hip headers into C callers (cgo / JNI / ctypes). */
typedef void *dgsStream_t;

/* include/gspmm.h:13  enum REDUCEOP { SUM, MAX, MIN, MEAN } */
enum { DGS_SUM = 0, DGS_MAX = 1, DGS_MIN = 2, DGS_MEAN = 3 };

enum {
  DGS_OK = 0,
  DGS_EINVAL = -1,    /* bad enum / negative size / null required pointer */
  DGS_EWORKSPACE = -2,/* workspace too small (query dgs_*_workspace_bytes) */
  DGS_ELAUNCH = -3,   /* hipGetLastError() after a launch was not hipSuccess */
  DGS_ERANGE = -4     /* sizes exceed int32 indexing of the CSR arrays */
};

/* ABI version (major*1000+minor) and the gfx target the kernels were compiled for. */
int dgs_version(void);
const char *dgs_arch(void);
const char *dgs_strerror(int code);
/* The DGS_* tuning overrides (tests / experiments) are read from the environment once per process; this re-reads them.
 * Thread-safe with respect to concurrent launches (they keep the snapshot they started with). */
void dgs_reload_tuning(void);

/*
 * CSR SpMM with reduce:  C[r,:] = reduce_{p in row r} val[p] * B[col[p],:]
 * Replaces: spmm_cuda() host launcher, src/cuda/spmm_cuda.cu:14-253, i.e. kernel
 *           csrspmm_seqreduce_rowbalance_kernel, include/cuda/spmm_cuda.cuh:10-55 ("algorithm 0",
 *           the numerical contract; algorithms 1/2 are other schedules of the same maths, and every
 *           `algorithm` value returns the algorithm-0 result here - it is kept as a tuning hint).
 *   rowptr[M+1], col[nnz] int32; val[nnz] or NULL (= weight 1, cuda_util.cuh:140-146);
 *   B[K,N], C[M,N] row-major fp32;
 *   E[M,N] int32 arg COLUMN ids (-1 for empty rows), REQUIRED for MAX/MIN, optional (may be NULL) else.
 *   MAX/MIN: values and E bit-exact vs algorithm 0 (first occurrence in CSR order wins ties; identities
 *   (float)INT_MIN / (float)INT_MAX; empty row -> 0 / -1).  SUM/MEAN: sequential CSR order for rows up
 *   to the split threshold (bit-exact vs an fmaf chain), fixed-tree split above it (<=1e-5 rel).
 *   workspace: dgs_spmm_csr_workspace_bytes() bytes, 256-B aligned; contents undefined on entry/exit.
 *   algorithm: low byte = the reference's algorithm id (a hint, ignored); higher bits = scheduling hints:
 *     DGS_ALG_SHARED_GPU  other kernels run concurrently on this GPU (e.g. an overlapped RCCL collective): do not take
 *                         the column-panel sweep, whose soft barrier assumes one workgroup per CU, all co-resident
 *                         (correct but several times slower when CUs are taken away).
 *     DGS_ALG_STRICT_SUM  sum / mean: every (row, feature) is ONE sequential fmaf chain in CSR order whatever the row length -
 *                         literally algorithm 0 (include/cuda/spmm_cuda.cuh:27-47 as nvcc contracts it), bit-exact for every
 *                         row; long rows are spread over waves by FEATURE slices instead of being cut.
 *                         Slower than the default on graphs with hub rows (a 50 k-nnz row is a 50 k-step dependent chain).
 *     DGS_ALG_STRICT_NOFMA  the same with the product rounded before the add (res + (w*x), no contraction): bit-exact
 *                         against the reference's host loop spmm_reference_host (example/util/sp_util.hpp:73-83) as g++
 *                         compiles it.  Never takes the column-panel sweep.  Implies strict order.
 *     DGS_ALG_NO_HUB_ROWS  sum / mean: the caller knows that no row of THIS matrix is longer than dgs_spmm_hub_threshold()
 *                         (one max over the row lengths, e.g. kept next to the CSC view): the launch takes the kernels without
 *                         the hub role - the plain single-launch kernel for small inputs (every graph the reference benchmarks:
 *                         Pubmed, PPI, p2p-Gnutella31, ca-CondMat ...), no idle hub workgroups in front of a plan-free fused
 *                         launch.  Same bits as without the hint when it is true; a false hint sends the long rows through the
 *                         tree (within ~1e-5 of the chain, not bit-identical to it).
 *     DGS_ALG_NO_HUB_COLS  the same statement about the COLUMNS of the matrix.  Ignored by every dgs_* entry; the torch
 *                         binding hands it on as NO_HUB_ROWS to the transposed product of the backward pass (and drops the
 *                         forward's NO_HUB_ROWS there).
 *   The strict bits are ignored by max / min (always exact) and by the accumulating entries.
 */
#define DGS_ALG_SHARED_GPU 0x100
#define DGS_ALG_STRICT_SUM 0x200
#define DGS_ALG_STRICT_NOFMA 0x400
#define DGS_ALG_NO_HUB_ROWS 0x800
#define DGS_ALG_NO_HUB_COLS 0x1000
size_t dgs_spmm_csr_workspace_bytes(int reduce_op, int64_t M, int64_t N, int64_t nnz);
int dgs_spmm_csr_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz,
                     const int32_t *rowptr, const int32_t *col, const float *val, const float *B,
                     float *C, int32_t *E, int algorithm, void *workspace, size_t workspace_bytes,
                     dgsStream_t stream);

/*
 * Which schedule dgs_spmm_csr_f32 runs for these sizes with 16-byte aligned operands (introspection for tests,
 * benchmarks and bug reports; a pure function of its arguments and of the DGS_PANEL* environment overrides):
 *   DGS_SCHED_SMALL  one launch (inputs up to 2^18 nnz / 2^16 rows),
 *   DGS_SCHED_ROWS   classify -> fused row-stream/unit kernel -> combine (the general schedule),
 *   DGS_SCHED_PANEL  column-panel sweep with LDS-resident accumulators (dense graphs: the dense operand does not
 *                    fit the L2s and every panel row is reused several times per XCD), long rows on the unit path.
 * No reference counterpart (the reference has one schedule per `algorithm` id, src/cuda/spmm_cuda.cu:14-253).
 */
#define DGS_SCHED_SMALL 0
#define DGS_SCHED_ROWS 1
#define DGS_SCHED_PANEL 2
int dgs_spmm_csr_schedule(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz);
/*
 * Numerical contract of sum / mean (the reference's result is ONE sequential fp32 chain per (row, feature) in CSR order,
 * include/cuda/spmm_cuda.cuh:27-47, host twin example/util/sp_util.hpp:73-83):
 *   rows of <= 64 nnz            that chain (fmaf), bit for bit
 *   rows of > hub threshold nnz  that chain (fmaf), bit for bit - the HUB rows, each worked by its own workgroups, chained by
 *                                one wave (every feature width and every schedule: general, column-panel, single-launch)
 *   rows in between              a fixed reduction tree (deterministic, closer to the exact sum than the chain; within ~7e-6
 *                                of the chain on non-negative data at the default threshold)
 * The threshold is DGS_HUB_CHAIN (default 16384, clamped to >= 1024, 0 = no hub chains: every row above 64 nnz takes the
 * tree); the chain's own rounding error grows like sqrt(nnz) and passes 1e-5 of the exact sum beyond ~3 10^4 nnz, which is
 * where a tree - however accurate - stops being within 1e-5 of the REFERENCE.  DGS_ALG_STRICT_SUM / _NOFMA chain every row.
 * "Within 1e-5 of the chain" is a statement about CONTINUOUS data (the reference's tests: U[0,1) features; measured with thousands of
 * rows at the threshold: max 6.9e-6).  Operands drawn from a handful of values (the reference's example drivers fill both with
 * {0, .1, .2}) give the chain itself a systematic rounding bias - 1.1e-5 of the exact sum at 2 048 nnz, 2.7e-5 at 16 384, on every
 * element of a row - which no accurate summation shares: callers that need the reference's bits on such data use the strict bits
 * (or DGS_HUB_CHAIN=1024).  DESIGN.md 4.1g, profiles/r05_chain_error_by_value_law.txt.
 * dgs_spmm_hub_threshold() returns the threshold in force on the current device (0 = off).
 *
 * Device gate.  The hub workgroup keeps two register sets of gathers in flight across workgroup barriers while one wave chains
 * out of LDS - behaviour of the compiler's wait counts and of the memory system that only the hardware can confirm.  So without
 * an explicit DGS_HUB_CHAIN the chains are ON only on a device where dgs_spmm_hub_selftest() has passed in this process: it runs
 * the default sum on generated matrices with hub rows - every family of the hub workgroup (feature slices of 16 / 8 / 4 / 2 / 1
 * 16-byte lanes = N 256 / 128 / 64, 32, 16 / 8 / 4, and of 16 / 8 / 4 / 1 scalar lanes = N 20 / 7 / 3 / 1), the general and the
 * single-launch schedule, fourteen shapes (dgs_spmm_selftest_hub_shapes()) - against a one-thread-per-element
 * sequential fmaf kernel and demands identical bits on the hub rows (and 1e-5 elsewhere).  It is one of the TWO entry points of
 * this library that synchronise (`stream`, once) - call it once per device at start-up (dgsparse's Python layer and torch binding
 * do, at the first use of a device); until it has passed, rows above 64 nnz take the fixed tree on that device (the round-3
 * schedule).  Returns 1 = passed (chains on from now on), 0 = FAILED (chains stay off on this device; results of the tree are
 * still within the contract's 1e-5 except on rows of > ~3 10^4 nnz), < 0 = DGS_E* (state unchanged).  DGS_HUB_CHAIN=n in the
 * environment bypasses the gate both ways (n = 0: off, n > 0: on at that threshold) and the test is then skipped (returns 1, no
 * launch, no synchronisation).
 *   scratch: dgs_spmm_hub_selftest_bytes() bytes of device memory (~38 MB), 256-B aligned, contents undefined on entry and exit.
 * dgs_spmm_hub_gate(): 1 / 0 / -1 = passed / not run / failed on the current device.  dgs_spmm_hub_gate_assume(v) sets it (returns the
 * previous state): for hosts that cache the verdict of an identical (library binary, device model, runtime) triple across processes
 * - every DataLoader worker and rank pays the test otherwise.  A bypass like DGS_HUB_CHAIN; dgsparse uses it only under DGS_GATE_CACHE.
 *
 * The IN-KERNEL FOLD (off by default).  Rows that are cut into several units leave partial rows in the workspace; a combine launch
 * behind the fused launch folds them.  With DGS_FOLD=1 the unit wave that completes a row (an arrival counter per row and feature
 * tile, zeroed per call) folds it inside the fused launch - one launch less per call, the fold overlapping the rest of the launch.
 * The partial rows then cross from one workgroup to another - possibly on another XCD, behind another L2 - through agent-scope
 * (sc1, write-through) stores and sc1 loads ordered around the counter's atomic, every partial row on 128-byte lines of its own
 * (no line is shared between two rows' slots).  Same results either way: the fold order is the fixed unit order in both.  It is not a default because no hardware measurement says it is faster (round 6), and because only the
 * hardware can confirm the hand-over: dgs_spmm_fold_selftest() runs sum, max and min over generated matrices with ~600 multi-unit
 * rows (2 .. 59 units) both ways - for every family of partial row the launchers can pick (whole-line slots; slots that share a
 * 128-byte line two, four, eight, ten to a line; scalar-lane slots; two feature tiles - nine families), `rounds` times each, the last round with a
 * streaming kernel loading the fabric from a second stream (flags bit 0) - and demands identical bits (values and arg ids).
 * flags bits 8 .. : run only these families (diagnosis; only a full run raises the gate).  Returns 1 / 0 / < 0 like the hub test;
 * synchronises `stream` once; same scratch.  DGS_FOLD=2 ("auto") makes dgs_spmm_hub_selftest() run it (3 rounds, loaded) and
 * folds in the kernel exactly where it passed (dgs_spmm_fold_gate(): 1 / 0 / -1).
 * dgs_spmm_selftest_detail(out, n): the mismatch counters of the last tests of this process - [0] hub total, [1] fold total,
 * [2 + f] fold family f (dgs_spmm_selftest_families() of them), [16 + h] hub shape h (dgs_spmm_selftest_hub_shapes() of them).
 * No reference counterpart (the reference has one kernel per algorithm id and no self-checks).
 */
int dgs_spmm_hub_threshold(void);
size_t dgs_spmm_hub_selftest_bytes(void);
int dgs_spmm_hub_selftest(void *scratch, size_t scratch_bytes, dgsStream_t stream);
int dgs_spmm_hub_gate(void);
int dgs_spmm_hub_gate_assume(int verdict);
int dgs_spmm_fold_gate(void);
int dgs_spmm_fold_selftest(void *scratch, size_t scratch_bytes, int rounds, int flags, dgsStream_t stream);
int dgs_spmm_selftest_families(void);
int dgs_spmm_selftest_hub_shapes(void);
int dgs_spmm_selftest_detail(int32_t *out, int n);

/*
 * Cached locality plan of the row-stream schedule (new; the reference keeps no per-matrix state - the closest thing is
 * the CSC view dgsparse/storage.py:159-174 computes once per Storage, and the plan has the same lifetime).
 * dgs_spmm_plan_build() analyses the sparsity pattern once (column histogram -> 8 column slices with equal reference
 * counts; rows longer than 256 nnz are cut at slice boundaries; the unit table is sorted by (slice, first column)) and
 * dgs_spmm_csr_plan_f32() then runs the SAME kernels as dgs_spmm_csr_f32 over the plan's tables: XCD x walks column
 * slice x, so its L2 sees an eighth of the dense operand, and the call needs no classify pass and no memset.
 * The plan depends on (rowptr, col) only - not on values, N or the reduce op - and is never written by a call, so one
 * plan may serve concurrent calls on different streams.  Results obey the same contract as the plan-free call (the
 * split points of long rows differ, so sum/mean of rows > 64 nnz may differ in the last bits between the two; max/min
 * values and E are identical).  Graphs that take the SMALL or PANEL schedule ignore the plan.
 *   plan       dgs_spmm_plan_bytes() bytes of device memory, 256-B aligned, owned by the caller;
 *   workspace  dgs_spmm_plan_workspace_bytes() bytes, only during the build;
 *   info       host struct filled by the build (it blocks on `stream` once for that); pass it back to the calls.
 */
typedef struct dgsSpmmPlanInfo {
  int32_t n_units;      /* entries of the unit table */
  int32_t n_long;       /* multi-unit rows */
  int32_t n_pslots;     /* partial rows a call needs in its workspace */
  int32_t n_hub;        /* hub rows (longer than DGS_HUB_CHAIN = 16384 nnz): sum / mean chain them whole, see dgs_spmm_csr_f32 */
  int32_t tslice;       /* rows longer than this were cut at column-slice boundaries */
  int32_t xcd_start[9]; /* first unit of each XCD's share */
  int32_t off_long;     /* byte offset of the long-row table inside the plan buffer; 0 = the build-time layout */
  int32_t off_hub;      /* ... of the hub-row table */
} dgsSpmmPlanInfo;
size_t dgs_spmm_plan_bytes(int64_t M, int64_t K, int64_t nnz);
size_t dgs_spmm_plan_workspace_bytes(int64_t M, int64_t K, int64_t nnz);
int dgs_spmm_plan_build(int64_t M, int64_t K, int64_t nnz, const int32_t *rowptr, const int32_t *col, void *plan,
                        size_t plan_bytes, void *workspace, size_t workspace_bytes, dgsSpmmPlanInfo *info,
                        dgsStream_t stream);
/* The same build with the column histogram supplied: col_prefix[c] = number of entries with column < c for c = 0 .. K (the
 * colptr of the CSC view; NULL = count it here).  Saves the nnz atomicAdds of the histogram, ~60 % of the build time. */
int dgs_spmm_plan_build2(int64_t M, int64_t K, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                         const int32_t *col_prefix, void *plan, size_t plan_bytes, void *workspace, size_t workspace_bytes,
                         dgsSpmmPlanInfo *info, dgsStream_t stream);
/* Build without blocking: call dgs_spmm_plan_build with info == NULL (no host synchronisation at all), copy the first
 * DGS_PLAN_HEADER_BYTES of the plan buffer to (pinned) host memory behind it on the same stream, and hand that copy to
 * dgs_spmm_plan_info_from_header once the copy has completed (event query): it fills *info exactly as the blocking build
 * does.  dgsparse.Storage builds its plans this way, on a side stream, from the k-th use of a matrix on. */
#define DGS_PLAN_HEADER_BYTES 256
int dgs_spmm_plan_info_from_header(const void *host_header, size_t bytes, dgsSpmmPlanInfo *info);
/* Calls BEFORE the counts are known: dgs_spmm_plan_provisional_info fills *info with upper bounds computed from four sums over
 * the row lengths - rows longer than t1 / than tslice (dgs_spmm_plan_thresholds): how many, and how many nnz they hold.  A
 * build queued with info == NULL followed, on the SAME stream, by dgs_spmm_csr_plan_f32 calls with the build buffer and this
 * provisional info is correct without any host synchronisation (the kernels take the real counts from the plan's device
 * header; the bounds only size grids and the partial-row workspace). */
void dgs_spmm_plan_thresholds(int32_t *t1, int32_t *tslice);
int dgs_spmm_plan_provisional_info(int64_t nnz, int64_t rows_gt_t1, int64_t nnz_gt_t1, int64_t rows_gt_tslice,
                                   int64_t nnz_gt_tslice, dgsSpmmPlanInfo *info);
/* The build needs a buffer sized for the worst case (~2.9 bytes per nnz); the tables it leaves are ~16 bytes per UNIT
 * (1M x 1M / 16 M nnz: 46 MB vs 1.7 MB).  dgs_spmm_plan_compact copies them into a buffer of dgs_spmm_plan_compact_bytes
 * and updates *info to describe the copy (pass that buffer + info to the calls; the build buffer can be freed). */
size_t dgs_spmm_plan_compact_bytes(const dgsSpmmPlanInfo *info);
int dgs_spmm_plan_compact(const void *plan, dgsSpmmPlanInfo *info, void *compact, size_t compact_bytes, int64_t nnz,
                          dgsStream_t stream);
size_t dgs_spmm_csr_plan_workspace_bytes(int reduce_op, int64_t M, int64_t N, int64_t nnz,
                                         const dgsSpmmPlanInfo *info);
int dgs_spmm_csr_plan_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                          const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                          const void *plan, const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes,
                          dgsStream_t stream);

/*
 * The general forward entry: everything dgs_spmm_csr_f32 and dgs_spmm_csr_plan_f32 do (plan / info may be NULL) plus a fused
 * EPILOGUE for sum / mean, applied where a finished output row leaves the kernels:
 *     C[r, f] = relu(row_scale[r] * (A.B)[r, f] + bias[f])          bias[N], row_scale[M] nullable, relu 0 / 1
 * with the roundings of the separate elementwise ops (multiply, add, clamp; no contraction), so the result equals the
 * unfused sequence bit for bit while the M x N result is written once instead of written, read and written again.  New (the
 * reference's GCN layer runs spmm_sum and torch.relu as two passes, dgsparse/nn/gcnconv.py:10-35).  DGS_EINVAL for max / min
 * and for the strict-order bits together with an epilogue.  workspace: dgs_spmm_csr_plan_workspace_bytes with a plan on the
 * row-stream schedule, dgs_spmm_csr_workspace_bytes otherwise.
 * Strict-order bits WITH a plan (round 5): the plan also carries every row longer than 64 nnz sorted by length, which is all the
 * strict schedule needs - the call is one launch (no memset, no classify pass), the chains and so the bits are those of the
 * plan-free strict call.  (An experiment override of the strict class thresholds, DGS_STRICT_MID / _HUB, makes the call ignore
 * the plan and want the plan-free workspace: pass the larger of the two sizes when such overrides are in play.)
 */
int dgs_spmm_csr_ex_f32(int reduce_op, int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                        const int32_t *col, const float *val, const float *B, float *C, int32_t *E, int algorithm,
                        const float *bias, const float *row_scale, int relu, const void *plan, const dgsSpmmPlanInfo *info,
                        void *workspace, size_t workspace_bytes, dgsStream_t stream);

/*
 * Accumulating SpMM (sum):  C[rowmap[r],:] += sum_{p in row r} val[p] * B[col[p],:]   (rowmap == NULL: C[r,:] += ...)
 * New (the reference has no accumulating product; its nnz-balanced algorithms atomically add into a caller-zeroed C,
 * src/ge-spmm/gespmm.cc:65-90, which is the closest thing).  Rows of A without entries leave C untouched, so a product
 * over a COMPACT matrix (only the rows that have entries, rowmap = their row numbers) touches those rows of C only:
 * dgsparse.dist adds the halo product into the local one this way, without a temporary and without an M x N add pass.
 * Every element of C receives exactly one add per call (rows are reduced first, in the schedule's usual order), so the
 * result is deterministic.  plan/info: optional cached plan of (rowptr, col) (NULL: plan-free); workspace as for the
 * corresponding plain call (dgs_spmm_csr_plan_workspace_bytes with a plan, dgs_spmm_csr_workspace_bytes without).
 * Never takes the column-panel schedule.
 */
int dgs_spmm_csr_acc_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                         const float *val, const float *B, float *C, const int32_t *rowmap, const void *plan,
                         const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes, dgsStream_t stream);

/*
 * Accumulating SpMM (max with arg ids):  (C, E)[rowmap[r],:] = better of { what they hold, max over row r of A } where the
 * arg of this product is written as col + col_off.  For a matrix whose columns were split into two products over one
 * extended index space - ids [0, n_local) = columns of the first product, [n_local, ...) = the second's, of which the
 * first h_lo slots stand for columns that PRECEDE the first product's in the original rows (dgsparse.dist: this rank's
 * columns / halo slots in global order, h_lo of them owned by lower ranks).  Ties go to the entry that comes first in
 * that original column order, value and arg together - exactly algorithm 0's rule on the undivided row provided the row's
 * columns are sorted (include/cuda/spmm_cuda.cuh:38-41).  E == -1 means "no arg yet" (empty so far).  New; lets the
 * multi-GPU max overlap its local product with the halo exchange.  MIN cannot be merged by column key (its macro keeps
 * the LATER operand's bits on a tie (+0.0 / -0.0) while E names the first): see dgs_spmm_csr_acc_min_f32 below.
 */
int dgs_spmm_csr_acc_max_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                             const float *val, const float *B, float *C, int32_t *E, const int32_t *rowmap,
                             int32_t col_off, int32_t n_local, int32_t h_lo, const void *plan,
                             const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes, dgsStream_t stream);
/* The same for MIN, which keeps the LATER operand's bits on a tie while E names the FIRST minimum: a pair can be put in
 * front of or behind what (C, E) hold, nothing else.  precedes != 0: the columns of this product all come BEFORE the ones
 * (C, E) already cover in the row, 0: all AFTER; the commit is algorithm 0's MIN step on the two pairs in that order (an
 * output row no earlier product had entries for must hold the empty-row 0 / -1).  Exact unless a product is NaN - see
 * dgs_spmm_min_merge_f32 for the detector and the redo that go with it (new; multi-GPU path). */
int dgs_spmm_csr_acc_min_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                             const float *val, const float *B, float *C, int32_t *E, const int32_t *rowmap,
                             int32_t col_off, int32_t precedes, const void *plan, const dgsSpmmPlanInfo *info,
                             void *workspace, size_t workspace_bytes, dgsStream_t stream);
/* Both sides in ONE launch (round 5, "around" form; new, multi-GPU path - the reference is single-device).  The
 * matrix is written in the row order of the undivided row: [columns that precede | ONE virtual entry | columns that follow].
 * It has K columns, of which the ids [virt_lo, virt_lo + virt_n) are VIRTUAL: entry (r, virt_lo + rowmap[r]) - weight 1 if
 * val is given - stands for everything (C, E)[rowmap[r]] already cover, its "dense row" is that row of C, so the old value
 * goes through the row's own MIN chain at its place and every tie rule (value bits of the LAST minimum, arg of the FIRST:
 * include/cuda/spmm_cuda.cuh:38-41 with the MIN macro of include/gspmm.h:133-146) is the undivided row's own.  B has
 * K - virt_n rows: column c < virt_lo is row c of B, c >= virt_lo + virt_n is row c - virt_n (a sorted shard row stays
 * sorted, so a plan cuts it by column slice like any other).  Result: (C, E)[rowmap[r]] = the MIN over the whole row; E keeps
 * what it held where the virtual entry is the first minimum, else the winner's B row + col_off.  A row without a virtual
 * entry (its output holds the empty-row 0 / -1) replaces its output pair.  No two rows may share an output row, and only
 * row r may name the virtual column virt_lo + rowmap[r]; ids must stay below 2^31.  In place: C is read (as virtual rows)
 * and written by the same launch.  Exact unless a product is NaN, like the two-launch form. */
int dgs_spmm_csr_acc_min_around_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                    const float *val, const float *B, float *C, int32_t *E, const int32_t *rowmap,
                                    int32_t col_off, int32_t virt_lo, int32_t virt_n, const void *plan,
                                    const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes,
                                    dgsStream_t stream);

/*
 * Masked SpMM = backward of max/min w.r.t. the dense operand, run on the CSC arrays of A:
 *   out[j,:] = sum_{p in [ptr[j],ptr[j+1])} [E[idx[p],:] == j] * val[p] * G[idx[p],:]
 * Replaces: spmm_cuda_with_mask(), src/cuda/spmm_cuda.cu:255-303 /
 *           csrspmm_seqreduce_rowbalance_with_mask_kernel, include/cuda/spmm_cuda.cuh:400-433
 *           (the formula, not that kernel's stale-variable bug).
 *   ptr[Mout+1], idx[nnz] = colptr,row of A; val = values permuted to CSC order or NULL;
 *   G[Min,N] grad of the SpMM output; E[Min,N] saved arg ids; out[Mout,N].  Same schedule (and workspace
 *   protocol) as dgs_spmm_csr_f32, so hub columns with 10^4+ entries are split like long rows.
 */
size_t dgs_spmm_csr_mask_workspace_bytes(int64_t Mout, int64_t N, int64_t nnz);
int dgs_spmm_csr_mask_f32(int64_t Mout, int64_t Min, int64_t N, int64_t nnz, const int32_t *ptr,
                          const int32_t *idx, const float *val, const float *G, const int32_t *E,
                          float *out, void *workspace, size_t workspace_bytes, dgsStream_t stream);

/*
 * CSR SDDMM:  out[e] = sum_k D1[row(e),k] * D2[col(e),k]   (/ deg(row(e)) when reduce_op==DGS_MEAN)
 * Replaces: sddmm_cuda_csr(), src/cuda/spmm_cuda.cu:331-361 / sddmmCSR{2,1}Scale<REDUCE>,
 *           include/cuda/sddmm_cuda.cuh:222-401; and the standalone C entry sddmm_cuda_csr,
 *           src/sddmm/sddmm.h:10.   D1[M,F], D2[K,F], out[nnz].  reduce_op in {DGS_SUM, DGS_MEAN}.
 */
int dgs_sddmm_csr_f32(int reduce_op, int64_t M, int64_t K, int64_t F, int64_t nnz,
                      const int32_t *rowptr, const int32_t *col, const float *D1, const float *D2,
                      float *out, dgsStream_t stream);

/* The same product over the cached locality plan of (rowptr, col) - the plan dgs_spmm_plan_build makes for the SpMM, whose
 * backward w.r.t. the edge values this SDDMM is (src/spmm.cpp:52-80): rows up to 64 nnz in row blocks (the D1 slice is read
 * once per row, not once per nnz), longer rows as the plan's units in column-slice order, one slice per XCD; no workspace.
 * plan / info as for dgs_spmm_csr_plan_f32 (provisional info included).  Shapes the fused kernel does not cover (dense graphs
 * on the column-panel schedule, F outside 32..256 or not a multiple of 4, tiny inputs) silently take dgs_sddmm_csr_f32. */
int dgs_sddmm_csr_plan_f32(int reduce_op, int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                           const int32_t *col, const float *D1, const float *D2, float *out, const void *plan,
                           const dgsSpmmPlanInfo *info, dgsStream_t stream);

/* Which schedule dgs_sddmm_csr_f32 / dgs_sddmm_csr_mask_f32 (masked != 0) take for these sizes with 16-byte aligned
 * operands: DGS_SCHED_ROWS (nnz-balanced kernel) or DGS_SCHED_PANEL (column-panel sweep, dense graphs).  Introspection,
 * no reference counterpart; see dgs_spmm_csr_schedule. */
int dgs_sddmm_csr_schedule(int64_t M, int64_t K, int64_t F, int64_t nnz, int masked);

/*
 * Masked SDDMM = backward of max/min w.r.t. the sparse values:
 *   out[e] = sum_k [E[row(e),k] == col(e)] * D1[row(e),k] * D2[col(e),k]
 * Replaces: sddmm_cuda_csr_with_mask(), src/cuda/spmm_cuda.cu:363-382 / sddmmCSR1Scale_with_mask,
 *           include/cuda/sddmm_cuda.cuh:403-507.
 */
int dgs_sddmm_csr_mask_f32(int64_t M, int64_t K, int64_t F, int64_t nnz, const int32_t *rowptr,
                           const int32_t *col, const float *D1, const float *D2, const int32_t *E,
                           float *out, dgsStream_t stream);

/*
 * Both gradients of SpMM-max/min in one pass over the forward's arg ids (either output may be NULL):
 *   gX[j,f] = sum_{(i,j) in A} [E[i,f] == j] * A[i,j] * gC[i,f]      (what dgs_spmm_csr_mask_f32 computes from the CSC arrays)
 *   gW[e]   = sum_f [E[row(e),f] == col(e)] * gC[row(e),f] * X[col(e),f]        (what dgs_sddmm_csr_mask_f32 computes)
 * Replaces: the pair spmm_cuda_with_mask() + sddmm_cuda_csr_with_mask(), src/cuda/spmm_cuda.cu:255-303,363-382, as
 *           they are called from SpMMMax/SpMMMin::backward, src/spmm.cpp:100-214.
 * Every (i,f) has one arg column, so both sums are scatters with M*N sources (instead of nnz*N gathered row pairs);
 * needs only the CSR arrays.  gX [K,N] and gW [nnz] are zeroed inside.  Accumulation uses fp32 atomics: results
 * agree with the masked kernels to rounding but are not bit-reproducible from run to run - callers that need
 * determinism use the two masked entry points above.
 */
int dgs_spmm_arg_backward_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                              const int32_t *col, const float *val, const int32_t *E, const float *gC,
                              const float *X, float *gX, float *gW, dgsStream_t stream);

/* Generalised SpMM of the reference's gspmm-fp demo module (src/gspmm-fp/gspmm.cc:9-28, GSpMM_u_e / GSpMM_u):
 *   C[r,:] = reduce_p compute(val[p], B[col[p],:]),  compute_op: 0 ADD a+b, 1 SUB b-a, 2 MUL a*b, 3 DIV b/a
 * (enum COMPUTEOP, src/gspmm-fp/gspmm.h:16); val == NULL means weight 1 (GSpMM_u).  No arg output.  MUL+sum/mean runs
 * the full SpMM schedule (and needs its workspace); everything else a compact sequential-row kernel. */
enum { DGS_COMPUTE_ADD = 0, DGS_COMPUTE_SUB = 1, DGS_COMPUTE_MUL = 2, DGS_COMPUTE_DIV = 3 };
size_t dgs_gspmm_csr_workspace_bytes(int reduce_op, int compute_op, int64_t M, int64_t N, int64_t nnz);
int dgs_gspmm_csr_f32(int reduce_op, int compute_op, int64_t M, int64_t K, int64_t N, int64_t nnz,
                      const int32_t *rowptr, const int32_t *col, const float *val, const float *B, float *C,
                      void *workspace, size_t workspace_bytes, dgsStream_t stream);

/* COO SDDMM: out[e] = sum_k D1[rowind[e],k] * D2[colind[e],k].
 * Replaces: sddmm_cuda_coo(), src/cuda/spmm_cuda.cu:305-329 (sddmmCOO{4,2,1}Scale, include/cuda/sddmm_cuda.cuh:13-220)
 *           and the standalone C entry src/sddmm/sddmm.h:7. */
int dgs_sddmm_coo_f32(int64_t F, int64_t nnz, const int32_t *rowind, const int32_t *colind, const float *D1,
                      const float *D2, float *out, dgsStream_t stream);

/*
 * Stable CSR -> CSC transpose (integer permutation; exact for any nnz < 2^31, unlike the reference's
 * float-encoded permutation, dgsparse/storage.py:164-169, which breaks at nnz >= 2^24).
 * Replaces: csr2csc_cuda(), src/cuda/spmm_cuda.cu:384-414 / csr2cscKernel (cusparseCsr2cscEx2),
 *           include/cuda/csr2csc.cuh:8-26.
 *   colptr[Kcols+1], row[nnz] outputs; cscval[nnz] (needs val) and perm[nnz] (CSC slot -> CSR slot)
 *   are optional outputs (NULL to skip).  Rectangular matrices supported (reference: square only).
 */
size_t dgs_csr2csc_workspace_bytes(int64_t M, int64_t Kcols, int64_t nnz);
int dgs_csr2csc_i32(int64_t M, int64_t Kcols, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                    const float *val, int32_t *colptr, int32_t *row, float *cscval, int32_t *perm,
                    void *workspace, size_t workspace_bytes, dgsStream_t stream);

/* Row gather / scatter-add used by the multi-GPU halo exchange (new; no reference counterpart):
 *   gather:      dst[i,:] = src[ids[i],:]          scatter_add: dst[ids[i],:] += src[i,:] (ids unique) */
int dgs_gather_rows_f32(int64_t n_ids, int64_t N, const int32_t *ids, const float *src, float *dst,
                        dgsStream_t stream);
int dgs_scatter_add_rows_f32(int64_t n_ids, int64_t N, const int32_t *ids, const float *src, float *dst,
                             dgsStream_t stream);
/* In-place relabel of arg ids: ids[i] = map[ids[i]] where ids[i] >= 0 (-1 = "no arg" stays).  The multi-GPU path
 * computes max/min in the extended [local | halo] column space and hands back GLOBAL column ids (new). */
int dgs_relabel_i32(int64_t n, int32_t *ids, const int32_t *map, dgsStream_t stream);
/* Overlapped MIN of the multi-GPU path for shards whose rows have sorted columns (new; csrc/dist_merge.hip has the
 * argument for exactness).  (C, E) hold the min over the columns this rank owns (the product that ran while the halo was
 * travelling); (Ch, Eh) [2R, N] hold the min over the halo entries of the R rows that have any, row 2r = those that come
 * BEFORE the local columns of shard row rowmap[r] (lower ranks), row 2r + 1 = those AFTER (rowptr2 [2R + 1] is that
 * matrix's row pointer; arg ids are halo slots, col_off is added).  loc_rowptr [rows + 1] is the row pointer of the local
 * product (a row without local entries holds the empty-row 0 / -1, which must not take part).  The three pairs are folded
 * in CSR order with algorithm 0's own MIN step, in place.  nonfinite (device int, may be NULL): when *nonfinite != 0 a
 * product may be NaN and MIN stops being mergeable, so the R rows are recomputed sequentially over the whole shard
 * (rowptr, col in extended ids, val or NULL, B = the [local | halo] buffer) instead.  rowptr2 == NULL (then Ch, Eh,
 * loc_rowptr are ignored and nonfinite is required): redo-only call - the merge was done by dgs_spmm_csr_acc_min_f32.
 * dgs_nonfinite_flag_f32 ORs 1 into *flag when x[0, n) holds a NaN or an infinity (the caller zeroes the flag). */
int dgs_nonfinite_flag_f32(int64_t n, const float *x, int32_t *flag, dgsStream_t stream);
int dgs_spmm_min_merge_f32(int64_t R, int64_t N, const int32_t *rowmap, const int32_t *rowptr2, const float *Ch,
                           const int32_t *Eh, int32_t col_off, const int32_t *loc_rowptr, float *C, int32_t *E,
                           const int32_t *nonfinite, const int32_t *rowptr, const int32_t *col, const float *val,
                           const float *B, dgsStream_t stream);

/* ---- GE-SpMM / SDDMM compatibility entry points (same names and argument order as the reference's
 *      standalone C libraries; default stream, void return) ------------------------------------- */

/* src/ge-spmm/gespmm.h:9-16 */
struct SpMatCsrDescr_t {
  int nrow;
  int ncol;
  int nnz;
  int *indptr;
  int *indices;
  float *data;
};
/* src/ge-spmm/gespmm.h:18-30 (only row-major "transpose_BC=true" layouts exist here) */
enum gespmmAlg_t {
  GESPMM_ALG_SEQREDUCE_ROWBALANCE = 0,
  GESPMM_ALG_PARREDUCE_ROWBALANCE,
  GESPMM_ALG_SEQREDUCE_NNZBALANCE,
  GESPMM_ALG_PARREDUCE_NNZBALANCE,
  GESPMM_ALG_SEQREDUCE_ROWBALANCE_NON_TRANSPOSE,
  GESPMM_ALG_PARREDUCE_ROWBALANCE_NON_TRANSPOSE,
  GESPMM_ALG_SEQREDUCE_NNZBALANCE_NON_TRANSPOSE,
  GESPMM_ALG_PARREDUCE_NNZBALANCE_NON_TRANSPOSE,
  GESPMM_ALG_ROWCACHING_ROWBALANCE,
  GESPMM_ALG_ROWCACHING_NNZBALANCE,
  GESPMM_ALG_DEFAULT
};
/* src/ge-spmm/gespmm.h:32 */
void gespmmCsrSpMM(const struct SpMatCsrDescr_t spmatA, float *B, const int N, float *C,
                   bool transpose_BC, enum gespmmAlg_t alg);
/* src/ge-spmm/gespmm.h:38-41 */
void spmm_cuda(int nrowA, int ncolB, int *rowptr, int *colind, float *values, float *dense, float *out);
void spmm_cuda_no_edge_value(int nrowA, int ncolB, int *rowptr, int *colind, float *values, float *dense,
                             float *out);
/* src/ge-spmm/gespmm.h:34 (selector) and :64-84 (per-algorithm entry points, row-major B/C) */
enum gespmmAlg_t gespmmAlgSel(int dense_ncol, bool transpose_BC);
void csrspmm_parreduce_rowbalance(const struct SpMatCsrDescr_t spmatA, const float *B, const int N, float *C);
void csrspmm_parreduce_nnzbalance(const struct SpMatCsrDescr_t spmatA, const float *B, const int N, float *C);
void csrspmm_seqreduce_rowbalance(const struct SpMatCsrDescr_t spmatA, const float *B, const int N, float *C);
void csrspmm_seqreduce_nnzbalance(const struct SpMatCsrDescr_t spmatA, const float *B, const int N, float *C);
void csrspmm_rowcaching_rowbalance(const struct SpMatCsrDescr_t spmatA, const float *B, const int N, float *C);
void csrspmm_rowcaching_nnzbalance(const struct SpMatCsrDescr_t spmatA, const float *B, const int N, float *C);
/* src/sddmm/sddmm.h:7-10 */
void sddmm_cuda_coo(int k, int nnz, int *rowind, int *colind, float *D1, float *D2, float *out);
void sddmm_cuda_csr(int m, int k, int nnz, int *rowptr, int *colind, float *D1, float *D2, float *out);

#ifdef __cplusplus
}
#endif
#endif /* DGSPARSE_HIP_H */
