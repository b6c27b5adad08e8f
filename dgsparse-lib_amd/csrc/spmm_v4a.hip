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
      !a.accumulate)
    return spmm_run_strict(fm.G, fm.V, a);  // (with a.plan: over the plan's strict table)
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

// ---- device self-tests: the hub chains, and the in-kernel fold (include/dgsparse_hip.h "Device gate") ---------------------------
// Generated inputs (no host buffers: everything is a hash of the index).  Hub test: the default sum with the chains forced on
// against a reference kernel that is beyond suspicion - one thread per (row, feature), one fmaf chain in CSR order - on every
// family of hub workgroup the launchers can pick (kHubShapes below; the general and the single-launch schedule).  Fold test (its own entry point, round 6): sum, max and min over matrices with hundreds of multi-unit
// rows (2 .. 59 units each), folded inside the fused launch, against the SAME launches with the combine kernel behind them
// (identical trees: identical bits, values and arg ids) - for every family of PARTIAL ROW the launchers can pick (whole-line slots,
// slots that share a 128-byte line two / four / eight to a line, scalar-lane slots written with 4-byte agent-scope atomics, two
// feature tiles with their own arrival counters), each several times over, with a streaming kernel loading the fabric from a second
// stream when asked (the case the MI355X guide prices at ~1.1 us per returning atomic: hand-offs fail under uneven load first).
namespace dgs {
namespace selftest {
struct Shape {
  int M, K, N;
  int lens[4];                // lengths of rows 0 .. 3
  int nmid, midlen, midlen2;  // rows 4 .. 4 + nmid - 1: midlen (even ones) / midlen2 (odd ones)
  int tail;                   // every other row
  int nnz() const {
    return lens[0] + lens[1] + lens[2] + lens[3] + ((nmid + 1) / 2) * midlen + (nmid / 2) * midlen2 + (M - 4 - nmid) * tail;
  }
};
constexpr Shape general(int M, int N, int nmid, int midlen) { return Shape{M, 8192, N, {20000, kHubChain + 1, kHubChain, 5000}, nmid, midlen, midlen, 2}; }
constexpr Shape single(int N) { return Shape{40, 8192, N, {20000, 17000, 70, 3}, 0, 0, 0, 3}; }
// hub test.  General schedule: two hub rows (one of them threshold + 1), a row of exactly the threshold and tree rows; the
// production-sized one first (> 2^16 rows), then one per FAMILY of hub workgroup on a few thousand rows (> 2^18 nnz: still the general
// schedule).  A family is strict_hub_coop<V, GP>, GP = strict_hub_gp(G, V) the lanes of a feature slice: 16-byte lanes with 16 / 8 / 4 /
// 2 / 1 lanes per slice (N = 256 / 128 / 64, 32, 16 / 8 / 4) and scalar lanes with 16 / 8 / 4 / 1 (N = 20 / 7 / 3 / 1).  Single-launch
// schedule (spmm_small_hub: the same workgroup behind the row stream): 4-lane, 2-lane and scalar 16-lane slices.
constexpr Shape kHubShapes[] = {general(66000, 64, 0, 0),       general(4096, 128, 1000, 240), general(4096, 256, 1000, 240),
                                general(4096, 32, 1000, 240),   general(4096, 16, 1000, 240),  general(4096, 8, 1000, 240),
                                general(4096, 4, 1000, 240),    general(4096, 20, 1000, 240),  general(4096, 7, 1000, 240),
                                general(4096, 3, 1000, 240),    general(4096, 1, 1000, 240),   single(64),
                                single(20),                     single(8)};
constexpr int kNumHubShapes = sizeof(kHubShapes) / sizeof(kHubShapes[0]);
static_assert(kNumHubShapes <= 32, "per-shape counters sit at [16, 48) of the scratch header");
// fold test: per family 604 multi-unit rows - 59, 36, 12 and 3 units, 300 of 2 (the minimum) and 300 of 6 - whose ~2 500 partial
// rows are written and folded by workgroups all over the chip.  Family = feature width = (lanes per row group, lane vector, tiles).
constexpr Shape fold_shape_of(int N) { return Shape{2048, 8192, N, {15000, 9000, 3000, 700}, 600, 300, 1500, 2}; }
constexpr int kFoldWidths[] = {64,    // G 16, 16-byte lanes: 256-byte slots (whole lines)
                               32,    // G 8: 128-byte slots
                               16,    // G 4: 64-byte slots, two to a line - written by different XCDs at different times
                               8,     // G 2: four to a line
                               4,     // G 1: eight to a line
                               20,    // scalar lanes (N % 4 != 0): 4-byte agent-scope atomic stores / loads, 80-byte slots
                               256,   // two feature tiles of 32 lanes (narrowed): one arrival counter per row AND tile
                               128,   // G 32: 512-byte slots
                               3};    // scalar lanes, 12-byte slots: ten to a line, every word its own fabric write
constexpr int kNumFoldFamilies = sizeof(kFoldWidths) / sizeof(kFoldWidths[0]);
static_assert(kNumFoldFamilies <= 14, "per-family counters sit at [2, 16) of the scratch header");
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float unit_float(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
__global__ __launch_bounds__(kBlock) void gen(Shape sh, int nnz, int *__restrict__ rowptr, int *__restrict__ col,
                                              float *__restrict__ val, float *__restrict__ B) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i <= sh.M) {
    int p = 0;
    for (int r = 0; r < 4 && r < i; r++) p += sh.lens[r];
    if (i > 4) {
      const int m = (int)min((int64_t)sh.nmid, i - 4);
      p += ((m + 1) / 2) * sh.midlen + (m / 2) * sh.midlen2;
    }
    if (i > 4 + sh.nmid) p += ((int)i - 4 - sh.nmid) * sh.tail;
    rowptr[i] = p;
  }
  if (i < nnz) {
    col[i] = (int)(hash32(2u * (unsigned)i + 1u) % (unsigned)sh.K);
    val[i] = unit_float(hash32(2u * (unsigned)i));
  }
  if (i < (int64_t)sh.K * sh.N) B[i] = unit_float(hash32((unsigned)i + 0x9e3779b9u));
}
// one workgroup per row, one thread per feature: algorithm 0 as written (include/cuda/spmm_cuda.cuh:27-47), fmaf-contracted
__global__ __launch_bounds__(kBlock) void reference(int N, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                   const float *__restrict__ val, const float *__restrict__ B,
                                                   float *__restrict__ C) {
  const int r = blockIdx.x, f = threadIdx.x;
  if (f >= N) return;
  const int rs = rowptr[r], re = rowptr[r + 1];
  float acc = 0.0f;
  int p = rs;
  for (; p + 8 <= re; p += 8) {  // eight independent loads in flight, the chain itself strictly in order
    float w[8], x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      w[u] = val[p + u];
      x[u] = B[(int64_t)col[p + u] * N + f];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc = __builtin_fmaf(w[u], x[u], acc);
  }
  for (; p < re; p++) acc = __builtin_fmaf(val[p], B[(int64_t)col[p] * N + f], acc);
  C[(int64_t)r * N + f] = acc;
}
// rows the contract chains (<= T1 nnz, > threshold): identical bits; rows in between: 1e-5 relative
__global__ __launch_bounds__(kBlock) void compare(int M, int N, int thub, const int *__restrict__ rowptr,
                                                  const float *__restrict__ C, const float *__restrict__ R,
                                                  int *__restrict__ bad, int *__restrict__ bad2) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int r = (int)(i / N), len = rowptr[r + 1] - rowptr[r];
  const float c = C[i], ref = R[i];
  const bool ok = (len <= kT1 || len > thub) ? (__float_as_uint(c) == __float_as_uint(ref))
                                             : (fabsf(c - ref) <= 1e-5f * fabsf(ref) + 1e-30f);
  if (!ok) {
    atomicAdd(bad, 1);
    atomicAdd(bad2, 1);
  }
}
__global__ __launch_bounds__(kBlock) void compare_bits(int64_t n, const unsigned *__restrict__ a, const unsigned *__restrict__ b,
                                                       int *__restrict__ bad, int *__restrict__ bad2) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n && a[i] != b[i]) {
    atomicAdd(bad, 1);
    atomicAdd(bad2, 1);
  }
}
// Fabric load for the fold test: every workgroup streams the scratch region `sweeps` times with 16-byte non-temporal loads (reads
// only - whatever the products write meanwhile is just data to it) and leaves a checksum so that the loads cannot be dropped.
__global__ __launch_bounds__(kBlock) void fabric_load(const dgs_f4 *__restrict__ src, int64_t n16, int sweeps, float *__restrict__ sink) {
  float s = 0.0f;
  for (int k = 0; k < sweeps; k++)
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (int64_t)gridDim.x * kBlock) {
      const dgs_f4 v = __builtin_nontemporal_load(src + i);
      s += v[0] + v[3];
    }
  if (s == 123.456f) sink[threadIdx.x & 15] = s;  // (never, as far as the compiler can tell: keeps the loads)
}
// counters at the head of the scratch: [0] hub test, [1] fold test, [2 + f] fold family f, [16 + h] hub shape h; floats 48 .. 63: the sink
constexpr int kCntFold = 2, kCntHub = 16, kCntSink = 48;
struct Layout {
  size_t rowptr, col, val, B, C, R, E1, E2, ws, ws_bytes, total;
};
static Layout layout(const Shape &sh, bool with_arg) {
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  Layout L;
  size_t o = 256;  // the counters
  L.rowptr = o;  o += up((size_t)(sh.M + 1) * 4);
  L.col = o;     o += up((size_t)sh.nnz() * 4);
  L.val = o;     o += up((size_t)sh.nnz() * 4);
  L.B = o;       o += up((size_t)sh.K * sh.N * 4);
  L.C = o;       o += up((size_t)sh.M * sh.N * 4);
  L.R = o;       o += up((size_t)sh.M * sh.N * 4);
  L.E1 = o;      o += with_arg ? up((size_t)sh.M * sh.N * 4) : 0;
  L.E2 = o;      o += with_arg ? up((size_t)sh.M * sh.N * 4) : 0;
  L.ws = o;
  L.ws_bytes = dgs_spmm_csr_workspace_bytes(with_arg ? DGS_MAX : DGS_SUM, sh.M, sh.N, sh.nnz());
  L.total = o + up(L.ws_bytes);
  return L;
}
static void generate(const Shape &sh, const Layout &L, char *base, hipStream_t st) {
  int64_t n = (int64_t)sh.K * sh.N;
  if (n < sh.nnz()) n = sh.nnz();
  if (n < sh.M + 1) n = sh.M + 1;
  hipLaunchKernelGGL(gen, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, sh, sh.nnz(),
                     reinterpret_cast<int *>(base + L.rowptr), reinterpret_cast<int *>(base + L.col),
                     reinterpret_cast<float *>(base + L.val), reinterpret_cast<float *>(base + L.B));
}
static int product(const Shape &sh, const Layout &L, char *base, int op, int hints, float *C, int *E, hipStream_t st) {
  if (hipMemsetAsync(C, 0xFF, (size_t)sh.M * sh.N * 4, st) != hipSuccess) return DGS_ELAUNCH;  // NaN: an unwritten element fails
  if (E && hipMemsetAsync(E, 0x7F, (size_t)sh.M * sh.N * 4, st) != hipSuccess) return DGS_ELAUNCH;
  const FeatMap fm = feat_map(sh.N, true);
  SpmmArgs a{sh.M, sh.K, sh.N, sh.nnz(), reinterpret_cast<int *>(base + L.rowptr), reinterpret_cast<int *>(base + L.col),
             reinterpret_cast<float *>(base + L.val), reinterpret_cast<float *>(base + L.B), C, E, fm.tiles,
             L.ws_bytes ? base + L.ws : nullptr, st, op};
  a.hints = hints;
  return run(fm, a);
}
static int hub_shape(const Shape &sh, int idx, char *base, hipStream_t st) {
  const Layout L = layout(sh, false);
  float *C = reinterpret_cast<float *>(base + L.C), *R = reinterpret_cast<float *>(base + L.R);
  int *cnt = reinterpret_cast<int *>(base);
  generate(sh, L, base, st);
  const int rc = product(sh, L, base, DGS_SUM, kHintForceHub | kHintNoFold, C, nullptr, st);
  if (rc != DGS_OK) return rc;
  static_assert(kBlock >= 256, "one thread per feature, N <= 256");
  hipLaunchKernelGGL(reference, dim3((unsigned)sh.M), dim3((unsigned)(sh.N <= kWave ? kWave : kBlock)), 0, st, sh.N, reinterpret_cast<int *>(base + L.rowptr),
                     reinterpret_cast<int *>(base + L.col), reinterpret_cast<float *>(base + L.val),
                     reinterpret_cast<float *>(base + L.B), R);
  hipLaunchKernelGGL(compare, dim3((unsigned)(((int64_t)sh.M * sh.N + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, sh.M, sh.N,
                     kHubChain, reinterpret_cast<int *>(base + L.rowptr), C, R, cnt, cnt + kCntHub + idx);
  return check_launch();
}
// One family of the fold test: `rounds` folded products of each reduce against ONE product with the combine launch behind it.  The
// last round of each reduce runs under the fabric load (second stream `ld`, released by an event once the inputs exist).
static int fold_family(int fam, int rounds, hipStream_t ld, hipEvent_t ev, char *base, size_t scratch_bytes, hipStream_t st) {
  const Shape sh = fold_shape_of(kFoldWidths[fam]);
  const Layout L = layout(sh, true);
  float *C = reinterpret_cast<float *>(base + L.C), *R = reinterpret_cast<float *>(base + L.R);
  int *E1 = reinterpret_cast<int *>(base + L.E1), *E2 = reinterpret_cast<int *>(base + L.E2);
  int *cnt = reinterpret_cast<int *>(base);
  const int64_t n = (int64_t)sh.M * sh.N;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
  generate(sh, L, base, st);
  const int ops[3] = {DGS_SUM, DGS_MAX, DGS_MIN};
  for (int op : ops) {
    int *e1 = op == DGS_SUM ? nullptr : E1, *e2 = op == DGS_SUM ? nullptr : E2;
    int rc = product(sh, L, base, op, kHintNoFold, R, e2, st);
    if (rc != DGS_OK) return rc;
    for (int k = 0; k < rounds; k++) {
      if (ld && k == rounds - 1) {
        if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(ld, ev, 0) != hipSuccess) return DGS_ELAUNCH;
        hipLaunchKernelGGL(fabric_load, dim3(1024), dim3(kBlock), 0, ld, reinterpret_cast<const dgs_f4 *>(base + 256),
                           (int64_t)((scratch_bytes - 256) / 16), 24, reinterpret_cast<float *>(base) + kCntSink);
      }
      rc = product(sh, L, base, op, kHintForceFold, C, e1, st);
      if (rc != DGS_OK) return rc;
      hipLaunchKernelGGL(compare_bits, grid, dim3(kBlock), 0, st, n, reinterpret_cast<unsigned *>(C), reinterpret_cast<unsigned *>(R), cnt + 1, cnt + kCntFold + fam);
      if (e1) hipLaunchKernelGGL(compare_bits, grid, dim3(kBlock), 0, st, n, reinterpret_cast<unsigned *>(e1), reinterpret_cast<unsigned *>(e2), cnt + 1, cnt + kCntFold + fam);
    }
  }
  return check_launch();
}
static int g_detail[64];  // the counters of the last self-test of this process (dgs_spmm_selftest_detail)
}  // namespace selftest
}  // namespace dgs

extern "C" size_t dgs_spmm_hub_selftest_bytes(void) {
  size_t m = 0;
  for (const selftest::Shape &sh : selftest::kHubShapes) {
    const size_t t = selftest::layout(sh, false).total;
    m = t > m ? t : m;
  }
  for (int w : selftest::kFoldWidths) {
    const size_t t = selftest::layout(selftest::fold_shape_of(w), true).total;
    m = t > m ? t : m;
  }
  return m;
}
extern "C" int dgs_spmm_hub_gate(void) { return hub_gate(); }
// For hosts that keep the verdict of an IDENTICAL (library binary, device model, runtime) triple across processes - DataLoader workers,
// the ranks of a job - instead of paying the self-test in each: sets the hub gate of the current device (1 passed / -1 failed / 0 not
// run), returns the previous state.  As much a bypass as DGS_HUB_CHAIN=16384 is; dgsparse's Python layer uses it only with
// DGS_GATE_CACHE set, and only for verdicts this library wrote itself (dgsparse/_capi.py: _gate_cache_*).
extern "C" int dgs_spmm_hub_gate_assume(int verdict) {
  const int prev = hub_gate();
  hub_gate_set(verdict > 0 ? 1 : (verdict < 0 ? -1 : 0));
  return prev;
}
extern "C" int dgs_spmm_fold_gate(void) { return fold_gate(); }
extern "C" int dgs_spmm_selftest_families(void) { return selftest::kNumFoldFamilies; }
extern "C" int dgs_spmm_selftest_hub_shapes(void) { return selftest::kNumHubShapes; }
extern "C" int dgs_spmm_selftest_detail(int32_t *out, int n) {
  for (int i = 0; i < n && i < 64; i++) out[i] = selftest::g_detail[i];
  return n < 64 ? n : 64;
}
static int selftest_fetch(char *base, hipStream_t st, int lo, int hi, int lo2 = 0, int hi2 = 0) {
  int cnt[64];
  if (hipMemcpyAsync(cnt, base, sizeof(cnt), hipMemcpyDeviceToHost, st) != hipSuccess) return DGS_ELAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return DGS_ELAUNCH;
  for (int i = lo; i < hi; i++) selftest::g_detail[i] = cnt[i];
  for (int i = lo2; i < hi2; i++) selftest::g_detail[i] = cnt[i];
  return DGS_OK;
}
// The fold test.  rounds >= 1 folded products per family and reduce (the library's own default: 3); flags bit 0: load the fabric
// from a second stream during the last round of each; bits 8 .. 8 + families - 1: run only these families (0 = all - only a full
// run sets the gate).  Returns 1 = identical bits everywhere (a full run: fold gate up), 0 = FAILED (gate down), < 0 = DGS_E*.
extern "C" int dgs_spmm_fold_selftest(void *scratch, size_t scratch_bytes, int rounds, int flags, dgsStream_t stream) {
  if (!scratch || scratch_bytes < dgs_spmm_hub_selftest_bytes() || !is_aligned16(scratch) || rounds < 1 || rounds > 1000) return DGS_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *base = static_cast<char *>(scratch);
  if (hipMemsetAsync(base, 0, 256, st) != hipSuccess) return DGS_ELAUNCH;
  hipStream_t ld = nullptr;
  hipEvent_t ev = nullptr;
  if (flags & 1) {
    if (hipStreamCreateWithFlags(&ld, hipStreamNonBlocking) != hipSuccess) return DGS_ELAUNCH;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(ld);
      return DGS_ELAUNCH;
    }
  }
  const unsigned mask = ((unsigned)flags >> 8) & ((1u << selftest::kNumFoldFamilies) - 1u);
  int rc = DGS_OK;
  for (int f = 0; f < selftest::kNumFoldFamilies && rc == DGS_OK; f++)
    if (!mask || (mask >> f & 1)) rc = selftest::fold_family(f, rounds, ld, ev, base, dgs_spmm_hub_selftest_bytes(), st);
  if (ld) {  // the load kernels only read the scratch, but it must outlive them
    (void)hipStreamSynchronize(ld);
    (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(ld);
  }
  if (rc == DGS_OK) rc = selftest_fetch(base, st, 1, 16);
  if (rc != DGS_OK) return rc;
  const bool ok = selftest::g_detail[1] == 0;
  if (!mask || !ok) fold_gate_set(ok ? 1 : -1);
  return ok ? 1 : 0;
}
// The hub test (+ the fold test when DGS_FOLD=2 asks the device to decide).  With an explicit DGS_HUB_CHAIN the gate is not
// consulted (hub_threshold), so the hub test is skipped - state unchanged, returns 1; the same for the fold with DGS_FOLD != 2:
// a process that pins both pays nothing here.
extern "C" int dgs_spmm_hub_selftest(void *scratch, size_t scratch_bytes, dgsStream_t stream) {
  const bool want_hub = tuning().hub_chain == kTuneUnset, want_fold = tuning().fold == 2;
  if (!want_hub && !want_fold) return 1;
  if (!scratch || scratch_bytes < dgs_spmm_hub_selftest_bytes() || !is_aligned16(scratch)) return DGS_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *base = static_cast<char *>(scratch);
  int verdict = 1;
  if (want_hub) {
    if (hipMemsetAsync(base, 0, 256, st) != hipSuccess) return DGS_ELAUNCH;  // the mismatch counters
    for (int h = 0; h < selftest::kNumHubShapes; h++) {
      const int rc = selftest::hub_shape(selftest::kHubShapes[h], h, base, st);  // (stream order: one shape's arrays are dead when the next one's are written)
      if (rc != DGS_OK) return rc;
    }
    const int rc = selftest_fetch(base, st, 0, 1, selftest::kCntHub, selftest::kCntHub + selftest::kNumHubShapes);
    if (rc != DGS_OK) return rc;
    hub_gate_set(selftest::g_detail[0] == 0 ? 1 : -1);
    verdict = selftest::g_detail[0] == 0 ? 1 : 0;
  }
  if (want_fold) {
    const int rc = dgs_spmm_fold_selftest(scratch, scratch_bytes, 3, 1, stream);
    if (rc < 0) return rc;
  }
  return verdict;
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
  a.hints = public_hints(algorithm);
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
  // (a strict call uses the plan's strict table unless an experiment override moved the class thresholds)
  const bool planned = plan && info && (!strict || strict_over_plan_ok()) && nnz > 0 && !tiny_problem(M, nnz) &&
                       dgs_spmm_csr_schedule(reduce_op, M, K, N, nnz) == DGS_SCHED_ROWS;
  if (planned && !is_aligned16(plan)) return DGS_EINVAL;
  const size_t need = planned ? dgs_spmm_csr_plan_workspace_bytes(reduce_op, M, N, nnz, info)
                              : dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz);
  if (need > 0 && (!workspace || workspace_bytes < need)) return DGS_EWORKSPACE;
  const bool al = is_aligned16(B) && is_aligned16(C) && (!arg || is_aligned16(E)) && (need == 0 || is_aligned16(workspace)) &&
                  (!bias || is_aligned16(bias));
  const FeatMap fm = feat_map(N, al);
  SpmmArgs a{M, K, N, nnz, rowptr, col, val, B, C, arg ? E : nullptr, fm.tiles, need ? workspace : nullptr, st, reduce_op};
  a.hints = public_hints(algorithm);
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
// virt_n != 0: the "around" form - both sides in one launch, the old pair a virtual entry of its row.
static int acc_min_launch(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                          const float *val, const float *B, float *C, int32_t *E, const int32_t *rowmap, int32_t col_off,
                          int32_t precedes, int32_t virt_lo, int32_t virt_n, const void *plan, const dgsSpmmPlanInfo *info,
                          void *workspace, size_t workspace_bytes, dgsStream_t stream) {
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
  if (virt_n) a.acc.vc = VirtCols{C, virt_lo, virt_n};
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

extern "C" int dgs_spmm_csr_acc_min_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                        const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                        const int32_t *rowmap, int32_t col_off, int32_t precedes, const void *plan,
                                        const dgsSpmmPlanInfo *info, void *workspace, size_t workspace_bytes,
                                        dgsStream_t stream) {
  return acc_min_launch(M, K, N, nnz, rowptr, col, val, B, C, E, rowmap, col_off, precedes, 0, 0, plan, info, workspace,
                        workspace_bytes, stream);
}

// The "around" form (spmm_impl.h AccArg): the matrix has K columns, of which [virt_lo, virt_lo + virt_n) are virtual - entry
// (r, virt_lo + rowmap[r]) stands for what (C, E)[rowmap[r]] hold, at its place in the row - and B has K - virt_n rows: column
// c < virt_lo is row c of B, c >= virt_lo + virt_n is row c - virt_n.  (C, E)[rowmap[r]] = the MIN over the whole row in row
// order, E = the arg the output held where the virtual entry is the first minimum, else the winning column mapped like the rows
// of B (c or c - virt_n) + col_off.  A row WITHOUT a virtual entry replaces its output pair.  Two rows must not share an
// output row, and the only row that may name virtual column virt_lo + j is the one with rowmap[r] == j.
extern "C" int dgs_spmm_csr_acc_min_around_f32(int64_t M, int64_t K, int64_t N, int64_t nnz, const int32_t *rowptr,
                                               const int32_t *col, const float *val, const float *B, float *C, int32_t *E,
                                               const int32_t *rowmap, int32_t col_off, int32_t virt_lo, int32_t virt_n,
                                               const void *plan, const dgsSpmmPlanInfo *info, void *workspace,
                                               size_t workspace_bytes, dgsStream_t stream) {
  if (M == 0 || N == 0) return DGS_OK;  // a rank without rows (n_local = 0 gives virt_n = 0 too): nothing to fold, not an error
  if (virt_lo < 0 || virt_n <= 0 || (int64_t)virt_lo + virt_n > K) return DGS_EINVAL;
  return acc_min_launch(M, K, N, nnz, rowptr, col, val, B, C, E, rowmap, col_off, 0, virt_lo, virt_n, plan, info, workspace,
                        workspace_bytes, stream);
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
