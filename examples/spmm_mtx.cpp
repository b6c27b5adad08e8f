// spmm_mtx.cpp -- standalone driver over the C ABI, the counterpart of the reference's example/ge-spmm/spmm.cu
// and example/sddmm/sddmm.cu: load a MatrixMarket file (pattern only, symmetrised/sorted like read_mtx_file,
// example/util/sp_util.hpp:171-251), fill values and the dense operand with {0, .1, .2} (sp_util.hpp:44-48), run
// SpMM (sum/max/min/mean) and SDDMM through libdgsparse_hip.so, check against a host loop, then time 10 warm-up +
// 100 launches with hipEvents and print GFLOP/s = 2*nnz*N/t (spmm.cu:145-164).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/spmm_mtx.cpp -Ldgsparse-lib_amd/dgsparse -ldgsparse_hip \
//         -Wl,-rpath,$PWD/dgsparse-lib_amd/dgsparse -o examples/spmm_mtx
//   examples/spmm_mtx graph.mtx [N=64]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dgsparse_hip.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

static bool load_mtx(const char *path, int &nrow, int &ncol, std::vector<int> &indptr, std::vector<int> &indices) {
  FILE *f = fopen(path, "r");
  if (!f) return false;
  char line[1024];
  if (!fgets(line, sizeof line, f)) return false;
  std::string banner(line);
  std::transform(banner.begin(), banner.end(), banner.begin(), ::tolower);
  if (banner.find("%%matrixmarket") != 0 || banner.find("coordinate") == std::string::npos) return false;
  const bool pattern = banner.find("pattern") != std::string::npos;
  const bool symmetric = banner.find("symmetric") != std::string::npos;
  do {
    if (!fgets(line, sizeof line, f)) return false;
  } while (line[0] == '%');
  long nnz;
  if (sscanf(line, "%d %d %ld", &nrow, &ncol, &nnz) != 3) return false;
  std::vector<long long> key;
  key.reserve(symmetric ? 2 * nnz : nnz);
  for (long i = 0; i < nnz; i++) {
    int r, c;
    if (!fgets(line, sizeof line, f) || sscanf(line, "%d %d", &r, &c) != 2) return false;
    (void)pattern;  // values, when present, are dropped like the reference does
    key.push_back((long long)(r - 1) * ncol + (c - 1));
    if (symmetric) key.push_back((long long)(c - 1) * ncol + (r - 1));
  }
  fclose(f);
  std::sort(key.begin(), key.end());
  if (symmetric) key.erase(std::unique(key.begin(), key.end()), key.end());
  indptr.assign(nrow + 1, 0);
  indices.resize(key.size());
  for (size_t i = 0; i < key.size(); i++) {
    indptr[key[i] / ncol + 1]++;
    indices[i] = (int)(key[i] % ncol);
  }
  for (int r = 0; r < nrow; r++) indptr[r + 1] += indptr[r];
  return true;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s matrix.mtx [N]\n", argv[0]);
    return 1;
  }
  const int N = argc > 2 ? atoi(argv[2]) : 64;
  int M, K;
  std::vector<int> indptr, indices;
  if (!load_mtx(argv[1], M, K, indptr, indices)) {
    fprintf(stderr, "could not read %s\n", argv[1]);
    return 1;
  }
  const long nnz = (long)indices.size();
  printf("matrix %s: %d x %d, nnz %ld, N = %d, library abi %d (%s)\n", argv[1], M, K, nnz, N, dgs_version(), dgs_arch());
  srand(0);
  std::vector<float> val(nnz), B((size_t)K * N), D1((size_t)M * N), C((size_t)M * N), Cref((size_t)M * N);
  for (auto &v : val) v = (float)(rand() % 3) / 10;
  for (auto &v : B) v = (float)(rand() % 3) / 10;
  for (auto &v : D1) v = (float)(rand() % 3) / 10;

  int *d_ptr, *d_idx, *d_E;
  float *d_val, *d_B, *d_D1, *d_C, *d_out;
  void *d_ws;
  HIP_OK(hipMalloc(&d_ptr, (M + 1) * sizeof(int)));
  HIP_OK(hipMalloc(&d_idx, std::max<long>(nnz, 1) * sizeof(int)));
  HIP_OK(hipMalloc(&d_val, std::max<long>(nnz, 1) * sizeof(float)));
  HIP_OK(hipMalloc(&d_out, std::max<long>(nnz, 1) * sizeof(float)));
  HIP_OK(hipMalloc(&d_B, B.size() * sizeof(float)));
  HIP_OK(hipMalloc(&d_D1, D1.size() * sizeof(float)));
  HIP_OK(hipMalloc(&d_C, C.size() * sizeof(float)));
  HIP_OK(hipMalloc(&d_E, C.size() * sizeof(int)));
  const size_t wsb = std::max(dgs_spmm_csr_workspace_bytes(DGS_MAX, M, N, nnz), (size_t)256);
  HIP_OK(hipMalloc(&d_ws, wsb));
  HIP_OK(hipMemcpy(d_ptr, indptr.data(), (M + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_idx, indices.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_val, val.data(), nnz * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_B, B.data(), B.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_D1, D1.data(), D1.size() * sizeof(float), hipMemcpyHostToDevice));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));

  // Device gate of the hub chains (include/dgsparse_hip.h): a C caller runs the library's self-test once per device, then tells
  // the launches what it knows about the matrix - here: the longest row (every matrix the reference ships is far below the
  // threshold, so sum / mean keep the plain single-launch kernel).
  {
    void *d_st;
    const size_t sb = dgs_spmm_hub_selftest_bytes();
    HIP_OK(hipMalloc(&d_st, sb));
    const int v = dgs_spmm_hub_selftest(d_st, sb, st);
    printf("hub-chain self-test on this device: %s\n", v == 1 ? "passed" : (v == 0 ? "FAILED (chains off)" : dgs_strerror(v)));
    HIP_OK(hipFree(d_st));
  }
  int maxlen = 0;
  for (int r = 0; r < M; r++) maxlen = std::max(maxlen, indptr[r + 1] - indptr[r]);
  const int thub = dgs_spmm_hub_threshold();
  const int alg = (thub > 0 && maxlen <= thub) ? DGS_ALG_NO_HUB_ROWS : 0;
  printf("longest row %d nnz, hub threshold %d -> algorithm hints 0x%x\n", maxlen, thub, alg);

  // Cached locality plan (include/dgsparse_hip.h): built once from (rowptr, col) when the shape takes the row-stream
  // schedule; the same calls then run over the plan's tables.
  void *d_plan = nullptr, *d_pws = nullptr;
  dgsSpmmPlanInfo pinfo;
  size_t pwsb = 0;
  const bool planned = nnz > 0 && dgs_spmm_csr_schedule(DGS_SUM, M, K, N, nnz) == DGS_SCHED_ROWS;
  if (planned) {
    const size_t pb = dgs_spmm_plan_bytes(M, K, nnz), bb = dgs_spmm_plan_workspace_bytes(M, K, nnz);
    void *d_big, *d_bws;
    HIP_OK(hipMalloc(&d_big, pb));
    HIP_OK(hipMalloc(&d_bws, bb));
    int rc = dgs_spmm_plan_build(M, K, nnz, d_ptr, d_idx, d_big, pb, d_bws, bb, &pinfo, st);
    if (rc) {
      fprintf(stderr, "dgs_spmm_plan_build: %s\n", dgs_strerror(rc));
      return 3;
    }
    const size_t cb = dgs_spmm_plan_compact_bytes(&pinfo);
    HIP_OK(hipMalloc(&d_plan, cb));
    rc = dgs_spmm_plan_compact(d_big, &pinfo, d_plan, cb, nnz, st);
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipFree(d_big));
    HIP_OK(hipFree(d_bws));
    if (rc) return 3;
    pwsb = dgs_spmm_csr_plan_workspace_bytes(DGS_MAX, M, N, nnz, &pinfo);
    HIP_OK(hipMalloc(&d_pws, pwsb));
    printf("plan: %d units, %d long rows, %d partial rows, %zu bytes\n", pinfo.n_units, pinfo.n_long, pinfo.n_pslots, cb);
  }

  const char *names[4] = {"sum", "max", "min", "mean"};
  int bad_total = 0;
  for (int op = 0; op < 4; op++) {
    // host check (sequential CSR order, the algorithm-0 semantics)
    for (int r = 0; r < M; r++)
      for (int f = 0; f < N; f++) {
        const int s = indptr[r], e = indptr[r + 1];
        float res = op == DGS_MAX ? (float)INT32_MIN : op == DGS_MIN ? (float)INT32_MAX : 0.f;
        double acc = 0.0;  // sum/mean: exact yardstick (a sequential fp32 chain drifts by > 1e-5 on rows of 10^3+ nnz)
        for (int p = s; p < e; p++) {
          const float t = val[p] * B[(size_t)indices[p] * N + f];
          if (op == DGS_MAX) res = res < t ? t : res;
          else if (op == DGS_MIN) res = res < t ? res : t;
          else acc += (double)t;
        }
        if (op == DGS_SUM || op == DGS_MEAN) res = (float)(op == DGS_MEAN && e > s ? acc / (double)(e - s) : acc);
        Cref[(size_t)r * N + f] = e > s ? res : 0.f;
      }
    int rc = dgs_spmm_csr_f32(op, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, d_E, alg, d_ws, wsb, st);
    if (rc) {
      fprintf(stderr, "dgs_spmm_csr_f32: %s\n", dgs_strerror(rc));
      return 3;
    }
    HIP_OK(hipMemcpyAsync(C.data(), d_C, C.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    long bad = 0;
    for (size_t i = 0; i < C.size(); i++)
      if (fabsf(C[i] - Cref[i]) > 1e-5f * fabsf(Cref[i]) + 2e-6f) bad++;
    bad_total += bad != 0;
    for (int i = 0; i < 10; i++) dgs_spmm_csr_f32(op, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, d_E, alg, d_ws, wsb, st);
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; i++) dgs_spmm_csr_f32(op, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, d_E, alg, d_ws, wsb, st);
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("[SpMM-%s] check %s (%ld mismatches)  time %.6f ms  throughput %.2f GFLOP/s\n", names[op],
           bad ? "FAILED" : "passed", bad, ms / 100, 2.0 * nnz * N / (ms / 100) * 1e-6);
    if (planned) {  // the same product over the cached plan
      rc = dgs_spmm_csr_plan_f32(op, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, d_E, d_plan, &pinfo, d_pws, pwsb, st);
      if (rc) {
        fprintf(stderr, "dgs_spmm_csr_plan_f32: %s\n", dgs_strerror(rc));
        return 3;
      }
      HIP_OK(hipMemcpyAsync(C.data(), d_C, C.size() * sizeof(float), hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      long pbad = 0;
      for (size_t i = 0; i < C.size(); i++)
        if (fabsf(C[i] - Cref[i]) > 1e-5f * fabsf(Cref[i]) + 2e-6f) pbad++;
      bad_total += pbad != 0;
      for (int i = 0; i < 10; i++)
        dgs_spmm_csr_plan_f32(op, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, d_E, d_plan, &pinfo, d_pws, pwsb, st);
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < 100; i++)
        dgs_spmm_csr_plan_f32(op, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, d_E, d_plan, &pinfo, d_pws, pwsb, st);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      printf("[SpMM-%s, plan] verification %s (%ld mismatches)  time %.6f ms  throughput %.2f GFLOP/s\n", names[op],
             pbad ? "FAILED" : "ok", pbad, ms / 100, 2.0 * nnz * N / (ms / 100) * 1e-6);
    }
  }
  {  // strict-order sum (DGS_ALG_STRICT_NOFMA): the reference's own host loop (example/util/sp_util.hpp:73-83: C += val * B, a
     // sequential fp32 chain per (row, feature), product rounded before the add) reproduced BIT FOR BIT for every row length
    for (int r = 0; r < M; r++)
      for (int f = 0; f < N; f++) {
        volatile float acc = 0.f;  // volatile: no host-side contraction or reassociation of the chain
        for (int p = indptr[r]; p < indptr[r + 1]; p++) {
          volatile float t = val[p] * B[(size_t)indices[p] * N + f];
          acc = acc + t;
        }
        Cref[(size_t)r * N + f] = acc;
      }
    int rc = dgs_spmm_csr_f32(DGS_SUM, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, nullptr, DGS_ALG_STRICT_NOFMA, d_ws, wsb, st);
    if (rc) {
      fprintf(stderr, "dgs_spmm_csr_f32 (strict): %s\n", dgs_strerror(rc));
      return 3;
    }
    HIP_OK(hipMemcpyAsync(C.data(), d_C, C.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    long bad = 0;
    for (size_t i = 0; i < C.size(); i++) bad += memcmp(&C[i], &Cref[i], sizeof(float)) != 0;
    bad_total += bad != 0;
    for (int i = 0; i < 10; i++)
      dgs_spmm_csr_f32(DGS_SUM, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, nullptr, DGS_ALG_STRICT_NOFMA, d_ws, wsb, st);
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; i++)
      dgs_spmm_csr_f32(DGS_SUM, M, K, N, nnz, d_ptr, d_idx, d_val, d_B, d_C, nullptr, DGS_ALG_STRICT_NOFMA, d_ws, wsb, st);
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("[SpMM-sum, strict order] bit-exact vs the sequential host loop: %s (%ld of %zu elements differ)  time %.6f ms\n",
           bad ? "FAILED" : "yes", bad, C.size(), ms / 100);
  }
  {  // SDDMM
    std::vector<float> out(nnz), ref(nnz);
    for (int r = 0; r < M; r++)
      for (int p = indptr[r]; p < indptr[r + 1]; p++) {
        float acc = 0;
        for (int k = 0; k < N; k++) acc += D1[(size_t)r * N + k] * B[(size_t)indices[p] * N + k];
        ref[p] = acc;
      }
    int rc = dgs_sddmm_csr_f32(DGS_SUM, M, K, N, nnz, d_ptr, d_idx, d_D1, d_B, d_out, st);
    if (rc) return 3;
    HIP_OK(hipMemcpyAsync(out.data(), d_out, nnz * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    long bad = 0;
    for (long i = 0; i < nnz; i++)
      if (fabsf(out[i] - ref[i]) > 1e-5f * fabsf(ref[i]) + 2e-6f) bad++;
    bad_total += bad != 0;
    for (int i = 0; i < 10; i++) dgs_sddmm_csr_f32(DGS_SUM, M, K, N, nnz, d_ptr, d_idx, d_D1, d_B, d_out, st);
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; i++) dgs_sddmm_csr_f32(DGS_SUM, M, K, N, nnz, d_ptr, d_idx, d_D1, d_B, d_out, st);
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("[SDDMM] check %s (%ld mismatches)  time %.6f ms  throughput %.2f GFLOP/s\n", bad ? "FAILED" : "passed", bad,
           ms / 100, 2.0 * nnz * N / (ms / 100) * 1e-6);
  }
  return bad_total ? 4 : 0;
}
