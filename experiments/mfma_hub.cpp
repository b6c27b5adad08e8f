// mfma_hub.cpp -- the MFMA question of north_star, answered with a number.
// "MFMA only for the dense-feature tile of blocked/ELL-packed rows where it is genuinely a dense panel GEMM": the only
// candidate in BASELINE.json's configurations is the hub-row block of the Reddit-shaped graph (rows > 4096 nnz: 330 rows,
// 2.23 M nnz, 2.9 % dense on average, 9.3 % at most, over K = 232 965 columns, N = 128).  This program builds that block
// (same degree law, uniform columns), ELL-packs it the way an MFMA kernel needs it -- blocks of 32 rows, the UNION of
// their columns as the k dimension, a dense 32 x U tile of values (zeros where a row lacks the column) -- and times
//   (a) v_mfma_f32_32x32x2_f32 over the gathered B panel: one wave = one 32-row x 32-feature output tile, k split over
//       workgroups, partial tiles added with fp32 atomics;
//   (b) the sparse unit path on the same rows (one wave per 256-nnz unit, 32-lane groups x float4, 8 gathers in
//       flight, xor-butterfly, fp32 atomics for the partial rows) -- a lean copy of csrc/spmm_impl.h's unit loop;
// and checks (a) against (b).  f32 MFMA runs at the f32 VECTOR rate on gfx950 (MI355X_MICROARCH.md: 157 TF both), so a
// tile that is d dense reaches at most d x peak: the packed hub block is 4-5 % dense.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 experiments/mfma_hub.cpp -o /tmp/mh && /tmp/mh
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int N = 128;

// (a) grid = (row blocks, k chunks); 256 threads = 4 waves, wave w owns features [32w, 32w+32).
// At[blk][k][32] = value of row (32*blk + i) at union column k (k-major so that a wave reads 2 x 32 floats per step);
// ucol[blk][k] = the column id.  acc layout of v_mfma_f32_32x32x2_f32: lane l holds column j = l % 32 and rows
// i = 8*(v/4) + 4*(l/32) + v%4 for v = 0..15.
__global__ __launch_bounds__(256) void k_mfma(const float *__restrict__ At, const int *__restrict__ ucol,
                                              const int *__restrict__ ustart, const float *__restrict__ B,
                                              float *__restrict__ C, int kchunk) {
  const int blk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u0 = ustart[blk], u1 = ustart[blk + 1];
  const int k0 = u0 + blockIdx.y * kchunk, k1 = min(u1, k0 + kchunk);
  if (k0 >= k1) return;
  f16v acc = {0};
  const int j = lane & 31, kh = lane >> 5;
  const float *Bp = B + wave * 32 + j;
  for (int k = k0; k < k1; k += 16) {  // 8 MFMAs per iteration, all loads issued first
    float a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int kk = k + 2 * q + kh;
      const bool ok = kk < k1;
      a[q] = ok ? At[(size_t)kk * 32 + j] : 0.f;
      b[q] = ok ? Bp[(size_t)ucol[kk] * N] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
  }
  float *Cb = C + (size_t)blk * 32 * N + wave * 32 + j;
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int i = 8 * (v / 4) + 4 * kh + (v % 4);
    unsafeAtomicAdd(Cb + (size_t)i * N, acc[v]);
  }
}

// (b) one wave per unit of <= 256 nnz: 2 groups of 32 lanes x float4, interleaved nnz, 8 gathers in flight
__global__ __launch_bounds__(256) void k_units(const int *__restrict__ urow, const int *__restrict__ ubeg,
                                               const int *__restrict__ ulen, int nunits, const int *__restrict__ col,
                                               const float *__restrict__ val, const float *__restrict__ B,
                                               float *__restrict__ C) {
  const int lane = threadIdx.x & 63, g = lane >> 5, l = lane & 31;
  for (int u = blockIdx.x * 4 + (threadIdx.x >> 6); u < nunits; u += gridDim.x * 4) {
    const int p0 = ubeg[u], n = ulen[u];
    f4 acc = {0, 0, 0, 0};
    for (int j = g; j < n; j += 16) {
      f4 x[8];
      float w[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int p = j + 2 * q;
        const bool ok = p < n;
        w[q] = ok ? val[p0 + p] : 0.f;
        const int c = ok ? col[p0 + p] : 0;
        x[q] = *reinterpret_cast<const f4 *>(B + (size_t)c * N + l * 4);
      }
#pragma unroll
      for (int q = 0; q < 8; q++) acc += w[q] * x[q];
    }
    for (int v = 0; v < 4; v++) acc[v] += __shfl_xor(acc[v], 32, 64);
    if (g == 0) {
      float *Cp = C + (size_t)urow[u] * N + l * 4;
      for (int v = 0; v < 4; v++) unsafeAtomicAdd(Cp + v, acc[v]);
    }
  }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int K = 232965, H = 330, dmax = 21657;
  std::mt19937_64 rng(1);
  // hub degrees: Pareto tail above 4096 (alpha = 3.5 as the Reddit-shaped generator), capped
  std::vector<int> deg(H);
  long nnz = 0;
  for (int i = 0; i < H; i++) {
    const double u = std::uniform_real_distribution<double>(0, 1)(rng);
    deg[i] = std::min(dmax, (int)(4096.0 * std::pow(1.0 - u, -1.0 / 2.5)));
    nnz += deg[i];
  }
  std::vector<int> rowptr(H + 1, 0), col;
  std::vector<float> val;
  std::vector<char> mark(K);
  for (int i = 0; i < H; i++) {
    std::fill(mark.begin(), mark.end(), 0);
    int got = 0;
    while (got < deg[i]) {
      const int c = (int)(rng() % K);
      if (!mark[c]) { mark[c] = 1; got++; }
    }
    for (int c = 0; c < K; c++)
      if (mark[c]) { col.push_back(c); val.push_back((float)(rng() % 1000) / 1000.f); }
    rowptr[i + 1] = (int)col.size();
  }
  printf("hub block: %d rows, %ld nnz (mean %.0f, %.1f %% dense), K = %d, N = %d\n", H, nnz, (double)nnz / H,
         100.0 * nnz / H / K, K, N);
  // ELL-pack: blocks of 32 rows, union of columns, dense k-major tile
  const int nblk = (H + 31) / 32;
  std::vector<int> ustart(nblk + 1, 0), ucol;
  std::vector<float> At;
  for (int b = 0; b < nblk; b++) {
    std::vector<int> slot(K, -1);
    std::vector<int> u;
    for (int i = 32 * b; i < std::min(H, 32 * b + 32); i++)
      for (int p = rowptr[i]; p < rowptr[i + 1]; p++)
        if (slot[col[p]] < 0) { slot[col[p]] = 0; }
    for (int c = 0; c < K; c++)
      if (slot[c] == 0) { slot[c] = (int)u.size(); u.push_back(c); }
    const size_t base = At.size();
    At.resize(base + u.size() * 32, 0.f);
    for (int i = 32 * b; i < std::min(H, 32 * b + 32); i++)
      for (int p = rowptr[i]; p < rowptr[i + 1]; p++) At[base + (size_t)slot[col[p]] * 32 + (i - 32 * b)] = val[p];
    ucol.insert(ucol.end(), u.begin(), u.end());
    ustart[b + 1] = (int)ucol.size();
  }
  const double tile_fill = (double)nnz / ((double)ucol.size() * 32);
  printf("ELL pack: %d blocks of 32 rows, %zu union columns in total (%.0f per block), tile fill %.1f %%, dense tile %.1f MB\n",
         nblk, ucol.size(), (double)ucol.size() / nblk, 100 * tile_fill, At.size() * 4 / 1e6);
  // units for (b)
  std::vector<int> urow, ubeg, ulen;
  for (int i = 0; i < H; i++)
    for (int p = rowptr[i]; p < rowptr[i + 1]; p += 256) {
      urow.push_back(i); ubeg.push_back(p); ulen.push_back(std::min(256, rowptr[i + 1] - p));
    }
  float *dB, *dAt, *dval, *dC1, *dC2;
  int *ducol, *dustart, *dcol, *durow, *dubeg, *dulen;
  hipMalloc(&dB, (size_t)K * N * 4);
  std::vector<float> hB((size_t)K * N);
  for (auto &x : hB) x = (float)(rng() % 2000) / 2000.f;
  hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
#define UP(d, h) hipMalloc(&d, h.size() * sizeof(h[0])); hipMemcpy(d, h.data(), h.size() * sizeof(h[0]), hipMemcpyHostToDevice)
  UP(dAt, At); UP(ducol, ucol); UP(dustart, ustart); UP(dcol, col); UP(dval, val); UP(durow, urow); UP(dubeg, ubeg); UP(dulen, ulen);
  const size_t cbytes = (size_t)nblk * 32 * N * 4;
  hipMalloc(&dC1, cbytes);
  hipMalloc(&dC2, cbytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timeit = [&](auto fn, const char *what, double flops) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
      hipEventRecord(e0);
      fn();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) best = std::min(best, ms);
    }
    printf("%-58s %8.3f ms  %7.2f TFLOP/s executed, %6.2f TFLOP/s useful (2*nnz*N)\n", what, best, flops / best / 1e9,
           2.0 * nnz * N / best / 1e9);
    return best;
  };
  int maxu = 0;
  for (int b = 0; b < nblk; b++) maxu = std::max(maxu, ustart[b + 1] - ustart[b]);
  for (int ksplit : {16, 64, 128, 256, 512}) {
    const int kchunk = ((maxu + ksplit - 1) / ksplit + 15) & ~15;
    char what[128];
    snprintf(what, sizeof what, "(a) MFMA 32x32x2 f32, ELL-packed, k split %d", ksplit);
    timeit([&] {
      hipMemsetAsync(dC1, 0, cbytes, 0);
      hipLaunchKernelGGL(k_mfma, dim3(nblk, ksplit), dim3(256), 0, 0, dAt, ducol, dustart, dB, dC1, kchunk);
    }, what, 2.0 * ucol.size() * 32 * N);
  }
  timeit([&] {
    hipMemsetAsync(dC2, 0, cbytes, 0);
    hipLaunchKernelGGL(k_units, dim3(1024), dim3(256), 0, 0, durow, dubeg, dulen, (int)urow.size(), dcol, dval, dB, dC2);
  }, "(b) sparse unit path (wave per 256 nnz, 8 gathers in flight)", 2.0 * nnz * N);
  std::vector<float> c1((size_t)nblk * 32 * N), c2(c1.size());
  hipMemcpy(c1.data(), dC1, cbytes, hipMemcpyDeviceToHost);
  hipMemcpy(c2.data(), dC2, cbytes, hipMemcpyDeviceToHost);
  double worst = 0;
  for (size_t i = 0; i < (size_t)H * N; i++) worst = std::max(worst, std::fabs((double)c1[i] - c2[i]) / std::max(1.0, std::fabs((double)c2[i])));
  printf("max rel difference MFMA vs sparse: %.2e  (both fp32, different summation orders)\n", worst);
  return worst < 1e-4 ? 0 : 1;
}
