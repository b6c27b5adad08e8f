// gather_sizes.cpp -- micro-benchmark: which level of the hierarchy bounds the B-row gather?
// 16M gathers of ROWB-byte rows (one dwordx4 per lane, ROWB/16 lanes per row), 8 in flight per lane, uniformly random
// rows of a table of 2 MB .. 2 GB.  2-4 MB sits in every XCD's L2, <= ~128 MB can live in the 256 MB Infinity Cache
// (memory side, shared by the XCDs), larger tables come from HBM.  If MALL hits were cheaper than HBM on the
// L2-miss path the 32..128 MB rows would run well above the 512 MB+ ones.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 experiments/gather_sizes.cpp -o /tmp/gs && /tmp/gs
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int LPR>  // lanes per row: 8 = 128-B rows (N=32), 16 = 256-B (N=64), 32 = 512-B (N=128)
__global__ __launch_bounds__(256) void k_gather(const float *B, const int *idx, int per_group, int rowmask, float *out) {
  const int l = threadIdx.x & (LPR - 1);
  const long grp = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
  f4 acc = {0, 0, 0, 0};
  for (int i = 0; i < per_group; i += 8) {
    f4 x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const long k = grp * per_group + i + u;
      x[u] = *reinterpret_cast<const f4 *>(B + (long)(idx[k] & rowmask) * (LPR * 4) + l * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += x[u];
  }
  if (acc[0] == 123.456f) out[grp] = acc[1] + acc[2] + acc[3];
}

template <int LPR>
static void run(const float *B, const int *idx, float *out, long total, long maxrows) {
  const int per_group = 64;
  const long groups = total / per_group;
  const int blocks = (int)(groups * LPR / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (long rows = 8192 * 16 / LPR; rows <= maxrows; rows *= 2) {
    const double mb = rows * LPR * 16.0 / 1048576.0;
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_gather<LPR>, dim3(blocks), dim3(256), 0, 0, B, idx, per_group, (int)(rows - 1), out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("row %4d B  table %8.1f MB  %7.3f ms  %6.2f TB/s gathered\n", LPR * 16, mb, best,
           total * LPR * 16.0 / best / 1e9);
  }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const long total = 1 << 24;
  const long bytes = 2l << 30;
  float *B, *out;
  int *idx;
  hipMalloc(&B, bytes);
  hipMalloc(&out, total);
  hipMalloc(&idx, total * 4);
  hipMemset(B, 0, bytes);
  std::vector<int> h(total);
  srand(1);
  for (long i = 0; i < total; i++) h[i] = (int)(((long)rand() * 32768 + rand()) & 0x7fffffff);
  hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
  run<8>(B, idx, out, total, bytes / 128);
  run<16>(B, idx, out, total, bytes / 256);
  run<32>(B, idx, out, total / 2, bytes / 512);
  return 0;
}
