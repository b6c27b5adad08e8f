// gather_policy.cpp -- micro-benchmark: can a cache-policy bit on the COLD gathers protect a hot set in the XCD L2?
// Every wave alternates instructions of hot gathers (256-byte rows drawn from a HOT_MB table, default policy) and cold
// gathers (rows of a 1 GB table, policy under test: default / nt / sc1 / sc0 sc1 / nt sc1 / nt sc0 sc1).  The hot set is
// sized so that it fits the 4 MiB L2 alone but not together with the cold stream under LRU.  If a policy keeps the cold
// rows out of the L2, the hot half runs at L2 speed and the total approaches (cold bytes) / fabric rate.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result experiments/gather_policy.cpp -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void cold_load(f4 &x, const float *p) {
  if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(x) : "v"(p) : "memory");
  if constexpr (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(x) : "v"(p) : "memory");
}

// 16 lanes per 256-byte row; per step 4 hot + 4 cold gathers in flight per lane.  MODE 0: both, 1: hot only, 2: cold only
template <int POL, int MODE>
__global__ __launch_bounds__(256) void k(const float *hot, const float *cold, const int *idx, int per_group, int hotmask,
                                        int coldmask, float *out) {
  const int l = threadIdx.x & 15;
  const long grp = ((long)blockIdx.x * 256 + threadIdx.x) / 16;
  f4 acc = {0, 0, 0, 0};
  for (int i = 0; i < per_group; i += 8) {
    f4 h[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long k0 = grp * per_group + i + 2 * u;
      const float *ph = hot + (long)(idx[k0] & hotmask) * 64 + l * 4;
      const float *pc = cold + (long)(idx[k0 + 1] & coldmask) * 64 + l * 4;
      if (MODE != 2) cold_load<0>(h[u], ph); else h[u] = f4{0, 0, 0, 0};
      if (MODE != 1) cold_load<POL>(c[u], pc); else c[u] = f4{0, 0, 0, 0};
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])::"memory");
#pragma unroll
    for (int u = 0; u < 4; u++) acc += h[u] + c[u];
  }
  if (acc[0] == 123.456f) out[grp] = acc[1] + acc[2] + acc[3];
}

template <int POL>
static void run(const char *name, const float *hot, const float *cold, const int *idx, float *out, long total, int hot_rows,
                int cold_rows) {
  const int per_group = 64;
  const long groups = total / per_group;
  const int blocks = (int)(groups * 16 / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best[3] = {1e30f, 1e30f, 1e30f};
  for (int mode = 0; mode < 3; mode++)
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL((k<POL, 0>), dim3(blocks), dim3(256), 0, 0, hot, cold, idx, per_group, hot_rows - 1, cold_rows - 1, out);
      if (mode == 1) hipLaunchKernelGGL((k<POL, 1>), dim3(blocks), dim3(256), 0, 0, hot, cold, idx, per_group, hot_rows - 1, cold_rows - 1, out);
      if (mode == 2) hipLaunchKernelGGL((k<POL, 2>), dim3(blocks), dim3(256), 0, 0, hot, cold, idx, per_group, hot_rows - 1, cold_rows - 1, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best[mode]) best[mode] = ms;
    }
  const double half = total / 2 * 256.0;
  printf("cold policy %-12s hot %7.3f MB: mixed %.3f ms | hot alone %.3f ms (%.1f TB/s) | cold alone %.3f ms (%.2f TB/s) | mixed - cold alone = %.3f ms\n",
         name, hot_rows * 256.0 / 1048576.0, best[0], best[1], half / best[1] / 1e9, best[2], half / best[2] / 1e9,
         best[0] - best[2]);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const long total = 1 << 24;
  const long cbytes = 1l << 30;
  float *hot, *cold, *out;
  int *idx;
  hipMalloc(&hot, 64l << 20);
  hipMalloc(&cold, cbytes);
  hipMalloc(&out, total);
  hipMalloc(&idx, total * 4);
  hipMemset(hot, 0, 64l << 20);
  hipMemset(cold, 0, cbytes);
  std::vector<int> h(total);
  srand(1);
  for (long i = 0; i < total; i++) h[i] = (int)(((long)rand() * 32768 + rand()) & 0x7fffffff);
  hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
  const int cold_rows = (int)(cbytes / 256);
  for (int hot_rows : {32, 64, 128, 256, 4096, 8192, 16384}) {  // 8 .. 64 KB (the 32 KB L1?), then 1, 2, 4 MB
    run<0>("default", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<1>("nt", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<2>("sc1", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<3>("sc0 sc1", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<4>("sc1 nt", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<5>("sc0 sc1 nt", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<6>("sc0", hot, cold, idx, out, total, hot_rows, cold_rows);
    run<7>("sc0 nt", hot, cold, idx, out, total, hot_rows, cold_rows);
  }
  return 0;
}
