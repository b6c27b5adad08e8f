// load_policy.cpp -- micro-benchmark behind DESIGN.md section 7: how do gfx950 cache-policy bits on the B-row gather
// behave?  16M gathers of 256-B rows (16 lanes x dwordx4) from a 256 MB table, 4 rows per wave instruction.
//   mode A: all gathers use policy P (uniform random rows)               -> throughput of the policy itself
//   mode B: half the gathers hit a 3 MB hot set with PLAIN loads, the other half go to random rows with policy P
//           -> does P keep the hot set resident (time drops) and at what cost
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 experiments/load_policy.cpp -o /tmp/lp && /tmp/lp
#include <hip/hip_runtime.h>
#ifndef VAR
#define VAR 0
#endif

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define LOADER(NAME, MODS)                                                                           \
  __device__ __forceinline__ f4 NAME(const float *p) {                                               \
    f4 r;                                                                                            \
    asm volatile("global_load_dwordx4 %0, %1, off " MODS "\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory"); \
    return r;                                                                                        \
  }
LOADER(ld_plain, "")
LOADER(ld_nt, "nt")
LOADER(ld_sc0, "sc0")
LOADER(ld_sc1, "sc1")
LOADER(ld_sc0sc1, "sc0 sc1")
LOADER(ld_ntsc1, "sc1 nt")
LOADER(ld_ntsc0, "sc0 nt")
LOADER(ld_all, "sc0 sc1 nt")

// 8 independent gathers in flight per lane: issue 8 loads without waiting, then one wait
#define KERNEL(NAME, MODS)                                                                                         \
  __global__ __launch_bounds__(256) void NAME(const float *B, const int *idx, const int *hotidx, int per_group,    \
                                              int mode, float *out) {                                              \
    const int lane = threadIdx.x & 63, l = lane & 15;                                                              \
    const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;                                                  \
    f4 acc = {0, 0, 0, 0};                                                                                         \
    for (int i = 0; i < per_group; i += 8) {                                                                       \
      f4 x[8];                                                                                                     \
      const float *p[8];                                                                                           \
      for (int u = 0; u < 8; u++) {                                                                                \
        const long k = grp * per_group + i + u;                                                                    \
        const bool hot = (mode == 1) && (u & 1);                                                                   \
        p[u] = B + (long)(hot ? hotidx[k] : idx[k]) * 64 + l * 4;                                                  \
      }                                                                                                            \
      if (mode == 1) {                                                                                             \
        asm volatile("global_load_dwordx4 %0, %8, off " MODS "\n\tglobal_load_dwordx4 %1, %9, off\n\t"             \
                     "global_load_dwordx4 %2, %10, off " MODS "\n\tglobal_load_dwordx4 %3, %11, off\n\t"           \
                     "global_load_dwordx4 %4, %12, off " MODS "\n\tglobal_load_dwordx4 %5, %13, off\n\t"           \
                     "global_load_dwordx4 %6, %14, off " MODS "\n\tglobal_load_dwordx4 %7, %15, off\n\t"           \
                     "s_waitcnt vmcnt(0)"                                                                          \
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]),  \
                       "=&v"(x[7])                                                                                 \
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])      \
                     : "memory");                                                                                  \
      } else {                                                                                                     \
        asm volatile("global_load_dwordx4 %0, %8, off " MODS "\n\tglobal_load_dwordx4 %1, %9, off " MODS "\n\t"    \
                     "global_load_dwordx4 %2, %10, off " MODS "\n\tglobal_load_dwordx4 %3, %11, off " MODS "\n\t"  \
                     "global_load_dwordx4 %4, %12, off " MODS "\n\tglobal_load_dwordx4 %5, %13, off " MODS "\n\t"  \
                     "global_load_dwordx4 %6, %14, off " MODS "\n\tglobal_load_dwordx4 %7, %15, off " MODS "\n\t"  \
                     "s_waitcnt vmcnt(0)"                                                                          \
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]),  \
                       "=&v"(x[7])                                                                                 \
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])      \
                     : "memory");                                                                                  \
      }                                                                                                            \
      for (int u = 0; u < 8; u++) acc += x[u];                                                                     \
    }                                                                                                              \
    if (acc[0] == 123.456f) out[grp] = acc[1] + acc[2] + acc[3];                                                   \
  }
KERNEL(k_plain, "")
KERNEL(k_nt, "nt")
KERNEL(k_sc0, "sc0")
KERNEL(k_sc1, "sc1")
KERNEL(k_sc0sc1, "sc0 sc1")
KERNEL(k_ntsc1, "sc1 nt")
KERNEL(k_ntsc0, "sc0 nt")
KERNEL(k_all, "sc0 sc1 nt")

// mode C: the hot/cold choice differs per 16-lane group inside one wave, so every gather becomes two exec-masked
// instructions (plain for the hot lanes, nt for the cold ones) -- what a per-nnz hint costs in a 4-rows-per-wave kernel
__device__ __forceinline__ void ld_split(f4 &r, const float *p, int hot) {
  unsigned long long sv;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "v_cmp_ne_u32 vcc, 0, %3\n\t"
      "s_and_b64 exec, %1, vcc\n\t"
      "global_load_dwordx4 %0, %2, off\n\t"
      "s_andn2_b64 exec, %1, vcc\n\t"
#if VAR == 1
      "global_load_dwordx4 %0, %2, off\n\t"
#else
      "global_load_dwordx4 %0, %2, off nt\n\t"
#endif
      "s_mov_b64 exec, %1"
      : "=&v"(r), "=&s"(sv)
      : "v"(p), "v"(hot)
      : "memory", "vcc", "scc");
}
__global__ __launch_bounds__(256) void k_split(const float *B, const int *idx, const int *hotidx, int per_group, int mode,
                                               float *out) {
  const int lane = threadIdx.x & 63, l = lane & 15;
  const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  f4 acc = {0, 0, 0, 0};
  for (int i = 0; i < per_group; i += 8) {
    f4 x[8];
    const float *p[8];
    int hot[8];
    for (int u = 0; u < 8; u++) {
      const long k = grp * per_group + i + u;
      hot[u] = (int)((k * 2654435761u >> 13) & 1);  // per-group pseudo-random, ~half hot
      p[u] = B + (long)(hot[u] ? hotidx[k] : idx[k]) * 64 + l * 4;
    }
    // the compiler does not count the asm loads below in its vmcnt bookkeeping: make it finish its own index loads first
    for (int u = 0; u < 8; u++) asm volatile("" : "+v"(p[u]), "+v"(hot[u]));
    for (int u = 0; u < 8; u++) ld_split(x[u], p[u], hot[u]);
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"memory");
    for (int u = 0; u < 8; u++) acc += x[u];
  }
  if (acc[0] == 123.456f) out[grp] = acc[1] + acc[2] + acc[3];
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const bool only_c = argc > 1;
  const long rows = 1 << 20, total = 1 << 24;  // 256 MB table, 16M gathers
  const int per_group = 64;
  float *B, *out;
  int *idx, *hotidx;
  hipMalloc(&B, rows * 64 * 4);
  hipMalloc(&out, (total / per_group) * 4);
  hipMalloc(&idx, total * 4);
  hipMalloc(&hotidx, total * 4);
  hipMemset(B, 0, rows * 64 * 4);
  std::vector<int> h(total), hh(total);
  srand(1);
  for (long i = 0; i < total; i++) {
    h[i] = (int)(((long)rand() * 32768 + rand()) % rows);
    hh[i] = (int)((rand() % 12288) * 85 % rows);  // 12288 hot rows = 3 MB, spread over the table
  }
  hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice);
  hipMemcpy(hotidx, hh.data(), total * 4, hipMemcpyHostToDevice);
  const long groups = total / per_group;
  const int blocks = (int)(groups * 16 / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct V { const char *name; void (*k)(const float *, const int *, const int *, int, int, float *); };
  V vs[] = {{"plain", k_plain}, {"nt", k_nt}, {"sc0", k_sc0}, {"sc1", k_sc1}, {"sc0 sc1", k_sc0sc1},
            {"sc1 nt", k_ntsc1}, {"sc0 nt", k_ntsc0}, {"sc0 sc1 nt", k_all}};
  for (int mode = 0; mode < 2 && !only_c; mode++)
    for (auto &v : vs) {
      for (int it = 0; it < 3; it++) hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, B, idx, hotidx, per_group, mode, out);
      hipEventRecord(e0);
      for (int it = 0; it < 10; it++) hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, 0, B, idx, hotidx, per_group, mode, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("mode %s  policy %-11s  %.3f ms  %.2f TB/s of gathers\n", mode ? "B(hot/cold)" : "A(all)     ", v.name, ms / 10,
             total * 256.0 / (ms / 10 * 1e-3) / 1e12);
    }
  {
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, B, idx, hotidx, per_group, 2, out);
    hipEventRecord(e0);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, B, idx, hotidx, per_group, 2, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode C(per-group hot/cold, exec-masked pair)  %.3f ms  %.2f TB/s of gathers\n", ms / 10,
           total * 256.0 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
