// lds_dma_gather.cpp -- micro-benchmarks behind two round-4 design questions.
//
// Part 1  "does an LDS-landing gather (global_load_lds_dwordx4, gfx950 LDS-DMA) beat the VGPR gather window?"
//         Random 256-/512-byte row gathers from a hot (2 MB: fits one XCD's L2) or cold (1 GB) table, summed per lane.
//           vgpr<U>  rolling window of U wave-level gather instructions in registers (what spmm_fused does today)
//           glds<D>  per-wave LDS ring of D one-KiB slots filled by global_load_lds_dwordx4, consumed by ds_read_b128;
//                    the index tile arrives by LDS-DMA too, so every VMEM instruction of the wave is a counted DMA
//         at full occupancy (the fabric-bound regime of the headline graph) and at ONE workgroup per CU (the
//         latency-bound regime of mid-size graphs, where memory-level parallelism per wave is the only knob).
//
// Part 2  "how fast can ONE workgroup feed a sequential chain over a 50 k-nnz hub row while the rest of the chip
//         saturates the fabric?"  A consumer workgroup per XCD walks a list of L random rows in rounds: W gather waves
//         (8 gather instructions each in flight) -> LDS tile -> wave 0 runs a dependent fma chain over the tile (one
//         link per row, 64 features = 64 lanes).  Around it: background workgroups gathering cold rows flat out, and
//         optionally a PREFETCH workgroup on the same XCD that touches the consumer's rows DIST rounds ahead with
//         loads whose result is never used (no registers, no LDS: the lines just land in the XCD's L2).
//         Reported: ns per chain link.  Little's law says (bytes in flight per row) / latency under load is the rate;
//         the prefetcher is the only way to put megabytes in flight for one row.
//
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 experiments/lds_dma_gather.cpp -o /tmp/ldg && /tmp/ldg
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

// one LDS-DMA instruction: every lane's 16 bytes at gsrc land at lds_dst + lane * 16 (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(uintptr_t)p; }

constexpr int kChunk = 256;  // indices per tile

// ---------------------------------------------------------------------------------------------------------------
// Part 1a: VGPR window.  G lanes per row (16: 256-byte rows, 32: 512-byte rows); U instructions in flight.
template <int G, int U>
__global__ __launch_bounds__(256) void gather_vgpr(const float *__restrict__ table, const int *__restrict__ idx, int per_wave,
                                                   float *__restrict__ out) {
  constexpr int RPI = 64 / G, RF = G * 4, JN = kChunk / RPI;
  __shared__ int tile[4][kChunk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * 4 + wave;
  const int g = lane / G, l = lane % G;
  const int *my = idx + w * per_wave;
  const float *tl = table + l * 4;
  f4 acc = {0, 0, 0, 0};
  for (int c0 = 0; c0 < per_wave; c0 += kChunk) {
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < kChunk; t += 64) tile[wave][t] = __builtin_nontemporal_load(my + c0 + t);
    __builtin_amdgcn_wave_barrier();
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      x[u] = *reinterpret_cast<const f4 *>(tl + (long)tile[wave][u * RPI + g] * RF);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int j = 0; j < JN; j += U) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        acc += x[u];
        int jn = j + u + U;
        jn = jn < JN ? jn : JN - 1;  // tail: repeats the last row (loaded, never added: the loop ends first)
        x[u] = *reinterpret_cast<const f4 *>(tl + (long)tile[wave][jn * RPI + g] * RF);
      }
    }
  }
  reinterpret_cast<f4 *>(out)[w * 64 + lane] = acc;
}

// Part 1b: LDS-DMA ring of D slots per wave.
template <int G, int D>
__global__ __launch_bounds__(256) void gather_glds(const float *__restrict__ table, const int *__restrict__ idx, int per_wave,
                                                   float *__restrict__ out) {
  constexpr int RPI = 64 / G, RF = G * 4, JN = kChunk / RPI;
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * 4 + wave;
  const int g = lane / G, l = lane % G;
  char *ring = smem + wave * (D * 1024 + 2 * 1024);
  int *itile = reinterpret_cast<int *>(ring + D * 1024);
  const unsigned ring_o = __builtin_amdgcn_readfirstlane(lds_off(ring));
  const unsigned it_o = __builtin_amdgcn_readfirstlane(lds_off(itile));
  const int *my = idx + w * per_wave;
  const float *tl = table + l * 4;
  f4 acc = {0, 0, 0, 0};
  const int nchunks = per_wave / kChunk;
  for (int sl = 0; sl < D; sl++) *reinterpret_cast<f4 *>(ring + sl * 1024 + lane * 16) = f4{0, 0, 0, 0};
  glds16(my + lane * 4, it_o);
  wait_vm<0>();
  unsigned n = 0;  // gathers issued so far
  f4 pend = {0, 0, 0, 0};
  for (int c = 0; c < nchunks; c++) {
    if (c + 1 < nchunks) glds16(my + (c + 1) * kChunk + lane * 4, it_o + ((c + 1) & 1) * 1024);
    const int *it = itile + (c & 1) * kChunk;
    int r = it[g];
    for (int j = 0; j < JN; j++) {
      const int rn = it[(j + 1 < JN ? j + 1 : j) * RPI + g];  // next instruction's row: its LDS latency runs under this step
      glds16(tl + (long)r * RF, ring_o + (n & (D - 1)) * 1024);
      n++;
      acc += pend;
      pend = f4{0, 0, 0, 0};
      // gather n - D (D - 1 younger ones, + at most the index tile: stricter, still safe) has landed; the first D - 1 steps
      // read the zeros the ring was initialised with (branch-free steady state)
      wait_vm<D - 1>();
      pend = *reinterpret_cast<const f4 *>(ring + (n & (D - 1)) * 1024 + lane * 16);  // added one step later
      r = rn;
    }
  }
  acc += pend;
  wait_vm<0>();
  for (unsigned k = 1; k < D && k <= n; k++) acc += *reinterpret_cast<const f4 *>(ring + ((n - D + k) & (D - 1)) * 1024 + lane * 16);
  reinterpret_cast<f4 *>(out)[w * 64 + lane] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// Part 2: hub-row consumer + background + prefetcher.
struct SimCtl {
  int progress[8];   // rounds finished by consumer x
  int done;          // consumers finished
  int pad[7];
  long long cyc[8];  // consumer x: wall-clock ticks start -> end
  int xcc[16];       // XCC id seen by consumer x / prefetcher x
};

__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15); }

// junk load: one dword per lane, result never used (prefetch into the XCD's L2)
__device__ __forceinline__ void touch(const void *p) {
  unsigned junk;
  asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(p) : "memory");
}

// W gather waves + 1 chain wave (wave 0).  Rows are 256 bytes (64 floats = 64 chain lanes).  MODE 0: chain + gathers,
// 1: gathers only (no chain), 2: chain only (rows come from a fixed L2-resident row: no miss ever)
template <int W>
__device__ void consumer(const float *__restrict__ table, const int *__restrict__ list, int L, SimCtl *ctl, int x, int mode,
                         float *out, float *lds) {
  constexpr int RPR = W * 32;  // rows per round: 8 instructions x 4 rows per gather wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 16, l = lane % 16;
  float acc = 0.f;
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) ctl->xcc[x] = xcc_id();
  const int rounds = L / RPR;
  f4 xr[8];
  const float *tl = table + l * 4;
  const int gw = wave - 1;  // gather wave index
  auto issue = [&](int r) {
    if (r >= rounds) return;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      int row = list[r * RPR + gw * 32 + q * 4 + g];
      if (mode == 2) row &= 63;
      xr[q] = *reinterpret_cast<const f4 *>(tl + (long)row * 64);
    }
  };
  if (wave > 0) issue(0);
  for (int r = 0; r < rounds; r++) {
    float *buf = lds + (r & 1) * RPR * 64;
    if (wave > 0) {
#pragma unroll
      for (int q = 0; q < 8; q++) *reinterpret_cast<f4 *>(buf + (gw * 32 + q * 4 + g) * 64 + l * 4) = xr[q];
      issue(r + 1);
    }
    __syncthreads();
    if (wave == 0) {
      if (lane == 0) __hip_atomic_store(&ctl->progress[x], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mode != 1) {
        __builtin_amdgcn_s_setprio(3);
        const float *xb = buf + lane;
#pragma unroll 2
        for (int k = 0; k < RPR; k += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; u++) v[u] = xb[(k + u) * 64];
#pragma unroll
          for (int u = 0; u < 16; u++) acc = __builtin_fmaf(1.0000001f, v[u], acc);
        }
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  if (wave == 0) out[x * 64 + lane] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    ctl->cyc[x] = wall_clock64() - t0;
    __hip_atomic_fetch_add(&ctl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// prefetcher for consumer x: 4 waves, each touches a quarter of a round's rows (2 lines of 128 B per row)
template <int W>
__device__ void prefetcher(const float *__restrict__ table, const int *__restrict__ list, int L, SimCtl *ctl, int x, int dist) {
  constexpr int RPR = W * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) ctl->xcc[8 + x] = xcc_id();
  const int rounds = L / RPR;
  int p = 1;
  for (int guard = 0; guard < (1 << 21) && p < rounds; guard++) {
    const int prog = __hip_atomic_load(&ctl->progress[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prog >= rounds) break;
    if (p < prog + 1) p = prog + 1;
    if (p > prog + dist) {
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    // this wave's share of round p: rows [wave * RPR/4, (wave + 1) * RPR/4), 32 rows (64 lines) per instruction
    for (int i = wave * (RPR / 4); i < (wave + 1) * (RPR / 4); i += 32) {
      const int row = list[p * RPR + i + (lane >> 1)];
      touch(table + (long)row * 64 + (lane & 1) * 32);
    }
    p++;
  }
  wait_vm<0>();
}

__device__ void background(const float *__restrict__ table, const int *__restrict__ idx, int n_idx, SimCtl *ctl, float *out) {
  const int lane = threadIdx.x & 63;
  const int g = lane / 16, l = lane % 16;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  f4 acc = {0, 0, 0, 0};
  const float *tl = table + l * 4;
  long pos = (wid * 7919) % (n_idx - 4096);
  // bounded (~1 s): a consumer that never finishes must not hang the GPU
  for (int guard = 0; guard < 20000 && __hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8; guard++) {
    for (int it = 0; it < 16; it++) {
      f4 x[6];
#pragma unroll
      for (int u = 0; u < 6; u++) x[u] = *reinterpret_cast<const f4 *>(tl + (long)idx[pos + u * 4 + g] * 64);
#pragma unroll
      for (int u = 0; u < 6; u++) acc += x[u];
      pos += 24;
      if (pos >= n_idx - 4096) pos = 0;
    }
  }
  if (acc[0] == 123.456f) out[wid] = acc[1];
}

template <int W>
__global__ __launch_bounds__((W + 1) * 64) void hubsim(const float *__restrict__ table, const int *__restrict__ lists, int L,
                                                       const int *__restrict__ bgidx, int n_bg, SimCtl *ctl, int mode, int dist,
                                                       int bg_on, float *out) {
  extern __shared__ float lds[];
  const int b = blockIdx.x;
  if (b < 8) consumer<W>(table, lists + (long)b * L, L, ctl, b, mode, out, lds);
  else if (b < 16) {
    if (dist > 0 && threadIdx.x < 256) prefetcher<W>(table, lists + (long)(b - 8) * L, L, ctl, b - 8, dist);
  } else if (bg_on && threadIdx.x < 256) background(table, bgidx, n_bg, ctl, out + 1024);
}

// ---------------------------------------------------------------------------------------------------------------
static float time_ms(hipEvent_t e0, hipEvent_t e1) {
  float ms;
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

template <int G, int U>
static double run_vgpr(const float *table, const int *idx, long total, int blocks, float *out) {
  const int per_wave = (int)(total / ((long)blocks * 4)) / kChunk * kChunk;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_vgpr<G, U>), dim3(blocks), dim3(256), 0, 0, table, idx, per_wave, out);
    CK(hipEventRecord(e1));
    const float ms = time_ms(e0, e1);
    if (rep && ms < best) best = ms;
  }
  return (double)per_wave * blocks * 4 * G * 16 / (best * 1e-3) / 1e12;  // TB/s of gathered rows
}
template <int G, int D>
static double run_glds(const float *table, const int *idx, long total, int blocks, float *out) {
  const int per_wave = (int)(total / ((long)blocks * 4)) / kChunk * kChunk;
  const size_t lds = 4 * (D * 1024 + 2048);
  auto kern = gather_glds<G, D>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, table, idx, per_wave, out);
    CK(hipEventRecord(e1));
    const float ms = time_ms(e0, e1);
    if (rep && ms < best) best = ms;
  }
  return (double)per_wave * blocks * 4 * G * 16 / (best * 1e-3) / 1e12;
}

// Part 1b helpers (round 5): the same kernels with the LDS footprint the PRODUCT row stream would have - `pad` bytes of dynamic
// LDS on top of the kernel's own (spmm_fused holds 20 544 B per workgroup; hipcc lets a launch reserve dynamic LDS the kernel
// never declares) - so the occupancy per CU is the product's, not the microbenchmark's.  Returns microseconds per launch.
template <int G, int U>
static double us_vgpr(const float *table, const int *idx, int per_wave, int blocks, float *out, size_t pad) {
  auto kern = gather_vgpr<G, U>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), pad, 0, table, idx, per_wave, out);
    CK(hipEventRecord(e1));
    const float ms = time_ms(e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best * 1e3;
}
template <int G, int D>
static double us_glds(const float *table, const int *idx, int per_wave, int blocks, float *out, size_t pad) {
  const size_t lds = 4 * (D * 1024 + 2048) + pad;
  auto kern = gather_glds<G, D>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, table, idx, per_wave, out);
    CK(hipEventRecord(e1));
    const float ms = time_ms(e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best * 1e3;
}

static double checksum(const float *d_out, long n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d_out, n * 4, hipMemcpyDeviceToHost));
  double s = 0;
  for (long i = 0; i < n; i++) s += h[i];
  return s;
}

int main(int argc, char **argv) {
  const bool only2 = argc > 1 && !strcmp(argv[1], "part2");
  const bool only1 = argc > 1 && !strcmp(argv[1], "part1");
  const long cold_rows = 1L << 22;  // x 256 B = 1 GB
  float *table;
  CK(hipMalloc(&table, cold_rows * 256));
  {
    std::vector<float> h((size_t)cold_rows * 64);
    std::mt19937 rng(1);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(rng() & 1023) * (1.0f / 1024.0f);
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  const long total = 1L << 24;  // gathered rows per run (4 GB at 256 B)
  std::vector<int> hidx(total);
  std::mt19937 rng(7);
  for (long i = 0; i < total; i++) hidx[i] = (int)(rng() & (cold_rows - 1));
  int *idx_cold, *idx_hot, *idx_cold512, *idx_hot512;
  CK(hipMalloc(&idx_cold, total * 4));
  CK(hipMalloc(&idx_hot, total * 4));
  CK(hipMalloc(&idx_cold512, total * 4));
  CK(hipMalloc(&idx_hot512, total * 4));
  CK(hipMemcpy(idx_cold, hidx.data(), total * 4, hipMemcpyHostToDevice));
  {
    std::vector<int> t(total);
    for (long i = 0; i < total; i++) t[i] = hidx[i] & 8191;  // 8192 x 256 B = 2 MB
    CK(hipMemcpy(idx_hot, t.data(), total * 4, hipMemcpyHostToDevice));
    for (long i = 0; i < total; i++) t[i] = hidx[i] & (cold_rows / 2 - 1);  // 512-byte rows
    CK(hipMemcpy(idx_cold512, t.data(), total * 4, hipMemcpyHostToDevice));
    for (long i = 0; i < total; i++) t[i] = hidx[i] & 4095;
    CK(hipMemcpy(idx_hot512, t.data(), total * 4, hipMemcpyHostToDevice));
  }
  float *out;
  CK(hipMalloc(&out, 64L << 20));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);

  if (!only2) {
    printf("\nPart 1: gathered TB/s (rows x row bytes / time); in-flight = rows per wave\n");
    printf("%-34s %10s %10s %10s %10s\n", "variant", "cold full", "hot full", "cold 1wg/CU", "hot 1wg/CU");
    // correctness: vgpr and glds over the same indices must agree
    {
      run_vgpr<16, 4>(table, idx_cold, 1 << 20, 64, out);
      const double a = checksum(out, 64L * 4 * 64 * 4);
      run_glds<16, 8>(table, idx_cold, 1 << 20, 64, out);
      const double b = checksum(out, 64L * 4 * 64 * 4);
      printf("checksum vgpr %.6f glds %.6f %s\n", a, b, a == b ? "EQUAL" : "DIFFERENT");
    }
#define ROW(name, call_c, call_h)                                                                                    \
  {                                                                                                                  \
    const int full = cus * 16, one = cus;                                                                            \
    double a, b, c, d;                                                                                               \
    { const int blocks = full; a = call_c; b = call_h; }                                                             \
    { const int blocks = one; c = call_c; d = call_h; }                                                              \
    printf("%-34s %10.2f %10.2f %10.2f %10.2f\n", name, a, b, c, d);                                                 \
    fflush(stdout);                                                                                                  \
  }
    ROW("256B vgpr U=2 (8 rows)", (run_vgpr<16, 2>(table, idx_cold, total, blocks, out)), (run_vgpr<16, 2>(table, idx_hot, total, blocks, out)));
    ROW("256B vgpr U=4 (16 rows)", (run_vgpr<16, 4>(table, idx_cold, total, blocks, out)), (run_vgpr<16, 4>(table, idx_hot, total, blocks, out)));
    ROW("256B vgpr U=8 (32 rows)", (run_vgpr<16, 8>(table, idx_cold, total, blocks, out)), (run_vgpr<16, 8>(table, idx_hot, total, blocks, out)));
    ROW("256B vgpr U=16 (64 rows)", (run_vgpr<16, 16>(table, idx_cold, total, blocks, out)), (run_vgpr<16, 16>(table, idx_hot, total, blocks, out)));
    ROW("256B glds D=2 (8 rows)", (run_glds<16, 2>(table, idx_cold, total, blocks, out)), (run_glds<16, 2>(table, idx_hot, total, blocks, out)));
    ROW("256B glds D=4 (16 rows)", (run_glds<16, 4>(table, idx_cold, total, blocks, out)), (run_glds<16, 4>(table, idx_hot, total, blocks, out)));
    ROW("256B glds D=8 (32 rows)", (run_glds<16, 8>(table, idx_cold, total, blocks, out)), (run_glds<16, 8>(table, idx_hot, total, blocks, out)));
    ROW("256B glds D=16 (64 rows)", (run_glds<16, 16>(table, idx_cold, total, blocks, out)), (run_glds<16, 16>(table, idx_hot, total, blocks, out)));
    ROW("256B glds D=32 (128 rows)", (run_glds<16, 32>(table, idx_cold, total, blocks, out)), (run_glds<16, 32>(table, idx_hot, total, blocks, out)));
    ROW("512B vgpr U=4 (8 rows)", (run_vgpr<32, 4>(table, idx_cold512, total, blocks, out)), (run_vgpr<32, 4>(table, idx_hot512, total, blocks, out)));
    ROW("512B vgpr U=8 (16 rows)", (run_vgpr<32, 8>(table, idx_cold512, total, blocks, out)), (run_vgpr<32, 8>(table, idx_hot512, total, blocks, out)));
    ROW("512B glds D=8 (16 rows)", (run_glds<32, 8>(table, idx_cold512, total, blocks, out)), (run_glds<32, 8>(table, idx_hot512, total, blocks, out)));
    ROW("512B glds D=16 (32 rows)", (run_glds<32, 16>(table, idx_cold512, total, blocks, out)), (run_glds<32, 16>(table, idx_hot512, total, blocks, out)));
    ROW("512B glds D=32 (64 rows)", (run_glds<32, 32>(table, idx_cold512, total, blocks, out)), (run_glds<32, 32>(table, idx_hot512, total, blocks, out)));
  }

  if (!only2) {
    // Part 1b (round 5): would an LDS-landing window pay IN THE ROW STREAM of a mid-size launch?  The arxiv-shaped
    // configuration (C2) hits L2 on 54 % of its gathers, runs ~660 workgroups of 4 waves for ~45 us, and spmm_fused sits at 5
    // workgroups per CU (20.5 KB of LDS, 96 VGPRs).  An LDS ring costs occupancy: 4 waves x D KB on top of those 20.5 KB.
    // So: the two kernels of part 1 on a list that is hot (2 MB table) with probability 0.54, else cold, in a C2-sized grid (664
    // workgroups x 4 waves x 256 or 512 rows) and in a grid that fills the chip for a while (16 per CU), each with the LDS the
    // product would carry.  What to read: glds must beat 'vgpr U=8, 20.5 KB pad' (the product's window) by >= 10 % in the first two
    // columns to be worth a kernel variant; if it does not, the item is closed with these numbers.
    std::vector<int> t(total);
    std::mt19937 r2(11);
    for (long i = 0; i < total; i++) t[i] = ((r2() % 100) < 54) ? (hidx[i] & 8191) : hidx[i];
    int *idx_mix;
    CK(hipMalloc(&idx_mix, total * 4));
    CK(hipMemcpy(idx_mix, t.data(), total * 4, hipMemcpyHostToDevice));
    const size_t prod = 20544;
    printf("\nPart 1b: 54 %% hot / 46 %% cold list, product-sized LDS per workgroup; microseconds per launch (lower is better)\n");
    printf("%-44s %12s %12s %12s\n", "variant (LDS per workgroup)", "664wg x 256", "664wg x 512", "16/CU x 512");
#define ROWB(name, call)                                                                        \
  {                                                                                             \
    double a, b, c;                                                                             \
    { const int blocks = 664, pw = 256; a = call; }                                             \
    { const int blocks = 664, pw = 512; b = call; }                                             \
    { const int blocks = cus * 16, pw = 512; c = call; }                                        \
    printf("%-44s %12.2f %12.2f %12.2f\n", name, a, b, c);                                      \
    fflush(stdout);                                                                             \
  }
    ROWB("vgpr U=4, 4 KB + 20.5 KB pad (5-6 wg/CU)", (us_vgpr<16, 4>(table, idx_mix, pw, blocks, out, prod)));
    ROWB("vgpr U=8, 4 KB + 20.5 KB pad (product)", (us_vgpr<16, 8>(table, idx_mix, pw, blocks, out, prod)));
    ROWB("vgpr U=16, 4 KB + 20.5 KB pad", (us_vgpr<16, 16>(table, idx_mix, pw, blocks, out, prod)));
    ROWB("glds D=4, 24 KB + 20.5 KB (3 wg/CU)", (us_glds<16, 4>(table, idx_mix, pw, blocks, out, prod)));
    ROWB("glds D=8, 40 KB + 20.5 KB (2 wg/CU)", (us_glds<16, 8>(table, idx_mix, pw, blocks, out, prod)));
    ROWB("glds D=16, 72 KB + 20.5 KB (1 wg/CU)", (us_glds<16, 16>(table, idx_mix, pw, blocks, out, prod)));
    ROWB("glds D=4, 24 KB, no pad (6 wg/CU)", (us_glds<16, 4>(table, idx_mix, pw, blocks, out, 0)));
    ROWB("glds D=8, 40 KB, no pad (4 wg/CU)", (us_glds<16, 8>(table, idx_mix, pw, blocks, out, 0)));
    CK(hipFree(idx_mix));
  }

  if (!only1) {
    printf("\nPart 2: one consumer workgroup per XCD, L = 51200 rows of 256 B each; ns per chain link (mean / max over the 8)\n");
    const int L = 51200;
    int *lists;
    CK(hipMalloc(&lists, 8L * L * 4));
    CK(hipMemcpy(lists, hidx.data() + 12345, 8L * L * 4, hipMemcpyHostToDevice));
    SimCtl *ctl;
    CK(hipMalloc(&ctl, sizeof(SimCtl)));
    double clk_ghz = 0.1;  // wall_clock64 ticks at 100 MHz
    auto sim = [&](auto kern, int W, int mode, int dist, int bg_on, const char *name) {
      const size_t lds = (size_t)2 * W * 32 * 64 * 4;
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      double best_mean = 1e30, best_max = 1e30;
      int xcc[16] = {};
      float best_ms = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(ctl, 0, sizeof(SimCtl)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(bg_on ? 16 + cus * 4 : 16), dim3((W + 1) * 64), lds, 0, table, lists, L, idx_cold,
                           (int)total, ctl, mode, dist, bg_on, out);
        CK(hipEventRecord(e1));
        const float ms = time_ms(e0, e1);
        SimCtl h;
        CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
        double mean = 0, mx = 0;
        for (int x = 0; x < 8; x++) {
          const double ns = (double)h.cyc[x] / clk_ghz / L;
          mean += ns / 8;
          if (ns > mx) mx = ns;
        }
        if (mean < best_mean) {
          best_mean = mean;
          best_max = mx;
          best_ms = ms;
          memcpy(xcc, h.xcc, sizeof(xcc));
        }
      }
      printf("%-52s W=%d  %7.2f / %7.2f ns per link   kernel %.3f ms   xcc cons", name, W, best_mean, best_max, best_ms);
      for (int x = 0; x < 8; x++) printf(" %d", xcc[x]);
      if (dist > 0) {
        printf("  pf");
        for (int x = 0; x < 8; x++) printf(" %d", xcc[8 + x]);
      }
      printf("\n");
      fflush(stdout);
    };
#define SIMW(W)                                                                              \
  sim(hubsim<W>, W, 2, 0, 0, "chain only (rows always hit), idle chip");                     \
  sim(hubsim<W>, W, 1, 0, 0, "gathers only, idle chip");                                     \
  sim(hubsim<W>, W, 0, 0, 0, "chain + gathers, idle chip");                                  \
  sim(hubsim<W>, W, 1, 0, 1, "gathers only, saturated fabric");                              \
  sim(hubsim<W>, W, 0, 0, 1, "chain + gathers, saturated fabric");                           \
  sim(hubsim<W>, W, 0, 4, 1, "chain + gathers, saturated, prefetch 4 rounds ahead");         \
  sim(hubsim<W>, W, 0, 16, 1, "chain + gathers, saturated, prefetch 16 rounds ahead");       \
  sim(hubsim<W>, W, 0, 64, 1, "chain + gathers, saturated, prefetch 64 rounds ahead");       \
  sim(hubsim<W>, W, 0, 16, 0, "chain + gathers, idle chip, prefetch 16 rounds ahead");
    SIMW(4)
    SIMW(8)
  }
  return 0;
}
