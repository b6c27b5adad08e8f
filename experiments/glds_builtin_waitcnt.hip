// experiments/glds_builtin_waitcnt.hip - does hipcc's waitcnt pass order a ds_read behind the LDS-DMA that fills the slot it reads?
// (round 5, no GPU needed: an ISA question).  A per-wave ring of D one-KiB slots, each filled by __builtin_amdgcn_global_load_lds
// (16 bytes per lane, gfx950) and consumed by a ds_read_b128 one trip of the loop later - the shape an LDS-landing gather
// window in the SpMM row stream would have.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S experiments/glds_builtin_waitcnt.hip -o - | less
// Answer with ROCm 7.2.0 (profiles/r05_glds_builtin_isa.txt): NO.  The loop body is [8 x ds_read_b128] [lgkmcnt(0)] [adds]
// [vmcnt(0) - for the eight idx loads, which happen to be global loads here] [8 x global_load_lds_dwordx4] [branch]: nothing
// makes the ds_reads at the top of the next trip wait for the DMA issued at the bottom of this one; with the indices coming from
// LDS (as in the product kernels) the incidental vmcnt(0) goes away as well.  The builtin is therefore only usable with
// hand-placed s_waitcnt vmcnt(n) (inline asm, as experiments/lds_dma_gather.cpp does) - which the CPU emulation cannot model
// and hipcc's own vmcnt bookkeeping does not see.  Consequence recorded in DESIGN.md section 7.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(256) void k(const float *__restrict__ table, const int *__restrict__ idx, int n, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) char ring[4][D][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l = lane & 15;
  const float *tl = table + l * 4;
  f4 acc = {0, 0, 0, 0};
  // prologue
#pragma unroll
  for (int d = 0; d < D; d++) {
    int r = idx[d * 4 + g];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tl + (long)r * 64), (__attribute__((address_space(3))) void *)&ring[wave][d][0], 16, 0, 0);
  }
  for (int j = D; j < n; j += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      f4 x = *reinterpret_cast<const f4 *>(&ring[wave][d][lane * 16]);
      acc += x;
      int r = idx[(j + d) * 4 + g];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tl + (long)r * 64), (__attribute__((address_space(3))) void *)&ring[wave][d][0], 16, 0, 0);
    }
  }
  reinterpret_cast<f4 *>(out)[blockIdx.x * 256 + threadIdx.x] = acc;
}
template __global__ void k<8>(const float *, const int *, int, float *);
