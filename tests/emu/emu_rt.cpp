// tests/emu/emu_rt.cpp -- TEST INFRASTRUCTURE ONLY: the scheduler behind tests/emu/hip/hip_runtime.h.
//
// One workgroup at a time; every work-item is a fiber on its own stack (hand-rolled x86-64 context switch: glibc's swapcontext
// makes a system call per switch).  A fiber runs until it reaches a wave collective, a wave barrier or a workgroup barrier, or
// returns from the kernel; the scheduler then runs the next one.  A rendezvous is released when every LIVE lane of the wave
// (every live work-item of the workgroup) has arrived - work-items that returned from the kernel no longer count, which is what
// the hardware's exec mask / wave count do.  When nothing can run and something still waits, that is a bug in the kernel's
// barrier protocol: the emulator prints who waits where and aborts.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <utility>
#include <vector>

#include "hip/hip_runtime.h"

// AddressSanitizer build (make ASAN=1): the fibers switch stacks behind the sanitizer's back, so every switch is announced
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/common_interface_defs.h>
#define EMU_ASAN 1
#endif
#endif
#ifndef EMU_ASAN
#define EMU_ASAN 0
#endif

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

// dynamic LDS of the column-panel kernels (DGS_DYN_SHARED in dgs_common.h)
namespace dgs {
alignas(16) char panel_dyn[160 * 1024];
alignas(16) char sd_dyn[160 * 1024];
}  // namespace dgs

namespace emu {

Item *cur = nullptr;

namespace {
constexpr size_t kStack = 512 * 1024;
enum State { RUN, WAIT_WAVE, WAIT_WSYNC, WAIT_BLOCK, DONE };
struct Fiber {
  Item item;
  void *sp = nullptr;
  char *stack = nullptr;
  void *fake = nullptr;  // ASAN: this fiber's fake-stack handle while it is switched out
  State st = DONE;
  int wave = 0, lane = 0;
};
void *sched_fake = nullptr;
const void *sched_bottom = nullptr;
size_t sched_size = 0;
struct Wave {
  unsigned char buf[64][16], snap[64][16];
  unsigned long long live = 0, arrived = 0, snap_live = 0;
  unsigned bytes = 0;
};
std::vector<Fiber> fibers;
std::vector<Wave> waves;
void *sched_sp = nullptr;
Fiber *me = nullptr;
void (*k_fn)(void *) = nullptr;
void *k_ctx = nullptr;
int block_live = 0, block_arrived = 0;

// The order in which the runnable work-items of a workgroup get their turn.  Between two rendezvous points a fiber runs
// undisturbed, so a missing barrier BETWEEN waves only shows when the reader happens to run before the writer (RAW) or the
// over-writer before the reader (WAR): one fixed order hides one of the two.  DGS_EMU_ORDER = fwd (default: ascending
// work-item id) | rev | rand:<seed> (a fresh permutation of the waves and of the lanes in each wave for every sweep).
enum Order { FWD, REV, RAND };
Order order_mode = FWD;
unsigned long long order_state = 1;
bool order_read = false;
std::vector<unsigned> order_buf;
unsigned order_next(unsigned n) {  // xorshift64*: deterministic for a seed
  order_state ^= order_state >> 12;
  order_state ^= order_state << 25;
  order_state ^= order_state >> 27;
  return (unsigned)(((order_state * 2685821657736338717ull) >> 33) % n);
}
void order_init() {
  if (order_read) return;
  order_read = true;
  const char *e = getenv("DGS_EMU_ORDER");
  if (!e || !*e || !strcmp(e, "fwd")) return;
  if (!strcmp(e, "rev")) order_mode = REV;
  else if (!strncmp(e, "rand:", 5)) {
    order_mode = RAND;
    order_state = strtoull(e + 5, nullptr, 10) * 0x9e3779b97f4a7c15ull + 0x1234567ull;
    if (!order_state) order_state = 1;
  } else {
    fprintf(stderr, "emu: DGS_EMU_ORDER=%s (fwd | rev | rand:<seed>)\n", e);
    abort();
  }
}
const std::vector<unsigned> &sweep_order(unsigned nthr) {
  if (order_buf.size() != nthr || order_mode == RAND) {
    order_buf.resize(nthr);
    for (unsigned t = 0; t < nthr; t++) order_buf[t] = order_mode == REV ? nthr - 1 - t : t;
  }
  if (order_mode == RAND) {
    const unsigned nw = (nthr + 63) / 64;
    // waves first (Fisher-Yates over whole waves), then the lanes inside each wave
    for (unsigned w = nw; w > 1; w--) {
      const unsigned o = order_next(w);
      if (o != w - 1)
        for (unsigned l = 0; l < 64; l++) {
          const unsigned a = (w - 1) * 64 + l, b = o * 64 + l;
          if (a < nthr && b < nthr) std::swap(order_buf[a], order_buf[b]);
        }
    }
    for (unsigned w = 0; w < nw; w++) {
      const unsigned n = min(64u, nthr - w * 64);
      for (unsigned l = n; l > 1; l--) std::swap(order_buf[w * 64 + l - 1], order_buf[w * 64 + order_next(l)]);
    }
  }
  return order_buf;
}

void yield() {
  Fiber *f = me;
#if EMU_ASAN
  __sanitizer_start_switch_fiber(&f->fake, sched_bottom, sched_size);
#endif
  emu_switch(&f->sp, sched_sp);
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(f->fake, nullptr, nullptr);
#endif
}

void release_wave(Wave &w, int wi, State from) {
  for (int l = 0; l < 64; l++)
    if ((size_t)(wi * 64 + l) < fibers.size() && fibers[wi * 64 + l].st == from) fibers[wi * 64 + l].st = RUN;
  w.arrived = 0;
}
void try_release_wave(int wi) {
  Wave &w = waves[wi];
  if (!w.arrived || (w.arrived & w.live) != w.live) return;
  // every live lane waits: they must all wait for the same kind of rendezvous
  State kind = DONE;
  for (int l = 0; l < 64; l++)
    if ((w.live >> l) & 1ull) {
      const State s = fibers[wi * 64 + l].st;
      if (kind == DONE) kind = s;
      else if (kind != s) return;  // mixed: left to the deadlock report
    }
  if (kind == WAIT_WAVE) {
    memcpy(w.snap, w.buf, sizeof(w.snap));
    w.snap_live = w.live;
    release_wave(w, wi, WAIT_WAVE);
  } else if (kind == WAIT_WSYNC) {
    release_wave(w, wi, WAIT_WSYNC);
  }
}
void try_release_block() {
  if (block_live > 0 && block_arrived == block_live) {
    for (auto &f : fibers)
      if (f.st == WAIT_BLOCK) f.st = RUN;
    block_arrived = 0;
  }
}

extern "C" void emu_fiber_entry() {
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &sched_bottom, &sched_size);
#endif
  k_fn(k_ctx);
  Fiber *f = me;
  f->st = DONE;
  Wave &w = waves[f->wave];
  w.live &= ~(1ull << f->lane);
  block_live--;
  try_release_wave(f->wave);
  try_release_block();
#if EMU_ASAN
  __sanitizer_start_switch_fiber(nullptr, sched_bottom, sched_size);  // this fiber ends here
#endif
  emu_switch(&f->sp, sched_sp);
  abort();  // never resumed
}

void start_fiber(Fiber &f) {
  if (!f.stack) {
    f.stack = static_cast<char *>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (f.stack == MAP_FAILED) {
      perror("emu: mmap");
      abort();
    }
  }
  // stack image emu_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into the entry with rsp = 16 n + 8
  uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
  void **sp = reinterpret_cast<void **>(top - 8);  // the slot `ret` would have left behind: keeps the ABI's alignment at entry
  *--sp = reinterpret_cast<void *>(&emu_fiber_entry);
  for (int i = 0; i < 6; i++) *--sp = nullptr;
  f.sp = sp;
  f.st = RUN;
}

// Nothing can run.  Before calling it a deadlock: a wave in which SOME lanes wait at a collective while every other live lane
// sits at a barrier is what divergence looks like on the hardware - e.g. lane groups with group-uniform loop bounds, the groups
// that have more iterations shuffling among themselves while the others (exec-masked) are already past the loop.  Release the
// collective for the lanes that arrived: the others count as inactive for it (ballot bit 0, undefined as a shuffle source).
bool release_partial_collectives() {
  bool any = false;
  for (size_t wi = 0; wi < waves.size(); wi++) {
    Wave &w = waves[wi];
    unsigned long long at = 0;
    for (int l = 0; l < 64 && wi * 64 + l < fibers.size(); l++)
      if (fibers[wi * 64 + l].st == WAIT_WAVE) at |= 1ull << l;
    if (!at) continue;
    memcpy(w.snap, w.buf, sizeof(w.snap));
    w.snap_live = at;
    for (int l = 0; l < 64; l++)
      if ((at >> l) & 1ull) fibers[wi * 64 + l].st = RUN;
    w.arrived &= ~at;
    any = true;
  }
  return any;
}

[[noreturn]] void deadlock(const dim3 &b) {
  fprintf(stderr, "emu: DEADLOCK in block (%u, %u): no work-item can run\n", b.x, b.y);
  const char *nm[] = {"run", "wave collective", "wave barrier", "workgroup barrier", "done"};
  for (size_t w = 0; w < waves.size(); w++) {
    fprintf(stderr, "  wave %zu live %016llx:", w, waves[w].live);
    int cnt[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 64 && w * 64 + l < fibers.size(); l++) cnt[fibers[w * 64 + l].st]++;
    for (int s = 0; s < 5; s++)
      if (cnt[s]) fprintf(stderr, " %d x %s", cnt[s], nm[s]);
    fprintf(stderr, "\n");
  }
  abort();
}
}  // namespace

void wave_collective(const void *in, unsigned bytes, void *all, unsigned long long *live) {
  if (bytes > 16) {
    fprintf(stderr, "emu: collective of %u bytes\n", bytes);
    abort();
  }
  Wave &w = waves[me->wave];
  memcpy(w.buf[me->lane], in, bytes);
  w.bytes = bytes;
  w.arrived |= 1ull << me->lane;
  me->st = WAIT_WAVE;
  try_release_wave(me->wave);
  while (me->st != RUN) yield();
  unsigned char *o = static_cast<unsigned char *>(all);
  for (int l = 0; l < 64; l++) memcpy(o + (size_t)l * bytes, w.snap[l], bytes);
  *live = w.snap_live;
}

void wave_sync() {
  Wave &w = waves[me->wave];
  w.arrived |= 1ull << me->lane;
  me->st = WAIT_WSYNC;
  try_release_wave(me->wave);
  while (me->st != RUN) yield();
}

void block_sync() {
  block_arrived++;
  me->st = WAIT_BLOCK;
  try_release_block();
  while (me->st != RUN) yield();
}

void launch_impl(dim3 grid, dim3 block, void (*fn)(void *), void *ctx) {
  const unsigned nthr = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1 || nthr == 0 || nthr > 1024) {
    fprintf(stderr, "emu: block shape %u x %u x %u\n", block.x, block.y, block.z);
    abort();
  }
  order_init();
  if (fibers.size() < nthr) fibers.resize(nthr);
  const unsigned nw = (nthr + 63) / 64;
  k_fn = fn;
  k_ctx = ctx;
  Item *const saved_cur = cur;
  Fiber *const saved_me = me;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        waves.assign(nw, Wave{});
        block_live = (int)nthr;
        block_arrived = 0;
        for (unsigned t = 0; t < nthr; t++) {
          Fiber &f = fibers[t];
          f.item.tid = uint3{t, 0, 0};
          f.item.bid = uint3{bx, by, bz};
          f.item.bdim = uint3{block.x, 1, 1};
          f.item.gdim = uint3{grid.x, grid.y, grid.z};
          f.wave = (int)(t / 64);
          f.lane = (int)(t % 64);
          waves[f.wave].live |= 1ull << f.lane;
          start_fiber(f);
        }
        for (unsigned t = nthr; t < fibers.size(); t++) fibers[t].st = DONE;
        int remaining = (int)nthr;
        while (remaining > 0) {
          bool ran = false;
          const std::vector<unsigned> &ord = sweep_order(nthr);
          for (unsigned i = 0; i < nthr; i++) {
            Fiber &f = fibers[ord[i]];
            if (f.st != RUN) continue;
            ran = true;
            me = &f;
            cur = &f.item;
#if EMU_ASAN
            __sanitizer_start_switch_fiber(&sched_fake, f.stack, kStack);
#endif
            emu_switch(&sched_sp, f.sp);
#if EMU_ASAN
            __sanitizer_finish_switch_fiber(sched_fake, nullptr, nullptr);
#endif
            if (f.st == DONE) remaining--;
          }
          if (!ran && !release_partial_collectives()) deadlock(dim3(bx, by, bz));
        }
      }
  cur = saved_cur;
  me = saved_me;
}

}  // namespace emu
