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

#include <string>

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

// LDS: every __shared__ object of the emulated kernels is a static in the link section "emu_lds" (hip/hip_runtime.h), and so is the
// dynamic LDS of the column-panel kernels (DGS_DYN_SHARED in dgs_common.h).  One workgroup at a time owns the section; when
// several are resident (DGS_EMU_BLOCKS > 1) the scheduler saves and restores it around every workgroup switch.
namespace dgs {
alignas(16) __attribute__((section("emu_lds"))) char panel_dyn[160 * 1024];
alignas(16) __attribute__((section("emu_lds"))) char sd_dyn[160 * 1024];
}  // namespace dgs
extern "C" char __start_emu_lds[], __stop_emu_lds[];

namespace emu {

Item *cur = nullptr;

namespace {
struct Block;
}
void mem_wave_end(Block *b, int wave);  // relaxed-memory mode (below)
void mem_kernel_end();
void mem_init();

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
  bool relaxed = false;  // gave its turn away in a spin-wait (s_sleep): no progress of its own
  int relaxes = 0;       // ... how often during the workgroup's current turn
};
struct Wave {
  unsigned char buf[64][16], snap[64][16];
  unsigned long long live = 0, arrived = 0, snap_live = 0;
  unsigned bytes = 0;
};
// One resident workgroup.  DGS_EMU_BLOCKS = 1 (default): workgroups run one after the other, each to completion, in dispatch
// order - enough for kernels whose workgroups do not wait for one another.  DGS_EMU_BLOCKS = n: up to n workgroups are resident
// and take turns (a workgroup gives its turn away when all it did in a sweep was spin: `s_sleep` is the yield point), which is
// what a kernel with hand-overs BETWEEN workgroups needs - the soft barrier of the column-panel sweep, the slice-by-slice hub
// chains - and DGS_EMU_BLOCK_ORDER = fwd | rev | rand:<seed> is the order in which they are dispatched (a consumer resident
// before its producer has to wait, not fail).
struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  std::vector<char> lds;  // the section's contents while another workgroup owns it (full copy: sanitizer builds)
  std::vector<std::pair<unsigned, std::vector<char>>> lds_pages;  // ... or just the pages the launch has written (dirty tracking)
  dim3 bid;
  size_t dispatch = 0;    // position in the launch's dispatch order: CU = dispatch % CUs, XCD = dispatch % 8 (memory model)
  int live = 0, arrived = 0, remaining = 0;
  bool fresh = true;      // has not run yet: its LDS is whatever the previous owner left (as on the hardware)
};
void *sched_fake = nullptr;
const void *sched_bottom = nullptr;
size_t sched_size = 0;
void *sched_sp = nullptr;
Fiber *me = nullptr;
Block *blk = nullptr;
void (*k_fn)(void *) = nullptr;
void *k_ctx = nullptr;
std::vector<char *> stack_pool;

// The order in which the runnable work-items of a workgroup get their turn.  Between two rendezvous points a fiber runs
// undisturbed, so a missing barrier BETWEEN waves only shows when the reader happens to run before the writer (RAW) or the
// over-writer before the reader (WAR): one fixed order hides one of the two.  DGS_EMU_ORDER = fwd (default: ascending
// work-item id) | rev | rand:<seed> (a fresh permutation of the waves and of the lanes in each wave for every sweep).
enum Order { FWD, REV, RAND };
Order order_mode = FWD, block_order = FWD;
unsigned long long order_state = 1, block_state = 1;
int max_resident = 1, preempt_n = 0;
unsigned long long preempt_state = 0x9e3779b97f4a7c15ull;
bool order_read = false;
std::vector<unsigned> order_buf;
unsigned xs_next(unsigned long long &st, unsigned n) {  // xorshift64*: deterministic for a seed
  st ^= st >> 12;
  st ^= st << 25;
  st ^= st >> 27;
  return (unsigned)(((st * 2685821657736338717ull) >> 33) % n);
}
void parse_order(const char *name, Order &mode, unsigned long long &state) {
  const char *e = getenv(name);
  if (!e || !*e || !strcmp(e, "fwd")) return;
  if (!strcmp(e, "rev")) mode = REV;
  else if (!strncmp(e, "rand:", 5)) {
    mode = RAND;
    state = strtoull(e + 5, nullptr, 10) * 0x9e3779b97f4a7c15ull + 0x1234567ull;
    if (!state) state = 1;
  } else {
    fprintf(stderr, "emu: %s=%s (fwd | rev | rand:<seed>)\n", name, e);
    abort();
  }
}
void order_init() {
  if (order_read) return;
  order_read = true;
  parse_order("DGS_EMU_ORDER", order_mode, order_state);
  parse_order("DGS_EMU_BLOCK_ORDER", block_order, block_state);
  if (const char *e = getenv("DGS_EMU_BLOCKS")) max_resident = atoi(e) > 1 ? atoi(e) : 1;
  if (const char *e = getenv("DGS_EMU_PREEMPT")) preempt_n = atoi(e) > 0 ? atoi(e) : 0;
  mem_init();
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
      const unsigned o = xs_next(order_state, w);
      if (o != w - 1)
        for (unsigned l = 0; l < 64; l++) {
          const unsigned a = (w - 1) * 64 + l, b = o * 64 + l;
          if (a < nthr && b < nthr) std::swap(order_buf[a], order_buf[b]);
        }
    }
    for (unsigned w = 0; w < nw; w++) {
      const unsigned n = min(64u, nthr - w * 64);
      for (unsigned l = n; l > 1; l--) std::swap(order_buf[w * 64 + l - 1], order_buf[w * 64 + xs_next(order_state, l)]);
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
    if ((size_t)(wi * 64 + l) < blk->fibers.size() && blk->fibers[wi * 64 + l].st == from) blk->fibers[wi * 64 + l].st = RUN;
  w.arrived = 0;
}
void try_release_wave(int wi) {
  Wave &w = blk->waves[wi];
  if (!w.arrived || (w.arrived & w.live) != w.live) return;
  // every live lane waits: they must all wait for the same kind of rendezvous
  State kind = DONE;
  for (int l = 0; l < 64; l++)
    if ((w.live >> l) & 1ull) {
      const State s = blk->fibers[wi * 64 + l].st;
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
  if (blk->live > 0 && blk->arrived == blk->live) {
    for (auto &f : blk->fibers)
      if (f.st == WAIT_BLOCK) f.st = RUN;
    blk->arrived = 0;
  }
}

extern "C" void emu_fiber_entry() {
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &sched_bottom, &sched_size);
#endif
  k_fn(k_ctx);
  Fiber *f = me;
  f->st = DONE;
  Wave &w = blk->waves[f->wave];
  w.live &= ~(1ull << f->lane);
  blk->live--;
  try_release_wave(f->wave);
  try_release_block();
#if EMU_ASAN
  __sanitizer_start_switch_fiber(nullptr, sched_bottom, sched_size);  // this fiber ends here
#endif
  emu_switch(&f->sp, sched_sp);
  abort();  // never resumed
}

char *get_stack() {
  if (!stack_pool.empty()) {
    char *s = stack_pool.back();
    stack_pool.pop_back();
    return s;
  }
  char *s = static_cast<char *>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
  if (s == MAP_FAILED) {
    perror("emu: mmap");
    abort();
  }
  return s;
}

void start_fiber(Fiber &f) {
  if (!f.stack) f.stack = get_stack();
  // stack image emu_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into the entry with rsp = 16 n + 8
  uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
  void **sp = reinterpret_cast<void **>(top - 8);  // the slot `ret` would have left behind: keeps the ABI's alignment at entry
  *--sp = reinterpret_cast<void *>(&emu_fiber_entry);
  for (int i = 0; i < 6; i++) *--sp = nullptr;
  f.sp = sp;
  f.st = RUN;
  f.relaxed = false;
}

// Nothing can run.  Before calling it a deadlock: a wave in which SOME lanes wait at a collective while every other live lane
// sits at a barrier is what divergence looks like on the hardware - e.g. lane groups with group-uniform loop bounds, the groups
// that have more iterations shuffling among themselves while the others (exec-masked) are already past the loop.  Release the
// collective for the lanes that arrived: the others count as inactive for it (ballot bit 0, undefined as a shuffle source).
bool release_partial_collectives() {
  bool any = false;
  for (size_t wi = 0; wi < blk->waves.size(); wi++) {
    Wave &w = blk->waves[wi];
    unsigned long long at = 0;
    for (int l = 0; l < 64 && wi * 64 + l < blk->fibers.size(); l++)
      if (blk->fibers[wi * 64 + l].st == WAIT_WAVE) at |= 1ull << l;
    if (!at) continue;
    memcpy(w.snap, w.buf, sizeof(w.snap));
    w.snap_live = at;
    for (int l = 0; l < 64; l++)
      if ((at >> l) & 1ull) blk->fibers[wi * 64 + l].st = RUN;
    w.arrived &= ~at;
    any = true;
  }
  return any;
}

[[noreturn]] void deadlock(const char *why) {
  fprintf(stderr, "emu: DEADLOCK in block (%u, %u): %s\n", blk->bid.x, blk->bid.y, why);
  const char *nm[] = {"run", "wave collective", "wave barrier", "workgroup barrier", "done"};
  for (size_t w = 0; w < blk->waves.size(); w++) {
    fprintf(stderr, "  wave %zu live %016llx:", w, blk->waves[w].live);
    int cnt[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 64 && w * 64 + l < blk->fibers.size(); l++) cnt[blk->fibers[w * 64 + l].st]++;
    for (int s = 0; s < 5; s++)
      if (cnt[s]) fprintf(stderr, " %d x %s", cnt[s], nm[s]);
    fprintf(stderr, "\n");
  }
  abort();
}

void init_block(Block &b, dim3 bid, dim3 block, dim3 grid, unsigned nthr, size_t dispatch = 0) {
  b.bid = bid;
  b.dispatch = dispatch;
  b.fibers.resize(nthr);
  b.waves.assign((nthr + 63) / 64, Wave{});
  b.live = b.remaining = (int)nthr;
  b.arrived = 0;
  b.fresh = true;
  for (unsigned t = 0; t < nthr; t++) {
    Fiber &f = b.fibers[t];
    f.item.tid = uint3{t, 0, 0};
    f.item.bid = uint3{bid.x, bid.y, bid.z};
    f.item.bdim = uint3{block.x, 1, 1};
    f.item.gdim = uint3{grid.x, grid.y, grid.z};
    f.wave = (int)(t / 64);
    f.lane = (int)(t % 64);
    b.waves[f.wave].live |= 1ull << f.lane;
    start_fiber(f);
  }
}

// Runs the current workgroup until it is finished or until it has nothing to do but spin.  "Nothing but spin" = a sweep in which
// every work-item that ran ended in a spin-wait, or - the work-items of a spinning workgroup need not be in step: the one that
// released a barrier runs ahead of the others by a loop iteration for good - every runnable work-item has spun at least twice
// during this turn.  Returns true when the turn looked like progress (a work-item finished, or nobody spun at all).
bool run_block() {
  bool finished_any = false, relaxed_any = false;
  int spins = 0;
  const unsigned nthr = (unsigned)blk->fibers.size();
  for (auto &f : blk->fibers) f.relaxes = 0;
  while (blk->remaining > 0) {
    bool ran = false, worked = false, relaxed_now = false;
    const std::vector<unsigned> &ord = sweep_order(nthr);
    for (unsigned i = 0; i < nthr; i++) {
      Fiber &f = blk->fibers[ord[i]];
      if (f.st != RUN) continue;
      ran = true;
      f.relaxed = false;
      me = &f;
      cur = &f.item;
#if EMU_ASAN
      __sanitizer_start_switch_fiber(&sched_fake, f.stack, kStack);
#endif
      emu_switch(&sched_sp, f.sp);
#if EMU_ASAN
      __sanitizer_finish_switch_fiber(sched_fake, nullptr, nullptr);
#endif
      if (f.st == DONE) {
        blk->remaining--;
        finished_any = true;
        if (!blk->waves[f.wave].live) mem_wave_end(blk, f.wave);  // a wave's outstanding stores complete when it ends
      }
      if (f.relaxed) {
        f.relaxes++;
        relaxed_now = relaxed_any = true;
      } else {
        worked = true;
      }
    }
    if (!ran && !release_partial_collectives()) deadlock("no work-item can run");
    if (!ran) continue;
    // DGS_EMU_PREEMPT=n: a resident workgroup loses its turn after a sweep with probability 1/n - workgroups then interleave at the
    // granularity of rendezvous points instead of running to completion one after the other (hand-overs BETWEEN workgroups that
    // never spin - the in-kernel fold's arrival tickets - see no interleaving otherwise)
    if (max_resident > 1 && preempt_n > 0 && blk->remaining > 0 && xs_next(preempt_state, (unsigned)preempt_n) == 0) return true;
    bool all_spun = relaxed_now;
    if (all_spun && worked)
      for (auto &f : blk->fibers)
        if (f.st == RUN && f.relaxes < 2) {
          all_spun = false;
          break;
        }
    if (!all_spun) {
      if (worked) spins = 0;
      continue;
    }
    if (max_resident > 1) return finished_any;  // somebody else's turn
    if (++spins > 1000000) return false;  // alone and spinning for good (a bounded spin - the panel sweep's soft barrier - runs out long before)
  }
  return finished_any || !relaxed_any;
}
}  // namespace

// ---- relaxed-memory mode (DGS_EMU_MEM=relaxed) ---------------------------------------------------------------------------------
// What the plain emulation cannot get wrong: the ORDER and VISIBILITY of global-memory accesses between workgroups.  Every access
// completes at once there, so a hand-over that drops its write-through bit, its drain or its L1 bypass still "works".  This mode
// models what MI355X_MICROARCH.md ("Workgroup dispatch, XCD placement & inter-workgroup visibility") states about the hardware,
// for the accesses the kernels make through load_vec / store_vec, the raw-buffer builtins, __hip_atomic_load / _store and the atomic
// read-modify-writes, inside WATCHED address ranges (emu_mem_watch: the SpMM workspace - partial rows, tables, counters):
//   * a wave's stores sit in the wave's queue until the wave drains it (`s_waitcnt vmcnt(0)`: drain_vmem) or ends; a read-modify-
//     write atomic goes to memory at once - it can overtake the stores in front of it (why the fold drains before it counts in);
//   * a performed sc1 store is written through to memory (visible everywhere); a performed PLAIN store stays dirty in the L2 of
//     its workgroup's XCD (dispatch index % 8) - other XCDs read the old memory until the kernel ends (no release fence in these
//     kernels);
//   * a PLAIN load goes through the L1 of its workgroup's CU (dispatch index % CUs): the first plain load of a 128-byte line pins
//     the moment, later plain loads of the line see memory as of that moment - never refreshed by other CUs' stores, for the whole
//     launch (no acquire in these kernels); an sc1 load (buffer load with the sc1 bit, agent-scope atomic load) bypasses L1 and
//     reads the XCD's L2 / memory as of now;
//   * a wave sees its own queued stores; the waves of one CU see each other's performed stores.
// Besides returning the (possibly stale) value the model COUNTS the reads that were served stale data although a newer value
// existed (emu_mem_report): "unperformed" - the newest store to the word was still in another workgroup's queue or dirty in another
// XCD's L2 - and "l1_stale" - a plain load that hit a line pinned before another workgroup's newer write-through.  A correct
// hand-over has zero of both; tests/emu/mutation_check.py shows that dropping sc1 on the stores, the drain, or sc1 on the loads
// of the in-kernel fold gives wrong bits or a non-zero count.
}  // namespace emu
#include <algorithm>
#include <unordered_map>
namespace emu {
namespace {
bool g_mem_on = false;
struct Range { uintptr_t lo, hi; };
std::vector<Range> g_watch;
struct Perf { unsigned long long t; unsigned val; int cu; size_t wg; };
struct WordHist { unsigned orig; std::vector<Perf> w; };
struct Pending { uintptr_t word; unsigned val; int sc1; };
struct Dirty { unsigned long long t; unsigned val; int cu; size_t wg; };
std::unordered_map<uintptr_t, WordHist> g_hist;                 // performed write-throughs of this launch, per watched word
std::unordered_map<uintptr_t, Dirty> g_dirty[8];                // performed plain stores, per XCD (dirty L2 lines)
std::unordered_map<uintptr_t, unsigned long long> g_l1[1024];  // per CU: line -> moment of the first plain load
std::unordered_map<unsigned long long, std::vector<Pending>> g_queue;  // per (workgroup, wave)
std::unordered_map<uintptr_t, int> g_unperf;                   // watched word -> number of queued stores to it (any wave)
unsigned long long g_now = 1;
unsigned long long g_cnt_unperformed = 0, g_cnt_l1stale = 0, g_cnt_loads = 0, g_cnt_stores = 0;
int g_ncu = 256;
bool watched(uintptr_t a) {
  for (const Range &r : g_watch)
    if (a >= r.lo && a < r.hi) return true;
  return false;
}
unsigned long long wave_key_of(const Block *b, int wave) { return ((unsigned long long)b->dispatch << 8) | (unsigned)wave; }  // (<= 16 waves per workgroup)
unsigned long long wave_key() { return wave_key_of(blk, me->wave); }
int cur_cu() { return (int)(blk->dispatch % (size_t)g_ncu); }
int cur_xcd() { return (int)(blk->dispatch % 8); }
void perform(const Block *b, const Pending &p) {
  const int cu = (int)(b->dispatch % (size_t)g_ncu), xcd = (int)(b->dispatch % 8);
  const unsigned long long t = ++g_now;
  auto u = g_unperf.find(p.word);
  if (u != g_unperf.end() && --u->second <= 0) g_unperf.erase(u);
  if (p.sc1) {
    WordHist &h = g_hist[p.word];
    if (h.w.empty()) memcpy(&h.orig, reinterpret_cast<void *>(p.word), 4);
    h.w.push_back(Perf{t, p.val, cu, b->dispatch});
    memcpy(reinterpret_cast<void *>(p.word), &p.val, 4);
    g_dirty[xcd].erase(p.word);  // write-through drops the line from the XCD's L2
  } else {
    g_dirty[xcd][p.word] = Dirty{t, p.val, cu, b->dispatch};
  }
}
void drain_queue(const Block *b, int wave) {
  auto it = g_queue.find(wave_key_of(b, wave));
  if (it == g_queue.end()) return;
  for (const Pending &p : it->second) perform(b, p);
  g_queue.erase(it);
}
unsigned load_word(uintptr_t a, int sc1) {
  g_cnt_loads++;
  // 1. the wave's own queued stores
  auto q = g_queue.find(wave_key());
  if (q != g_queue.end())
    for (size_t i = q->second.size(); i-- > 0;)
      if (q->second[i].word == a) return q->second[i].val;
  const int cu = cur_cu(), xcd = cur_xcd();
  unsigned long long T = g_now;
  if (!sc1) {
    auto &l1 = g_l1[cu];
    auto ins = l1.emplace(a >> 7, g_now);
    T = ins.first->second;
  }
  // 2. value as of T: this XCD's dirty line, else memory's history
  unsigned val;
  bool newer_elsewhere = false, stale_l1 = false;
  auto d = g_dirty[xcd].find(a);
  auto h = g_hist.find(a);
  bool have = false;
  unsigned long long tv = 0;
  if (d != g_dirty[xcd].end() && (d->second.t <= T || d->second.cu == cu)) {
    val = d->second.val;
    tv = d->second.t;
    have = true;
  }
  if (h != g_hist.end()) {
    const WordHist &wh = h->second;
    unsigned mv = wh.orig;
    unsigned long long mt = 0;
    for (const Perf &p : wh.w)
      if (p.t <= T || p.cu == cu) { mv = p.val; mt = p.t; }
    if (!have || mt > tv) { val = mv; tv = mt; have = true; }
    if (!wh.w.empty() && wh.w.back().t > tv && wh.w.back().wg != blk->dispatch) stale_l1 = true;
  }
  if (!have) memcpy(&val, reinterpret_cast<void *>(a), 4);
  // 3. hazards: a newer value exists that this read cannot see
  if (g_unperf.count(a)) newer_elsewhere = true;  // (own-wave stores returned above; another wave's store is still in its queue)
  for (int x = 0; x < 8 && !newer_elsewhere; x++)
    if (x != xcd) {
      auto o = g_dirty[x].find(a);
      if (o != g_dirty[x].end() && o->second.t > tv) newer_elsewhere = true;
    }
  if (newer_elsewhere) g_cnt_unperformed++;
  if (stale_l1) g_cnt_l1stale++;
  return val;
}
}  // namespace

void mem_init() {
  const char *e = getenv("DGS_EMU_MEM");
  g_mem_on = e && !strcmp(e, "relaxed");
  if (const char *c = getenv("DGS_EMU_CUS")) g_ncu = atoi(c) > 0 && atoi(c) <= 1024 ? atoi(c) : 256;
}
bool mem_on() { return g_mem_on && blk != nullptr && !g_watch.empty(); }
void mem_load(const void *p, void *out, unsigned bytes, int sc1) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (!watched(a) || (a & 3) || (bytes & 3)) {
    memcpy(out, p, bytes);
    return;
  }
  for (unsigned i = 0; i < bytes; i += 4) {
    const unsigned v = load_word(a + i, sc1);
    memcpy(static_cast<char *>(out) + i, &v, 4);
  }
}
void mem_store(void *p, const void *src, unsigned bytes, int sc1) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (!watched(a) || (a & 3) || (bytes & 3)) {
    memcpy(p, src, bytes);
    return;
  }
  std::vector<Pending> &q = g_queue[wave_key()];
  for (unsigned i = 0; i < bytes; i += 4) {
    unsigned v;
    memcpy(&v, static_cast<const char *>(src) + i, 4);
    q.push_back(Pending{a + i, v, sc1});
    g_unperf[a + i]++;
    g_cnt_stores++;
  }
}
// s_waitcnt vmcnt(0) is a WAVE instruction: when it retires, every lane's earlier stores have been issued AND performed.  The fibers
// of a wave are not in lockstep, so "every lane's earlier stores have been issued" has to be said explicitly - in BOTH memory modes: a
// lane that draws an arrival ticket behind the drain must not do so while its wave-mates' fibers have not reached their stores yet
// (found by the round-6 campaign with resident workgroups in random fiber order: a unit wave with no ticket in flight counted its
// last partial row in before lanes 1 .. 63 had written theirs, and a wave of the SAME workgroup folded the row - one arg id wrong.
// An artefact of the emulation: on the hardware the wave's stores precede the wait in program order).  Then the queue is performed.
void mem_drain() {
  if (blk == nullptr) return;
  static const bool nosync = getenv("DGS_EMU_DRAIN_NOSYNC") != nullptr;  // (the pre-fix behaviour, to reproduce the artefact)
  if (!nosync || mem_on()) wave_sync();
  if (mem_on()) drain_queue(blk, me->wave);
}
void mem_wave_end(Block *b, int wave) {
  if (g_mem_on) drain_queue(b, wave);
}
void mem_kernel_end() {  // the end of a kernel is a release / the start of the next an acquire: everything lands, every cache is cold
  if (!g_mem_on) return;
  for (auto &kv : g_queue) (void)kv;  // (queues are empty: every wave ended)
  g_queue.clear();
  g_unperf.clear();
  for (int x = 0; x < 8; x++) {
    // dirty lines are written back in the order they were written (two XCDs that wrote one word: the later one wins, as on a
    // write-back of both)
    for (auto &kv : g_dirty[x]) {
      bool newest = true;
      for (int y = 0; y < 8; y++)
        if (y != x) {
          auto o = g_dirty[y].find(kv.first);
          if (o != g_dirty[y].end() && o->second.t > kv.second.t) newest = false;
        }
      auto h = g_hist.find(kv.first);
      if (h != g_hist.end() && !h->second.w.empty() && h->second.w.back().t > kv.second.t) newest = false;
      if (newest) memcpy(reinterpret_cast<void *>(kv.first), &kv.second.val, 4);
    }
  }
  for (int x = 0; x < 8; x++) g_dirty[x].clear();
  g_hist.clear();
  for (int c = 0; c < g_ncu; c++) g_l1[c].clear();
}
}  // namespace emu
extern "C" void emu_mem_watch(const void *base, size_t bytes) {  // bytes = 0: forget every range
  if (!bytes) emu::g_watch.clear();
  else emu::g_watch.push_back(emu::Range{reinterpret_cast<uintptr_t>(base), reinterpret_cast<uintptr_t>(base) + bytes});
}
extern "C" void emu_mem_report(unsigned long long *out /* [4]: unperformed, l1_stale, loads, stores */, int clear) {
  out[0] = emu::g_cnt_unperformed;
  out[1] = emu::g_cnt_l1stale;
  out[2] = emu::g_cnt_loads;
  out[3] = emu::g_cnt_stores;
  if (clear) emu::g_cnt_unperformed = emu::g_cnt_l1stale = emu::g_cnt_loads = emu::g_cnt_stores = 0;
}
namespace emu {

void relax() {  // s_sleep inside a spin-wait: give the turn away (to the other work-items, then to the other workgroups)
  me->relaxed = true;
  yield();
}

void wave_collective(const void *in, unsigned bytes, void *all, unsigned long long *live) {
  if (bytes > 16) {
    fprintf(stderr, "emu: collective of %u bytes\n", bytes);
    abort();
  }
  Wave &w = blk->waves[me->wave];
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
  Wave &w = blk->waves[me->wave];
  w.arrived |= 1ull << me->lane;
  me->st = WAIT_WSYNC;
  try_release_wave(me->wave);
  while (me->st != RUN) yield();
}

void block_sync() {
  blk->arrived++;
  me->st = WAIT_BLOCK;
  try_release_block();
  while (me->st != RUN) yield();
}

// Launch log: the source text of the kernel argument of every hipLaunchKernelGGL since the last emu_launch_log(clear = 1), one
// "name grid.x grid.y" line each - how the tests see WHICH kernels a C-ABI call took and how many launches it cost.
static std::string g_launch_log;
void log_launch(const char *kern, dim3 grid) {
  if (g_launch_log.size() < (1u << 20)) g_launch_log += std::string(kern) + " " + std::to_string(grid.x) + " " + std::to_string(grid.y) + "\n";
}
}  // namespace emu
extern "C" size_t emu_launch_log(char *buf, size_t cap, int clear) {
  const size_t n = emu::g_launch_log.size() < cap ? emu::g_launch_log.size() : (cap ? cap - 1 : 0);
  if (buf && cap) {
    memcpy(buf, emu::g_launch_log.data(), n);
    buf[n] = 0;
  }
  if (clear) emu::g_launch_log.clear();
  return n;
}
namespace emu {
// Dirty-page tracking of the LDS section while several workgroups are resident.  Every __shared__ object of every template
// instantiation is a distinct static, ~21 MB in all, of which one launch touches a few pages; copying the whole section at every
// workgroup switch made pre-emptive schedules (DGS_EMU_PREEMPT) 20x slower than the kernels themselves.  The section is
// write-protected at the start of a multi-resident launch; the first write to a page faults once, the handler notes the page and
// unprotects it; a workgroup switch saves / restores only the noted pages.  (Not under ASAN / UBSAN builds: the sanitizers own SIGSEGV.)
}  // namespace emu
#include <signal.h>
#include <unistd.h>
namespace emu {
namespace {
constexpr size_t kPage = 4096;
bool lds_track = false;
unsigned lds_dirty[8192];
volatile unsigned lds_ndirty = 0;
struct sigaction lds_old_sa;
char *lds_lo() { return reinterpret_cast<char *>(reinterpret_cast<uintptr_t>(__start_emu_lds) & ~(uintptr_t)(kPage - 1)); }
char *lds_hi() { return reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(__stop_emu_lds) + kPage - 1) & ~(uintptr_t)(kPage - 1)); }
void lds_segv(int sig, siginfo_t *si, void *uc) {
  char *a = static_cast<char *>(si->si_addr);
  if (lds_track && a >= __start_emu_lds && a < __stop_emu_lds && lds_ndirty < 8192) {
    const unsigned pg = (unsigned)((a - lds_lo()) / kPage);
    lds_dirty[lds_ndirty] = pg;
    lds_ndirty = lds_ndirty + 1;
    // a page that straddles the section's ends holds other data too: it simply becomes writable like before
    mprotect(lds_lo() + (size_t)pg * kPage, kPage, PROT_READ | PROT_WRITE);
    return;
  }
  sigaction(SIGSEGV, &lds_old_sa, nullptr);  // not ours: let the default / previous handler have it on the retry
  (void)sig;
  (void)uc;
}
bool lds_track_begin() {
#if EMU_ASAN
  return false;
#else
  if (getenv("DGS_EMU_UBSAN") || getenv("DGS_EMU_NO_LDS_TRACK")) return false;
  // only whole pages strictly inside the section are protected (its first / last partial pages are shared with other data and
  // are treated as always dirty)
  char *lo = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(__start_emu_lds) + kPage - 1) & ~(uintptr_t)(kPage - 1));
  char *hi = reinterpret_cast<char *>(reinterpret_cast<uintptr_t>(__stop_emu_lds) & ~(uintptr_t)(kPage - 1));
  if (hi <= lo) return false;
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = lds_segv;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  if (sigaction(SIGSEGV, &sa, &lds_old_sa) != 0) return false;
  lds_ndirty = 0;
  // the partial first / last pages: always saved
  if (lo > __start_emu_lds) lds_dirty[lds_ndirty++] = 0;
  if (hi < __stop_emu_lds) lds_dirty[lds_ndirty++] = (unsigned)((hi - lds_lo()) / kPage);
  lds_track = true;
  if (mprotect(lo, (size_t)(hi - lo), PROT_READ) != 0) {
    lds_track = false;
    sigaction(SIGSEGV, &lds_old_sa, nullptr);
    return false;
  }
  return true;
#endif
}
void lds_track_end() {
  if (!lds_track) return;
  lds_track = false;
  mprotect(lds_lo(), (size_t)(lds_hi() - lds_lo()), PROT_READ | PROT_WRITE);
  sigaction(SIGSEGV, &lds_old_sa, nullptr);
}
// the bytes of page pg that belong to the section
void lds_page_span(unsigned pg, char *&p, size_t &n) {
  char *a = lds_lo() + (size_t)pg * kPage, *b = a + kPage;
  if (a < __start_emu_lds) a = __start_emu_lds;
  if (b > __stop_emu_lds) b = __stop_emu_lds;
  p = a;
  n = (size_t)(b - a);
}
void lds_save(Block *b) {
  b->lds_pages.clear();
  const unsigned n = lds_ndirty;
  for (unsigned i = 0; i < n; i++) {
    char *p;
    size_t len;
    lds_page_span(lds_dirty[i], p, len);
    b->lds_pages.emplace_back(lds_dirty[i], std::vector<char>(p, p + len));
  }
}
void lds_restore(const Block *b) {
  for (const auto &pg : b->lds_pages) {
    char *p;
    size_t len;
    lds_page_span(pg.first, p, len);
    memcpy(p, pg.second.data(), len);  // (a page that was dirty when b was saved is writable by now)
  }
}
}  // namespace

void launch_impl(dim3 grid, dim3 block, void (*fn)(void *), void *ctx) {
  const unsigned nthr = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1 || nthr == 0 || nthr > 1024) {
    fprintf(stderr, "emu: block shape %u x %u x %u\n", block.x, block.y, block.z);
    abort();
  }
  order_init();
  // saved across a nested launch (none today) and restored at the end
  void (*const saved_fn)(void *) = k_fn;
  void *const saved_ctx = k_ctx;
  Item *const saved_cur = cur;
  Fiber *const saved_me = me;
  Block *const saved_blk = blk;
  k_fn = fn;
  k_ctx = ctx;
  // dispatch order: x fastest, then y, then z (the hardware's), permuted as a whole on request
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  std::vector<size_t> order(nblocks);
  for (size_t i = 0; i < nblocks; i++) order[i] = block_order == REV ? nblocks - 1 - i : i;
  if (block_order == RAND)
    for (size_t i = nblocks; i > 1; i--) std::swap(order[i - 1], order[xs_next(block_state, (unsigned)(i < 0xffffffffu ? i : 0xffffffffu))]);
  auto bid_of = [&](size_t i) { return dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y))); };
  const size_t lds_bytes = (size_t)(__stop_emu_lds - __start_emu_lds);
  if (max_resident <= 1) {
    Block b;
    blk = &b;
    for (size_t i = 0; i < nblocks; i++) {
      init_block(b, bid_of(order[i]), block, grid, nthr, i);
      run_block();
      if (b.remaining > 0) deadlock("its work-items wait for another workgroup, and workgroups run one at a time (DGS_EMU_BLOCKS)");
    }
    for (auto &f : b.fibers) stack_pool.push_back(f.stack);
  } else {
    std::vector<Block *> res;
    size_t next = 0, turn = 0;
    Block *owner = nullptr;  // whose LDS is in the section
    const bool tracking = lds_track_begin();
    int idle_rounds = 0;
    while (next < nblocks || !res.empty()) {
      while (next < nblocks && (int)res.size() < max_resident) {
        Block *b = new Block;
        blk = b;
        init_block(*b, bid_of(order[next]), block, grid, nthr, next);
        next++;
        res.push_back(b);
      }
      if (turn >= res.size()) turn = 0;
      Block *b = res[turn];
      if (owner != b) {
        if (tracking) {
          if (owner) lds_save(owner);
          if (!b->fresh) lds_restore(b);
        } else {
          if (owner) {
            owner->lds.resize(lds_bytes);
            memcpy(owner->lds.data(), __start_emu_lds, lds_bytes);
          }
          if (!b->fresh) memcpy(__start_emu_lds, b->lds.data(), lds_bytes);
        }
        owner = b;
      }
      b->fresh = false;
      blk = b;
      const bool progress = run_block();
      if (getenv("DGS_EMU_TRACE")) fprintf(stderr, "turn %zu block %u,%u progress %d remaining %d resident %zu\n", turn, b->bid.x, b->bid.y, (int)progress, b->remaining, res.size());
      if (b->remaining == 0) {
        for (auto &f : b->fibers) stack_pool.push_back(f.stack);
        res.erase(res.begin() + (long)turn);
        if (owner == b) owner = nullptr;
        delete b;
        idle_rounds = 0;
        continue;  // (the next one slid into this turn)
      }
      idle_rounds = progress ? 0 : idle_rounds + 1;
      // (a turn that ends in a spin counts as idle even when the workgroup did real work before it: hence the generous limit)
      const int idle_limit = 200 * (int)res.size() + 8;
      if (idle_rounds > idle_limit && next >= nblocks) deadlock("every resident workgroup spins and none is left to dispatch");
      if (idle_rounds > idle_limit && (int)res.size() >= max_resident)
        deadlock("every resident workgroup spins and there is no room to dispatch the one they wait for (DGS_EMU_BLOCKS)");
      turn++;
    }
    if (tracking) lds_track_end();
  }
  mem_kernel_end();
  k_fn = saved_fn;
  k_ctx = saved_ctx;
  cur = saved_cur;
  me = saved_me;
  blk = saved_blk;
}

}  // namespace emu
