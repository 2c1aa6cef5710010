// tests/emu/emu_runtime.cc -- TEST INFRASTRUCTURE: the wave64 emulator behind tests/emu/include/hip/hip_runtime.h (read that header first).
//
// A launch runs its workgroups one after another on the calling thread (one launch at a time, process-wide).  Inside a workgroup every
// lane is a fiber with a stack of its own; the scheduler runs the lanes of a wave one after another until each has reached a
// cross-lane operation (all of them the SAME one, checked by operation kind and source line), evaluates it and goes round again; a
// wave that reaches a workgroup barrier or s_sleep gives way to the next wave.  A launch whose lanes can make no progress (a wave
// split over different operations, a barrier some wave never reaches) aborts the process with a description: such code would hang or
// misbehave on the GPU too, or relies on divergent cross-lane semantics the emulator does not model (see MGPU_ANY in mgpu_device.hpp).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <elf.h>
#include <map>

#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// AddressSanitizer has to be told about stack switches it did not make (python tests/emu/build_emu.py asan)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#include <sanitizer/asan_interface.h>
#endif
#endif
#ifndef EMU_ASAN
#define EMU_ASAN 0
#endif
// ... and ThreadSanitizer likewise (python tests/emu/build_emu.py tsan): every lane is a fiber of its own to it
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
extern "C" {
void *__tsan_get_current_fiber(void);
void *__tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void *fiber);
void __tsan_switch_to_fiber(void *fiber, unsigned flags);
}
#endif
#endif
#ifndef EMU_TSAN
#define EMU_TSAN 0
#endif

namespace emu {

thread_local Ctx *g_cur = nullptr;

namespace {

constexpr size_t kStackBytes = 256 * 1024;
constexpr size_t kLdsBytes = 160 * 1024;
constexpr int kTurn = 24; // cross-lane operations a wave gets per turn (about one step of the wave-scheduled kernels)

enum State : int { RUNNABLE = 0, AT_WAVE_OP, AT_SYNC, ASLEEP, DONE };

struct FiberImpl {
  Ctx ctx;
  void *sp = nullptr;     // saved stack pointer while the fiber is not running
  void *stack = nullptr;  // mmap'ed
  State state = DONE;
  int op = 0, site = 0;
  unsigned long long arg = 0, result = 0;
  int src = 0;
  void *fake = nullptr; // AddressSanitizer's fake-stack handle of this fiber while it is switched out
  void *tsan = nullptr; // ThreadSanitizer's fiber
};

struct Sched {
  std::vector<FiberImpl> fibers; // block threads
  void *main_sp = nullptr;
  FiberImpl *running = nullptr;
  const std::function<void()> *entry = nullptr;
  unsigned char *lds = nullptr;
  void *main_fake = nullptr;          // AddressSanitizer: the scheduler's own handle, and its stack as the fibers see it
  const void *main_bottom = nullptr;
  size_t main_size = 0;
  void *main_tsan = nullptr;
};
thread_local Sched *g_sched = nullptr;

// ---- context switch (x86-64 SysV: callee-saved rbx, rbp, r12-r15) ----------------------------------------------------------------
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

void yield_to_scheduler() {
  Sched *s = g_sched;
  FiberImpl *f = s->running;
#if EMU_ASAN
  __sanitizer_start_switch_fiber(f->state == DONE ? nullptr : &f->fake, s->main_bottom, s->main_size);
#endif
#if EMU_TSAN
  __tsan_switch_to_fiber(s->main_tsan, 0);
#endif
  emu_switch(&f->sp, s->main_sp);
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(f->fake, &g_sched->main_bottom, &g_sched->main_size);
#endif
}

extern "C" void emu_fiber_main() {
  Sched *s = g_sched;
  FiberImpl *f = s->running;
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &s->main_bottom, &s->main_size);
#endif
  (*s->entry)();
  f->state = DONE;
  yield_to_scheduler();
  fprintf(stderr, "emu: a finished fiber was resumed\n");
  abort();
}

void prepare(FiberImpl &f) {
  // stack top: [6 callee-saved zeros][return address = emu_fiber_main][alignment slot]
  uintptr_t top = (uintptr_t)f.stack + kStackBytes;
  top &= ~(uintptr_t)15;
  void **sp = (void **)top;
  *--sp = nullptr;                    // keeps (rsp + 8) % 16 == 0 at emu_fiber_main's entry, as after a call
  *--sp = (void *)&emu_fiber_main;    // `ret` of emu_switch jumps here
  for (int i = 0; i < 6; ++i) *--sp = nullptr;
  f.sp = sp;
}

[[noreturn]] void die(Sched &S, const char *what) {
  fprintf(stderr, "emu: %s\n", what);
  const size_t n = S.fibers.size();
  for (size_t w = 0; w < (n + 63) / 64; ++w) {
    fprintf(stderr, "  wave %zu:", w);
    int last_state = -1, last_op = -1, last_site = -1, run = 0;
    for (size_t l = w * 64; l < std::min(n, w * 64 + 64); ++l) {
      const FiberImpl &f = S.fibers[l];
      if ((int)f.state == last_state && f.op == last_op && f.site == last_site) {
        ++run;
        continue;
      }
      if (run) fprintf(stderr, " x%d", run);
      fprintf(stderr, " [lane %zu: state %d op %d line %d]", l - w * 64, (int)f.state, f.op, f.site);
      last_state = f.state; last_op = f.op; last_site = f.site; run = 1;
    }
    if (run) fprintf(stderr, " x%d", run);
    fprintf(stderr, "\n");
  }
  abort();
}

void run_fiber(Sched &S, FiberImpl &f) {
  S.running = &f;
  g_cur = &f.ctx;
#if EMU_ASAN
  __sanitizer_start_switch_fiber(&S.main_fake, f.stack, kStackBytes);
#endif
#if EMU_TSAN
  __tsan_switch_to_fiber(f.tsan, 0);
#endif
  emu_switch(&S.main_sp, f.sp);
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(S.main_fake, nullptr, nullptr);
#endif
  S.running = nullptr;
}

// Runs wave w until every live lane is at a workgroup barrier, asleep or done -- or until it has been through `budget` cross-lane
// operations: the waves of a workgroup take turns (a persistent wave that never meets a barrier would otherwise do the whole launch's
// work alone while its neighbours never leave the prologue).  Returns true if anything ran.
bool run_wave(Sched &S, size_t w, int budget) {
  const size_t n = S.fibers.size(), l0 = w * 64, l1 = std::min(n, l0 + 64);
  bool progressed = false;
  for (int ops = 0;;) {
    if (ops >= budget) return true;
    bool ran = false;
    // The order in which a wave's lanes run between two cross-lane operations is the emulator's choice -- the hardware runs them in
    // lockstep, and code that is correct there for any reason other than an explicit wave barrier depends on it.  MGPU_EMU_LANE_ORDER =
    // reverse | random makes such dependences visible (the k_trace finding of round 5); the default is lane 0 first.
    static const int order = [] {
      const char *e = getenv("MGPU_EMU_LANE_ORDER");
      return !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "random") ? 2 : 0));
    }();
    static thread_local unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    const size_t cnt = l1 - l0;
    size_t rot = 0, stride = 1;
    if (order == 2) {
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      rot = (size_t)(lcg >> 33) % cnt;
      stride = (cnt == 64) ? ((size_t)((lcg >> 20) & 31u) * 2 + 1) : 1; // odd: a permutation of 64
    }
    for (size_t k = 0; k < cnt; ++k) {
      const size_t i = order == 1 ? cnt - 1 - k : (order == 2 ? (rot + k * stride) % cnt : k);
      FiberImpl &f = S.fibers[l0 + i];
      if (f.state == RUNNABLE) {
        run_fiber(S, f);
        ran = true;
      }
    }
    progressed |= ran;
    // where does the wave stand?
    int live = 0, at_op = 0, at_sync = 0, asleep = 0;
    int op = 0, site = 0;
    bool same = true;
    for (size_t l = l0; l < l1; ++l) {
      const FiberImpl &f = S.fibers[l];
      if (f.state == DONE) continue;
      ++live;
      if (f.state == AT_WAVE_OP) {
        if (at_op == 0) { op = f.op; site = f.site; }
        else if (f.op != op || f.site != site) same = false;
        ++at_op;
      } else if (f.state == AT_SYNC) ++at_sync;
      else if (f.state == ASLEEP) ++asleep;
    }
    if (live == 0) return progressed;
    if (at_op == live) {
      if (!same) die(S, "the lanes of a wave reached DIFFERENT cross-lane operations (divergent control flow around a wave operation)");
      // evaluate
      unsigned long long ballot = 0;
      int first = -1;
      for (size_t l = l0; l < l1; ++l) {
        FiberImpl &f = S.fibers[l];
        if (f.state != AT_WAVE_OP) continue;
        if (first < 0) first = (int)(l - l0);
        if (op == OP_BALLOT && f.arg) ballot |= 1ull << (l - l0);
      }
      for (size_t l = l0; l < l1; ++l) {
        FiberImpl &f = S.fibers[l];
        if (f.state != AT_WAVE_OP) continue;
        switch (op) {
        case OP_BALLOT: f.result = ballot; break;
        case OP_FIRST: f.result = S.fibers[l0 + first].arg; break;
        case OP_SHFL: {
          const int src = f.src;
          const bool ok = src >= 0 && src < 64 && l0 + src < l1 && S.fibers[l0 + src].state == AT_WAVE_OP;
          f.result = ok ? S.fibers[l0 + src].arg : f.arg;
          break;
        }
        default: break;
        }
      }
      for (size_t l = l0; l < l1; ++l)
        if (S.fibers[l].state == AT_WAVE_OP) S.fibers[l].state = RUNNABLE;
      progressed = true;
      ++ops;
      continue;
    }
    if (at_op != 0 && (at_sync != 0 || asleep != 0))
      die(S, "a wave is split between a cross-lane operation and a barrier / sleep");
    if (asleep == live) { // s_sleep: the wave gives way; it is runnable again on its next turn
      for (size_t l = l0; l < l1; ++l)
        if (S.fibers[l].state == ASLEEP) S.fibers[l].state = RUNNABLE;
      return true;
    }
    if (at_sync == live) return progressed;
    if (asleep != 0 || at_sync != 0) die(S, "a wave is split between a barrier and a sleep / the end of the kernel");
    if (!ran) die(S, "a wave made no progress");
  }
}

// One turn of a workgroup: every wave gets its turn, then a completed barrier is released.  Returns false when nothing could run.
enum TurnResult : int { TURN_DONE = 0, TURN_PROGRESS, TURN_STUCK };
TurnResult block_turn(Sched &S) {
  const size_t n = S.fibers.size(), waves = (n + 63) / 64;
  bool any = false;
  for (size_t w = 0; w < waves; ++w) any |= run_wave(S, w, kTurn);
  size_t live = 0, at_sync = 0;
  for (const FiberImpl &f : S.fibers) {
    if (f.state == DONE) continue;
    ++live;
    if (f.state == AT_SYNC) ++at_sync;
  }
  if (live == 0) return TURN_DONE;
  if (at_sync == live) {
    for (FiberImpl &f : S.fibers)
      if (f.state == AT_SYNC) f.state = RUNNABLE; // (different __syncthreads sites in one barrier are legal on the hardware)
    return TURN_PROGRESS;
  }
  if (at_sync != 0 && !any) die(S, "__syncthreads: some lanes wait at the barrier while the others have left the kernel or cannot reach it");
  return any ? TURN_PROGRESS : TURN_STUCK;
}

void run_block(Sched &S) {
  for (;;) {
    const TurnResult r = block_turn(S);
    if (r == TURN_DONE) return;
    if (r == TURN_STUCK) die(S, "the workgroup made no progress");
  }
}

std::mutex g_launch_mutex;
std::vector<void *> g_stack_pool; // guarded by g_launch_mutex

void block_at(int op, int site, State st) {
  FiberImpl *f = g_sched->running;
  f->op = op;
  f->site = site;
  f->state = st;
  yield_to_scheduler();
}

} // namespace

unsigned long long wave_ballot(int pred, int site) {
  FiberImpl *f = g_sched->running;
  f->arg = (unsigned long long)pred;
  block_at(OP_BALLOT, site, AT_WAVE_OP);
  return f->result;
}
unsigned long long wave_shfl(unsigned long long v, int src_lane, int site) {
  FiberImpl *f = g_sched->running;
  f->arg = v;
  f->src = src_lane;
  block_at(OP_SHFL, site, AT_WAVE_OP);
  return f->result;
}
unsigned long long wave_first(unsigned long long v, int site) {
  FiberImpl *f = g_sched->running;
  f->arg = v;
  block_at(OP_FIRST, site, AT_WAVE_OP);
  return f->result;
}
void wave_barrier(int site) { block_at(OP_WAVE_BARRIER, site, AT_WAVE_OP); }
void block_sync(int site) { block_at(OP_SYNC, site, AT_SYNC); }
void wave_sleep() { block_at(OP_SLEEP, 0, ASLEEP); }
unsigned long long clock_ticks() { // 100 MHz, as the device's wall clock
  return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10ull;
}

// static LDS lives in one linker section (hip_runtime.h: __shared__); resident kernels keep one image of it per workgroup
extern "C" char __start_emu_lds[], __stop_emu_lds[];
// which static belongs to which kernel (generated by build_emu.py's second link pass; absent in the first)
extern "C" {
struct EmuLdsSym { const char *name; unsigned long off, size; };
extern const EmuLdsSym emu_lds_table[] __attribute__((weak));
}

namespace {
struct Block {
  Sched S;
  unsigned char *lds = nullptr;
  std::vector<char> statics; // this workgroup's image of the resident kernel's static LDS (the ranges below, back to back)
  bool done = false;
};
struct Range { size_t off, size; };
// the static LDS variables of the kernel called `name` ("k_trace_server<CAP>" -> every k_trace_server<..>(...)::variable), merged
std::vector<Range> statics_of(const char *name) {
  std::vector<Range> r;
  std::string key(name);
  key = key.substr(0, key.find('<'));
  if (&emu_lds_table[0] != nullptr && !key.empty())
    for (const EmuLdsSym *e = emu_lds_table; e->name; ++e) {
      const std::string n(e->name);
      const size_t at = n.find(key);
      if (at != std::string::npos && (n[at + key.size()] == '<' || n[at + key.size()] == '(') && (at == 0 || n[at - 1] == ':' || n[at - 1] == ' '))
        r.push_back(Range{e->off, e->size});
    }
  if (r.empty()) r.push_back(Range{0, (size_t)(__stop_emu_lds - __start_emu_lds)}); // no table: the whole section
  return r;
}
void copy_in(const std::vector<Range> &rs, const std::vector<char> &img) {
  size_t at = 0;
  for (const Range &r : rs) {
    memcpy(__start_emu_lds + r.off, img.data() + at, r.size);
    at += r.size;
  }
}
void copy_out(const std::vector<Range> &rs, std::vector<char> &img) {
  size_t at = 0;
  for (const Range &r : rs) {
    memcpy(img.data() + at, __start_emu_lds + r.off, r.size);
    at += r.size;
  }
}
std::mutex g_resident_mutex;
std::vector<std::thread> g_resident; // guarded by g_resident_mutex

void take_stacks(Sched &S) {
  for (FiberImpl &f : S.fibers) {
    if (!g_stack_pool.empty()) {
      f.stack = g_stack_pool.back();
      g_stack_pool.pop_back();
    } else {
      f.stack = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (f.stack == MAP_FAILED) {
        perror("emu: mmap of a fiber stack");
        abort();
      }
    }
#if EMU_TSAN
    f.tsan = __tsan_create_fiber(0);
#endif
  }
}
void give_stacks(Sched &S) {
  for (FiberImpl &f : S.fibers) {
    g_stack_pool.push_back(f.stack);
#if EMU_TSAN
    __tsan_destroy_fiber(f.tsan);
#endif
  }
}
void arm(Sched &S, unsigned bx, unsigned by, unsigned bz, dim3 grid, dim3 block, unsigned char *lds) {
  const size_t nthreads = S.fibers.size();
  for (size_t t = 0; t < nthreads; ++t) {
    FiberImpl &f = S.fibers[t];
    f.ctx.thread_idx = Idx{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
    f.ctx.block_idx = Idx{bx, by, bz};
    f.ctx.block_dim = Idx{block.x, block.y, block.z};
    f.ctx.grid_dim = Idx{grid.x, grid.y, grid.z};
    f.ctx.dyn_shared = lds;
    f.ctx.lane = (int)(t & 63);
    f.state = RUNNABLE;
    prepare(f);
  }
}
bool is_resident(const char *name) {
  const char *e = getenv("MGPU_EMU_RESIDENT");
  const std::string pat = e ? e : "k_trace_server";
  return !pat.empty() && std::string(name).find(pat) != std::string::npos;
}

// every workgroup to completion, one after another, on the calling thread
void run_serial(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &lane_entry) {
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  Sched S;
  S.entry = &lane_entry;
  S.fibers.resize(nthreads);
  static unsigned char *lds = nullptr;
  if (!lds) lds = (unsigned char *)aligned_alloc(256, kLdsBytes);
  S.lds = lds;
  take_stacks(S);
#if EMU_TSAN
  S.main_tsan = __tsan_get_current_fiber();
#endif
  Sched *outer = g_sched;
  Ctx *outer_cur = g_cur;
  g_sched = &S;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        memset(lds, 0xCD, shmem); // LDS comes up with whatever the last workgroup left: make reliance on that visible
        // dynamic LDS beyond what the launch asked for does not exist: a canary behind it (and, under AddressSanitizer, poison)
        const size_t guard = std::min<size_t>(kLdsBytes - shmem, 4096);
        memset(lds + shmem, 0x5A, guard);
#if EMU_ASAN
        __asan_poison_memory_region(lds + shmem, kLdsBytes - shmem);
#endif
        arm(S, bx, by, bz, grid, block, lds);
        run_block(S);
#if EMU_ASAN
        __asan_unpoison_memory_region(lds + shmem, kLdsBytes - shmem);
#endif
        for (size_t i = 0; i < guard; ++i)
          if (lds[shmem + i] != 0x5A) {
            fprintf(stderr, "emu: a kernel wrote dynamic LDS byte %zu of a launch that asked for %zu bytes (block %zu threads)\n", shmem + i, shmem, nthreads);
            abort();
          }
      }
  g_sched = outer;
  g_cur = outer_cur;
  give_stacks(S);
}

// all workgroups alive, taking turns; static LDS swapped with the workgroup
void run_resident(const char *name, dim3 grid, dim3 block, size_t shmem, const std::function<void()> &lane_entry) {
  const size_t nthreads = (size_t)block.x * block.y * block.z, nblocks = (size_t)grid.x * grid.y * grid.z;
  const std::vector<Range> ranges = statics_of(name);
  size_t nstat = 0;
  for (const Range &r : ranges) nstat += r.size;
  std::vector<Block> blocks(nblocks);
  size_t i = 0;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx, ++i) {
        Block &b = blocks[i];
        b.S.entry = &lane_entry;
        b.S.fibers.resize(nthreads);
        b.lds = (unsigned char *)aligned_alloc(256, (shmem + 255 + 4096) & ~(size_t)255);
        memset(b.lds, 0xCD, shmem);
        take_stacks(b.S);
#if EMU_TSAN
        b.S.main_tsan = __tsan_get_current_fiber();
#endif
        arm(b.S, bx, by, bz, grid, block, b.lds);
        b.statics.resize(nstat);
        copy_out(ranges, b.statics);
      }
  Sched *outer = g_sched;
  Ctx *outer_cur = g_cur;
  unsigned long long rounds = 0;
  for (;;) {
    ++rounds;
    bool live = false, progress = false;
    for (Block &b : blocks) {
      if (b.done) continue;
      copy_in(ranges, b.statics);
      g_sched = &b.S;
      const TurnResult r = block_turn(b.S);
      copy_out(ranges, b.statics);
      if (r == TURN_DONE) b.done = true;
      else live = true;
      if (r != TURN_STUCK) progress = true;
    }
    if (!live) break;
    if (!progress) die(blocks[0].S, "a resident kernel made no progress in any workgroup");
  }
  g_sched = outer;
  g_cur = outer_cur;
  if (getenv("MGPU_EMU_TRACE")) fprintf(stderr, "emu: resident kernel %s left after %llu rounds (%zu bytes of static LDS per workgroup)\n", name, rounds, nstat);
  for (Block &b : blocks) {
    give_stacks(b.S);
    free(b.lds);
  }
}
} // namespace

void launch(const char *name, dim3 grid, dim3 block, size_t shmem, const std::function<void()> &lane_entry,
            const std::function<std::function<void()>()> &owning_copy) {
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024 || shmem > kLdsBytes) {
    fprintf(stderr, "emu: bad launch of %s (block %zu threads, %zu bytes of dynamic LDS)\n", name, nthreads, shmem);
    abort();
  }
  static const bool trace = getenv("MGPU_EMU_TRACE") != nullptr;
  const bool resident = is_resident(name);
  if (trace) fprintf(stderr, "emu: launch %s grid %u x %u x %u, block %zu, dynamic LDS %zu%s\n", name, grid.x, grid.y, grid.z, nthreads, shmem, resident ? " (resident)" : "");
  if (!resident) {
    std::lock_guard<std::mutex> lock(g_launch_mutex);
    run_serial(grid, block, shmem, lane_entry);
    return;
  }
  std::lock_guard<std::mutex> rl(g_resident_mutex);
  const std::string kname(name);
  const std::function<void()> owned = owning_copy();
  g_resident.emplace_back([=]() {
    std::lock_guard<std::mutex> lock(g_launch_mutex); // one kernel at a time: the static LDS section is the process's
    run_resident(kname.c_str(), grid, block, shmem, owned);
  });
}

} // namespace emu
namespace isa {
bool run(const char *mangled, dim3 grid, dim3 block, size_t shmem, const void *kernarg, size_t kernarg_bytes);
}
namespace emu {
// name of a function with INTERNAL linkage (kernels in an anonymous namespace are not in the dynamic symbol table dladdr reads): from the
// library file's own .symtab
static const char *local_symbol_name(const Dl_info &info, const void *fn) {
  static std::mutex mu;
  static std::map<std::string, std::map<uintptr_t, std::string>> tables;
  std::lock_guard<std::mutex> lk(mu);
  if (!info.dli_fname) return nullptr;
  auto it = tables.find(info.dli_fname);
  if (it == tables.end()) {
    std::map<uintptr_t, std::string> tab;
    FILE *f = fopen(info.dli_fname, "rb");
    if (f) {
      std::vector<unsigned char> img;
      fseek(f, 0, SEEK_END);
      const long n = ftell(f);
      fseek(f, 0, SEEK_SET);
      img.resize(n > 0 ? (size_t)n : 0);
      if (n > 0 && fread(img.data(), 1, (size_t)n, f) == (size_t)n && n > (long)sizeof(Elf64_Ehdr)) {
        const Elf64_Ehdr *eh = reinterpret_cast<const Elf64_Ehdr *>(img.data());
        const Elf64_Shdr *sh = reinterpret_cast<const Elf64_Shdr *>(img.data() + eh->e_shoff);
        for (int i = 0; i < eh->e_shnum; ++i)
          if (sh[i].sh_type == SHT_SYMTAB) {
            const Elf64_Sym *sym = reinterpret_cast<const Elf64_Sym *>(img.data() + sh[i].sh_offset);
            const char *str = reinterpret_cast<const char *>(img.data() + sh[sh[i].sh_link].sh_offset);
            for (size_t k = 0; k < sh[i].sh_size / sizeof(Elf64_Sym); ++k)
              if (ELF64_ST_TYPE(sym[k].st_info) == STT_FUNC && sym[k].st_value) tab[(uintptr_t)sym[k].st_value] = str + sym[k].st_name;
          }
      }
      fclose(f);
    }
    it = tables.emplace(info.dli_fname, std::move(tab)).first;
  }
  auto jt = it->second.find((uintptr_t)fn - (uintptr_t)info.dli_fbase);
  return jt == it->second.end() ? nullptr : jt->second.c_str();
}

bool launch_isa(const void *fn, dim3 grid, dim3 block, size_t shmem, const void *kernarg, size_t kernarg_bytes) {
  Dl_info info;
  if (!dladdr(fn, &info)) return false;
  if (!info.dli_sname) info.dli_sname = local_symbol_name(info, fn);
  if (!info.dli_sname) return false;
  if (is_resident(info.dli_sname)) return false; // resident kernels run beside the host on a thread of their own: the C++ path's business
  static const bool trace = getenv("MGPU_EMU_TRACE") != nullptr;
  std::lock_guard<std::mutex> lock(g_launch_mutex); // one kernel at a time, like the C++ path
  const bool ran = isa::run(info.dli_sname, grid, block, shmem, kernarg, kernarg_bytes);
  if (trace) fprintf(stderr, "emu: %s %s grid %u block %u, dynamic LDS %zu\n", ran ? "ISA launch of" : "no ISA for", info.dli_sname, grid.x, block.x, shmem);
  return ran;
}

void wait_resident() {
  std::vector<std::thread> mine;
  {
    std::lock_guard<std::mutex> rl(g_resident_mutex);
    mine.swap(g_resident);
  }
  for (std::thread &t : mine) t.join();
}

} // namespace emu

extern "C" { unsigned long long emu_stats[64]; unsigned char *emu_log = nullptr; unsigned long long emu_log_n = 0, emu_log_cap = 0; } // scratch counters of -DMGPU_EMU_STATS builds (read by the tests through ctypes)

// ---- runtime API ---------------------------------------------------------------------------------------------------------------------
struct emuStream { int dummy; };
struct emuEvent { std::chrono::steady_clock::time_point t; bool recorded; };

static thread_local int g_device = 0;
static int emu_devices() {
  const char *e = getenv("MGPU_EMU_DEVICES");
  const int n = e ? atoi(e) : 1;
  return n < 1 ? 1 : n;
}
hipError_t hipMalloc(void **p, size_t bytes) {
  *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, 0xA5, bytes); // device memory comes up dirty
  return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
  *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, bytes, k); }
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; ++r) memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t) {
  return hipMemcpy2D(dst, dpitch, src, spitch, width, height, k);
}
hipError_t hipMemset(void *p, int v, size_t bytes) { if (bytes) memset(p, v, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t) { return hipMemset(p, v, bytes); }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)16 << 30; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu_devices()) return hipErrorInvalidValue; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = emu_devices(); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "wave64 emulator (tests/emu)");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
  const char *e = getenv("MGPU_EMU_CUS");
  p->multiProcessorCount = e && atoi(e) > 0 ? atoi(e) : 1;
  p->totalGlobalMem = (size_t)16 << 30;
  p->sharedMemPerBlock = 160 * 1024;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() { emu::wait_resident(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new emuStream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { emu::wait_resident(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emuEvent{std::chrono::steady_clock::now(), false}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
