// tests/emu/hipemu.cpp -- TEST INFRASTRUCTURE ONLY: block/thread scheduler of the CPU HIP stand-in
// (see include/hip/hip_runtime.h).
//
// A thread block runs as blockDim cooperative fibers (ucontext) on ONE OS thread: a fiber runs until
// it waits (at __syncthreads() or at a wavefront rendezvous inside a shuffle / vote) or returns, then the
// next fiber runs; a waiting fiber yields again each round until the barrier it waits at is complete.  Blocks are independent, so a pool of OS threads executes different
// blocks in parallel; `__shared__` is `static thread_local`, i.e. private to the OS thread and thus
// to the block it is currently running.
//
// On x86-64 the fibers switch with a dozen instructions (callee-saved registers + stack pointer); glibc's swapcontext
// makes a signal-mask system call per switch, which dominated the suite once the kernels used wavefront shuffles (two
// yields each).  Other hosts keep ucontext.
#include <hip/hip_runtime.h>
#include <ucontext.h>

#if defined(__x86_64__)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch, @function
hipemu_switch:
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
    .size hipemu_switch, .-hipemu_switch
)");
#endif

#include <condition_variable>
#include <mutex>

namespace hipemu {

thread_local Fiber* t_current = nullptr;
thread_local double t_shfl[1024];
dim3 g_blockDim, g_gridDim;

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;

struct Worker {
  std::vector<Fiber> fibers;
  std::vector<char*> stacks;
  // barrier state of the block being run: block-wide (__syncthreads) and per wavefront (shuffles, votes)
  unsigned live = 0, bar_arrived = 0, bar_gen = 0;
  unsigned wave_live[32] = {}, wave_arrived[32] = {}, wave_gen[32] = {};
#if defined(__x86_64__)
  void* sched_sp = nullptr;
#else
  ucontext_t sched;
#endif
  const std::function<void()>* body = nullptr;
};

thread_local Worker* t_worker = nullptr;

unsigned linear_tid(const Fiber* f) { return f->tid.x + g_blockDim.x * (f->tid.y + g_blockDim.y * f->tid.z); }

void fiber_entry() {
  Fiber* f = t_current;
  (*t_worker->body)();
  f->done = true;
  {   // a thread that has returned no longer takes part in barriers: release those its exit completes
    Worker& w = *t_worker;
    const unsigned wv = linear_tid(f) >> 6;
    --w.live;
    --w.wave_live[wv];
    if (w.bar_arrived && w.bar_arrived == w.live) {
      w.bar_arrived = 0;
      ++w.bar_gen;
    }
    if (w.wave_arrived[wv] && w.wave_arrived[wv] == w.wave_live[wv]) {
      w.wave_arrived[wv] = 0;
      ++w.wave_gen[wv];
    }
  }
#if defined(__x86_64__)
  hipemu_switch(&f->sp, t_worker->sched_sp);   // never resumed
  __builtin_trap();
#else
  swapcontext(&f->ctx, &t_worker->sched);
#endif
}

void run_block(Worker& w, dim3 grid, dim3 block, unsigned long b, const std::function<void()>& body) {
  const unsigned n = block.x * block.y * block.z;
  if (w.fibers.size() < n) w.fibers.resize(n);
  while (w.stacks.size() < n) w.stacks.push_back(static_cast<char*>(malloc(STACK_BYTES)));
  w.body = &body;
  t_worker = &w;
  const dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long)grid.x * grid.y)));
  for (unsigned t = 0; t < n; ++t) {
    Fiber& f = w.fibers[t];
    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    f.bid = bidx;
    f.done = false;
#if defined(__x86_64__)
    // initial frame: six zeroed callee-saved registers, then the entry point as hipemu_switch's return address, then a
    // null "return address" of the entry point itself (it never returns), so that it starts with rsp = 16 n + 8 as the
    // ABI has it at any function entry.
    void** sp = reinterpret_cast<void**>(reinterpret_cast<uintptr_t>(w.stacks[t] + STACK_BYTES) & ~(uintptr_t)15);
    *--sp = nullptr;
    *--sp = reinterpret_cast<void*>(&fiber_entry);
    for (int k = 0; k < 6; ++k) *--sp = nullptr;
    f.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = w.stacks[t];
    f.ctx.uc_stack.ss_size = STACK_BYTES;
    f.ctx.uc_link = &w.sched;
    makecontext(&f.ctx, fiber_entry, 0);
#endif
  }
  w.live = n;
  w.bar_arrived = 0;
  for (unsigned k = 0; k < 32; ++k) {
    w.wave_arrived[k] = 0;
    w.wave_live[k] = (64 * k < n) ? ((n - 64 * k < 64) ? n - 64 * k : 64) : 0;
  }
  while (w.live) {
    // one round: every live fiber runs until it next yields (waiting at a barrier, it yields again until released)
    for (unsigned t = 0; t < n; ++t) {
      Fiber& f = w.fibers[t];
      if (f.done) continue;
      t_current = &f;
#if defined(__x86_64__)
      hipemu_switch(&w.sched_sp, f.sp);
#else
      swapcontext(&w.sched, &f.ctx);
#endif
    }
  }
  t_current = nullptr;
}

}  // namespace

void fiber_yield() {
  Fiber* f = t_current;
#if defined(__x86_64__)
  hipemu_switch(&f->sp, t_worker->sched_sp);
#else
  swapcontext(&f->ctx, &t_worker->sched);
#endif
}

// __syncthreads(): wait until every live thread of the block has arrived.
void block_barrier() {
  Worker& w = *t_worker;
  const unsigned g = w.bar_gen;
  if (++w.bar_arrived == w.live) {
    w.bar_arrived = 0;
    ++w.bar_gen;
    return;
  }
  while (w.bar_gen == g) fiber_yield();
}

// Rendezvous of the live lanes of the calling thread's wavefront (shuffles / votes: every lane of the WAVEFRONT must reach
// the call; other wavefronts of the block may be elsewhere, as on the hardware).
void wave_barrier() {
  Worker& w = *t_worker;
  const unsigned wv = linear_tid(t_current) >> 6;
  const unsigned g = w.wave_gen[wv];
  if (++w.wave_arrived[wv] == w.wave_live[wv]) {
    w.wave_arrived[wv] = 0;
    ++w.wave_gen[wv];
    return;
  }
  while (w.wave_gen[wv] == g) fiber_yield();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
  if (nthreads == 0 || nblocks == 0) return;
  g_blockDim = block;
  g_gridDim = grid;
  unsigned nworkers = std::thread::hardware_concurrency();
  if (nworkers < 1) nworkers = 1;
  if (nworkers > 16) nworkers = 16;
  if (nworkers > nblocks) nworkers = (unsigned)nblocks;
  std::atomic<unsigned long> next{0};
  auto work = [&]() {
    static thread_local Worker w;  // stacks are reused across launches on pooled threads; fresh threads allocate
    for (;;) {
      const unsigned long b = next.fetch_add(1);
      if (b >= nblocks) break;
      run_block(w, grid, block, b, body);
    }
    for (char* s : w.stacks) free(s);
    w.stacks.clear();
    w.fibers.clear();
  };
  if (nworkers == 1) {
    work();
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(nworkers);
  for (unsigned i = 0; i < nworkers; ++i) pool.emplace_back(work);
  for (auto& th : pool) th.join();
}

}  // namespace hipemu
