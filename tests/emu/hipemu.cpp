// tests/emu/hipemu.cpp -- TEST INFRASTRUCTURE ONLY: block/thread scheduler of the CPU HIP stand-in
// (see include/hip/hip_runtime.h).
//
// A thread block runs as blockDim cooperative fibers (ucontext) on ONE OS thread: a fiber runs until
// it reaches __syncthreads() (or returns), then the next fiber runs; when every live fiber has
// arrived the round restarts.  Blocks are independent, so a pool of OS threads executes different
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
#if defined(__x86_64__)
  void* sched_sp = nullptr;
#else
  ucontext_t sched;
#endif
  const std::function<void()>* body = nullptr;
};

thread_local Worker* t_worker = nullptr;

void fiber_entry() {
  Fiber* f = t_current;
  (*t_worker->body)();
  f->done = true;
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
  unsigned live = n;
  while (live) {
    // one round: every live fiber runs to its next barrier (or to the end)
    for (unsigned t = 0; t < n; ++t) {
      Fiber& f = w.fibers[t];
      if (f.done) continue;
      t_current = &f;
#if defined(__x86_64__)
      hipemu_switch(&w.sched_sp, f.sp);
#else
      swapcontext(&w.sched, &f.ctx);
#endif
      if (f.done) --live;
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
