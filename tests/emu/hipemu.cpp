// tests/emu/hipemu.cpp -- TEST INFRASTRUCTURE ONLY: block/thread scheduler of the CPU HIP
// stand-in (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

namespace hipemu {

thread_local dim3 t_threadIdx, t_blockIdx;
dim3 g_blockDim, g_gridDim;
pthread_barrier_t* g_barrier = nullptr;

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
  if (nthreads == 0 || nblocks == 0) return;
  g_blockDim = block;
  g_gridDim = grid;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, nthreads);
  g_barrier = &bar;
  std::vector<std::thread> pool;
  pool.reserve(nthreads);
  for (unsigned t = 0; t < nthreads; ++t) {
    pool.emplace_back([=, &bar, &body]() {
      t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      for (unsigned long b = 0; b < nblocks; ++b) {
        t_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y),
                          (unsigned)(b / ((unsigned long)grid.x * grid.y)));
        body();
        pthread_barrier_wait(&bar);  // no thread enters the next block while LDS is in use
      }
    });
  }
  for (auto& th : pool) th.join();
  pthread_barrier_destroy(&bar);
  g_barrier = nullptr;
}

}  // namespace hipemu
