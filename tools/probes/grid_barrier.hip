// tools/probes/grid_barrier.hip -- what does a grid-wide barrier cost on MI355X, against a kernel boundary?  (VERDICT round 5, item 7b:
// a persistent per-level demons kernel would replace the two dependent launches of an iteration by two grid barriers.)
//   hipcc -O2 --offload-arch=gfx950 -o tools/probes/grid_barrier tools/probes/grid_barrier.hip && tools/probes/grid_barrier
// (a) K barriers inside ONE launch of B co-resident blocks of 512 threads: sense-reversing counter at agent scope, thread 0 of a
//     block arrives and spins (s_sleep), the block joins behind __syncthreads; microseconds per barrier;
// (b) K empty kernels of the same shape launched back to back on one stream: microseconds per dependent launch.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void __launch_bounds__(512) k_barriers(unsigned* counter, unsigned* sense, int rounds, float* sink) {
  float acc = (float)threadIdx.x;
  unsigned local = 0;
  for (int r = 0; r < rounds; ++r) {
    acc = acc * 1.0001f + 1.0f;   // (something between the barriers)
    __syncthreads();
    if (threadIdx.x == 0) {
      local ^= 1u;
      __threadfence();
      if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sense, local, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(sense, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (acc == -1.0f) sink[0] = acc;
}
__global__ void __launch_bounds__(512) k_empty(float* sink) {
  if (threadIdx.x == 1023) sink[0] = 1.0f;
}

int main() {
  unsigned* words;
  float* sink;
  hipMalloc(&words, 256);
  hipMalloc(&sink, 4);
  hipStream_t s;
  hipStreamCreate(&s);
  const int rounds = 2000;
  for (int blocks : {16, 32, 66, 132, 264, 495}) {
    hipMemsetAsync(words, 0, 256, s);
    hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(512), 0, s, words, words + 32, 10, sink);   // warm-up
    hipStreamSynchronize(s);
    hipMemsetAsync(words, 0, 256, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(512), 0, s, words, words + 32, rounds, sink);
    hipStreamSynchronize(s);
    const double bar_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(512), 0, s, sink);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < rounds; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(512), 0, s, sink);
    hipStreamSynchronize(s);
    const double launch_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    printf("%3d blocks x 512 threads: grid barrier %.2f us each; empty dependent launch %.2f us each\n", blocks, bar_us, launch_us);
  }
  return 0;
}
