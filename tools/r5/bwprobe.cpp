// round 5: what the box streams -- read-only, copy and three-array kernels with 16-byte accesses, HIP-event timed.
//   hipcc -O2 --offload-arch=gfx950 -o bwprobe bwprobe.cpp && ./bwprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* out, size_t n) {
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1234.5f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_copy_nt(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { typedef float v4 __attribute__((ext_vector_type(4))); const float4 t = a[i]; v4 w = {t.x, t.y, t.z, t.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4*>(b) + i); }
}
__global__ void __launch_bounds__(256) k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + 2.0f * y.x, x.y + 2.0f * y.y, x.z + 2.0f * y.z, x.w + 2.0f * y.w);
  }
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

// kernel B's store pattern: 512 blocks of 512 threads, a block owns a 64 x 16 tile of a 512 x 512 plane and marches 128
// planes; per plane it writes its tile of four arrays (D'x, D'y, D'z, M o D').  W = 2: a thread owns two x-neighbours and
// issues four 8-byte stores (today); W = 4: the lanes of a pair have exchanged halves, each issues two 16-byte stores (even
// lanes arrays 0 and 1, odd lanes arrays 2 and 3).
template <int W>
__global__ void __launch_bounds__(512) k_store_march(float* __restrict__ base, size_t array_floats, int nz) {
  const int tile = blockIdx.x & 255, chunk = blockIdx.x >> 8;            // 8 x 32 tiles, 2 chunks
  const int tx = tile & 7, ty = tile >> 3;
  const int lane2 = threadIdx.x & 31, row = threadIdx.x >> 5;              // 32 threads x 2 voxels = 64 columns, 16 rows
  const size_t plane = 512 * 512;
  for (int z = chunk * (nz / 2); z < (chunk + 1) * (nz / 2); ++z) {
    const size_t rowoff = (size_t)z * plane + (size_t)(ty * 16 + row) * 512 + tx * 64;
    if (W == 2) {
      const float2 v = make_float2((float)z, 1.0f);
#pragma unroll
      for (int a = 0; a < 4; ++a) *reinterpret_cast<float2*>(base + a * array_floats + rowoff + 2 * lane2) = v;
    } else {
      const float4 v = make_float4((float)z, 1.0f, 2.0f, 3.0f);
      const int pair = lane2 >> 1, odd = lane2 & 1;
#pragma unroll
      for (int a = 0; a < 2; ++a) *reinterpret_cast<float4*>(base + (size_t)(2 * odd + a) * array_floats + rowoff + 4 * pair) = v;
    }
  }
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 1024;   // MB per array
  const size_t n = mb * 1024 * 1024 / 16;
  float4 *a, *b, *c; float* out;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int blocks : {2048, 8192, 65536}) {
    auto run = [&](const char* name, double bytes, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("%-28s %6d blocks  %8.3f ms  %7.0f GB/s\n", name, blocks, ms, bytes / (ms * 1e-3) / 1e9);
    };
    run("read (sum)", (double)n * 16, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, out, n); });
    run("write", (double)n * 16, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n); });
    run("copy (1 read + 1 write)", (double)n * 32, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    run("copy, non-temporal store", (double)n * 32, [&] { hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    run("triad (2 reads + 1 write)", (double)n * 48, [&] { hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, 0, a, b, c, n); });
  }
  {
    const int nz = 256;
    const size_t af = (size_t)512 * 512 * nz;
    float* big;
    CK(hipMalloc(&big, 4 * af * sizeof(float)));
    CK(hipMemset(big, 0, 4 * af * sizeof(float)));
    for (int rep = 0; rep < 2; ++rep)
      for (int w : {2, 4}) {
        auto launch = [&] {
          if (w == 2) hipLaunchKernelGGL(k_store_march<2>, dim3(512), dim3(512), 0, 0, big, af, nz);
          else hipLaunchKernelGGL(k_store_march<4>, dim3(512), dim3(512), 0, 0, big, af, nz);
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 10; ++r) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("marching stores of kernel B's shape, %2d-byte stores: %8.3f ms  %7.0f GB/s (16 B/voxel written)\n", 4 * w, ms, 4.0 * af * 4 / (ms * 1e-3) / 1e9);
      }
  }
  return 0;
}
