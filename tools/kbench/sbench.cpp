// tools/kbench/sbench.cpp -- stand-alone kernel throughput (measurement tooling, not product): times C-ABI calls of a
// build of libplatipy_hip.so on a synthetic nx x ny x nz volume with HIP events and prints algorithmic GB/s.
//   sbench <lib.so> nx ny nz [reps]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "../../include/platipy_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_fill(float* a, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    a[i] = scale * ((float)(x & 0xffffff) / 16777216.0f - 0.5f);
  }
}
// smooth displacement field (what registration produces): a few mm, varying over tens of voxels
__global__ void k_fill_smooth(float* d, int nx, int ny, int nz, float amp) {
  const size_t N = (size_t)nx * ny * nz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
    d[i] = amp * __sinf(0.013f * y + 0.021f * z + 0.007f * x);
    d[N + i] = 0.8f * amp * __cosf(0.017f * x + 0.009f * z);
    d[2 * N + i] = 0.6f * amp * __sinf(0.015f * x + 0.019f * y);
  }
}
__global__ void k_fill_u8(unsigned char* a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (i / 7) & 1;
}

template <typename T> T sym(void* h, const char* n) { void* p = dlsym(h, n); if (!p) { fprintf(stderr, "missing %s\n", n); exit(2); } return reinterpret_cast<T>(p); }

int main(int argc, char** argv) {
  if (argc < 5) return 1;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  const int nx = atoi(argv[2]), ny = atoi(argv[3]), nz = atoi(argv[4]), reps = argc > 5 ? atoi(argv[5]) : 5;
  const size_t N = (size_t)nx * ny * nz;
  auto create = sym<int (*)(int, void*, pp_ctx**)>(h, "pp_create");
  auto last_error = sym<const char* (*)(const pp_ctx*)>(h, "pp_last_error");
  auto sync = sym<int (*)(pp_ctx*)>(h, "pp_sync");
  auto smooth_field = sym<int (*)(pp_ctx*, float*, const int*, const double*, double, int)>(h, "pp_smooth_field_f32");
  auto dgauss = sym<int (*)(pp_ctx*, const float*, float*, const int*, const double*, const double*, double, int, int)>(h, "pp_discrete_gaussian_f32");
  auto warp = sym<int (*)(pp_ctx*, const float*, const float*, const pp_geom*, float, float*)>(h, "pp_warp_f32");
  auto resample = sym<int (*)(pp_ctx*, const float*, const pp_geom*, const pp_geom*, const double*, const double*, const float*, int, double, float*)>(h, "pp_resample_f32");
  auto resample_u8 = sym<int (*)(pp_ctx*, const uint8_t*, const pp_geom*, const pp_geom*, const double*, const double*, const float*, int, double, uint8_t*)>(h, "pp_resample_u8");
  auto compose = sym<int (*)(pp_ctx*, float*, const float*, const pp_geom*)>(h, "pp_compose_field_f32");
  auto rgauss = sym<int (*)(pp_ctx*, float*, const pp_geom*, const double*)>(h, "pp_recursive_gaussian_field_f32");
  auto fuse_acc = sym<int (*)(pp_ctx*, const float*, const uint8_t*, float*, float*, size_t)>(h, "pp_fuse_accumulate_u8");
  auto fuse_div = sym<int (*)(pp_ctx*, const float*, const float*, float*, size_t)>(h, "pp_fuse_divide_f32");
  auto wlocal = sym<int (*)(pp_ctx*, const float*, const float*, const int*, const double*, double, double, float*)>(h, "pp_weight_map_local_f32");
  auto defaults = sym<void (*)(pp_demons_params*)>(h, "pp_demons_default_params");
  auto execute = sym<int (*)(pp_ctx*, const float*, const float*, const pp_geom*, const pp_demons_params*, float*, pp_demons_stats*)>(h, "pp_demons_execute_f32");

  float *A, *B, *D, *D2;
  unsigned char *L, *L2;
  CK(hipMalloc(&A, N * 4)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&D, 3 * N * 4)); CK(hipMalloc(&D2, 3 * N * 4)); CK(hipMalloc(&L, N)); CK(hipMalloc(&L2, N));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, N, 1000.0f, 1u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B, N, 1000.0f, 2u);
  hipLaunchKernelGGL(k_fill_smooth, dim3(4096), dim3(256), 0, 0, D, nx, ny, nz, 4.0f);
  hipLaunchKernelGGL(k_fill_smooth, dim3(4096), dim3(256), 0, 0, D2, nx, ny, nz, 1.5f);
  hipLaunchKernelGGL(k_fill_u8, dim3(4096), dim3(256), 0, 0, L, N);
  CK(hipDeviceSynchronize());
  hipStream_t st;
  CK(hipStreamCreate(&st));
  pp_ctx* ctx = nullptr;
  if (create(0, st, &ctx)) return 2;
  pp_geom g;
  g.size[0] = nx; g.size[1] = ny; g.size[2] = nz;
  for (int i = 0; i < 3; ++i) { g.spacing[i] = 1.0; g.origin[i] = 0.0; }
  for (int i = 0; i < 9; ++i) g.direction[i] = (i % 4 == 0) ? 1.0 : 0.0;
  const int size[3] = {nx, ny, nz};
  const double sp[3] = {1, 1, 1};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes_per_voxel, const std::function<int()>& f) {
    if (f()) { printf("%-44s FAILED: %s\n", name, last_error(ctx)); return; }
    sync(ctx);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-44s %8.3f ms  %7.0f GB/s (%.0f B/voxel algorithmic)\n", name, ms, bytes_per_voxel * N / (ms * 1e-3) / 1e9, bytes_per_voxel);
    fflush(stdout);
  };
  const double s15[3] = {1.5, 1.5, 1.5}, v1[3] = {1, 1, 1}, v64[3] = {64, 64, 64}, v16[3] = {16, 16, 16};
  run("smooth_field sigma 1.5 vox (9 passes, r=2)", 72, [&] { return smooth_field(ctx, D, size, s15, 0.1, 30); });
  run("discrete_gaussian var 1 (3 passes, r=3)", 24, [&] { return dgauss(ctx, A, B, size, sp, v1, 0.01, 32, 1); });
  run("discrete_gaussian var 16 (3 passes, r~13)", 24, [&] { return dgauss(ctx, A, B, size, sp, v16, 0.01, 128, 1); });
  run("discrete_gaussian var 64 width<=32 (r=16)", 24, [&] { return dgauss(ctx, A, B, size, sp, v64, 0.01, 32, 1); });
  run("warp (same grid, linear)", 20, [&] { return warp(ctx, A, D, &g, -1000.0f, B); });
  run("resample f32 linear through field", 20, [&] { return resample(ctx, A, &g, &g, nullptr, nullptr, D, 2, -1000.0, B); });
  run("resample u8 nearest through field", 14, [&] { return resample_u8(ctx, L, &g, &g, nullptr, nullptr, D, 1, 0.0, L2); });
  {
    const double ang = 0.05, Aa[9] = {cos(ang), -sin(ang), 0.0, sin(ang), cos(ang), 0.0, 0.0, 0.0, 1.02}, ta[3] = {3.0, -2.0, 1.5};
    run("resample f32 linear through an affine", 8, [&] { return resample(ctx, A, &g, &g, Aa, ta, nullptr, 2, -1000.0, B); });
    run("resample u8 nearest through an affine", 2, [&] { return resample_u8(ctx, L, &g, &g, Aa, ta, nullptr, 1, 0.0, L2); });
  }
  run("resample f32 linear identity", 8, [&] { return resample(ctx, A, &g, &g, nullptr, nullptr, nullptr, 2, 0.0, B); });
  {
    auto resample_field = sym<int (*)(pp_ctx*, const float*, const pp_geom*, const pp_geom*, float*)>(h, "pp_resample_field_f32");
    pp_geom gc = g;
    gc.size[0] = nx / 4; gc.size[1] = ny / 4; gc.size[2] = nz / 4;
    for (int i = 0; i < 3; ++i) { gc.spacing[i] = 4.0; gc.origin[i] = 1.5; }
    // (the coarse field: the first nx/4 * ny/4 * nz/4 * 3 floats of D2, any values)
    run("resample_field x4 up-sampling (12 B written)", 12, [&] { return resample_field(ctx, D2, &gc, &g, D); });
    hipLaunchKernelGGL(k_fill_smooth, dim3(4096), dim3(256), 0, st, D, nx, ny, nz, 4.0f);
  }
  run("compose_field", 36, [&] { return compose(ctx, D, D2, &g); });
  run("recursive_gaussian_field sigma 1.5", 72, [&] { return rgauss(ctx, D, &g, s15); });
  run("weight_map_local sigma 2", 40, [&] { return wlocal(ctx, A, B, size, sp, 2.0, 1e-5, D); });
  run("fuse_accumulate_u8", 21, [&] { return fuse_acc(ctx, A, L, B, D, N); });
  run("fuse_divide", 12, [&] { return fuse_div(ctx, A, B, D, N); });
  pp_demons_params p;
  defaults(&p);
  p.iterations = 4; p.smooth_update = 1; p.smooth_displacement = 1; p.max_rms_error = 0.0;
  for (int i = 0; i < 3; ++i) p.sigma_d_vox[i] = 1.5;
  p.variant = PP_DEMONS_STAGED;
  run("demons staged, 4 iterations", 4 * 196, [&] { return execute(ctx, A, B, &g, &p, D, nullptr); });
  p.variant = PP_DEMONS_FUSED;
  run("demons fused, 4 iterations (compulsory 64)", 4 * 64, [&] { return execute(ctx, A, B, &g, &p, D, nullptr); });
  p.iterations = 5;
  run("demons fused, 5 iterations (odd: + copy)", 5 * 64 + 24, [&] { return execute(ctx, A, B, &g, &p, D, nullptr); });
  return 0;
}
