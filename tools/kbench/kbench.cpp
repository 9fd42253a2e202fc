// tools/kbench/kbench.cpp -- kernel A/B harness for the fused demons iteration (measurement tooling, not product).
//
// dlopens a build of libplatipy_hip.so given on the command line, runs pp_demons_execute_f32 on a synthetic
// nx x ny x nz pair for `iters` iterations with per-kernel HIP-event profiling on, and prints one line per
// configuration: per-kernel mean launch time, whole-iteration time, Mvoxel/s and a checksum of the field (equal
// checksums across builds = bit-identical fields).  Environment knobs of the library (PP_FUSED_GEN, PP_FUSED_TILE,
// PP_FUSED_ZCHUNK, ...) are read per call, so one process sweeps them:
//   kbench <lib.so> nx ny nz iters "ENV1=a,ENV2=b" ["ENV1=c" ...]
// Build: hipcc -O2 --offload-arch=gfx950 -o kbench kbench.cpp -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/platipy_amd.h"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

__device__ float hash01(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(x & 0xffffff) / 16777216.0f;
}
__device__ float scene(float x, float y, float z) {
  return 300.0f * __sinf(0.045f * x) * __cosf(0.037f * y) * __sinf(0.051f * z + 0.3f) + 150.0f * __cosf(0.011f * (x + y + z)) - 200.0f;
}
__global__ void k_init(float* F, float* M, int nx, int ny, int nz) {
  const size_t N = (size_t)nx * ny * nz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
    const float dx = 2.5f * __sinf(0.013f * y + 0.021f * z), dy = 2.0f * __cosf(0.017f * x + 0.009f * z), dz = 1.5f * __sinf(0.015f * x + 0.019f * y);
    F[i] = scene((float)x, (float)y, (float)z) + 10.0f * (hash01((unsigned)i) - 0.5f);
    M[i] = scene(x + dx, y + dy, z + dz) + 10.0f * (hash01((unsigned)i * 2654435761u + 17u) - 0.5f);
  }
}
__global__ void k_checksum(const float* a, size_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += (unsigned long long)__float_as_uint(a[i]) * (unsigned long long)((i % 1000003u) + 1u);
  atomicAdd(out, s);
}

// PMC calibration kernels with known byte counts (MI355X_MICROARCH.md: calibrate FETCH_SIZE / WRITE_SIZE on your own
// access widths): a 16 B/lane copy and a 4 B/lane copy of n floats (reads 4n bytes, writes 4n bytes each).
__global__ void k_cal_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_cal_copy4(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <typename T>
T sym(void* h, const char* name) {
  void* p = dlsym(h, name);
  if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
  return reinterpret_cast<T>(p);
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: kbench lib nx ny nz iters env-spec...\n"); return 1; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  auto create = sym<int (*)(int, void*, pp_ctx**)>(h, "pp_create");
  auto last_error = sym<const char* (*)(const pp_ctx*)>(h, "pp_last_error");
  auto defaults = sym<void (*)(pp_demons_params*)>(h, "pp_demons_default_params");
  auto execute = sym<int (*)(pp_ctx*, const float*, const float*, const pp_geom*, const pp_demons_params*, float*, pp_demons_stats*)>(h, "pp_demons_execute_f32");
  auto prof_enable = sym<int (*)(pp_ctx*, int)>(h, "pp_profile_enable");
  auto prof_read = sym<int (*)(pp_ctx*, pp_profile_entry*, int)>(h, "pp_profile_read");
  auto sync = sym<int (*)(pp_ctx*)>(h, "pp_sync");

  const int nx = atoi(argv[2]), ny = atoi(argv[3]), nz = atoi(argv[4]), iters = atoi(argv[5]);
  const size_t N = (size_t)nx * ny * nz;
  float *F, *M, *D;
  unsigned long long* cs;
  CK(hipMalloc(&F, N * 4)); CK(hipMalloc(&M, N * 4)); CK(hipMalloc(&D, 3 * N * 4)); CK(hipMalloc(&cs, 8));
  hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, F, M, nx, ny, nz);
  CK(hipDeviceSynchronize());
  if (getenv("KB_CALIBRATE")) {   // D (3N floats) as scratch: copy N floats F -> D with both access widths
    hipLaunchKernelGGL(k_cal_copy16, dim3(8192), dim3(256), 0, 0, (const float4*)F, (float4*)D, N / 4);
    hipLaunchKernelGGL(k_cal_copy4, dim3(8192), dim3(256), 0, 0, (const float*)F, D + N, N);
    CK(hipDeviceSynchronize());
    printf("calibration: k_cal_copy16 and k_cal_copy4 each read %zu and write %zu bytes\n", N * 4, N * 4);
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  pp_ctx* ctx = nullptr;
  if (create(0, st, &ctx) != 0) { fprintf(stderr, "pp_create failed\n"); return 2; }
  pp_geom g;
  g.size[0] = nx; g.size[1] = ny; g.size[2] = nz;
  for (int i = 0; i < 3; ++i) { g.spacing[i] = 1.0; g.origin[i] = 0.0; }
  for (int i = 0; i < 9; ++i) g.direction[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (const char* sp = getenv("KB_SPACING")) sscanf(sp, "%lf,%lf,%lf", &g.spacing[0], &g.spacing[1], &g.spacing[2]);
  pp_demons_params p;
  defaults(&p);
  p.iterations = iters;
  for (int i = 0; i < 3; ++i) p.sigma_d_vox[i] = 1.5 / g.spacing[i];
  p.max_rms_error = 0.0;
  p.smooth_displacement = 1;
  p.smooth_update = 1;
  p.variant = PP_DEMONS_FUSED;
  if (getenv("KB_STAGED")) p.variant = PP_DEMONS_STAGED;

  for (int a = 6; a < argc; ++a) {
    // apply the env spec "K=V,K2=V2" (an entry "K=" unsets)
    std::string spec = argv[a];
    std::vector<std::string> keys;
    size_t pos = 0;
    while (pos < spec.size()) {
      size_t c = spec.find(',', pos);
      if (c == std::string::npos) c = spec.size();
      std::string kv = spec.substr(pos, c - pos);
      size_t e = kv.find('=');
      if (e != std::string::npos) {
        std::string k = kv.substr(0, e), v = kv.substr(e + 1);
        if (v.empty()) unsetenv(k.c_str()); else setenv(k.c_str(), v.c_str(), 1);
        keys.push_back(k);
      }
      pos = c + 1;
    }
    // (the library reads its switches once: take the snapshot again for this configuration; older builds have no such entry)
    if (auto reload = reinterpret_cast<void (*)(void)>(dlsym(h, "pp_reload_switches"))) reload();
    // warm-up (also sizes the workspace), then the timed run with per-kernel events
    pp_demons_params pw = p;
    pw.iterations = 2;
    if (execute(ctx, F, M, &g, &pw, D, nullptr) != 0) { fprintf(stderr, "execute failed: %s\n", last_error(ctx)); return 3; }
    sync(ctx);
    prof_enable(ctx, 1);
    pp_profile_entry ent[16];
    prof_read(ctx, ent, 16);
    auto t0 = std::chrono::steady_clock::now();
    if (execute(ctx, F, M, &g, &p, D, nullptr) != 0) { fprintf(stderr, "execute failed: %s\n", last_error(ctx)); return 3; }
    sync(ctx);
    auto t1 = std::chrono::steady_clock::now();
    const int ne = prof_read(ctx, ent, 16);
    prof_enable(ctx, 0);
    // un-profiled wall time of the same call
    auto t2 = std::chrono::steady_clock::now();
    if (execute(ctx, F, M, &g, &p, D, nullptr) != 0) return 3;
    sync(ctx);
    auto t3 = std::chrono::steady_clock::now();
    CK(hipMemsetAsync(cs, 0, 8, st));
    hipLaunchKernelGGL(k_checksum, dim3(2048), dim3(256), 0, st, (const float*)D, 3 * N, cs);
    unsigned long long hcs = 0;
    CK(hipMemcpyAsync(&hcs, cs, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const double ms_prof = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
    const double ms = std::chrono::duration<double, std::milli>(t3 - t2).count() / iters;
    printf("%-44s | %s | ms/iter %.4f (profiled %.4f) | %.0f Mvox/s | cs %016llx |", argv[1] + (strlen(argv[1]) > 44 ? strlen(argv[1]) - 44 : 0),
           spec.c_str(), ms, ms_prof, (double)N / ms / 1e3, hcs);
    for (int i = 0; i < ne && i < 16; ++i)
      if (ent[i].launches > 0 && ent[i].total_ms / ent[i].launches > 0.005) printf(" %s %.4f", ent[i].name, ent[i].total_ms / ent[i].launches);
    printf("\n");
    fflush(stdout);
    for (auto& k : keys) unsetenv(k.c_str());
    if (auto reload = reinterpret_cast<void (*)(void)>(dlsym(h, "pp_reload_switches"))) reload();
  }
  // -DPP_DRIFT builds: 100 MHz wall-clock stamps of every block at the quarter points of its march (the last launch of each
  // kernel): how far apart are the blocks that share an L2?  Spread = newest - oldest stamp over the blocks of one XCD
  // (block b -> XCD b % 8), in microseconds and in plane steps of that kernel.
  if (auto drift_read = reinterpret_cast<int (*)(unsigned long long*, int)>(dlsym(h, "pp_debug_drift_read"))) {
    std::vector<unsigned long long> buf(2 * 1024 * 4);
    if (auto xcc_read = reinterpret_cast<int (*)(unsigned*, int)>(dlsym(h, "pp_debug_drift_xcc_read"))) {
      std::vector<unsigned> xcc(2 * 1024);
      if (xcc_read(xcc.data(), (int)xcc.size()) > 0) {
        // HW_REG_XCC_ID (low 4 bits: the XCD) of blocks 0..23 of kernel A, and how many blocks b have XCC_ID & 15 == b % 8
        int agree = 0, nb = 0;
        for (int b = 0; b < 1024; ++b) if (buf.size() && xcc[b]) { ++nb; }
        printf("xcc ids of blocks 0..23 (kernel A, raw register & 0xff):");
        for (int b = 0; b < 24; ++b) printf(" %u", xcc[b] & 0xffu);
        for (int b = 0; b < 512; ++b) agree += ((xcc[b] & 15u) == (unsigned)(b % 8));
        printf("\n  blocks 0..511 with XCC_ID & 15 == b %% 8: %d of 512; distinct ids per residue:", agree);
        for (int r = 0; r < 8; ++r) { unsigned m = 0; for (int b = r; b < 512; b += 8) m |= 1u << (xcc[b] & 15u); printf(" %d:%#x", r, m); }
        printf("\n");
        (void)nb;
      }
    }
    if (drift_read(buf.data(), (int)buf.size()) > 0) {
      if (const char* dump = getenv("KB_DRIFT_DUMP")) {   // raw stamps: kernel block q0 q1 q2 q3 (10 ns ticks), for offline analysis
        if (FILE* fh = fopen(dump, "w")) {
          for (int k = 0; k < 2; ++k)
            for (int b = 0; b < 1024; ++b) {
              const unsigned long long* e = &buf[((size_t)k * 1024 + b) * 4];
              if (e[3]) fprintf(fh, "%d %d %llu %llu %llu %llu\n", k, b, e[0], e[1], e[2], e[3]);
            }
          fclose(fh);
        }
      }
      for (int k = 0; k < 2; ++k) {
        double step_us = 0.0; int nb = 0;
        for (int b = 0; b < 1024; ++b) {
          const unsigned long long* e = &buf[((size_t)k * 1024 + b) * 4];
          if (e[0] && e[3] > e[0]) { step_us += 0.01 * (double)(e[3] - e[0]); ++nb; }
        }
        if (!nb) continue;
        step_us /= nb;   // mean time from the first to the last quarter mark = 3/4 of a march
        printf("drift kernel %c: %d blocks, first-to-last quarter mark %.1f us\n", k ? 'B' : 'A', nb, step_us);
        // Do the two blocks that (presumably) share a CU -- XCD-local indices j and j + 32, i.e. blocks b and b + 256 of a
        // 512-block launch -- finish together?  d = end stamp of b + 256 minus end stamp of b.
        {
          double sum = 0.0, sabs = 0.0, lo = 1e30, hi = -1e30; int np = 0;
          double lo_first = 1e30, hi_first = -1e30, lo_second = 1e30, hi_second = -1e30;
          for (int b = 0; b < 256; ++b) {
            const unsigned long long e0 = buf[((size_t)k * 1024 + b) * 4 + 3], e1 = buf[((size_t)k * 1024 + b + 256) * 4 + 3];
            if (!e0 || !e1) continue;
            const double dd = 0.01 * ((double)e1 - (double)e0);
            sum += dd; sabs += dd < 0 ? -dd : dd; if (dd < lo) lo = dd; if (dd > hi) hi = dd; ++np;
            if ((b & 7) == 0) {   // XCD 0 only: spread inside each half of the run
              const double t0 = 0.01 * (double)e0, t1 = 0.01 * (double)e1;
              if (t0 < lo_first) lo_first = t0; if (t0 > hi_first) hi_first = t0;
              if (t1 < lo_second) lo_second = t1; if (t1 > hi_second) hi_second = t1;
            }
          }
          if (np) printf("  pairs (b, b + 256): end(b + 256) - end(b) mean %.1f us, mean |.| %.1f us, min %.1f, max %.1f over %d pairs; xcd 0: end spread of blocks j < 32 %.1f us, of j >= 32 %.1f us\n",
                         sum / np, sabs / np, lo, hi, np, hi_first - lo_first, hi_second - lo_second);
        }
        for (int x = 0; x < 8; ++x) {
          printf("  xcd %d:", x);
          for (int q = 0; q < 4; ++q) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int b = x; b < 1024; b += 8) {
              const unsigned long long v = buf[((size_t)k * 1024 + b) * 4 + q];
              if (!v) continue;
              if (v < lo) lo = v;
              if (v > hi) hi = v;
            }
            printf(" q%d spread %.2f us", q, hi >= lo ? 0.01 * (double)(hi - lo) : -1.0);
          }
          printf("\n");
        }
      }
    }
  }
  // -DPP_TRACE builds: per-wave shader-clock stamps at the plane loop's barriers (one interior block per kernel)
  if (auto trace_read = reinterpret_cast<int (*)(unsigned*, int)>(dlsym(h, "pp_debug_trace_read"))) {
    const int STEPS = 140, SLOTS = 6;
    std::vector<unsigned> buf(2 * 8 * STEPS * SLOTS);
    if (trace_read(buf.data(), (int)buf.size()) > 0 && getenv("KBENCH_PHASES")) {
      // small-grid kernels: phase stamps (steps 0..8, slot 0) of the first eight waves, clocks since wave 0's phase 0
      for (int k = 0; k < 2; ++k) {
        printf("phases kernel %c (0 entry, 1 halt read, 2 fold done, 8 images in LDS, 3 smoothing input ready, 4 x done, 5 y done, 6 z done, 7 end)\n", k ? 'B' : 'A');
        // rows 0..3: waves 0..3 of the middle block, rows 4..7: waves 0..3 of the last block; shader clocks / 10 ns ticks
        const unsigned base = buf[((k * 8 + 0) * STEPS + 0) * SLOTS + 0], wbase = buf[((k * 8 + 0) * STEPS + 0) * SLOTS + 1];
        for (int w = 0; w < 8; w += 2) {
          printf("  %s w%d:", w < 4 ? "mid " : "last", w & 3);
          for (int s : {0, 1, 2, 8, 3, 4, 5, 6, 7})
            printf(" p%d=%d/%.2fus", s, (int)(buf[((k * 8 + w) * STEPS + s) * SLOTS + 0] - base),
                   0.01 * (int)(buf[((k * 8 + w) * STEPS + s) * SLOTS + 1] - wbase));
          printf("\n");
        }
      }
    } else if (trace_read(buf.data(), (int)buf.size()) > 0) {
      for (int k = 0; k < 2; ++k) {
        printf("trace kernel %c: step | per wave: t(slot1)-t(slot0) ... (shader clocks since slot 0 of wave 0)\n", k ? 'B' : 'A');
        for (int s = 20; s < 32; ++s) {
          const unsigned base = buf[((k * 8 + 0) * STEPS + s) * SLOTS + 0];
          printf("  step %3d:", s);
          for (int w = 0; w < 8; ++w) {
            printf(" w%d[", w);
            for (int q = 0; q < (k ? 4 : 5); ++q) printf("%s%d", q ? " " : "", (int)(buf[((k * 8 + w) * STEPS + s) * SLOTS + q] - base));
            printf("]");
          }
          printf("\n");
        }
        const unsigned t0 = buf[((k * 8 + 0) * STEPS + 20) * SLOTS + 0], t1 = buf[((k * 8 + 0) * STEPS + 120) * SLOTS + 0];
        printf("  100 steps of wave 0: %u clocks = %.1f per step\n", t1 - t0, (t1 - t0) / 100.0);
      }
    }
  }
  return 0;
}
