// round 5: what a gather costs in the texture addresser.  One wavefront-wide memory instruction per pattern, many times, on a
// cache-resident array, all CUs busy: clocks per instruction per CU.
//   hipcc -O2 --offload-arch=gfx950 -o tabench tabench.cpp && ./tabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct f2u { float x, y; } __attribute__((packed, aligned(4)));
struct f3u { float x, y, z; } __attribute__((packed, aligned(4)));
struct f4u { float x, y, z, w; } __attribute__((packed, aligned(4)));

// MODE 0: 8-byte pair at 4-byte alignment, consecutive lanes one element apart (the warp's gather on a coherent field)
// MODE 1: 4-byte element, consecutive lanes (coalesced)
// MODE 2: 4-byte element, one lane in 8 active
// MODE 3: 8-byte pair, 8-byte aligned, consecutive lanes two elements apart
// MODE 4: 4-byte element, lanes scattered over rows (incoherent gather)
// MODE 5: 8-byte pair at 4-byte alignment, lanes scattered over rows
// MODE 6: 16-byte quad, 16-byte aligned, consecutive lanes
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ a, float* __restrict__ out, int iters, unsigned mask) {
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  unsigned base = (wave * 977u) & mask;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned row = (base + (unsigned)u * 4099u) & mask;
      if (MODE == 0) {
        const f2u v = *reinterpret_cast<const f2u*>(a + row + lane + 1u);
        acc += v.x + v.y;
      } else if (MODE == 1) {
        acc += a[row + lane + 1u];
      } else if (MODE == 2) {
        if ((lane & 7u) == 3u) acc += a[row + lane + 1u];
      } else if (MODE == 3) {
        const float2 v = *reinterpret_cast<const float2*>(a + ((row + 2u * lane) & ~1u));
        acc += v.x + v.y;
      } else if (MODE == 4) {
        acc += a[(row + lane * 521u + 1u) & mask];
      } else if (MODE == 5) {
        const f2u v = *reinterpret_cast<const f2u*>(a + ((row + lane * 521u + 1u) & mask));
        acc += v.x + v.y;
      } else if (MODE == 6) {
        const float4 v = *reinterpret_cast<const float4*>(a + ((row + 4u * lane) & ~3u));
        acc += v.x + v.y + v.z + v.w;
      } else {
        // kernel B's shape: lane l samples x = 2 l and 2 l + 1; `jog` = the displacement's integer part changes once in the wave
        const unsigned jog = (MODE >= 10 && MODE != 13) ? (lane >= 37u ? 1u : 0u) : 0u;
        const unsigned e = row + 2u * lane + 1u + jog;
        if (MODE == 7 || MODE == 10) {          // two pairs per lane (A's and B's corners): today's form
          const f2u v = *reinterpret_cast<const f2u*>(a + e);
          const f2u w = *reinterpret_cast<const f2u*>(a + e + 1u);
          acc += v.x + v.y + w.x + w.y;
        } else if (MODE == 8 || MODE == 11) {   // one 16-byte load at 4-byte alignment covers both
          const f4u v = *reinterpret_cast<const f4u*>(a + e);
          acc += v.x + v.y + v.z + v.w;
        } else if (MODE == 9 || MODE == 12) {   // the quad, plus a pair for one lane in 8 (row-incoherent lanes)
          const f4u v = *reinterpret_cast<const f4u*>(a + e);
          acc += v.x + v.y + v.z + v.w;
          if ((lane & 7u) == 3u) {
            const f2u w = *reinterpret_cast<const f2u*>(a + ((e + 4099u) & mask));
            acc += w.x + w.y;
          }
        } else {                                // 13: 12 bytes
          const f3u v = *reinterpret_cast<const f3u*>(a + e);
          acc += v.x + v.y + v.z;
        }
      }
    }
    base = (base + 64u * 131u) & mask;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
void run(const char* name, const float* a, float* out, unsigned mask) {
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, a, out, 10, mask);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, a, out, iters, mask);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double insts_per_cu = (double)blocks * 4 * iters * 8 / 256.0;   // wave-instructions per CU
  printf("%-62s %8.3f ms  %6.2f ns per wave-instruction per CU (%.1f clk at 2.1 GHz)\n", name, ms, ms * 1e6 / insts_per_cu, ms * 1e6 / insts_per_cu * 2.1);
}

int main(int argc, char** argv) {
  const unsigned words = argc > 1 ? (unsigned)atoi(argv[1]) : (1u << 18);   // 1 MB: L2-resident; 2^13: L1-resident
  float *a, *out;
  CK(hipMalloc(&a, (size_t)words * 4 + 4096));
  CK(hipMemset(a, 0, (size_t)words * 4 + 4096));
  CK(hipMalloc(&out, 64));
  const unsigned mask = words - 1u - 1023u;   // (keeps row + offsets inside the array: rows start below words - 1024)
  printf("array of %u floats\n", words);
  run<0>("pair, 4-byte aligned, consecutive lanes (the warp's gather)", a, out, mask);
  run<1>("element, consecutive lanes", a, out, mask);
  run<2>("element, one lane in 8 active", a, out, mask);
  run<3>("pair, 8-byte aligned, lanes two elements apart", a, out, mask);
  run<4>("element, lanes scattered", a, out, mask);
  run<5>("pair, 4-byte aligned, lanes scattered", a, out, mask);
  run<6>("quad, 16-byte aligned, consecutive lanes", a, out, mask);
  printf("kernel B's shape (lane l: x = 2 l, 2 l + 1); per lane-iteration, i.e. per 1 or 2 instructions:\n");
  run<7>("B: two pairs (today)", a, out, mask);
  run<8>("B: one 16-byte load, 4-byte aligned", a, out, mask);
  run<9>("B: 16-byte load + a pair for one lane in 8", a, out, mask);
  run<13>("B: one 12-byte load", a, out, mask);
  run<10>("B, one jog in the wave: two pairs (today)", a, out, mask);
  run<11>("B, one jog: one 16-byte load", a, out, mask);
  run<12>("B, one jog: 16-byte load + a pair for one lane in 8", a, out, mask);
  return 0;
}
