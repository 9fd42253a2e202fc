// probe: do 8-byte global / buffer loads from a 4-byte-aligned (odd dword) address return the right data on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
struct f2u { float x, y; } __attribute__((aligned(4)));
typedef unsigned u2 __attribute__((vector_size(8)));
__global__ void k(const float* __restrict__ a, float* __restrict__ o) {
  const char* base = reinterpret_cast<const char*>(a);
  unsigned off = (threadIdx.x * 3u + 1u) * 4u;   // odd / even dword offsets
  f2u v = *reinterpret_cast<const f2u*>(base + (size_t)off);
  o[2 * threadIdx.x] = v.x; o[2 * threadIdx.x + 1] = v.y;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, 4096, 0x00020000);
  u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  o[128 + 2 * threadIdx.x] = __builtin_bit_cast(float, w[0]); o[128 + 2 * threadIdx.x + 1] = __builtin_bit_cast(float, w[1]);
}
int main() {
  float h[1024], *d, *o, ho[256];
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  int bad_g = 0, bad_b = 0;
  for (int t = 0; t < 64; ++t) {
    int e = t * 3 + 1;
    if (ho[2 * t] != e || ho[2 * t + 1] != e + 1) bad_g++;
    if (ho[128 + 2 * t] != e || ho[128 + 2 * t + 1] != e + 1) bad_b++;
  }
  printf("global x2 unaligned: %d bad of 64; buffer x2 unaligned: %d bad of 64; lane1 global (%g,%g) buffer (%g,%g) expected (4,5)\n", bad_g, bad_b, ho[2], ho[3], ho[130], ho[131]);
  return 0;
}
