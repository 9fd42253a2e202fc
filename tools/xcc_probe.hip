#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)v;
}
int main() {
  const int nb = 4096;
  int* d; hipMalloc(&d, nb * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
    std::vector<int> h(nb); hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    int match = 0; int hist[16] = {0};
    for (int b = 0; b < nb; ++b) { int x = h[b] & 0xf; hist[x]++; if (x == (b & 7)) ++match; }
    printf("rep %d raw[0..15]:", rep); for (int b = 0; b < 16; ++b) printf(" %x", h[b]); printf("\n  xcc==b%%8 for %d of %d blocks; hist:", match, nb);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]); printf("\n");
  }
  return 0;
}
