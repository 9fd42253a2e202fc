// tools/probes/valu_rate.hip -- measured issue rates of the instruction classes the fused demons kernels are made of
// (measurement tooling).  Each kernel runs ITER iterations of 16 independent dependency chains of one instruction in
// every lane, 512-thread blocks, `bpc` blocks per CU (4 waves per SIMD at bpc = 2), and reports wave-instructions per
// cycle per SIMD at the measured wall time and the clock read from the device properties.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int ITER = 4096;

template <int KIND>
__global__ void __launch_bounds__(512) k_rate(float* out, float a, float b) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a * (float)(threadIdx.x + i);
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{v[2 * i], v[2 * i + 1]};
  const f2 pa = f2{a, a}, pb = f2{b, b};
  if constexpr (KIND == 15) asm volatile("s_mov_b64 vcc, 0x55555555" ::: "vcc");
  if constexpr (KIND == 21 || KIND == 25) asm volatile("s_mov_b64 s[20:21], 0x33333333" ::: "s20", "s21");
  if constexpr (KIND == 22) asm volatile("v_cmp_neq_f32 s[20:21], %0, %1\n\ts_nop 4" : : "v"(v[0]), "v"(v[1]) : "s20", "s21");
  for (int it = 0; it < ITER; ++it) {
    if constexpr (KIND == 0) {   // v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if constexpr (KIND == 1) {   // v_pk_fma_f32 (two fmas per lane)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
    } else if constexpr (KIND == 2) {   // v_cndmask_b32 with vcc
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
    } else if constexpr (KIND == 3) {   // DPP move (wave_shr:1)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
    } else if constexpr (KIND == 4) {   // v_mul_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
    } else if constexpr (KIND == 5) {   // v_pk_mul_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
    } else if constexpr (KIND == 6) {   // v_cmp_neq_f32 to an SGPR pair + v_cndmask from it (the select idiom)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_cmp_neq_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(a) : "s20", "s21");
    } else if constexpr (KIND == 7) {   // v_rcp_f32 (transcendental)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 8) {   // 1 dependent chain only (latency): 16 dependent fmas
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(a), "v"(b));
    } else if constexpr (KIND == 9) {   // s_and_b64 (SALU) interleaved 1:1 with v_fma
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_and_b64 s[20:21], s[20:21], exec" : "+v"(v[i]) : "v"(a), "v"(b) : "s20", "s21");
    } else if constexpr (KIND == 10) {  // v_pk_add_f32
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
    } else if constexpr (KIND == 11) {  // v_med3_f32 / v_max_f32 class
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
    } else if constexpr (KIND == 12) {  // v_cvt_i32_f32 + v_floor
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_floor_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 13) {  // integer v_add_u32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
    } else if constexpr (KIND == 15) {  // v_cndmask_b32 reading a vcc that was written ONCE before the loop (round 4: KIND 2's
                                        // clobber made the compiler put an s_nop behind every select -- it timed the pair)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a));
    } else if constexpr (KIND == 16) {  // v_fma_f32 + s_nop 0: the price of a hazard no-op behind a vector instruction
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if constexpr (KIND == 17) {  // the compiler's own select: v_cmp -> vcc, s_nop 1, v_cndmask ..., vcc (as in the shipped ISA)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_cmp_neq_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
    } else if constexpr (KIND == 18) {  // the same pair without the no-op (is the hazard real? compare the RESULT, not only the time)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_cmp_neq_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
    } else if constexpr (KIND == 19) {  // arithmetic blend instead of a select: v_fma with a 0/1 factor
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if constexpr (KIND == 20) {  // v_fma_f32 + s_nop 1
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 1" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if constexpr (KIND == 21) {  // mask in an SGPR pair written by the SCALAR unit just before each select
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("s_not_b64 s[20:21], s[20:21]\n\ts_nop 3\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(a) : "s20", "s21");
    } else if constexpr (KIND == 22) {  // mask written by a VALU compare ONCE before the loop (static, but vector-written)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(a));
    } else if constexpr (KIND == 23) {  // compare -> 8 unrelated vector instructions -> select on that compare's mask
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        asm volatile("v_cmp_neq_f32 s[20:21], %0, %1" : : "v"(v[8 * i]), "v"(a) : "s20", "s21");
#pragma unroll
        for (int j = 1; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[8 * i + j]) : "v"(a), "v"(b));
        asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[8 * i]) : "v"(a));
      }
    } else if constexpr (KIND == 24) {  // one compare feeding eight selects (the mask is read eight times)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        asm volatile("v_cmp_neq_f32 s[20:21], %0, %1" : : "v"(v[8 * i]), "v"(a) : "s20", "s21");
#pragma unroll
        for (int j = 0; j < 7; ++j) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[8 * i + j + 1]) : "v"(a));
      }
    } else if constexpr (KIND == 25) {  // v_fma_f32 with a scalar-register operand (are static SGPR reads slow in general?)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(v[i]) : "v"(b));
    } else if constexpr (KIND == 26) {  // v_cndmask with exec as the mask
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, exec" : "+v"(v[i]) : "v"(a));
    } else if constexpr (KIND == 14) {  // v_mul_u32_u24
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(a));
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int n_per_iter, int cus, double ghz, float* out) {
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(cus * bpc), dim3(512), 0, 0, out, 1.0001f, 0.5f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(cus * bpc), dim3(512), 0, 0, out, 1.0001f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD: blocks per CU * 8 waves / 4 SIMDs * ITER * n_per_iter
    const double wi = (double)bpc * 2.0 * ITER * n_per_iter;
    const double cyc = ms * 1e-3 * ghz * 1e9;
    printf("%-28s waves/SIMD %d : %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, bpc * 2, ms, cyc / wi, ghz);
  }
}

static float* g_out;
static int g_cus;
static double g_ghz;
template <int KIND>
void one(const char* name, int n_per_iter, int want) {
  if (want >= 0 && want != KIND) return;
  run<KIND>(name, n_per_iter, g_cus, g_ghz, g_out);
  fflush(stdout);
}

// valu_rate [kind]: every kind, or one of them (run each new kind in its own process under `timeout`: a probe that hangs -- one
// of the old kinds 8 / 9 / 14 did, and one of round 4's first mask-source kinds -- then costs seconds, not the GPU visit)
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int want = argc > 1 ? atoi(argv[1]) : -1;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  g_ghz = prop.clockRate * 1e-6;
  if (want < 0) printf("device %s, %d CUs, clockRate %.3f GHz\n", prop.name, g_cus, g_ghz);
  CK(hipMalloc(&g_out, 4096));
  one<0>("v_fma_f32", 16, want);
  one<1>("v_pk_fma_f32", 16, want);
  one<4>("v_mul_f32", 16, want);
  one<5>("v_pk_mul_f32", 16, want);
  one<10>("v_pk_add_f32", 16, want);
  one<2>("v_cndmask_b32 (vcc, undefined)", 16, want);
  one<6>("v_cmp + v_cndmask (sgpr)", 16, want);
  one<15>("v_cndmask_b32 (static vcc)", 16, want);
  one<17>("v_cmp vcc + s_nop 1 + cndmask", 16, want);
  one<18>("v_cmp vcc + cndmask (no nop)", 16, want);
  one<21>("s_not mask + cndmask (SALU-fresh)", 16, want);
  one<22>("cndmask, mask v_cmp'd once", 16, want);
  one<23>("v_cmp, 7 fma, cndmask", 16, want);
  one<24>("v_cmp + 7 cndmask on it", 16, want);
  one<26>("v_cndmask_b32 (exec)", 16, want);
  one<25>("v_fma_f32 with an SGPR operand", 16, want);
  one<16>("v_fma_f32 + s_nop 0", 16, want);
  one<20>("v_fma_f32 + s_nop 1", 16, want);
  one<3>("v_mov_b32_dpp wave_shr:1", 16, want);
  one<7>("v_rcp_f32", 16, want);
  one<11>("v_max_f32", 16, want);
  one<12>("v_floor_f32", 16, want);
  one<13>("v_add_u32", 16, want);
  return 0;
}
