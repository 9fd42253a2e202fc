// tests/emu/include/hip/hip_runtime.h
//
// TEST INFRASTRUCTURE ONLY.  A CPU stand-in for the small part of the HIP runtime that
// platipy_amd/csrc uses, so the *unmodified* kernel sources can be compiled with g++ and
// their indexing / LDS / barrier logic checked against the oracle in the CPU test suite
// (there is no GPU in the build container).  Each thread block runs as blockDim cooperative
// fibers on one OS thread (__syncthreads() yields to the next fiber); a pool of OS threads runs
// different blocks in parallel, and `__shared__` is thread-local to the OS thread, i.e. per block.
// It is never on the product's include path and nothing under platipy_amd/ refers to it.
#pragma once

#include <ucontext.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __constant__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uchar4 make_uchar4(unsigned char a, unsigned char b, unsigned char c, unsigned char d) {
  return uchar4{a, b, c, d};
}

namespace hipemu {
struct Fiber {
#if defined(__x86_64__)
  void* sp = nullptr;   // saved stack pointer (hand-written context switch: no signal-mask system call per yield)
#else
  ucontext_t ctx;
#endif
  dim3 tid, bid;
  bool done = false;
};
extern thread_local Fiber* t_current;
extern dim3 g_blockDim, g_gridDim;
void fiber_yield();
void block_barrier();
void wave_barrier();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (::hipemu::t_current->tid)
#define blockIdx (::hipemu::t_current->bid)
#define blockDim (::hipemu::g_blockDim)
#define gridDim (::hipemu::g_gridDim)

static inline void __syncthreads() { ::hipemu::block_barrier(); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  ::hipemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

// ---- runtime API subset ------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct hipemuEvent { std::chrono::steady_clock::time_point t; };
typedef hipemuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = nullptr;
  if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory;
  return hipSuccess;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

template <typename K>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }

// ---- device intrinsics subset ------------------------------------------------------
static inline float atomicAdd(float* p, float v) {
  float o, n;
  do { o = *p; n = o + v; } while (!__atomic_compare_exchange(p, &o, &n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
  return o;
}
static inline double atomicAdd(double* p, double v) {
  double o, n;
  do { o = *p; n = o + v; } while (!__atomic_compare_exchange(p, &o, &n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
  return o;
}
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
// scoped atomic load / store builtins of clang's HIP mode (the scope is meaningless on the CPU)
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T>
static inline T __hip_atomic_load(const T* p, int /*order*/, int /*scope*/) {
  T v;
  __atomic_load(const_cast<T*>(p), &v, __ATOMIC_SEQ_CST);
  return v;
}
template <class T>
static inline void __hip_atomic_store(T* p, T v, int /*order*/, int /*scope*/) {
  __atomic_store(p, &v, __ATOMIC_SEQ_CST);
}
template <class T>
static inline T __hip_atomic_fetch_add(T* p, T v, int /*order*/, int /*scope*/) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicMin(int* p, int v) {
  int o = *p;
  while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
static inline int atomicMax(int* p, int v) {
  int o = *p;
  while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int __float2int_rd(float x) { return (int)floorf(x); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fdividef(float a, float b) { return a / b; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))   // v_rcp_f32 (1 ulp) on the GPU
static inline void __builtin_amdgcn_sched_barrier(int) {}                  // (a scheduling hint: nothing to emulate)
static inline void __builtin_amdgcn_s_setprio(short) {}                    // (an issue-priority hint: nothing to emulate)
static inline void __builtin_amdgcn_s_sleep(int) {}                        // (a pause between two polls of a word another block writes)
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // (only ever applied to wave-uniform values)
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xffffffu) * (b & 0xffffffu)); }

// buffer resources: base pointer + 32-bit byte offsets + the range word (raw buffers: a lane whose per-lane offset reaches
// num_records is out of range -- its load returns 0, its store is dropped; the scalar offset takes no part in the check, the
// reading of the ISA the kernels are written to be independent of: they keep offset + scalar offset below 2^32 and arrays
// below 2^31 bytes).  The format word is ignored.
struct __amdgpu_buffer_rsrc_t { char* base; unsigned num_records; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int n, int) { return __amdgpu_buffer_rsrc_t{static_cast<char*>(p), (unsigned)n}; }
static inline bool hipemu_buf_in_range(const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned bytes) { return (unsigned long long)voff + bytes <= r.num_records; }
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
  unsigned v = 0;
  if (hipemu_buf_in_range(r, voff, 4)) memcpy(&v, r.base + (size_t)voff + soff, 4);
  return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
  if (hipemu_buf_in_range(r, voff, 4)) memcpy(r.base + (size_t)voff + soff, &v, 4);
}
typedef unsigned hipemu_u2 __attribute__((vector_size(8)));
static inline hipemu_u2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
  hipemu_u2 v = {0u, 0u};
  if (hipemu_buf_in_range(r, voff, 8)) memcpy(&v, r.base + (size_t)voff + soff, 8);
  return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b64(hipemu_u2 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
  if (hipemu_buf_in_range(r, voff, 8)) memcpy(r.base + (size_t)voff + soff, &v, 8);
}

typedef unsigned hipemu_u4 __attribute__((vector_size(16)));
static inline void __builtin_amdgcn_raw_buffer_store_b128(hipemu_u4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
  if (hipemu_buf_in_range(r, voff, 16)) memcpy(r.base + (size_t)voff + soff, &v, 16);
}

// Wavefront shuffle (64 lanes).  Every live lane of the WAVEFRONT must reach the call: the value is exchanged through a
// per-block table between two wavefront rendezvous.
namespace hipemu { extern thread_local double t_shfl[1024]; }
static inline float hipemu_shfl_from(float v, int delta) {   // value of lane (lane + delta), own value when that lane does not exist
  const unsigned t = threadIdx.x;
  ::hipemu::t_shfl[t] = (double)v;
  ::hipemu::wave_barrier();
  const int src = (int)(t & 63u) + delta;
  const unsigned idx = (t & ~63u) + (unsigned)src;
  const float r = (src >= 0 && src < 64 && idx < blockDim.x) ? (float)::hipemu::t_shfl[idx] : v;
  ::hipemu::wave_barrier();
  return r;
}
static inline float __shfl_up(float v, unsigned delta, int /*width*/ = 64) { return hipemu_shfl_from(v, -(int)delta); }
static inline float __shfl_down(float v, unsigned delta, int /*width*/ = 64) { return hipemu_shfl_from(v, (int)delta); }
// 32-bit integers travel through the same table as exact doubles
static inline int hipemu_shfl_int(int v, int src_lane_or_delta, bool relative) {
  const unsigned t = threadIdx.x;
  ::hipemu::t_shfl[t] = (double)v;
  ::hipemu::wave_barrier();
  const int src = relative ? (int)(t & 63u) + src_lane_or_delta : (src_lane_or_delta & 63);
  const unsigned idx = (t & ~63u) + (unsigned)src;
  const int r = (src >= 0 && src < 64 && idx < blockDim.x) ? (int)::hipemu::t_shfl[idx] : v;
  ::hipemu::wave_barrier();
  return r;
}
static inline int __shfl_up(int v, unsigned delta, int /*width*/ = 64) { return hipemu_shfl_int(v, -(int)delta, true); }
static inline int __shfl(int v, int src_lane, int /*width*/ = 64) { return hipemu_shfl_int(v, src_lane, false); }
// DPP whole-wave shifts by one lane (the two controls csrc/ uses): 0x138 = wave_shr:1 (value of lane - 1), 0x130 = wave_shl:1.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
  (void)old;
  float f;
  memcpy(&f, &src, 4);
  // (a float carries any 32-bit pattern through the table unchanged: it is only copied)
  const float r = ctrl == 0x138 ? hipemu_shfl_from(f, -1) : hipemu_shfl_from(f, 1);
  int out;
  memcpy(&out, &r, 4);
  return out;
}
static inline double __shfl_xor(double v, int lane_mask, int /*width*/ = 64) {
  const unsigned t = threadIdx.x;
  ::hipemu::t_shfl[t] = v;
  ::hipemu::wave_barrier();
  const unsigned src = (t & ~63u) | ((t ^ (unsigned)lane_mask) & 63u);
  const double r = src < blockDim.x ? ::hipemu::t_shfl[src] : v;
  ::hipemu::wave_barrier();
  return r;
}
// Wavefront ballot: bit l = predicate of lane l of the caller's wavefront (lanes beyond the block: 0).
static inline unsigned long long __ballot(int pred) {
  const unsigned t = threadIdx.x;
  ::hipemu::t_shfl[t] = pred ? 1.0 : 0.0;
  ::hipemu::wave_barrier();
  unsigned long long r = 0;
  for (unsigned l = (t & ~63u); l < (t & ~63u) + 64u && l < blockDim.x; ++l)
    if (::hipemu::t_shfl[l] != 0.0) r |= 1ull << (l & 63u);
  ::hipemu::wave_barrier();
  return r;
}
static inline int __ffsll(unsigned long long v) { return v ? __builtin_ctzll(v) + 1 : 0; }
// Wavefront vote: non-zero if the predicate holds in any live lane of the caller's wavefront.
static inline int __any(int pred) {
  const unsigned t = threadIdx.x;
  ::hipemu::t_shfl[t] = pred ? 1.0 : 0.0;
  ::hipemu::wave_barrier();
  int r = 0;
  for (unsigned l = (t & ~63u); l < (t & ~63u) + 64u && l < blockDim.x; ++l) r |= (::hipemu::t_shfl[l] != 0.0);
  ::hipemu::wave_barrier();
  return r;
}
