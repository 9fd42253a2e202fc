// platipy_amd/csrc/pp_internal.h -- shared host/device helpers of libplatipy_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/platipy_amd.h"

// ---------------------------------------------------------------------------------------
// context

struct pp_profiler;

struct pp_ctx {
  int device;
  hipStream_t stream;
  char* ws;          // device scratch, grown on demand
  size_t ws_bytes;
  pp_profiler* prof;  // NULL unless pp_profile_enable(ctx, 1)
  char* pinned;       // 4 KB of page-locked host memory for small read-backs (lazy)
  unsigned* ticket;   // device counter for last-block-finishes reductions (lazy, kept at 0 between launches)
  char* mailbox;      // 4 KB of page-locked, device-visible host memory a kernel writes small results into (lazy)
  unsigned long long mail_seq;  // sequence number of the last posted result
  double* hist;       // device ring of the last Execute's per-iteration {metric, RMS change} (lazy; pp_demons_history)
  int hist_cap;       // entries (iterations) the ring holds
  // Fixed-image samples of the metric lattice, kept while one optimiser level runs (pp_fixed_samples): the sample points
  // of the fixed image do not move between the level's hundreds of metric evaluations.
  float* fsamp;
  size_t fsamp_cap;        // floats
  int fsamp_scope;         // > 0 inside pp_linear_optimize_f32: the fixed image cannot change under the cache
  int fsamp_valid;
  struct {
    const float* fixed;
    const unsigned char* fmask;
    int fsize[3], vsize[3], stride;
    double Af[9], bf[3];
    const float* jitter;
    unsigned long long jitter_gen;
  } fsamp_key;
  // ITK's per-sample jitter of the metric lattice (pp_linear_set_sample_jitter): a caller-owned device array of 3 floats per
  // sample in virtual-index units, NULL = the lattice itself.  `jitter_gen` counts the calls, so that the fixed-sample cache
  // never serves samples taken under another jitter array that happens to live at the same address.
  const float* jitter;
  size_t jitter_samples;
  unsigned long long jitter_gen;
  // ITK's filtered gradient image of the moving image (pp_linear_set_moving_gradient): caller-owned, 3 volumes of mgrad_size
  // voxels in moving-index units, NULL = the interpolant's analytic gradient.
  const float* mgrad;
  int mgrad_size[3];
  // ... and its packed companion (pp_linear_set_moving_gradient_packed): (gx, gy, gz, intensity) per voxel of the SAME moving image,
  // what the value + gradient kernel gathers from when it is set (a quarter of the cache sectors per sample)
  const float* mgrad4;
  char err[512];
};

// The PP_* measurement / debugging switches, as the process environment held them when the library first needed
// one -- a snapshot, taken once (or again by pp_reload_switches): launchers never call getenv, which may race with a host
// program writing its environment from another thread.  -> the value, or NULL when the variable was unset, is not a known
// switch, or did not parse (numeric switches must be integers; a warning goes to stderr once).
const char* pp_env(const char* name);

// HIP-event bracket around one launch (no-ops while profiling is off).
void pp_prof_begin(pp_ctx* ctx, const char* kernel_name);
void pp_prof_end(pp_ctx* ctx);
struct pp_prof_scope {
  pp_ctx* c;
  pp_prof_scope(pp_ctx* ctx, const char* name) : c(ctx) { if (c->prof) pp_prof_begin(c, name); }
  ~pp_prof_scope() { if (c->prof) pp_prof_end(c); }
};

// Every entry point runs with the context's device current (allocations, event creation and launches on the legacy
// default stream all follow the calling thread's current device) and restores the caller's device on return, so a ctx
// may be driven from any thread -- a new Python thread starts on device 0 -- without disturbing the caller's state.
struct pp_device_guard {
  int prev = -1;
  bool switched = false;
  explicit pp_device_guard(const pp_ctx* ctx) {
    if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = (hipSetDevice(ctx->device) == hipSuccess);
  }
  ~pp_device_guard() {
    if (switched) (void)hipSetDevice(prev);
  }
  pp_device_guard(const pp_device_guard&) = delete;
  pp_device_guard& operator=(const pp_device_guard&) = delete;
};

int pp_fail(pp_ctx* ctx, int code, const char* fmt, ...);
// The metric entry points of pp_fusion.hip keep a device array of the fixed image's trilinear samples on the metric
// lattice (PP_FSAMP_INVALID bits where a sample falls outside the fixed buffer or its mask): built on first use inside a
// pp_fsamp_scope and reused while the key matches; outside such a scope they sample the fixed image themselves.
constexpr unsigned PP_FSAMP_INVALID = 0x7fc0deadu;
struct pp_fsamp_scope {
  pp_ctx* c;
  explicit pp_fsamp_scope(pp_ctx* ctx) : c(ctx) { ++c->fsamp_scope; }
  ~pp_fsamp_scope() {
    if (--c->fsamp_scope == 0) c->fsamp_valid = 0;
  }
};
// Reserve `bytes` of device scratch (256-B aligned slices are carved by the callers).
int pp_reserve(pp_ctx* ctx, size_t bytes);
// Copy `bytes` (<= 4096) from device memory to `host` through the context's page-locked staging buffer and wait
// for the stream: a pageable destination would make the runtime stage and block on its own, several times slower.
int pp_read_back(pp_ctx* ctx, const void* dev, void* host, size_t bytes);
// Device counter (zero between launches) for kernels whose last block folds the partial sums.
int pp_ticket(pp_ctx* ctx, unsigned** out);
// Device buffer for the per-iteration {metric, RMS change} pairs of one demons Execute (PP_HIST_CAP iterations).
constexpr int PP_HIST_CAP = 4096;
int pp_history_buffer(pp_ctx* ctx, double** out);
// Mailbox for kernels that hand a few numbers straight to the host: 4 KB of page-locked memory the device writes
// through its host pointer and the host polls -- no copy command and no interrupt-driven wake-up between a
// launch-latency-bound kernel and the host code that consumes it.
//   * Every number travels as ONE 16-byte store {value, tag} with tag = the launch's sequence number, and the host takes an
//     entry when its tag matches (tag first, then value).  Nothing depends on the order in which separate device stores
//     become visible in host memory.  (Until round 3 a payload was followed by a fence and a completion flag: once in a few
//     hundred affine registrations the host read a flag ahead of the payload it announced -- posted writes to host memory
//     may be reordered on the way -- took the previous launch's value for a candidate, and a line search branched
//     differently: tools/determinism_linear.py.)
//   * Up to PP_MAIL_WRITERS blocks of a launch post; writer k owns bytes [k * PP_MAIL_SLOT, (k + 1) * PP_MAIL_SLOT): its own
//     cache lines (the writers run on different XCDs, each behind its own L2).
struct alignas(16) pp_mail_entry {
  double value;
  unsigned long long tag;
};
constexpr size_t PP_MAIL_SLOT = 1024;         // bytes per writer: 64 entries
constexpr int PP_MAIL_WRITERS = 4;
constexpr int PP_MAIL_ENTRIES = (int)(PP_MAIL_SLOT / sizeof(pp_mail_entry));
__host__ __device__ inline pp_mail_entry* pp_mail_slot(void* mailbox, int writer) {
  return reinterpret_cast<pp_mail_entry*>(static_cast<char*>(mailbox) + (size_t)writer * PP_MAIL_SLOT);
}
//   * The tag word is not the bare sequence number but pp_mail_tag(seq, value): the sequence number mixed with the value's
//     own bits.  A 16-byte store is normally one transaction, but neither HIP nor PCIe promises it; if it ever tears, a new
//     tag beside the previous launch's value (or the reverse) does not verify and the host keeps polling -- the protocol
//     does not depend on the store being atomic.
__host__ __device__ inline unsigned long long pp_mail_tag(unsigned long long seq, double value) {
  return (seq * 0x9E3779B97F4A7C15ull) ^ __builtin_bit_cast(unsigned long long, value);
}
// one 16-byte store
__device__ __forceinline__ void pp_mail_post(pp_mail_entry* e, double value, unsigned long long seq) {
  typedef unsigned long long u64x2 __attribute__((vector_size(16)));
  u64x2 w;
  w[0] = __builtin_bit_cast(unsigned long long, value);
  w[1] = pp_mail_tag(seq, value);
  *reinterpret_cast<u64x2*>(e) = w;
}
// -> the mailbox and the sequence number of the next launch
int pp_mailbox(pp_ctx* ctx, char** mailbox, unsigned long long* seq);
// Wait for entries [0, n) of `writer` to carry `seq` and copy their values out (falls back to a stream synchronisation if
// they do not arrive within 200 ms; an entry still missing after that is an error).
int pp_mail_take(pp_ctx* ctx, int writer, int n, unsigned long long seq, double* out);

#define PP_HIP(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return pp_fail((ctx), PP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

#define PP_LAUNCH_CHECK(ctx, name)                                                        \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess)                                                                 \
      return pp_fail((ctx), PP_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

#define PP_REQUIRE(ctx, cond, msg)                            \
  do {                                                        \
    if (!(cond)) return pp_fail((ctx), PP_ERR_ARG, "%s", msg); \
  } while (0)

static inline size_t pp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline size_t pp_nvox(const int size[3]) { return (size_t)size[0] * size[1] * size[2]; }

// Scratch carving: a bump pointer over ctx->ws (valid after pp_reserve of the total).
struct pp_carver {
  char* base;
  size_t off;
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off = pp_align_up(off + count * sizeof(T), 256);
    return p;
  }
};

// ---------------------------------------------------------------------------------------
// geometry shared with kernels (all index-space; computed on the host in fp64)

struct pp_dims {
  int nx, ny, nz;
};

// out index -> input continuous index:  c = A * idx + b  (+ Md * D(idx) when a field is given)
struct pp_index_map {
  double A[9];
  double b[3];
  double Md[9];  // physical displacement (mm) -> input index units
};

int pp_geom_check(pp_ctx* ctx, const pp_geom* g, const char* what);
bool pp_geom_same_grid(const pp_geom* a, const pp_geom* b);
bool pp_geom_identity_dir(const pp_geom* g);
// c_in = P2I_in * (X(I2P_out * idx + origin_out) - origin_in) with X(p) = A p + t
void pp_make_index_map(const pp_geom* gin, const pp_geom* gout, const double* affine_A,
                       const double* affine_t, pp_index_map* m);

// Gaussian operator taps as floats for a kernel (host side); returns radius or < 0.
#define PP_MAX_RADIUS 127
struct pp_taps {
  int r;
  float w[2 * PP_MAX_RADIUS + 1];
};
int pp_make_taps(pp_ctx* ctx, double variance, double max_error, int max_kernel_width, pp_taps* t);

// Small-radius taps passed by value to the fused kernels.
#define PP_FUSED_MAX_R 5
struct pp_taps_small {
  float h[PP_FUSED_MAX_R + 1];  // symmetric kernel: h[0] = centre, h[k] = taps at distance k
};

// ---------------------------------------------------------------------------------------
// device helpers

__device__ __forceinline__ int pp_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Value of the previous / next lane of the wavefront (DPP wave_shr:1 / wave_shl:1, a VALU move: no LDS crossbar trip).
// Lane 0 / lane 63 keep their own value.  Every lane of the wavefront must execute the call.
#ifndef PP_B_XDPP
#define PP_B_XDPP 1
#endif
__device__ __forceinline__ float pp_lane_prev(float v) {
#if PP_B_XDPP
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x138, 0xf, 0xf, false));
#else
  return __shfl_up(v, 1);
#endif
}
__device__ __forceinline__ float pp_lane_next(float v) {
#if PP_B_XDPP
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x130, 0xf, 0xf, false));
#else
  return __shfl_down(v, 1);
#endif
}

// The value, opaque to the optimiser at this point (an empty asm with the value as a read-write VGPR operand): what is
// derived from it afterwards is computed where it is used instead of being hoisted out of the enclosing loop.  For
// rarely-taken paths whose hoisted per-lane predicates (two scalar registers each) would otherwise be held -- and
// spilled -- across the hot loop.
__device__ __forceinline__ unsigned pp_opaque(unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// Continuous index = base + frac with integer base and frac in [0,1): the inside-buffer test
// of itk::ImageFunction::IsInsideBuffer, [-0.5, n-0.5), done exactly on (base, frac).
__device__ __forceinline__ bool pp_inside1(int b, float f, int n) {
  return (b >= 0 || (b == -1 && f >= 0.5f)) && (b <= n - 2 || (b == n - 1 && f < 0.5f));
}

// One axis of itk::LinearInterpolateImageFunction's clamped lerp: given (base, frac) inside
// the buffer, the two sample indices and the weight (0 where ITK skips the axis).
__device__ __forceinline__ void pp_axis_setup(int b, float f, int n, int& i0, int& i1, float& w) {
  i0 = b < 0 ? 0 : b;
  i1 = i0 + 1 > n - 1 ? n - 1 : i0 + 1;
  w = b < 0 ? 0.0f : f;
}

// Trilinear sample of a scalar volume at (bx+fx, by+fy, bz+fz), already known inside.
// Lerps nest x, y, z in the form a + (b - a) * w, as the reference interpolator does.
template <typename T>
__device__ __forceinline__ float pp_trilinear(const T* __restrict__ im, int nx, int ny, int nz, int bx,
                                              float fx, int by, float fy, int bz, float fz) {
  int x0, x1, y0, y1, z0, z1;
  float wx, wy, wz;
  pp_axis_setup(bx, fx, nx, x0, x1, wx);
  pp_axis_setup(by, fy, ny, y0, y1, wy);
  pp_axis_setup(bz, fz, nz, z0, z1, wz);
  const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
  const T* p00 = im + z0 * sz + y0 * sy;
  const T* p10 = im + z0 * sz + y1 * sy;
  const T* p01 = im + z1 * sz + y0 * sy;
  const T* p11 = im + z1 * sz + y1 * sy;
  const float a000 = (float)p00[x0], a100 = (float)p00[x1];
  const float a010 = (float)p10[x0], a110 = (float)p10[x1];
  const float a001 = (float)p01[x0], a101 = (float)p01[x1];
  const float a011 = (float)p11[x0], a111 = (float)p11[x1];
  const float v00 = a000 + (a100 - a000) * wx;
  const float v10 = a010 + (a110 - a010) * wx;
  const float v01 = a001 + (a101 - a001) * wx;
  const float v11 = a011 + (a111 - a011) * wx;
  const float v0 = v00 + (v10 - v00) * wy;
  const float v1 = v01 + (v11 - v01) * wy;
  return v0 + (v1 - v0) * wz;
}

// The same sample with the two x corners of each row fetched as ONE access of 2 sizeof(T) bytes from an element-aligned
// position (global accesses run in unaligned mode on gfx950): a gather's cost is its memory instructions, and this
// halves them.  The pair starts at min(x0, nx - 2); on the last index -- where ITK's upper corner repeats the lower one
// -- both corners are its second element, so nothing is read past the row.  Needs nx >= 2.  Same lerps, same order.
template <typename T>
struct pp_pair {
  T x, y;
} __attribute__((packed, aligned(sizeof(T))));

template <typename T>
__device__ __forceinline__ float pp_trilinear_pairs(const T* __restrict__ im, int nx, int ny, int nz, int bx,
                                                    float fx, int by, float fy, int bz, float fz) {
  int x0, x1, y0, y1, z0, z1;
  float wx, wy, wz;
  pp_axis_setup(bx, fx, nx, x0, x1, wx);
  pp_axis_setup(by, fy, ny, y0, y1, wy);
  pp_axis_setup(bz, fz, nz, z0, z1, wz);
  const bool xlast = x0 > nx - 2;
  const size_t xs = (size_t)(xlast ? nx - 2 : x0);
  const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
  const pp_pair<T> p00 = *reinterpret_cast<const pp_pair<T>*>(im + z0 * sz + y0 * sy + xs);
  const pp_pair<T> p10 = *reinterpret_cast<const pp_pair<T>*>(im + z0 * sz + y1 * sy + xs);
  const pp_pair<T> p01 = *reinterpret_cast<const pp_pair<T>*>(im + z1 * sz + y0 * sy + xs);
  const pp_pair<T> p11 = *reinterpret_cast<const pp_pair<T>*>(im + z1 * sz + y1 * sy + xs);
  const float a000 = (float)(xlast ? p00.y : p00.x), a100 = (float)p00.y;
  const float a010 = (float)(xlast ? p10.y : p10.x), a110 = (float)p10.y;
  const float a001 = (float)(xlast ? p01.y : p01.x), a101 = (float)p01.y;
  const float a011 = (float)(xlast ? p11.y : p11.x), a111 = (float)p11.y;
  const float v00 = a000 + (a100 - a000) * wx;
  const float v10 = a010 + (a110 - a010) * wx;
  const float v01 = a001 + (a101 - a001) * wx;
  const float v11 = a011 + (a111 - a011) * wx;
  const float v0 = v00 + (v10 - v00) * wy;
  const float v1 = v01 + (v11 - v01) * wy;
  return v0 + (v1 - v0) * wz;
}

// Split idx + dv (dv = displacement in voxels) into integer base and fraction without the
// precision loss of forming the sum in fp32.
__device__ __forceinline__ void pp_split(int idx, float dv, int& b, float& f) {
  if (!(fabsf(dv) < 1.0e6f)) {  // absurd or NaN displacement: force "outside the buffer"
    b = -0x40000000;
    f = 0.0f;
    return;
  }
  const float fl = floorf(dv);
  b = idx + (int)fl;
  f = dv - fl;
}

// Block-wide sum of three doubles (deterministic tree); result valid on thread 0.
template <int NT>
__device__ __forceinline__ void pp_block_sum3(double& a, double& b, double& c, double* sm /* 3*NT */) {
  const int t = threadIdx.x;
  sm[t] = a;
  sm[NT + t] = b;
  sm[2 * NT + t] = c;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (t < s) {
      sm[t] += sm[t + s];
      sm[NT + t] += sm[NT + t + s];
      sm[2 * NT + t] += sm[2 * NT + t + s];
    }
    __syncthreads();
  }
  a = sm[0];
  b = sm[NT];
  c = sm[2 * NT];
}
