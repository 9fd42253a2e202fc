// platipy_amd/csrc/pp_iir.hip -- recursive (IIR) Gaussian smoothing.
//
// Replaces sitk.SmoothingRecursiveGaussian(dvf_total, sigma) (reference:
// platipy/imaging/registration/deformable.py:157-158), i.e. itk::RecursiveGaussianImageFilter:
// Deriche's 4th-order recursive approximation of the Gaussian, run as a causal plus an
// anti-causal pass per line with edge-value extension, along z, then x, then y; results of each
// directional pass are stored as fp32 (ITK's internal images are float), the recursion itself
// runs in fp64.  One thread owns one line; lanes always sit on consecutive x so global
// accesses stay coalesced -- for lines ALONG x the block transposes 256-row x 32-column chunks
// through LDS.  Runs once per pyramid level, not in the inner loop.
// For sigma <= 1.55 voxels (every pipeline setting) a pass is ONE sweep over the line in segments of
// 32 voxels -- causal recursion exact across segments, anti-causal state started 32 voxels ahead
// (k_rg_strided_seg2, k_rg_x_seg; 1.25 ms per 512 x 512 x 256 field = 3.9 TB/s); the two-sweep
// kernels (k_rg_strided, k_rg_x) take wider filters and in-place calls.
#include "pp_internal.h"
#include "pp_kernels.h"

#include <cstdlib>

namespace {

constexpr int NT = 256;

struct rg_coef {
  double n0, n1, n2, n3;
  double d1, d2, d3, d4;
  double m1, m2, m3, m4;
  double kn, km;  // steady-state gains SN/SD and SM/SD (edge extension)
};

// Deriche coefficients with ITK's normalisation (itk::RecursiveGaussianImageFilter::SetUp).  order 0: the Gaussian (unit DC
// gain, symmetric; what SmoothingRecursiveGaussian chains).  order 1: its first derivative (antisymmetric, normalised so that
// a ramp of slope 1 per VOXEL answers 1; `scale` multiplies the response -- sigma in physical units when the caller asked
// for NormalizeAcrossScale, and the sign of the spacing): the directional filter of itk::GradientRecursiveGaussianImageFilter,
// which ImageToImageMetricv4 runs over the moving image (sigma = the largest spacing) for its default gradient source.
void rg_setup(double sigma, double spacing, rg_coef* k, int order = 0, double scale = 1.0) {
  const double sd = sigma / std::fabs(spacing);
  const double W1 = 0.6681, L1 = -1.3932, W2 = 2.0787, L2 = -1.3732;
  const double A1 = order == 0 ? 1.3530 : -0.6724, B1 = order == 0 ? 1.8151 : -3.4327;
  const double A2 = order == 0 ? -0.3531 : 0.6724, B2 = order == 0 ? 0.0902 : 0.6100;
  const double c1 = std::cos(W1 / sd), c2 = std::cos(W2 / sd), s1 = std::sin(W1 / sd), s2 = std::sin(W2 / sd);
  const double e1 = std::exp(L1 / sd), e2 = std::exp(L2 / sd);
  k->d4 = e1 * e1 * e2 * e2;
  k->d3 = -2 * c1 * e1 * e2 * e2 + -2 * c2 * e2 * e1 * e1;
  k->d2 = 4 * c2 * c1 * e1 * e2 + e1 * e1 + e2 * e2;
  k->d1 = -2 * (e2 * c2 + e1 * c1);
  const double SD = 1.0 + k->d1 + k->d2 + k->d3 + k->d4;
  const double DD = k->d1 + 2 * k->d2 + 3 * k->d3 + 4 * k->d4;
  double n0 = A1 + A2;
  double n1 = e2 * (B2 * s2 - (A2 + 2 * A1) * c2) + e1 * (B1 * s1 - (A1 + 2 * A2) * c1);
  double n2 = 2 * e1 * e2 * ((A1 + A2) * c2 * c1 - (B1 * c2 * s1 + B2 * c1 * s2)) + A2 * e1 * e1 + A1 * e2 * e2;
  double n3 = e2 * e1 * e1 * (B2 * s2 - A2 * c2) + e1 * e2 * e2 * (B1 * s1 - A1 * c1);
  const double SN = n0 + n1 + n2 + n3;
  const double DN = n1 + 2 * n2 + 3 * n3;
  double norm;
  if (order == 0) {
    norm = 1.0 / (2 * SN / SD - n0);                       // alpha0
  } else {
    const double alpha1 = 2 * (SN * DD - DN * SD) / (SD * SD);
    norm = scale / alpha1;
  }
  n0 *= norm; n1 *= norm; n2 *= norm; n3 *= norm;
  k->n0 = n0; k->n1 = n1; k->n2 = n2; k->n3 = n3;
  const double sgn = order == 0 ? 1.0 : -1.0;              // ComputeRemainingCoefficients(symmetric = order 0)
  k->m1 = sgn * (n1 - k->d1 * n0);
  k->m2 = sgn * (n2 - k->d2 * n0);
  k->m3 = sgn * (n3 - k->d3 * n0);
  k->m4 = sgn * (-k->d4 * n0);
  k->kn = (n0 + n1 + n2 + n3) / SD;
  k->km = (k->m1 + k->m2 + k->m3 + k->m4) / SD;
}

struct rg_state {
  double x1, x2, x3, x4, y1, y2, y3, y4;
};

__device__ __forceinline__ void rg_init_causal(rg_state& s, double edge, const rg_coef& k) {
  s.x1 = s.x2 = s.x3 = s.x4 = edge;
  s.y1 = s.y2 = s.y3 = s.y4 = edge * k.kn;
}
__device__ __forceinline__ void rg_init_anti(rg_state& s, double edge, const rg_coef& k) {
  s.x1 = s.x2 = s.x3 = s.x4 = edge;
  s.y1 = s.y2 = s.y3 = s.y4 = edge * k.km;
}
// causal: y[i] = n0 x[i] + n1 x[i-1] + n2 x[i-2] + n3 x[i-3] - d1 y[i-1] - ... - d4 y[i-4]
__device__ __forceinline__ double rg_step_causal(rg_state& s, double x, const rg_coef& k) {
  const double y = (k.n0 * x + k.n1 * s.x1 + k.n2 * s.x2 + k.n3 * s.x3) - (k.d1 * s.y1 + k.d2 * s.y2 + k.d3 * s.y3 + k.d4 * s.y4);
  s.x3 = s.x2; s.x2 = s.x1; s.x1 = x;
  s.y4 = s.y3; s.y3 = s.y2; s.y2 = s.y1; s.y1 = y;
  return y;
}
// anti-causal: y[i] = m1 x[i+1] + m2 x[i+2] + m3 x[i+3] + m4 x[i+4] - d1 y[i+1] - ... - d4 y[i+4]
__device__ __forceinline__ double rg_step_anti(rg_state& s, double x, const rg_coef& k) {
  const double y = (k.m1 * s.x1 + k.m2 * s.x2 + k.m3 * s.x3 + k.m4 * s.x4) - (k.d1 * s.y1 + k.d2 * s.y2 + k.d3 * s.y3 + k.d4 * s.y4);
  s.x4 = s.x3; s.x3 = s.x2; s.x2 = s.x1; s.x1 = x;
  s.y4 = s.y3; s.y3 = s.y2; s.y2 = s.y1; s.y1 = y;
  return y;
}

// Lines along y (AXIS 1) or z (AXIS 2): thread = (x, other axis), strided walk.
template <int AXIS>
__global__ void __launch_bounds__(NT) k_rg_strided(const float* __restrict__ in, float* __restrict__ out, pp_dims d,
                                                   size_t cstride, rg_coef k) {
  in += (size_t)blockIdx.y * cstride;
  out += (size_t)blockIdx.y * cstride;
  const int nother = AXIS == 1 ? d.nz : d.ny;
  const size_t nlines = (size_t)d.nx * nother;
  const int len = AXIS == 1 ? d.ny : d.nz;
  const size_t stride = AXIS == 1 ? (size_t)d.nx : (size_t)d.nx * d.ny;
  for (size_t l = (size_t)blockIdx.x * NT + threadIdx.x; l < nlines; l += (size_t)gridDim.x * NT) {
    const int x = (int)(l % d.nx);
    const int o = (int)(l / d.nx);
    const size_t base = AXIS == 1 ? (size_t)o * d.nx * d.ny + x : (size_t)o * d.nx + x;
    rg_state s;
    rg_init_causal(s, (double)in[base], k);
    for (int i = 0; i < len; ++i) out[base + i * stride] = (float)rg_step_causal(s, (double)in[base + i * stride], k);
    rg_init_anti(s, (double)in[base + (size_t)(len - 1) * stride], k);
    for (int i = len - 1; i >= 0; --i) {
      const size_t a = base + i * stride;
      const double xi = (double)in[a];
      const double y = rg_step_anti(s, xi, k);
      out[a] = (float)((double)out[a] + y);
    }
  }
}

// ---- single sweep (round 3) -----------------------------------------------------------------------------------------
// The two-sweep walk above moves 20 bytes per voxel and axis (the anti-causal sweep re-reads the input and the causal
// result); a line is 1-2 KB per thread, so nothing of it stays on chip between the sweeps.  The anti-causal filter's
// response decays like exp(-1.37 k / sd) (sd = sigma in voxels): its state S voxels ahead of a point does not matter there
// once exp(-1.37 S / sd) is below double rounding.  So a thread walks its line ONCE in segments of S voxels held in
// registers: the causal recursion carries its exact state from segment to segment; the anti-causal recursion of a
// segment starts S voxels further on (at the true line end with ITK's edge initialisation when that is nearer), warms up
// over those S voxels -- the next segment, already in registers -- and then produces the segment.  8 bytes per voxel and
// axis.  With S = 32 the warm-up error is < 3e-13 of the signal for sd <= 1.55 (the pipelines' sigma = 1.5 voxels), far
// below the fp32 rounding of the stored result, which is therefore the two-sweep kernel's except for the one-in-millions
// value that sits within 1e-13 of a rounding boundary; larger sd keep the two-sweep kernels.
constexpr int RG_SEG = 32;
constexpr double RG_SEG_MAX_SD = 1.55;   // 1.3732 * 32 / 1.55 = 28.3: exp(-28.3) = 5e-13

template <int AXIS>
__global__ void __launch_bounds__(NT) k_rg_strided_seg(const float* __restrict__ in, float* __restrict__ out, pp_dims d,
                                                       size_t cstride, rg_coef k) {
  constexpr int S = RG_SEG;
  in += (size_t)blockIdx.y * cstride;
  out += (size_t)blockIdx.y * cstride;
  const int nother = AXIS == 1 ? d.nz : d.ny;
  const size_t nlines = (size_t)d.nx * nother;
  const int len = AXIS == 1 ? d.ny : d.nz;
  const size_t stride = AXIS == 1 ? (size_t)d.nx : (size_t)d.nx * d.ny;
  for (size_t l = (size_t)blockIdx.x * NT + threadIdx.x; l < nlines; l += (size_t)gridDim.x * NT) {
    const int x = (int)(l % d.nx);
    const int o = (int)(l / d.nx);
    const size_t base = AXIS == 1 ? (size_t)o * d.nx * d.ny + x : (size_t)o * d.nx + x;
    const float* __restrict__ src = in + base;
    float* __restrict__ dst = out + base;
    float wa[S], wb[S];   // two segments of the line: the one being produced and the one after it (roles alternate)
    auto load = [&](float (&w)[S], int a) {   // w[i] = x[a + i] where that exists
#pragma unroll
      for (int i = 0; i < S; ++i)
        if (a + i < len) w[i] = src[(size_t)(a + i) * stride];
    };
    rg_state sc;   // causal state, exact across segments
    // One segment: `m` = voxels a .. a + S - 1 (those below len), `la` = the S voxels after them.
    auto segment = [&](const float (&m)[S], const float (&la)[S], int a) {
      const int nm = len - a < S ? len - a : S;
      const int nl = len - (a + S) < 0 ? 0 : (len - (a + S) < S ? len - (a + S) : S);
      float c[S];
#pragma unroll
      for (int i = 0; i < S; ++i)
        if (i < nm) c[i] = (float)rg_step_causal(sc, (double)m[i], k);
      // anti-causal start: the last voxel in registers -- the line's last voxel (then this IS ITK's initialisation) or
      // S voxels past the segment (then the state has S voxels to converge)
      float edge = m[0];
#pragma unroll
      for (int i = 0; i < S; ++i) {
        if (i < nm) edge = m[i];
      }
#pragma unroll
      for (int i = 0; i < S; ++i) {
        if (i < nl) edge = la[i];
      }
      rg_state sa;
      rg_init_anti(sa, (double)edge, k);
#pragma unroll
      for (int i = S - 1; i >= 0; --i)
        if (i < nl) (void)rg_step_anti(sa, (double)la[i], k);
#pragma unroll
      for (int i = S - 1; i >= 0; --i)
        if (i < nm) dst[(size_t)(a + i) * stride] = (float)((double)c[i] + rg_step_anti(sa, (double)m[i], k));
    };
    load(wa, 0);
    load(wb, S);
    rg_init_causal(sc, (double)wa[0], k);
    for (int a = 0; a < len; a += 2 * S) {
      segment(wa, wb, a);
      if (a + S >= len) break;
      load(wa, a + 2 * S);
      segment(wb, wa, a + S);
      load(wb, a + 3 * S);
    }
  }
}

// Round 5: the same single sweep at FOUR waves per SIMD instead of one.  k_rg_strided_seg keeps two 32-voxel segments, the
// causal results of a segment and two filter states in registers, its loops fully unrolled: 256 VGPRs (AGPR spills included),
// ONE wave per SIMD -- a wave issues a burst of 32 loads, waits out the HBM latency with nothing else to run, computes, stores:
// 2.6 TB/s on 8 B/voxel, neither the fp64 pipe (~35 % busy) nor HBM anywhere near its limit.  Here the causal results of the
// segment in flight live in LDS (128 B a thread, [i][thread] layout: conflict-free), whole segments run a branch-free body and
// only the line's last, partial segments the guarded one: <= 128 VGPRs, four waves a SIMD.  Same operations in the same order
// per voxel EXCEPT at a line's end: the anti-causal state of the last segments is started up to 2 S - 1 voxels beyond the line on
// the replicated edge value, so a stored float can differ from k_rg_strided_seg's by one rounding there (equal in fp64 to
// < 3e-13 relative; the tests compare the two kernels with that tolerance, not bit for bit: test_recursive_gaussian_*).
// (Addressing: one buffer resource per volume, a 32-bit per-lane byte offset of the line's first voxel and the plane / row
// offset in the instruction's SCALAR offset -- with 64-bit per-element pointers the compiler kept 32 strided offsets alive in
// 64 more registers.  The host takes this kernel only when a component spans < 2^32 bytes.)
__device__ __forceinline__ float rg_bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void rg_bst(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
#ifndef PP_RG_SEG2_BLOCKS
#define PP_RG_SEG2_BLOCKS 2   // resident 256-thread blocks per CU the kernel is compiled for (measured, 512 x 512 x 256 field: 2 -> 1.31 ms, 3 -> 1.44 (168 VGPRs), 4 -> 1.96 (128 VGPRs + 208 B of scratch); the old kernel 1.87)
#endif
// (RG_AFTER: the conversion of sample i may not run ahead of the recursion -- left alone, the scheduler converts a whole
// segment up front and the 32 doubles cost 64 registers.)
#if defined(__HIP_DEVICE_COMPILE__)
#define RG_AFTER(x, dep) asm volatile("" : "+v"(x) : "v"(dep))
#else
#define RG_AFTER(x, dep) ((void)0)
#endif
template <int AXIS>
__global__ void __launch_bounds__(NT, PP_RG_SEG2_BLOCKS) k_rg_strided_seg2(const float* __restrict__ in, float* __restrict__ out, pp_dims d,
                                                           size_t cstride, rg_coef k) {
  constexpr int S = RG_SEG;
  __shared__ float cbuf[S][NT];   // causal result i of this thread's segment: cbuf[i][threadIdx.x]
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)blockIdx.y * cstride), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.y * cstride, 0, -1, 0x00020000);
  const int nother = AXIS == 1 ? d.nz : d.ny;
  const size_t nlines = (size_t)d.nx * nother;
  const int len = AXIS == 1 ? d.ny : d.nz;
  const unsigned stride4 = (AXIS == 1 ? (unsigned)d.nx : (unsigned)d.nx * (unsigned)d.ny) * 4u;   // bytes between a line's voxels
  const int t = threadIdx.x;
  {   // ONE line per thread, the grid covers the lines exactly (inside a grid-stride loop the compiler moves the two segment
      // arrays to scratch: 4.3 KB a lane); no barrier in this kernel, so surplus threads simply leave
    const size_t l = (size_t)blockIdx.x * NT + t;
    if (l >= nlines) return;
    const unsigned x = (unsigned)(l % d.nx);
    const unsigned o = (unsigned)(l / d.nx);
    const unsigned voff = (AXIS == 1 ? o * (unsigned)d.nx * (unsigned)d.ny + x : o * (unsigned)d.nx + x) * 4u;
    float wa[S], wb[S];
    // One body for every segment: voxels past the line's end are read as its LAST voxel (a clamped scalar offset) and never
    // stored.  The anti-causal state that ITK initialises at the line's end -- x = edge, y = edge * km, the filter's steady
    // state for a constant -- is then initialised up to 2 S - 1 voxels further on, on the replicated edge, and stays at that
    // steady state until it reaches the line (to double rounding: the stored floats are those of the guarded form but for
    // the one-in-millions value at a rounding boundary).
    auto load = [&](float (&w)[S], int a) __attribute__((always_inline)) {
      if (a + S <= len) {
        const unsigned s0 = (unsigned)a * stride4;
#pragma unroll
        for (int i = 0; i < S; ++i) w[i] = rg_bld(r_in, voff, s0 + (unsigned)i * stride4);
      } else {
#pragma unroll
        for (int i = 0; i < S; ++i) w[i] = rg_bld(r_in, voff, (unsigned)(a + i < len ? a + i : len - 1) * stride4);
      }
    };
    rg_state sc;
    auto segment = [&](const float (&m)[S], const float (&la)[S], int a) __attribute__((always_inline)) {
      const unsigned s0 = (unsigned)a * stride4;
      const bool whole = a + S <= len;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        double xd = (double)m[i];
        RG_AFTER(xd, sc.y1);
        cbuf[i][t] = (float)rg_step_causal(sc, xd, k);
      }
      rg_state sa;
      rg_init_anti(sa, (double)la[S - 1], k);
#pragma unroll
      for (int i = S - 1; i >= 0; --i) {
        double xd = (double)la[i];
        RG_AFTER(xd, sa.y1);
        (void)rg_step_anti(sa, xd, k);
      }
#pragma unroll
      for (int i = S - 1; i >= 0; --i) {
        double xd = (double)m[i];
        RG_AFTER(xd, sa.y1);
        const float y = (float)((double)cbuf[i][t] + rg_step_anti(sa, xd, k));
        if (whole || a + i < len) rg_bst(r_out, voff, s0 + (unsigned)i * stride4, y);
      }
    };
    load(wa, 0);
    load(wb, S);
    rg_init_causal(sc, (double)wa[0], k);
    for (int a = 0; a < len; a += 2 * S) {
      segment(wa, wb, a);
      if (a + S >= len) break;
      load(wa, a + 2 * S);
      segment(wb, wa, a + S);
      load(wb, a + 3 * S);
    }
  }
}

// Lines along x: a block owns 256 consecutive rows and walks them in 16-column chunks that are
// transposed through LDS (pitch 17 keeps the per-row accesses conflict-free).  VEC4: rows are 16-byte aligned
// (nx % 4 == 0, aligned base), so the chunk is moved with one 16-byte access per lane and 4 columns -- a quarter of the
// memory instructions of the scalar mover; the arithmetic is the same.
constexpr int CW = 16;
template <bool VEC4>
__global__ void __launch_bounds__(NT) k_rg_x(const float* __restrict__ in, float* __restrict__ out, pp_dims d,
                                             size_t cstride, rg_coef k) {
  __shared__ float tile[NT * (CW + 1)];
  __shared__ float tcau[NT * (CW + 1)];
  in += (size_t)blockIdx.y * cstride;
  out += (size_t)blockIdx.y * cstride;
  const size_t nrows = (size_t)d.ny * d.nz;
  const int t = threadIdx.x;
  // mover coordinates: VEC4: column group lc (4 columns), rows lr + 64 j;  scalar: column lc, rows lr + 16 j
  constexpr int LPR = VEC4 ? CW / 4 : CW;      // lanes per row
  constexpr int RPI = NT / LPR;                // rows per mover round
  const int lc = t % LPR, lr = t / LPR;
  const int nchunks = (d.nx + CW - 1) / CW;
  auto fetch = [&](const float* __restrict__ src, float* __restrict__ dst, size_t r0, int c0) {
#pragma unroll
    for (int j = 0; j < NT / RPI; ++j) {
      const int rr = lr + RPI * j;
      const size_t row = r0 + rr;
      if (VEC4) {
        if (row < nrows && c0 + 4 * lc < d.nx) {
          const float4 v = *reinterpret_cast<const float4*>(src + row * d.nx + c0 + 4 * lc);
          float* p = dst + rr * (CW + 1) + 4 * lc;
          p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        }
      } else {
        if (row < nrows && c0 + lc < d.nx) dst[rr * (CW + 1) + lc] = src[row * d.nx + c0 + lc];
      }
    }
  };
  auto put = [&](const float* __restrict__ src, size_t r0, int c0) {
#pragma unroll
    for (int j = 0; j < NT / RPI; ++j) {
      const int rr = lr + RPI * j;
      const size_t row = r0 + rr;
      if (VEC4) {
        if (row < nrows && c0 + 4 * lc < d.nx) {
          const float* p = src + rr * (CW + 1) + 4 * lc;
          *reinterpret_cast<float4*>(out + row * d.nx + c0 + 4 * lc) = make_float4(p[0], p[1], p[2], p[3]);
        }
      } else {
        if (row < nrows && c0 + lc < d.nx) out[row * d.nx + c0 + lc] = src[rr * (CW + 1) + lc];
      }
    }
  };
  for (size_t r0 = (size_t)blockIdx.x * NT; r0 < nrows; r0 += (size_t)gridDim.x * NT) {
    const size_t myrow = r0 + t;
    const bool have = myrow < nrows;
    rg_state s;
    if (have) rg_init_causal(s, (double)in[myrow * d.nx], k);
    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = ch * CW;
      __syncthreads();
      fetch(in, tile, r0, c0);
      __syncthreads();
      if (have) {
        const int w = d.nx - c0 < CW ? d.nx - c0 : CW;
        for (int c = 0; c < w; ++c) tile[t * (CW + 1) + c] = (float)rg_step_causal(s, (double)tile[t * (CW + 1) + c], k);
      }
      __syncthreads();
      put(tile, r0, c0);
    }
    if (have) rg_init_anti(s, (double)in[myrow * d.nx + d.nx - 1], k);
    for (int ch = nchunks - 1; ch >= 0; --ch) {
      const int c0 = ch * CW;
      __syncthreads();
      fetch(in, tile, r0, c0);
      fetch(out, tcau, r0, c0);
      __syncthreads();
      if (have) {
        const int w = d.nx - c0 < CW ? d.nx - c0 : CW;
        for (int c = w - 1; c >= 0; --c) {
          const double y = rg_step_anti(s, (double)tile[t * (CW + 1) + c], k);
          tcau[t * (CW + 1) + c] = (float)((double)tcau[t * (CW + 1) + c] + y);
        }
      }
      __syncthreads();
      put(tcau, r0, c0);
    }
  }
}

// Lines along x, single sweep: the segment scheme of k_rg_strided_seg with the two segments of a row in LDS (a block owns 256
// consecutive rows; 32-column chunks are transposed through LDS by the 16-byte mover, pitch 33 keeps the per-row walks
// conflict-free).  Per segment: causal recursion over the main half (results in registers), anti-causal warm-up over the
// other half, anti-causal over the main half with the sum written back into the tile, tile -> global, next chunk in.
struct rg_f4u {
  float x, y, z, w;
} __attribute__((aligned(4)));   // 16 bytes at 4-byte alignment (global_load / store_dwordx4 take it)
template <bool VEC4>
__global__ void __launch_bounds__(NT) k_rg_x_seg(const float* __restrict__ in, float* __restrict__ out, pp_dims d, size_t cstride, rg_coef k) {
  constexpr int S = RG_SEG, P = S + 1;
  __shared__ float half[2][NT * P];
  in += (size_t)blockIdx.y * cstride;
  out += (size_t)blockIdx.y * cstride;
  const size_t nrows = (size_t)d.ny * d.nz;
  const int t = threadIdx.x, len = d.nx;
  constexpr int LPR = S / 4;                   // lanes per row of the mover: a lane moves a quad of columns (VEC4: rows are 16-byte
                                               // aligned; otherwise 16-byte accesses at 4-byte alignment and element-wise row ends)
  constexpr int RPI = NT / LPR;                // rows per mover round
  const int lc = t % LPR, lr = t / LPR;
  auto load_quad = [&](size_t row, int cq, float (&q)[4]) {   // columns cq .. cq + 3 of a row (zeros past the row's end)
    q[0] = q[1] = q[2] = q[3] = 0.0f;
    if (row < nrows && cq + 3 < len) {
      if (VEC4) {
        const float4 v = *reinterpret_cast<const float4*>(in + row * len + cq);
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
      } else {
        const rg_f4u v = *reinterpret_cast<const rg_f4u*>(in + row * len + cq);
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
      }
    } else if (row < nrows) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cq + e < len) q[e] = in[row * len + cq + e];
    }
  };
  auto fetch = [&](float* __restrict__ dst, size_t r0, int c0) {
#pragma unroll
    for (int j = 0; j < NT / RPI; ++j) {
      const int rr = lr + RPI * j;
      float q[4];
      load_quad(r0 + rr, c0 + 4 * lc, q);
      float* p = dst + rr * P + 4 * lc;
      p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; p[3] = q[3];
    }
  };
  auto put = [&](const float* __restrict__ src, size_t r0, int c0) {
#pragma unroll
    for (int j = 0; j < NT / RPI; ++j) {
      const int rr = lr + RPI * j;
      const size_t row = r0 + rr;
      const int cq = c0 + 4 * lc;
      const float* p = src + rr * P + 4 * lc;
      if (row < nrows && cq + 3 < len) {
        if (VEC4) {
          *reinterpret_cast<float4*>(out + row * len + cq) = make_float4(p[0], p[1], p[2], p[3]);
        } else {
          rg_f4u v;
          v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
          *reinterpret_cast<rg_f4u*>(out + row * len + cq) = v;
        }
      } else if (row < nrows) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cq + e < len) out[row * len + cq + e] = p[e];
      }
    }
  };
  for (size_t r0 = (size_t)blockIdx.x * NT; r0 < nrows; r0 += (size_t)gridDim.x * NT) {
    const bool have = r0 + t < nrows;
    __syncthreads();
    fetch(half[0], r0, 0);
    fetch(half[1], r0, S);
    __syncthreads();
    rg_state sc;
    if (have) rg_init_causal(sc, (double)half[0][t * P], k);
    int cur = 0;
    for (int a = 0; a < len; a += S, cur ^= 1) {
      float* const m = half[cur] + t * P;
      const float* const la = half[cur ^ 1] + t * P;
      // Round 5: the chunk after next is REQUESTED here, ahead of this chunk's arithmetic, and lands in registers; it goes into
      // the tile this chunk frees once that has been written out.  (It used to be fetched behind the write-out, between two
      // barriers, with nothing to hide its latency.)
      const bool more = a + 2 * S < len;
      float pre[NT / RPI][4];
      if (more) {
#pragma unroll
        for (int j = 0; j < NT / RPI; ++j) load_quad(r0 + lr + RPI * j, a + 2 * S + 4 * lc, pre[j]);
      }
      if (have) {
        const int nm = len - a < S ? len - a : S;
        const int nl = len - (a + S) < 0 ? 0 : (len - (a + S) < S ? len - (a + S) : S);
        float c[S];
#pragma unroll
        for (int i = 0; i < S; ++i)
          if (i < nm) {
            double xd = (double)m[i];
            RG_AFTER(xd, sc.y1);
            c[i] = (float)rg_step_causal(sc, xd, k);
          }
        const float edge = nl > 0 ? la[nl - 1] : m[nm - 1];
        rg_state sa;
        rg_init_anti(sa, (double)edge, k);
        for (int i = nl - 1; i >= 0; --i) (void)rg_step_anti(sa, (double)la[i], k);
#pragma unroll
        for (int i = S - 1; i >= 0; --i)
          if (i < nm) {
            double xd = (double)m[i];
            RG_AFTER(xd, sa.y1);
            m[i] = (float)((double)c[i] + rg_step_anti(sa, xd, k));
          }
      }
      __syncthreads();
      put(half[cur], r0, a);
      __syncthreads();
      if (more) {
#pragma unroll
        for (int j = 0; j < NT / RPI; ++j) {
          float* q = half[cur] + (lr + RPI * j) * P + 4 * lc;
          q[0] = pre[j][0]; q[1] = pre[j][1]; q[2] = pre[j][2]; q[3] = pre[j][3];
        }
      }
      __syncthreads();
    }
  }
}

unsigned grid_for(size_t work) {
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > 65535u) blocks = 65535u;
  if (const char* e = pp_env("PP_RG_GRID")) {   // (measurement knob: fewer resident lines -> the causal results are re-read from cache)
    const size_t cap = (size_t)atoi(e);
    if (cap > 0 && blocks > cap) blocks = cap;
  }
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

int rg_pass(pp_ctx* ctx, int axis, const float* in, float* out, const pp_dims& d, int ncomp, double sigma, double spacing, int order = 0,
            double scale = 1.0) {
  const int len = axis == 0 ? d.nx : (axis == 1 ? d.ny : d.nz);
  if (len < 4) return pp_fail(ctx, PP_ERR_SIZE, "recursive Gaussian needs at least 4 voxels along axis %d (got %d)", axis, len);
  if (!(sigma > 0.0)) return pp_fail(ctx, PP_ERR_ARG, "recursive Gaussian: sigma must be positive");
  rg_coef k;
  rg_setup(sigma, spacing, &k, order, scale);
  const size_t cstride = (size_t)d.nx * d.ny * d.nz;
  if (axis == 0) {
    const bool v4 = d.nx % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) && cstride % 4 == 0;
    const bool seg = sigma / std::fabs(spacing) <= RG_SEG_MAX_SD && in != out && pp_env("PP_RG_TWO_SWEEP") == nullptr;
    if (seg && v4)
      hipLaunchKernelGGL(k_rg_x_seg<true>, dim3(grid_for((size_t)d.ny * d.nz), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
    else if (seg)
      hipLaunchKernelGGL(k_rg_x_seg<false>, dim3(grid_for((size_t)d.ny * d.nz), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
    else if (v4)
      hipLaunchKernelGGL(k_rg_x<true>, dim3(grid_for((size_t)d.ny * d.nz), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
    else
      hipLaunchKernelGGL(k_rg_x<false>, dim3(grid_for((size_t)d.ny * d.nz), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
  } else {
    const bool seg = sigma / std::fabs(spacing) <= RG_SEG_MAX_SD && in != out && pp_env("PP_RG_TWO_SWEEP") == nullptr;
    const bool small = cstride * sizeof(float) < ((size_t)1 << 32);   // k_rg_strided_seg2 addresses a component with 32-bit byte offsets
    if (axis == 1) {
      if (seg && small && !pp_env("PP_RG_SEG_V1")) hipLaunchKernelGGL((k_rg_strided_seg2<1>), dim3((unsigned)(((size_t)d.nx * d.nz + NT - 1) / NT), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
      else if (seg) hipLaunchKernelGGL((k_rg_strided_seg<1>), dim3(grid_for((size_t)d.nx * d.nz), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
      else hipLaunchKernelGGL((k_rg_strided<1>), dim3(grid_for((size_t)d.nx * d.nz), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
    } else {
      if (seg && small && !pp_env("PP_RG_SEG_V1")) hipLaunchKernelGGL((k_rg_strided_seg2<2>), dim3((unsigned)(((size_t)d.nx * d.ny + NT - 1) / NT), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
      else if (seg) hipLaunchKernelGGL((k_rg_strided_seg<2>), dim3(grid_for((size_t)d.nx * d.ny), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
      else hipLaunchKernelGGL((k_rg_strided<2>), dim3(grid_for((size_t)d.nx * d.ny), ncomp), dim3(NT), 0, ctx->stream, in, out, d, cstride, k);
    }
  }
  PP_LAUNCH_CHECK(ctx, "k_rg");
  return PP_OK;
}

// z -> x -> y, as SmoothingRecursiveGaussianImageFilter chains its directional filters.
int rg_all(pp_ctx* ctx, const float* in, float* out, float* tmp1, float* tmp2, const pp_geom* g, int ncomp, const double sigma[3]) {
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  int rc = rg_pass(ctx, 2, in, tmp1, d, ncomp, sigma[2], g->spacing[2]);
  if (rc) return rc;
  rc = rg_pass(ctx, 0, tmp1, tmp2, d, ncomp, sigma[0], g->spacing[0]);
  if (rc) return rc;
  return rg_pass(ctx, 1, tmp2, out, d, ncomp, sigma[1], g->spacing[1]);
}

}  // namespace

extern "C" {

int pp_recursive_gaussian_field_f32(pp_ctx* ctx, float* field, const pp_geom* g, const double sigma[3]) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, field && sigma, "pp_recursive_gaussian_field_f32: NULL argument");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  const size_t N = pp_nvox(g->size);
  rc = pp_reserve(ctx, 2 * pp_align_up(3 * N * sizeof(float), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* t1 = cv.take<float>(3 * N);
  float* t2 = cv.take<float>(3 * N);
  return rg_all(ctx, field, field, t1, t2, g, 3, sigma);
}

int pp_recursive_gaussian_pass_f32(pp_ctx* ctx, const float* in, float* out, const pp_geom* g, int axis, double sigma, int order,
                                   int normalize_across_scale) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && in != out, "pp_recursive_gaussian_pass_f32: two distinct volumes");
  PP_REQUIRE(ctx, axis >= 0 && axis < 3 && (order == 0 || order == 1), "pp_recursive_gaussian_pass_f32: axis 0..2, order 0 or 1");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  // first order: the response to a unit ramp per voxel is 1; ITK multiplies it by sigma (physical) when NormalizeAcrossScale
  // is on and by the sign of the spacing
  const double scale = order == 1 ? (normalize_across_scale ? sigma : 1.0) * (g->spacing[axis] < 0.0 ? -1.0 : 1.0) : 1.0;
  return rg_pass(ctx, axis, in, out, d, 1, sigma, g->spacing[axis], order, scale);
}

int pp_recursive_gaussian_f32(pp_ctx* ctx, const float* in, float* out, const pp_geom* g, const double sigma[3]) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && sigma, "pp_recursive_gaussian_f32: NULL argument");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  const size_t N = pp_nvox(g->size);
  rc = pp_reserve(ctx, 2 * pp_align_up(N * sizeof(float), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* t1 = cv.take<float>(N);
  float* t2 = cv.take<float>(N);
  return rg_all(ctx, in, out, t1, t2, g, 1, sigma);
}

}  // extern "C"
