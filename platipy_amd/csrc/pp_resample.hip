// platipy_amd/csrc/pp_resample.hip -- trilinear / nearest-neighbour gathers.
//
// Replaces itk::WarpImageFilter inside the demons loop, sitk.Resample / ResampleImageFilter
// (reference: registration/utils.py:176-190, :257-267; registration/deformable.py:130,137,140,
// 154,185,281-301) and itk::DisplacementFieldTransform.  All kernels are gathers: lanes run along x
// so the displacement planes, the output and -- for smooth fields -- the gathered neighbours are
// row-coalesced.  What bounds them is the number of gather instructions (an 8-byte corner pair costs
// ~15 clocks per wavefront and CU on consecutive lanes, ~30 otherwise: profiles/round5_tabench.txt) and,
// for the fp64 coordinate path, instruction issue -- not HBM bytes.
//
// Two coordinate paths:
//  * same-grid (hot loop, compose): continuous index = idx + D / spacing, formed as an integer
//    base plus an fp32 fraction so no precision is lost to the magnitude of idx;
//  * general (pyramids, linear transforms, propagation of masks): the ITK sequence
//    index -> physical -> transform -> physical -> continuous index, evaluated in fp64 with
//    contraction off, so nearest-neighbour decisions match the fp64 restatement bit for bit.  On
//    axis-aligned grids (every pipeline grid) k_resample_axis / k_resample_field_march evaluate the
//    same sequence without its exact-zero terms (round 5, below).
// Launch geometry: grid3_for (block width by least row padding) and, for the gathers through a field,
// the XCD-banded order of band_for.
#include "pp_internal.h"
#include "pp_kernels.h"
#include "pp_warp_sample.h"

namespace {

constexpr int NT = 256;

// Launch geometry of the per-voxel kernels: thread (x, y) of a (bx, 256 / bx) block, grid (x blocks, y blocks, z) -- no
// 64-bit division per voxel to recover (x, y, z) from a linear index.  bx = 64 / 128 / 256 lanes along x by row length.
// (Round 5: it was 256 for every row longer than 128 -- a third of the lanes idle at 341.)
struct pp_grid3 {
  dim3 grid, block;
};
pp_grid3 grid3_for(int nxv, int ny, int nz, unsigned bx_max = 256u) {
  // the widest of 64 / 128 / 256 lanes along x that pads the row least (341 voxels: 128 -> 384 lanes, not 256 -> 512)
  unsigned bx = 64u;
  for (unsigned c = 128u; c <= 256u && c <= bx_max; c *= 2u)
    if ((unsigned)((nxv + c - 1) / c) * c <= (unsigned)((nxv + bx - 1) / bx) * bx) bx = c;
  const unsigned by = NT / bx;
  pp_grid3 g;
  g.block = dim3(bx, by, 1);
  g.grid = dim3((unsigned)((nxv + bx - 1) / bx), (unsigned)((ny + by - 1) / by), (unsigned)nz);
  return g;
}

// The same blocks handed out XCD-aware.  Hardware sends block b of a launch to XCD b % 8, so with the plain (x, y, z) grid
// the rows y and y + 1 of a plane -- whose gathers read the same rows of the source -- run on different XCDs, every XCD
// walks every plane, and each L2 fetches what its neighbours also fetch (measured: the composition fetched 84 B / voxel
// from HBM / MALL for 24 compulsory, the warp through a field 25 for 16; banded 27.6 and 16.1 -- tools/r5/reg_pmc.sh).  The
// time follows only where the field is rough enough for the fetch to matter (sbench's composition 1.40 -> 1.01 ms; inside
// config 2's registration, whose finest-level field is smooth, the composition stays at 0.89 ms: it is bound by its
// twelve gather instructions per voxel, ~30 clocks each once lanes are not exactly consecutive).  Banded: a 1-D launch of 8 x per x nz blocks;
// block b works on plane (b / 8) / per and, within it, on tile (b % 8) x per + (b / 8) % per of the plane's gx x gy tiles --
// each XCD owns a band of rows of every plane, planes still go by in order (two planes of DRAM pages at a time).
struct pp_band {
  unsigned gx, gy, per, tiles;   // per == 0: plain 3-D grid
};
pp_band band_for(const pp_grid3& g, dim3* launch, bool wanted = true) {
  pp_band b{g.grid.x, g.grid.y, 0u, g.grid.x * g.grid.y};
  const char* e = pp_env("PP_RS_BAND");
  const bool on = wanted && !(e && atoi(e) == 0);
  if (on && b.tiles >= 64u && (size_t)((b.tiles + 7u) / 8u) * 8u * g.grid.z < ((size_t)1 << 31)) {
    b.per = (b.tiles + 7u) / 8u;
    *launch = dim3(8u * b.per * g.grid.z, 1, 1);
  } else {
    *launch = g.grid;
  }
  return b;
}
__device__ __forceinline__ bool pp_band_block(const pp_band& B, unsigned& bx, unsigned& by, unsigned& bz) {
  if (B.per == 0u) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    return true;
  }
  const unsigned b = blockIdx.x, q = b >> 3, t = (b & 7u) * B.per + q % B.per;
  bz = q / B.per;
  bx = t % B.gx;
  by = t / B.gx;
  return t < B.tiles;
}

// ---------------------------------------------------------------------------------------
// same-grid warp

// Straight-line version (pp_warp_sample.h): no branch per sample, 32-bit offsets, and the 4 x VEC corner loads of a thread
// are all issued before the first interpolation.  Same arithmetic as k_warp_same_grid below, which stays for volumes
// outside the helper's preconditions.
template <int VEC>
__global__ void __launch_bounds__(NT) k_warp_same_grid_sl(const float* __restrict__ moving, const float* __restrict__ field,
                                                          float* __restrict__ out, pp_dims d, pp_warp_scale sc, float edge,
                                                          const int* __restrict__ halt, pp_band B) {
  if (halt && *halt) return;
  const int nxv = d.nx / VEC;
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  unsigned bx_, by_, bz_;
  if (!pp_band_block(B, bx_, by_, bz_)) return;
  const int xv = bx_ * blockDim.x + threadIdx.x, y = by_ * blockDim.y + threadIdx.y, z = bz_;
  if (xv < nxv && y < d.ny) {
    const int x0 = xv * VEC;
    const size_t i = ((size_t)z * d.ny + y) * d.nx + x0;
    float dx[VEC], dy[VEC], dz[VEC], res[VEC];
    if (VEC == 4) {
      const float4 a = *reinterpret_cast<const float4*>(field + i);
      const float4 b = *reinterpret_cast<const float4*>(field + N + i);
      const float4 c = *reinterpret_cast<const float4*>(field + 2 * N + i);
      dx[0] = a.x; dx[1] = a.y; dx[2] = a.z; dx[3] = a.w;
      dy[0] = b.x; dy[1] = b.y; dy[2] = b.z; dy[3] = b.w;
      dz[0] = c.x; dz[1] = c.y; dz[2] = c.z; dz[3] = c.w;
    } else {
      dx[0] = field[i];
      dy[0] = field[N + i];
      dz[0] = field[2 * N + i];
    }
    const pp_warp_dims wd{d.nx, d.ny, d.nz, (unsigned)d.nx * 4u, (unsigned)d.nx * (unsigned)d.ny * 4u};
    const char* const rm = reinterpret_cast<const char*>(moving);
    pp_warp_pending g[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) fused2_warp_issue(rm, wd, x0 + v, dx[v] * sc.ix, y, dy[v] * sc.iy, z, dz[v] * sc.iz, true, g[v]);
#pragma unroll
    for (int v = 0; v < VEC; ++v) res[v] = fused2_warp_finish_edge(g[v], edge);
    if (VEC == 4)
      *reinterpret_cast<float4*>(out + i) = make_float4(res[0], res[1], res[2], res[3]);
    else
      out[i] = res[0];
  }
}

template <int VEC>
__global__ void __launch_bounds__(NT) k_warp_same_grid(const float* __restrict__ moving, const float* __restrict__ field,
                                                       float* __restrict__ out, pp_dims d, pp_warp_scale sc, float edge,
                                                       const int* __restrict__ halt) {
  if (halt && *halt) return;
  const int nxv = d.nx / VEC;
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  const int xv = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, z = blockIdx.z;
  if (xv < nxv && y < d.ny) {
    const int x0 = xv * VEC;
    const size_t i = ((size_t)z * d.ny + y) * d.nx + x0;
    float dx[VEC], dy[VEC], dz[VEC], res[VEC];
    if (VEC == 4) {
      const float4 a = *reinterpret_cast<const float4*>(field + i);
      const float4 b = *reinterpret_cast<const float4*>(field + N + i);
      const float4 c = *reinterpret_cast<const float4*>(field + 2 * N + i);
      dx[0] = a.x; dx[1] = a.y; dx[2] = a.z; dx[3] = a.w;
      dy[0] = b.x; dy[1] = b.y; dy[2] = b.z; dy[3] = b.w;
      dz[0] = c.x; dz[1] = c.y; dz[2] = c.z; dz[3] = c.w;
    } else {
      dx[0] = field[i];
      dy[0] = field[N + i];
      dz[0] = field[2 * N + i];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      int bx, by, bz;
      float fx, fy, fz;
      pp_split(x0 + v, dx[v] * sc.ix, bx, fx);
      pp_split(y, dy[v] * sc.iy, by, fy);
      pp_split(z, dz[v] * sc.iz, bz, fz);
      const bool inside = pp_inside1(bx, fx, d.nx) && pp_inside1(by, fy, d.ny) && pp_inside1(bz, fz, d.nz);
      res[v] = inside ? (d.nx >= 2 ? pp_trilinear_pairs(moving, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz)
                                   : pp_trilinear(moving, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz)) : edge;
    }
    if (VEC == 4)
      *reinterpret_cast<float4*>(out + i) = make_float4(res[0], res[1], res[2], res[3]);
    else
      out[i] = res[0];
  }
}

// total(x) += iter(x + total(x)), zero outside (deformable.py:154).  In place on `total`:
// every thread reads only its own voxel of `total`.
__global__ void __launch_bounds__(NT) k_compose_same_grid(float* __restrict__ total, const float* __restrict__ iter,
                                                          pp_dims d, pp_warp_scale sc, pp_band B) {
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  unsigned bx_, by_, bz_;
  if (!pp_band_block(B, bx_, by_, bz_)) return;
  const int x = bx_ * blockDim.x + threadIdx.x, y = by_ * blockDim.y + threadIdx.y, z = bz_;
  if (x < d.nx && y < d.ny) {
    const size_t i = ((size_t)z * d.ny + y) * d.nx + x;
    const float tx = total[i], ty = total[N + i], tz = total[2 * N + i];
    int bx, by, bz;
    float fx, fy, fz;
    pp_split(x, tx * sc.ix, bx, fx);
    pp_split(y, ty * sc.iy, by, fy);
    pp_split(z, tz * sc.iz, bz, fz);
    if (pp_inside1(bx, fx, d.nx) && pp_inside1(by, fy, d.ny) && pp_inside1(bz, fz, d.nz)) {
      if (d.nx >= 2) {
        total[i] = tx + pp_trilinear_pairs(iter, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz);
        total[N + i] = ty + pp_trilinear_pairs(iter + N, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz);
        total[2 * N + i] = tz + pp_trilinear_pairs(iter + 2 * N, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz);
      } else {
        total[i] = tx + pp_trilinear(iter, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz);
        total[N + i] = ty + pp_trilinear(iter + N, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz);
        total[2 * N + i] = tz + pp_trilinear(iter + 2 * N, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// general resample, fp64 coordinates

struct pp_xform {
  double i2p_out[9], o_out[3];
  double A[9], t[3];
  double p2i_in[9], o_in[3];
  int has_affine;
  int diag;   // host side: both grids axis-aligned, no linear transform (the marching kernels' case)
  int axis;   // host side: both grids axis-aligned (k_resample_axis, with or without a linear transform)
};

// itk::ImageBase::TransformIndexToPhysicalPoint -> MatrixOffsetTransformBase::TransformPoint ->
// DisplacementFieldTransform::TransformPoint -> TransformPhysicalPointToContinuousIndex, with
// each sum in the order ITK writes it.
__device__ __forceinline__ void pp_map_point(const pp_xform& X, int x, int y, int z, double ddx, double ddy, double ddz,
                                             double c[3]) {
#pragma clang fp contract(off)
  const double ix = (double)x, iy = (double)y, iz = (double)z;
  double p[3], q[3], v[3];
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    s += X.i2p_out[r * 3 + 0] * ix;
    s += X.i2p_out[r * 3 + 1] * iy;
    s += X.i2p_out[r * 3 + 2] * iz;
    p[r] = s + X.o_out[r];
  }
  if (X.has_affine) {
    for (int r = 0; r < 3; ++r) {
      double s = X.A[r * 3 + 0] * p[0] + X.A[r * 3 + 1] * p[1];
      s = s + X.A[r * 3 + 2] * p[2];
      q[r] = s + X.t[r];
    }
  } else {
    q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
  }
  q[0] += ddx;
  q[1] += ddy;
  q[2] += ddz;
  for (int r = 0; r < 3; ++r) v[r] = q[r] - X.o_in[r];
  for (int r = 0; r < 3; ++r) {
    double s = X.p2i_in[r * 3 + 0] * v[0] + X.p2i_in[r * 3 + 1] * v[1];
    c[r] = s + X.p2i_in[r * 3 + 2] * v[2];
  }
}

__device__ __forceinline__ bool pp_inside_d(const double c[3], const pp_dims& n) {
  return (c[0] >= -0.5 && c[0] < (double)n.nx - 0.5) && (c[1] >= -0.5 && c[1] < (double)n.ny - 0.5) &&
         (c[2] >= -0.5 && c[2] < (double)n.nz - 0.5);
}

template <typename T>
__device__ __forceinline__ T pp_cast_out(float v);
template <>
__device__ __forceinline__ float pp_cast_out<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint8_t pp_cast_out<uint8_t>(float v) {
  return v < 0.0f ? (uint8_t)0 : (v > 255.0f ? (uint8_t)255 : (uint8_t)v);  // clamp, then truncate
}

// itk::BSplineInterpolateImageFunction, spline order 3 (sitkBSpline): 4 x 4 x 4 cubic B-spline weights on the coefficient
// volume produced by the prefilter below; neighbour indices beyond the buffer are mirrored about the first / last voxel
// (period 2 n - 2), as ITK's DetermineRegionOfSupport / ApplyMirrorBoundaryConditions do.
__device__ __forceinline__ void pp_bspline3_axis(double c, int n, int idx[4], double w[4]) {
  const double fl = floor(c);
  const int i1 = (int)fl;
  const double t = c - fl;
  w[3] = (1.0 / 6.0) * t * t * t;
  w[0] = (1.0 / 6.0) + 0.5 * t * (t - 1.0) - w[3];
  w[2] = t + w[0] - 2.0 * w[3];
  w[1] = 1.0 - w[0] - w[2] - w[3];
  const int n2 = 2 * n - 2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = i1 - 1 + k;
    if (n == 1) {
      i = 0;
    } else {
      if (i < 0) i = -i - n2 * ((-i) / n2);
      else i = i - n2 * (i / n2);
      if (i >= n) i = n2 - i;
    }
    idx[k] = i;
  }
}
__device__ __forceinline__ float pp_bspline3_sample(const float* __restrict__ coef, const pp_dims& n, const double c[3]) {
  int ix[4], iy[4], iz[4];
  double wx[4], wy[4], wz[4];
  pp_bspline3_axis(c[0], n.nx, ix, wx);
  pp_bspline3_axis(c[1], n.ny, iy, wy);
  pp_bspline3_axis(c[2], n.nz, iz, wz);
  double acc = 0.0;
  for (int kz = 0; kz < 4; ++kz) {
    double az = 0.0;
    for (int ky = 0; ky < 4; ++ky) {
      const float* row = coef + ((size_t)iz[kz] * n.ny + iy[ky]) * n.nx;
      const double ay = wx[0] * (double)row[ix[0]] + wx[1] * (double)row[ix[1]] + wx[2] * (double)row[ix[2]] + wx[3] * (double)row[ix[3]];
      az += wy[ky] * ay;
    }
    acc += wz[kz] * az;
  }
  return (float)acc;
}

// itk::BSplineDecompositionImageFilter for spline order 3 along one axis, in place: the recursive (causal + anti-causal)
// filter with pole sqrt(3) - 2 and gain 6 that turns samples into B-spline coefficients, mirror boundaries (Unser's
// initialisation: the causal sum over a horizon where the pole's power falls below 1e-10, or the exact closed form on
// short lines).  One thread per line, fp64 recursion, fp32 storage; lines are `stride` apart element to element.
__global__ void __launch_bounds__(NT) k_bspline3_prefilter(float* __restrict__ data, pp_dims d, int axis) {
  const int len = axis == 0 ? d.nx : (axis == 1 ? d.ny : d.nz);
  const size_t stride = axis == 0 ? 1 : (axis == 1 ? (size_t)d.nx : (size_t)d.nx * d.ny);
  const size_t nlines = (size_t)d.nx * d.ny * d.nz / len;
  const double z = -0.26794919243112270647;   // sqrt(3) - 2
  for (size_t l = (size_t)blockIdx.x * NT + threadIdx.x; l < nlines; l += (size_t)gridDim.x * NT) {
    size_t base;
    if (axis == 0) base = l * (size_t)d.nx;
    else if (axis == 1) base = (l / d.nx) * (size_t)d.nx * d.ny + (l % d.nx);
    else base = l;
    float* p = data + base;
    if (len == 1) continue;
    const double lambda = (1.0 - z) * (1.0 - 1.0 / z);
    // causal initialisation
    double sum;
    const int horizon = 18 < len ? 18 : len;   // ceil(log(1e-10) / log|z|) = 18
    if (horizon < len) {
      double zn = z;
      sum = (double)p[0] * lambda;
      for (int k = 1; k < horizon; ++k) {
        sum += zn * (double)p[(size_t)k * stride] * lambda;
        zn *= z;
      }
    } else {
      const double iz = 1.0 / z;
      double zn = z, z2n = pow(z, (double)(len - 1));
      sum = ((double)p[0] + z2n * (double)p[(size_t)(len - 1) * stride]) * lambda;
      z2n *= z2n * iz;
      for (int k = 1; k <= len - 2; ++k) {
        sum += (zn + z2n) * (double)p[(size_t)k * stride] * lambda;
        zn *= z;
        z2n *= iz;
      }
      sum /= (1.0 - zn * zn);
    }
    double prev = sum;
    p[0] = (float)prev;   // stored in fp32, recursion continues in fp64
    double cm2 = 0.0, cm1 = prev;
    for (int k = 1; k < len; ++k) {
      const double cur = (double)p[(size_t)k * stride] * lambda + z * prev;
      p[(size_t)k * stride] = (float)cur;
      cm2 = prev;
      prev = cur;
      cm1 = cur;
    }
    // anti-causal initialisation and recursion
    double nxt = (z / (z * z - 1.0)) * (z * cm2 + cm1);
    p[(size_t)(len - 1) * stride] = (float)nxt;
    for (int k = len - 2; k >= 0; --k) {
      const double cur = z * (nxt - (double)p[(size_t)k * stride]);
      p[(size_t)k * stride] = (float)cur;
      nxt = cur;
    }
  }
}

template <typename T, int INTERP, bool HASFIELD>
__global__ void __launch_bounds__(NT) k_resample(const T* __restrict__ in, pp_dims din, const float* __restrict__ field,
                                                 T* __restrict__ out, pp_dims dout, pp_xform X, T default_value) {
  const size_t N = (size_t)dout.nx * dout.ny * dout.nz;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, z = blockIdx.z;
  if (x < dout.nx && y < dout.ny) {
    const size_t i = ((size_t)z * dout.ny + y) * dout.nx + x;
    double ddx = 0.0, ddy = 0.0, ddz = 0.0;
    if (HASFIELD) {
      ddx = (double)field[i];
      ddy = (double)field[N + i];
      ddz = (double)field[2 * N + i];
    }
    double c[3];
    pp_map_point(X, x, y, z, ddx, ddy, ddz, c);
    T res = default_value;
    if (pp_inside_d(c, din)) {
      if (INTERP == PP_INTERP_NEAREST) {
        const int qx = (int)floor(c[0] + 0.5), qy = (int)floor(c[1] + 0.5), qz = (int)floor(c[2] + 0.5);
        res = in[((size_t)qz * din.ny + qy) * din.nx + qx];
      } else if (INTERP == PP_INTERP_BSPLINE) {
        if (sizeof(T) == 4) res = pp_cast_out<T>(pp_bspline3_sample(reinterpret_cast<const float*>(in), din, c));
      } else {
        const double flx = floor(c[0]), fly = floor(c[1]), flz = floor(c[2]);
        res = pp_cast_out<T>(din.nx >= 2 ? pp_trilinear_pairs(in, din.nx, din.ny, din.nz, (int)flx, (float)(c[0] - flx), (int)fly,
                                                               (float)(c[1] - fly), (int)flz, (float)(c[2] - flz))
                                         : pp_trilinear(in, din.nx, din.ny, din.nz, (int)flx, (float)(c[0] - flx), (int)fly,
                                                        (float)(c[1] - fly), (int)flz, (float)(c[2] - flz)));
      }
    }
    out[i] = res;
  }
}

// sitk.Resample of the planar vector field onto another grid: identity transform, linear, 0.
__global__ void __launch_bounds__(NT) k_resample_field(const float* __restrict__ in, pp_dims din, float* __restrict__ out,
                                                       pp_dims dout, pp_xform X) {
  const size_t N = (size_t)dout.nx * dout.ny * dout.nz;
  const size_t Ni = (size_t)din.nx * din.ny * din.nz;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, z = blockIdx.z;
  if (x < dout.nx && y < dout.ny) {
    const size_t i = ((size_t)z * dout.ny + y) * dout.nx + x;
    double c[3];
    pp_map_point(X, x, y, z, 0.0, 0.0, 0.0, c);
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
    if (pp_inside_d(c, din)) {
      const double flx = floor(c[0]), fly = floor(c[1]), flz = floor(c[2]);
      const int bx = (int)flx, by = (int)fly, bz = (int)flz;
      const float fx = (float)(c[0] - flx), fy = (float)(c[1] - fly), fz = (float)(c[2] - flz);
      if (din.nx >= 2) {
        r0 = pp_trilinear_pairs(in, din.nx, din.ny, din.nz, bx, fx, by, fy, bz, fz);
        r1 = pp_trilinear_pairs(in + Ni, din.nx, din.ny, din.nz, bx, fx, by, fy, bz, fz);
        r2 = pp_trilinear_pairs(in + 2 * Ni, din.nx, din.ny, din.nz, bx, fx, by, fy, bz, fz);
      } else {
        r0 = pp_trilinear(in, din.nx, din.ny, din.nz, bx, fx, by, fy, bz, fz);
        r1 = pp_trilinear(in + Ni, din.nx, din.ny, din.nz, bx, fx, by, fy, bz, fz);
        r2 = pp_trilinear(in + 2 * Ni, din.nx, din.ny, din.nz, bx, fx, by, fy, bz, fz);
      }
    }
    out[i] = r0;
    out[N + i] = r1;
    out[2 * N + i] = r2;
  }
}

// ---------------------------------------------------------------------------------------
// Round 5: the gathers above for the common geometry -- both grids axis-aligned, no linear transform, every volume below
// 2^32 bytes.  Same values bit for bit (tests/test_kernels.py holds them to the general kernels); what changes is the cost
// around the arithmetic.  The general kernels spend ~90 fp64 operations on the two 3 x 3 index <-> physical products and
// ~35 64-bit address operations per voxel, and without a field to read they are bound by instruction issue, not memory (an
// identity resample moved 8 B / voxel at 1.9 TB/s).  On axis-aligned grids the off-diagonal products are exact zeros
// (x + (+-0) = x), so one multiply per axis gives the same bits; offsets are 32-bit byte offsets from wave-uniform bases.
// (Measured and dropped: the same kernels marching z in 64 x 4 tiles handed out XCD by XCD, so that the upper plane of
// corners of a step is the next step's lower plane in cache -- the warp through a field went 408 -> 478 us and the
// composition 863 -> 1134 us inside config 2's registration; plane-by-plane launch order keeps every block of the chip
// on two planes of DRAM pages.  Only the field up-sampling marches: its source is small and it keeps its corners in
// registers, below.)
bool rs_small(const pp_dims& d, size_t bytes_per_voxel) {
  return (size_t)d.nx * d.ny * d.nz * bytes_per_voxel < ((size_t)1 << 32);
}

template <typename T>
__device__ __forceinline__ T rs_ld(const T* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void rs_st(T* base, unsigned byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// The address half of pp_trilinear_pairs / pp_trilinear once for every volume sampled at the same point.
struct rs_corner {
  unsigned o00, o10, o01, o11;   // element offsets of the four x pairs (or, nx == 1, of the four single corners)
  float wx, wy, wz;
  bool xlast;
};
__device__ __forceinline__ void rs_corners(const pp_dims& n, int bx, float fx, int by, float fy, int bz, float fz, rs_corner& a) {
  int x0, x1, y0, y1, z0, z1;
  pp_axis_setup(bx, fx, n.nx, x0, x1, a.wx);
  pp_axis_setup(by, fy, n.ny, y0, y1, a.wy);
  pp_axis_setup(bz, fz, n.nz, z0, z1, a.wz);
  a.xlast = x0 > n.nx - 2;
  const unsigned xs = n.nx >= 2 ? (unsigned)(a.xlast ? n.nx - 2 : x0) : 0u;
  const unsigned r0 = (unsigned)z0 * (unsigned)n.ny, r1 = (unsigned)z1 * (unsigned)n.ny;
  a.o00 = (r0 + (unsigned)y0) * (unsigned)n.nx + xs;
  a.o10 = (r0 + (unsigned)y1) * (unsigned)n.nx + xs;
  a.o01 = (r1 + (unsigned)y0) * (unsigned)n.nx + xs;
  a.o11 = (r1 + (unsigned)y1) * (unsigned)n.nx + xs;
}
// the two x corners of a row: one access of 2 sizeof(T) bytes from an element-aligned position (pp_trilinear_pairs), or
// the single column's voxel twice
template <bool WIDE, typename T>
__device__ __forceinline__ void rs_row(const T* __restrict__ im, bool xlast, unsigned off, float& lo, float& hi) {
  if (WIDE) {
    const pp_pair<T> p = rs_ld(reinterpret_cast<const pp_pair<T>*>(im), off * (unsigned)sizeof(T));
    lo = (float)(xlast ? p.y : p.x);
    hi = (float)p.y;
  } else {
    lo = hi = (float)rs_ld(im, off * (unsigned)sizeof(T));
  }
}
__device__ __forceinline__ float rs_lerp(float a000, float a100, float a010, float a110, float a001, float a101, float a011, float a111,
                                         float wx, float wy, float wz) {
  const float v00 = a000 + (a100 - a000) * wx;
  const float v10 = a010 + (a110 - a010) * wx;
  const float v01 = a001 + (a101 - a001) * wx;
  const float v11 = a011 + (a111 - a011) * wx;
  const float v0 = v00 + (v10 - v00) * wy;
  const float v1 = v01 + (v11 - v01) * wy;
  return v0 + (v1 - v0) * wz;
}
template <bool WIDE, typename T>
__device__ __forceinline__ float rs_sample(const T* __restrict__ im, const rs_corner& a) {
  float a000, a100, a010, a110, a001, a101, a011, a111;
  rs_row<WIDE>(im, a.xlast, a.o00, a000, a100);   // (four loads in flight together: no branch between them)
  rs_row<WIDE>(im, a.xlast, a.o10, a010, a110);
  rs_row<WIDE>(im, a.xlast, a.o01, a001, a101);
  rs_row<WIDE>(im, a.xlast, a.o11, a011, a111);
  return rs_lerp(a000, a100, a010, a110, a001, a101, a011, a111, a.wx, a.wy, a.wz);
}

// pp_map_point on axis-aligned grids without a linear transform: per axis ((spacing_out * idx + origin_out) [+ d]) -
// origin_in, times 1 / spacing_in -- the terms pp_map_point adds besides these are products with exact zeros.
struct rs_axes {
  double s_out[3], o_out[3], o_in[3], p_in[3];
  double A[9], t[3];   // the linear transform between the two (k_resample_axis<..., AFFINE = true> only)
};
// The same with a linear transform q = A p + t between the two axis-aligned grids (the pipelines' affine propagation of
// images and labels): the index -> physical and physical -> index products are still diagonal, the 3 x 3 in the middle
// is evaluated as pp_map_point writes it.
template <bool HASFIELD>
__device__ __forceinline__ void rs_affine(const rs_axes& X, int x, int y, int z, double ddx, double ddy, double ddz, double c[3]) {
#pragma clang fp contract(off)
  double p[3], q[3];
  p[0] = X.s_out[0] * (double)x;
  p[0] = p[0] + X.o_out[0];
  p[1] = X.s_out[1] * (double)y;
  p[1] = p[1] + X.o_out[1];
  p[2] = X.s_out[2] * (double)z;
  p[2] = p[2] + X.o_out[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double s = X.A[r * 3 + 0] * p[0] + X.A[r * 3 + 1] * p[1];
    s = s + X.A[r * 3 + 2] * p[2];
    q[r] = s + X.t[r];
  }
  if (HASFIELD) {
    q[0] = q[0] + ddx;
    q[1] = q[1] + ddy;
    q[2] = q[2] + ddz;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double v = q[r] - X.o_in[r];
    c[r] = X.p_in[r] * v;
  }
}
template <bool HASFIELD>
__device__ __forceinline__ double rs_axis(const rs_axes& X, int r, int idx, double dd) {
#pragma clang fp contract(off)
  double p = X.s_out[r] * (double)idx;
  p = p + X.o_out[r];
  if (HASFIELD) p = p + dd;
  const double v = p - X.o_in[r];
  return X.p_in[r] * v;
}

// pp_inside_d without short-circuit evaluation: with it the compiler sinks each component's field load behind the test of
// the component before -- three dependent round trips to memory.
__device__ __forceinline__ bool rs_inside(const double c[3], const pp_dims& n) {
  const int ok = (int)(c[0] >= -0.5) & (int)(c[0] < (double)n.nx - 0.5) & (int)(c[1] >= -0.5) & (int)(c[1] < (double)n.ny - 0.5) &
                 (int)(c[2] >= -0.5) & (int)(c[2] < (double)n.nz - 0.5);
  return ok != 0;
}

// k_resample on axis-aligned grids (same launch geometry: grid3_for)
template <typename T, int INTERP, bool HASFIELD, bool WIDE, bool AFFINE>
__global__ void __launch_bounds__(NT) k_resample_axis(const T* __restrict__ in, pp_dims din, const float* __restrict__ field,
                                                      T* __restrict__ out, pp_dims dout, rs_axes X, T default_value, pp_band B) {
  unsigned bx_, by_, bz_;
  if (!pp_band_block(B, bx_, by_, bz_)) return;
  const int x = bx_ * blockDim.x + threadIdx.x, y = by_ * blockDim.y + threadIdx.y, z = bz_;
  if (x >= dout.nx || y >= dout.ny) return;
  const unsigned N4 = (unsigned)dout.nx * (unsigned)dout.ny * (unsigned)dout.nz * 4u;
  const unsigned i = ((unsigned)z * (unsigned)dout.ny + (unsigned)y) * (unsigned)dout.nx + (unsigned)x;
  double ddx = 0.0, ddy = 0.0, ddz = 0.0;
  if (HASFIELD) {
    ddx = (double)rs_ld(field, i * 4u);
    ddy = (double)rs_ld(field, N4 + i * 4u);
    ddz = (double)rs_ld(field, 2u * N4 + i * 4u);
  }
  double c[3];
  if (AFFINE) {
    rs_affine<HASFIELD>(X, x, y, z, ddx, ddy, ddz, c);
  } else {
    c[0] = rs_axis<HASFIELD>(X, 0, x, ddx);
    c[1] = rs_axis<HASFIELD>(X, 1, y, ddy);
    c[2] = rs_axis<HASFIELD>(X, 2, z, ddz);
  }
  T res = default_value;
  if (rs_inside(c, din)) {
    if (INTERP == PP_INTERP_NEAREST) {
      const int qx = (int)floor(c[0] + 0.5), qy = (int)floor(c[1] + 0.5), qz = (int)floor(c[2] + 0.5);
      res = rs_ld(in, (((unsigned)qz * (unsigned)din.ny + (unsigned)qy) * (unsigned)din.nx + (unsigned)qx) * (unsigned)sizeof(T));
    } else {
      const double flx = floor(c[0]), fly = floor(c[1]), flz = floor(c[2]);
      rs_corner a;
      rs_corners(din, (int)flx, (float)(c[0] - flx), (int)fly, (float)(c[1] - fly), (int)flz, (float)(c[2] - flz), a);
      res = pp_cast_out<T>(rs_sample<WIDE>(in, a));
    }
  }
  rs_st(out, i * (unsigned)sizeof(T), res);
}

// k_resample_field on axis-aligned grids, marching z.  A thread's x and y corners do not change along its column, and its
// z corners change once per (output planes per input plane) steps -- the pyramid's up-sampling is x2 to x4 -- for the whole
// block at once (the z coordinate depends on z alone): the twelve corner pairs of the three components stay in registers,
// a step onto the next input plane moves the upper six down and loads six, and most steps load nothing.
struct rs_grid {
  unsigned gx, gy, gz, zc;
};
constexpr int RS_TX = 64, RS_TY = NT / RS_TX;
rs_grid rs_grid_for(const pp_dims& d) {
  rs_grid g;
  g.gx = (unsigned)((d.nx + RS_TX - 1) / RS_TX);
  g.gy = (unsigned)((d.ny + RS_TY - 1) / RS_TY);
  g.zc = 32;
  const char* e = pp_env("PP_RS_ZCHUNK");
  if (e && atoi(e) > 0) g.zc = (unsigned)atoi(e);
  while (g.zc > 1 && (size_t)g.gx * g.gy * ((d.nz + g.zc - 1) / g.zc) < 4096) g.zc >>= 1;
  g.gz = (unsigned)((d.nz + g.zc - 1) / g.zc);
  return g;
}
template <bool WIDE>
__global__ void __launch_bounds__(NT) k_resample_field_march(const float* __restrict__ in, pp_dims din, float* __restrict__ out,
                                                             pp_dims dout, rs_axes X, rs_grid G) {
  const int x = (int)(blockIdx.x * RS_TX + threadIdx.x), y = (int)(blockIdx.y * RS_TY + threadIdx.y);
  const int zb = (int)(blockIdx.z * G.zc), ze = zb + (int)G.zc < dout.nz ? zb + (int)G.zc : dout.nz;
  if (x >= dout.nx || y >= dout.ny) return;
  const unsigned plane = (unsigned)dout.nx * (unsigned)dout.ny, N4 = plane * (unsigned)dout.nz * 4u;
  const unsigned Ni = (unsigned)din.nx * (unsigned)din.ny * (unsigned)din.nz;
  unsigned i = ((unsigned)zb * (unsigned)dout.ny + (unsigned)y) * (unsigned)dout.nx + (unsigned)x;
  double c[3];
  c[0] = rs_axis<false>(X, 0, x, 0.0);
  c[1] = rs_axis<false>(X, 1, y, 0.0);
  int cz0 = -1, cz1 = -1;               // the input planes held in lo / hi
  float lo[3][4], hi[3][4];             // per component: row y0 (x0, x1), row y1 (x0, x1)
  for (int z = zb; z < ze; ++z, i += plane) {
    c[2] = rs_axis<false>(X, 2, z, 0.0);
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
    if (rs_inside(c, din)) {
      const double flx = floor(c[0]), fly = floor(c[1]), flz = floor(c[2]);
      const int bx = (int)flx, by = (int)fly, bz = (int)flz;
      int x0, x1, y0, y1, z0, z1;
      float wx, wy, wz;
      pp_axis_setup(bx, (float)(c[0] - flx), din.nx, x0, x1, wx);
      pp_axis_setup(by, (float)(c[1] - fly), din.ny, y0, y1, wy);
      pp_axis_setup(bz, (float)(c[2] - flz), din.nz, z0, z1, wz);
      if (z0 != cz0 || z1 != cz1) {
        const bool xlast = x0 > din.nx - 2;
        const unsigned xs = WIDE ? (unsigned)(xlast ? din.nx - 2 : x0) : 0u;
        const bool shift = z0 == cz1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* comp = in + (size_t)k * Ni;
          if (shift) {
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[k][e] = hi[k][e];
          } else {
            rs_row<WIDE>(comp, xlast, ((unsigned)z0 * (unsigned)din.ny + (unsigned)y0) * (unsigned)din.nx + xs, lo[k][0], lo[k][1]);
            rs_row<WIDE>(comp, xlast, ((unsigned)z0 * (unsigned)din.ny + (unsigned)y1) * (unsigned)din.nx + xs, lo[k][2], lo[k][3]);
          }
          rs_row<WIDE>(comp, xlast, ((unsigned)z1 * (unsigned)din.ny + (unsigned)y0) * (unsigned)din.nx + xs, hi[k][0], hi[k][1]);
          rs_row<WIDE>(comp, xlast, ((unsigned)z1 * (unsigned)din.ny + (unsigned)y1) * (unsigned)din.nx + xs, hi[k][2], hi[k][3]);
        }
        cz0 = z0;
        cz1 = z1;
      }
      r0 = rs_lerp(lo[0][0], lo[0][1], lo[0][2], lo[0][3], hi[0][0], hi[0][1], hi[0][2], hi[0][3], wx, wy, wz);
      r1 = rs_lerp(lo[1][0], lo[1][1], lo[1][2], lo[1][3], hi[1][0], hi[1][1], hi[1][2], hi[1][3], wx, wy, wz);
      r2 = rs_lerp(lo[2][0], lo[2][1], lo[2][2], lo[2][3], hi[2][0], hi[2][1], hi[2][2], hi[2][3], wx, wy, wz);
    }
    rs_st(out, i * 4u, r0);
    rs_st(out, N4 + i * 4u, r1);
    rs_st(out, 2u * N4 + i * 4u, r2);
  }
}

unsigned grid_for(size_t work) {
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > 65535u * 8u) blocks = 65535u * 8u;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

void fill_xform(const pp_geom* gin, const pp_geom* gout, const double* A, const double* t, pp_xform* X) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) X->i2p_out[r * 3 + c] = gout->direction[r * 3 + c] * gout->spacing[c];
  pp_index_map m;
  pp_make_index_map(gin, gout, nullptr, nullptr, &m);
  memcpy(X->p2i_in, m.Md, sizeof(m.Md));
  for (int k = 0; k < 3; ++k) {
    X->o_out[k] = gout->origin[k];
    X->o_in[k] = gin->origin[k];
  }
  X->has_affine = (A != nullptr);
  X->axis = pp_geom_identity_dir(gin) && pp_geom_identity_dir(gout);
  X->diag = !A && X->axis;
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(X->A, A ? A : I3, sizeof(X->A));
  for (int k = 0; k < 3; ++k) X->t[k] = (A && t) ? t[k] : 0.0;
}

bool rs_generic_forced() { return pp_env("PP_RESAMPLE_GENERIC") != nullptr; }
rs_axes rs_axes_of(const pp_xform& X) {
  rs_axes a;
  for (int k = 0; k < 3; ++k) {
    a.s_out[k] = X.i2p_out[k * 3 + k];
    a.o_out[k] = X.o_out[k];
    a.o_in[k] = X.o_in[k];
    a.p_in[k] = X.p2i_in[k * 3 + k];
    a.t[k] = X.t[k];
  }
  for (int k = 0; k < 9; ++k) a.A[k] = X.A[k];
  return a;
}

template <typename T>
int resample_any(pp_ctx* ctx, const T* in, const pp_geom* gin, const pp_geom* gout, const double* A, const double* t,
                 const float* field, int interp, double default_value, T* out, const char* name) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out, "resample: NULL volume");
  PP_REQUIRE(ctx, (const void*)in != (const void*)out, "resample: in-place is not supported");
  int rc = pp_geom_check(ctx, gin, "input");
  if (rc) return rc;
  rc = pp_geom_check(ctx, gout, "output");
  if (rc) return rc;
  PP_REQUIRE(ctx, interp == PP_INTERP_NEAREST || interp == PP_INTERP_LINEAR || (interp == PP_INTERP_BSPLINE && sizeof(T) == 4),
             "resample: interpolator must be nearest, linear or (fp32 coefficient volumes) cubic B-spline");
  pp_xform X;
  fill_xform(gin, gout, A, t, &X);
  const pp_dims din{gin->size[0], gin->size[1], gin->size[2]};
  const pp_dims dout{gout->size[0], gout->size[1], gout->size[2]};
  const pp_grid3 g3 = grid3_for(dout.nx, dout.ny, dout.nz);
  const dim3 grid = g3.grid, block = g3.block;
  T dv;
  if (sizeof(T) == 1) {
    const double c = default_value < 0.0 ? 0.0 : (default_value > 255.0 ? 255.0 : default_value);
    dv = (T)c;
  } else {
    dv = (T)default_value;
  }
  if (X.axis && interp != PP_INTERP_BSPLINE && rs_small(din, sizeof(T)) && rs_small(dout, field ? 12 : sizeof(T)) && !rs_generic_forced()) {
    const rs_axes XA = rs_axes_of(X);
    dim3 launch;
    // (banded only where it measured faster or equal: linear through a field.  Labels through a field and resamples without
    // a field fetch nothing twice to begin with and ran 8 - 17 % slower banded.)
    const pp_band B = band_for(g3, &launch, field != nullptr && interp == PP_INTERP_LINEAR && sizeof(T) == 4);
#define PP_RSA2(I, F, W, AF) hipLaunchKernelGGL((k_resample_axis<T, I, F, W, AF>), launch, block, 0, ctx->stream, in, din, field, out, dout, XA, dv, B)
#define PP_RSA(I, F, W) do { if (X.has_affine) PP_RSA2(I, F, W, true); else PP_RSA2(I, F, W, false); } while (0)
    if (interp == PP_INTERP_NEAREST) {
      if (field) PP_RSA(PP_INTERP_NEAREST, true, true); else PP_RSA(PP_INTERP_NEAREST, false, true);
    } else if (din.nx >= 2) {
      if (field) PP_RSA(PP_INTERP_LINEAR, true, true); else PP_RSA(PP_INTERP_LINEAR, false, true);
    } else {
      if (field) PP_RSA(PP_INTERP_LINEAR, true, false); else PP_RSA(PP_INTERP_LINEAR, false, false);
    }
#undef PP_RSA2
#undef PP_RSA
    PP_LAUNCH_CHECK(ctx, name);
    return PP_OK;
  }
#define PP_RS(I, F) hipLaunchKernelGGL((k_resample<T, I, F>), grid, block, 0, ctx->stream, in, din, field, out, dout, X, dv)
  if (interp == PP_INTERP_NEAREST) {
    if (field) PP_RS(PP_INTERP_NEAREST, true); else PP_RS(PP_INTERP_NEAREST, false);
  } else if (interp == PP_INTERP_BSPLINE) {
    if (field) PP_RS(PP_INTERP_BSPLINE, true); else PP_RS(PP_INTERP_BSPLINE, false);
  } else {
    if (field) PP_RS(PP_INTERP_LINEAR, true); else PP_RS(PP_INTERP_LINEAR, false);
  }
#undef PP_RS
  PP_LAUNCH_CHECK(ctx, name);
  return PP_OK;
}

// sitk.TransformToDisplacementField for a linear transform q = A p + t (reference deformable.py:101-108), optionally
// plus a field already expressed on the grid: D(idx) = (A - I) p(idx) + t [+ add(idx)], coordinates in fp64.
struct pp_affine_disp {
  double i2p[9], origin[3], AmI[9], t[3];
};
__global__ void __launch_bounds__(NT) k_affine_displacement(pp_dims d, pp_affine_disp X, const float* __restrict__ add,
                                                            float* __restrict__ out) {
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < N; i += (size_t)gridDim.x * NT) {
    const double ix = (double)(i % d.nx), iy = (double)((i / d.nx) % d.ny), iz = (double)(i / ((size_t)d.nx * d.ny));
    double p[3];
    for (int r = 0; r < 3; ++r) p[r] = X.origin[r] + X.i2p[r * 3 + 0] * ix + X.i2p[r * 3 + 1] * iy + X.i2p[r * 3 + 2] * iz;
    for (int r = 0; r < 3; ++r) {
      const double v = X.AmI[r * 3 + 0] * p[0] + X.AmI[r * 3 + 1] * p[1] + X.AmI[r * 3 + 2] * p[2] + X.t[r];
      out[r * N + i] = (float)v + (add ? add[r * N + i] : 0.0f);
    }
  }
}

}  // namespace

int pp_warp_same_grid(pp_ctx* ctx, const float* moving, const float* field, const pp_dims& d, const pp_warp_scale& sc,
                      float edge_value, float* out, const int* halt_flag) {
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  const bool vec4 = (d.nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(field) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) && (N % 4 == 0);
  // preconditions of the straight-line sample (pp_warp_sample.h)
  const bool sl = N * sizeof(float) < ((size_t)1 << 32) && (size_t)d.ny * d.nz < ((size_t)1 << 24) && d.nx >= 2 && d.nx < (1 << 22) &&
                  d.ny < (1 << 22) && d.nz < (1 << 22) && pp_env("PP_WARP_LEGACY") == nullptr;
  if (vec4) {
    const pp_grid3 g3 = grid3_for(d.nx / 4, d.ny, d.nz);
    dim3 launch;
    const pp_band B = band_for(g3, &launch);
    if (sl)
      hipLaunchKernelGGL((k_warp_same_grid_sl<4>), launch, g3.block, 0, ctx->stream, moving, field, out, d, sc, edge_value, halt_flag, B);
    else
      hipLaunchKernelGGL((k_warp_same_grid<4>), g3.grid, g3.block, 0, ctx->stream, moving, field, out, d, sc, edge_value, halt_flag);
  } else {
    const pp_grid3 g3 = grid3_for(d.nx, d.ny, d.nz);
    dim3 launch;
    const pp_band B = band_for(g3, &launch);
    if (sl)
      hipLaunchKernelGGL((k_warp_same_grid_sl<1>), launch, g3.block, 0, ctx->stream, moving, field, out, d, sc, edge_value, halt_flag, B);
    else
      hipLaunchKernelGGL((k_warp_same_grid<1>), g3.grid, g3.block, 0, ctx->stream, moving, field, out, d, sc, edge_value, halt_flag);
  }
  PP_LAUNCH_CHECK(ctx, "k_warp_same_grid");
  return PP_OK;
}

extern "C" {

int pp_warp_f32(pp_ctx* ctx, const float* moving, const float* field, const pp_geom* g, float edge_value, float* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, moving && field && out, "pp_warp_f32: NULL volume");
  PP_REQUIRE(ctx, moving != out, "pp_warp_f32: in-place is not supported");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  if (!pp_geom_identity_dir(g))
    return pp_fail(ctx, PP_ERR_UNSUPPORTED, "pp_warp_f32: identity direction only; use pp_resample_f32");
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  const pp_warp_scale sc{(float)(1.0 / g->spacing[0]), (float)(1.0 / g->spacing[1]), (float)(1.0 / g->spacing[2])};
  return pp_warp_same_grid(ctx, moving, field, d, sc, edge_value, out, nullptr);
}

int pp_resample_f32(pp_ctx* ctx, const float* in, const pp_geom* gin, const pp_geom* gout, const double* affine_A,
                    const double* affine_t, const float* field, int interp, double default_value, float* out) {
  pp_device_guard dev_guard_(ctx);
  return resample_any<float>(ctx, in, gin, gout, affine_A, affine_t, field, interp, default_value, out, "k_resample<f32>");
}

int pp_resample_u8(pp_ctx* ctx, const uint8_t* in, const pp_geom* gin, const pp_geom* gout, const double* affine_A,
                   const double* affine_t, const float* field, int interp, double default_value, uint8_t* out) {
  pp_device_guard dev_guard_(ctx);
  return resample_any<uint8_t>(ctx, in, gin, gout, affine_A, affine_t, field, interp, default_value, out, "k_resample<u8>");
}

int pp_bspline_prefilter_f32(pp_ctx* ctx, const float* in, const int size[3], float* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && size, "pp_bspline_prefilter_f32: NULL argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_bspline_prefilter_f32: empty volume");
  const pp_dims d{size[0], size[1], size[2]};
  const size_t N = pp_nvox(size);
  if (in != out) PP_HIP(ctx, hipMemcpyAsync(out, in, N * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  for (int axis = 0; axis < 3; ++axis) {
    const int len = axis == 0 ? d.nx : (axis == 1 ? d.ny : d.nz);
    hipLaunchKernelGGL(k_bspline3_prefilter, dim3(grid_for(N / len)), dim3(NT), 0, ctx->stream, out, d, axis);
    PP_LAUNCH_CHECK(ctx, "k_bspline3_prefilter");
  }
  return PP_OK;
}

int pp_resample_field_f32(pp_ctx* ctx, const float* in, const pp_geom* gin, const pp_geom* gout, float* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && in != out, "pp_resample_field_f32: NULL or aliased field");
  int rc = pp_geom_check(ctx, gin, "input");
  if (rc) return rc;
  rc = pp_geom_check(ctx, gout, "output");
  if (rc) return rc;
  pp_xform X;
  fill_xform(gin, gout, nullptr, nullptr, &X);
  const pp_dims din{gin->size[0], gin->size[1], gin->size[2]};
  const pp_dims dout{gout->size[0], gout->size[1], gout->size[2]};
  if (X.diag && rs_small(din, 4) && rs_small(dout, 12) && !rs_generic_forced()) {
    const rs_grid G = rs_grid_for(dout);
    if (din.nx >= 2)
      hipLaunchKernelGGL(k_resample_field_march<true>, dim3(G.gx, G.gy, G.gz), dim3(RS_TX, RS_TY), 0, ctx->stream, in, din, out, dout,
                         rs_axes_of(X), G);
    else
      hipLaunchKernelGGL(k_resample_field_march<false>, dim3(G.gx, G.gy, G.gz), dim3(RS_TX, RS_TY), 0, ctx->stream, in, din, out, dout,
                         rs_axes_of(X), G);
    PP_LAUNCH_CHECK(ctx, "k_resample_field_march");
    return PP_OK;
  }
  const pp_grid3 g3 = grid3_for(dout.nx, dout.ny, dout.nz);
  hipLaunchKernelGGL(k_resample_field, g3.grid, g3.block, 0, ctx->stream, in, din, out, dout, X);
  PP_LAUNCH_CHECK(ctx, "k_resample_field");
  return PP_OK;
}

int pp_transform_to_field_f32(pp_ctx* ctx, const pp_geom* g, const double* affine_A, const double* affine_t,
                              const float* add_field, float* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, affine_A && affine_t && out, "pp_transform_to_field_f32: NULL argument");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  pp_affine_disp X;
  // p = origin + Dir * diag(spacing) * idx;  D = (A - I) p + t
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) X.i2p[r * 3 + c] = g->direction[r * 3 + c] * g->spacing[c];
    X.origin[r] = g->origin[r];
    for (int c = 0; c < 3; ++c) X.AmI[r * 3 + c] = affine_A[r * 3 + c] - (r == c ? 1.0 : 0.0);
    X.t[r] = affine_t[r];
  }
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  hipLaunchKernelGGL(k_affine_displacement, dim3(grid_for(pp_nvox(g->size))), dim3(NT), 0, ctx->stream, d, X, add_field, out);
  PP_LAUNCH_CHECK(ctx, "k_affine_displacement");
  return PP_OK;
}

int pp_compose_field_f32(pp_ctx* ctx, float* total, const float* iter, const pp_geom* g) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, total && iter && total != iter, "pp_compose_field_f32: NULL or aliased field");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  if (!pp_geom_identity_dir(g)) return pp_fail(ctx, PP_ERR_UNSUPPORTED, "pp_compose_field_f32: identity direction only");
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  const pp_warp_scale sc{(float)(1.0 / g->spacing[0]), (float)(1.0 / g->spacing[1]), (float)(1.0 / g->spacing[2])};
  // (measured, round 3: four voxels per thread with 16-byte accesses of `total` and the twelve corner loads of a voxel pair in
  // flight together ran 1.43 ms against this kernel's 1.14 ms at 512 x 512 x 256 -- one voxel per thread keeps more gathers
  // of more wavefronts in flight.)
  // 64 x 4 blocks: a block's four rows share their upper / lower corner rows in L1 (sbench 0.99 -> 0.96 ms, config 2's
  // registration 15.69 -> 15.53 ms against 256 x 1; PP_COMPOSE_BLOCK=128|256 for the other shapes)
  unsigned bxm = 64u;
  if (const char* e = pp_env("PP_COMPOSE_BLOCK")) bxm = (unsigned)atoi(e) >= 64u ? (unsigned)atoi(e) : 64u;
  const pp_grid3 g3 = grid3_for(d.nx, d.ny, d.nz, bxm);
  dim3 launch;
  const pp_band B = band_for(g3, &launch);
  hipLaunchKernelGGL(k_compose_same_grid, launch, g3.block, 0, ctx->stream, total, iter, d, sc, B);
  PP_LAUNCH_CHECK(ctx, "k_compose_same_grid");
  return PP_OK;
}

}  // extern "C"
