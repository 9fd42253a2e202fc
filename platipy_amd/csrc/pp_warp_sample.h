// pp_warp_sample.h -- straight-line trilinear sample (itk::LinearInterpolateImageFunction semantics) with 32-bit byte
// offsets, shared by the fused demons kernels (pp_demons_fused2.h) and the stand-alone warp (pp_resample.hip).
// Preconditions, checked by the callers on the host: the volume spans < 2^32 bytes, ny * nz < 2^24, 2 <= nx < 2^22,
// ny, nz < 2^22.
#pragma once
#include "pp_internal.h"

// 8 bytes from a 4-byte-aligned position: a global load through a wave-uniform base plus a 32-bit per-lane byte offset
// (global_load_dwordx2 v, v_off, s[base]; gfx950 runs global accesses in unaligned mode -- buffer loads of 8 bytes do
// not: they drop the low address bits).
struct pp_f2u {
  float x, y;
} __attribute__((aligned(4)));
__device__ __forceinline__ float2 pp_gld2(const char* base, unsigned byte_off) {
  const pp_f2u v = *reinterpret_cast<const pp_f2u*>(base + (size_t)byte_off);
  return make_float2(v.x, v.y);
}

// Straight-line itk::LinearInterpolateImageFunction sample; the arithmetic equals
// pp_trilinear's (lerps nested x, y, z as a + (b - a) w).  nx4 = nx * 4; volumes hold < 2^30 voxels.
struct pp_warp_dims {
  int nx, ny, nz;
  unsigned nx4, sz4;   // bytes per row / per plane
};
// One trilinear sample of the warp in two halves, so that the gathers of output plane n can stay in flight across the
// x pass of the next plane: `issue` forms the addresses and starts the four 8-byte loads, `finish` lerps.
struct pp_warp_pending {
  float2 p00, p10, p01, p11;
  float wx, wy, wz;
  unsigned flags;   // bit 0: inside the buffer, bit 1: x0 is the last index
};
// The address half of a sample on its own -- byte offsets of the four x-pairs, the three weights, the flags -- so that
// several volumes on the same grid (the three components of a displacement field) are gathered through one computation.
struct pp_warp_addr {
  unsigned o00, o10, o01, o11;
  float wx, wy, wz;
  unsigned flags;
};
__device__ __forceinline__ void fused2_warp_setup(const pp_warp_dims& wd, int xi, float dvx, int yi, float dvy, int zi, float dvz,
                                                  bool lane_ok, pp_warp_addr& a) {
  const float LIM = 8388608.0f;
  const float cvx = fminf(fmaxf(dvx, -LIM), LIM), cvy = fminf(fmaxf(dvy, -LIM), LIM), cvz = fminf(fmaxf(dvz, -LIM), LIM);
  const float flx = floorf(cvx), fly = floorf(cvy), flz = floorf(cvz);
  const int bx = xi + (int)flx, by = yi + (int)fly, bz = zi + (int)flz;
  const float fx = cvx - flx, fy = cvy - fly, fz = cvz - flz;
  const int nx_ = bx + (fx >= 0.5f ? 1 : 0), ny_ = by + (fy >= 0.5f ? 1 : 0), nz_ = bz + (fz >= 0.5f ? 1 : 0);
  const bool inside = lane_ok & ((unsigned)nx_ < (unsigned)wd.nx) & ((unsigned)ny_ < (unsigned)wd.ny) & ((unsigned)nz_ < (unsigned)wd.nz);
  const int x0 = pp_clampi(bx, 0, wd.nx - 1), y0 = pp_clampi(by, 0, wd.ny - 1), z0 = pp_clampi(bz, 0, wd.nz - 1);
  a.wx = bx < 0 ? 0.0f : fx;
  a.wy = by < 0 ? 0.0f : fy;
  a.wz = bz < 0 ? 0.0f : fz;
  const unsigned r00 = __umul24(__umul24((unsigned)z0, (unsigned)wd.ny) + (unsigned)y0, wd.nx4);
  const unsigned dy = y0 < wd.ny - 1 ? wd.nx4 : 0u, dz = z0 < wd.nz - 1 ? wd.sz4 : 0u;
  const bool xlast = x0 > wd.nx - 2;
  const unsigned c0 = (unsigned)(xlast ? wd.nx - 2 : x0) * 4u;
  a.o00 = r00 + c0;
  a.o10 = r00 + dy + c0;
  a.o01 = r00 + dz + c0;
  a.o11 = r00 + dz + dy + c0;
  a.flags = (inside ? 1u : 0u) | (xlast ? 2u : 0u);
}
__device__ __forceinline__ void fused2_warp_load(const char* rm, const pp_warp_addr& a, pp_warp_pending& g) {
  g.p00 = pp_gld2(rm, a.o00);
  g.p10 = pp_gld2(rm, a.o10);
  g.p01 = pp_gld2(rm, a.o01);
  g.p11 = pp_gld2(rm, a.o11);
  g.wx = a.wx;
  g.wy = a.wy;
  g.wz = a.wz;
  g.flags = a.flags;
}

// One trilinear sample of the warp in two halves: `issue` forms the addresses and starts the four 8-byte loads, `finish`
// lerps.  (pp_split and pp_inside1 in straight-line form, in fused2_warp_setup: the displacement is clamped to +-2^23
// voxels first -- one v_med3; also catches NaN -- beyond that the sample is outside any volume this kernel takes (nx, ny,
// nz < 2^22) either way and the integer conversions stay defined.  ITK's buffer test [-0.5, n - 0.5) on the continuous
// index is 0 <= round-half-up index <= n - 1, and the round-half-up index is base + (frac >= 0.5).  pp_axis_setup with the
// base index also clamped from above, so that outside lanes still form valid addresses; the upper corner of an axis
// repeats the lower one on the last index (ITK's clamp); 24-bit multiplies: z0 * ny + y0 < 2^24 and nx * 4 < 2^24 are
// checked on the host.  The two x corners of a row come from ONE 8-byte load that starts at min(x0, nx - 2), so on the
// last index both corners are its second element and nothing is read past the row.)
__device__ __forceinline__ void fused2_warp_issue(const char* rm, const pp_warp_dims& wd, int xi, float dvx, int yi, float dvy, int zi,
                                                  float dvz, bool lane_ok, pp_warp_pending& g) {
  pp_warp_addr a;
  fused2_warp_setup(wd, xi, dvx, yi, dvy, zi, dvz, lane_ok, a);
  fused2_warp_load(rm, a, g);
}
__device__ __forceinline__ float fused2_warp_finish(const pp_warp_pending& g) {
  const bool xlast = (g.flags & 2u) != 0;
  const float a000 = xlast ? g.p00.y : g.p00.x, a100 = g.p00.y;
  const float a010 = xlast ? g.p10.y : g.p10.x, a110 = g.p10.y;
  const float a001 = xlast ? g.p01.y : g.p01.x, a101 = g.p01.y;
  const float a011 = xlast ? g.p11.y : g.p11.x, a111 = g.p11.y;
  const float v00 = a000 + (a100 - a000) * g.wx;
  const float v10 = a010 + (a110 - a010) * g.wx;
  const float v01 = a001 + (a101 - a001) * g.wx;
  const float v11 = a011 + (a111 - a011) * g.wx;
  const float v0 = v00 + (v10 - v00) * g.wy;
  const float v1 = v01 + (v11 - v01) * g.wy;
  const float r = v0 + (v1 - v0) * g.wz;
  return (g.flags & 1u) ? r : FLT_MAX;
}
// (the same with a caller-chosen value outside the buffer)
__device__ __forceinline__ float fused2_warp_finish_edge(const pp_warp_pending& g, float edge) {
  const float r = fused2_warp_finish(g);
  return (g.flags & 1u) ? r : edge;
}
__device__ __forceinline__ float fused2_warp_sample(const char* rm, const pp_warp_dims& wd, int xi, float dvx, int yi, float dvy, int zi,
                                                    float dvz, bool lane_ok) {
  pp_warp_pending g;
  fused2_warp_issue(rm, wd, xi, dvx, yi, dvy, zi, dvz, lane_ok, g);
  return fused2_warp_finish(g);
}

