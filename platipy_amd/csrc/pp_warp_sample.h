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
// Both samples of a thread behind ONE wavefront vote (round 4).  When every lane's two base cells lie inside the volume with an
// upper neighbour on every axis -- all but the wavefronts that touch the volume's border -- the displacement clamp, the index
// clamps, the zero weight below index 0, the repeated corner of a last index and the buffer test are identities / true, and
// the address half of a sample is 3 floors, 3 conversions, 3 subtractions and 9 integer operations instead of ~60 vector
// instructions (the warp's address / weight arithmetic was ~190 of kernel B's ~300 vector instructions per thread and plane).
// Same values on that path: the clamps do not bind, the weights are the fractions, the four offsets are the slow path's with
// dy = row pitch, dz = plane pitch.  A NaN displacement fails the vote (s != s) and is sampled -- as outside -- by the slow path.
// Returns the vote; `fused2_warp_finish_pair` must be given it back.  Every lane of the wavefront must reach the call.
__device__ __forceinline__ bool fused2_warp_issue_pair(const char* rm, const pp_warp_dims& wd, int xi, int yi, int zi, float ax, float ay,
                                                       float az, float bx, float by, float bz, bool ok_a, bool ok_b, pp_warp_pending& ga,
                                                       pp_warp_pending& gb) {
  const float fax = floorf(ax), fay = floorf(ay), faz = floorf(az), fbx = floorf(bx), fby = floorf(by), fbz = floorf(bz);
  const int iax = xi + (int)fax, iay = yi + (int)fay, iaz = zi + (int)faz;
  const int ibx = xi + 1 + (int)fbx, iby = yi + (int)fby, ibz = zi + (int)fbz;
  const float s = ((ax + ay) + az) + ((bx + by) + bz);
  const unsigned lx = (unsigned)(wd.nx - 1), ly = (unsigned)(wd.ny - 1), lz = (unsigned)(wd.nz - 1);
  const bool interior = ok_a & ok_b & ((unsigned)iax < lx) & ((unsigned)iay < ly) & ((unsigned)iaz < lz) & ((unsigned)ibx < lx) &
                        ((unsigned)iby < ly) & ((unsigned)ibz < lz) & (s == s);
  pp_warp_addr a, b;
  const bool fast = !__any(!interior);
  if (fast) {
    a.wx = ax - fax; a.wy = ay - fay; a.wz = az - faz;
    b.wx = bx - fbx; b.wy = by - fby; b.wz = bz - fbz;
    const unsigned ra = __umul24(__umul24((unsigned)iaz, (unsigned)wd.ny) + (unsigned)iay, wd.nx4), ca = (unsigned)iax * 4u;
    const unsigned rb = __umul24(__umul24((unsigned)ibz, (unsigned)wd.ny) + (unsigned)iby, wd.nx4), cb = (unsigned)ibx * 4u;
    a.o00 = ra + ca; a.o10 = ra + wd.nx4 + ca; a.o01 = ra + wd.sz4 + ca; a.o11 = ra + wd.sz4 + wd.nx4 + ca;
    b.o00 = rb + cb; b.o10 = rb + wd.nx4 + cb; b.o01 = rb + wd.sz4 + cb; b.o11 = rb + wd.sz4 + wd.nx4 + cb;
    a.flags = 1u;
    b.flags = 1u;
  } else {
    fused2_warp_setup(wd, xi, ax, yi, ay, zi, az, ok_a, a);
    fused2_warp_setup(wd, xi + 1, bx, yi, by, zi, bz, ok_b, b);
  }
  fused2_warp_load(rm, a, ga);   // (the eight loads are common to both paths: no memory instruction inside the branch)
  fused2_warp_load(rm, b, gb);
  return fast;
}
// fused2_warp_finish on the voted path: every sample is inside and no lane sits on a last x index -- the same lerps without
// the five selects.
__device__ __forceinline__ float fused2_warp_finish_interior(const pp_warp_pending& g) {
  const float a000 = g.p00.x, a100 = g.p00.y;
  const float a010 = g.p10.x, a110 = g.p10.y;
  const float a001 = g.p01.x, a101 = g.p01.y;
  const float a011 = g.p11.x, a111 = g.p11.y;
  const float v00 = a000 + (a100 - a000) * g.wx;
  const float v10 = a010 + (a110 - a010) * g.wx;
  const float v01 = a001 + (a101 - a001) * g.wx;
  const float v11 = a011 + (a111 - a011) * g.wx;
  const float v0 = v00 + (v10 - v00) * g.wy;
  const float v1 = v01 + (v11 - v01) * g.wy;
  return v0 + (v1 - v0) * g.wz;
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

