// platipy_amd/csrc/pp_demons_fused2.h -- second generation of the two fused demons kernels
// (included by pp_demons.hip after the shared tile geometry / ESM helpers).
//
// Same schedule, tiles and arithmetic (operation for operation, so fields are bit-identical to the
// first generation) as k_fused_force_smooth / k_fused_add_smooth_warp; what changed is what the
// round-1 ISA spent its issue slots on:
//   * streaming loads, gathers and stores are buffer instructions (scalar resource + one 32-bit
//     per-lane byte offset): no 64-bit VALU address arithmetic per access;
//   * every plane-invariant index (LDS slots of the x pass, gather row strides) is computed once
//     per block, not once per plane;
//   * the trilinear warp is straight-line (clamped addresses, one select at the end) instead of a
//     nest of divergent exec-mask regions;
//   * the z window rotates by renaming (a wave-uniform switch on the plane phase) instead of moving
//     3 x OPT x 2R registers per plane;
//   * the per-plane pipeline is two barrier intervals instead of three (B) / four (A): work on
//     different LDS buffers that used to be separated by a barrier now shares an interval
//     (publish plane z+1 | y/z pass + warp of plane z;  x pass of plane z+1), so LDS-bound and
//     VALU/VMEM-bound phases of one block overlap;
//   * the ESM phase reads the fixed and warped tiles as one packed float2 tile (half the DS
//     instructions), and block reductions use wavefront shuffles + one 8-entry LDS hop.
#pragma once

// Register budget: 4 waves per SIMD = two 512-thread blocks per CU (<= 128 VGPRs); without the bound the scheduler
// spends up to ~170 registers on load latency it cannot use with one block per CU.
#ifndef PP_A_XASM
#define PP_A_XASM 1
#endif
#ifndef PP_B_XSHFL
#define PP_B_XSHFL 1
#endif
#ifndef PP_B_DEFER
#define PP_B_DEFER 1
#endif
#ifndef PP_GEN2_WAVES
#define PP_GEN2_WAVES 4
#endif

#include "pp_warp_sample.h"

typedef __amdgpu_buffer_rsrc_t pp_rsrc;
typedef unsigned pp_u2 __attribute__((vector_size(8)));

// Raw buffer resource over the whole 32-bit offset range (bounds are guaranteed by construction:
// every offset below is clamped into the volume).  0x00020000 = gfx9 DATA_FORMAT_32.
__device__ __forceinline__ pp_rsrc pp_make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
// Resource over the first 2^31 bytes only: a lane whose per-lane offset is PP_OOB (= 2^31) is outside it and the hardware
// drops its store -- the lane mask of a store without a branch around it.  (Raw buffers range-check the per-lane offset;
// whether the scalar offset takes part differs between ISA generations, so callers keep offset + scalar offset < 2^32 and
// whole arrays under 2^31 bytes: then a valid lane is inside and a masked lane outside under either rule.)
constexpr unsigned PP_OOB = 0x80000000u;
__device__ __forceinline__ pp_rsrc pp_make_rsrc_masked(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)PP_OOB, 0x00020000);
}
__device__ __forceinline__ float pp_bld(pp_rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// ... and the matching 8-byte store to a 4-byte-aligned position (rows of odd length: every other row starts off 8 bytes)
__device__ __forceinline__ void pp_gst2(char* base, unsigned byte_off, float a, float b) {
  pp_f2u v;
  v.x = a;
  v.y = b;
  *reinterpret_cast<pp_f2u*>(base + (size_t)byte_off) = v;
}
// The same load with a wave-uniform byte offset in the instruction's scalar-offset operand: one resource per ARRAY is
// built once per block and a plane / component is a 32-bit scalar (PP_SOFF=1, the default; callers guarantee that the whole
// array spans < 2^32 bytes).  Rebuilding a 64-bit base + resource per plane and array cost kernel A ~45 scalar instructions
// per plane and both kernels the scalar registers whose spills (v_readlane) sat in the plane loop.
#ifndef PP_SOFF
#define PP_SOFF 1
#endif
__device__ __forceinline__ float pp_blds(pp_rsrc r, unsigned byte_off, unsigned s_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, s_off, 0));
}
// Cache-policy bits of the buffer stores (2 = nt, streaming) for builds that force one policy; the product chooses per
// launch, see pp_bst2 below.
#ifndef PP_STORE_AUX
#define PP_STORE_AUX 0
#endif
__device__ __forceinline__ void pp_bst(pp_rsrc r, unsigned byte_off, float a) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), r, byte_off, 0, PP_STORE_AUX);
}
__device__ __forceinline__ void pp_bst2(pp_rsrc r, unsigned byte_off, float a, float b) {
  pp_u2 v;
  v[0] = __builtin_bit_cast(unsigned, a);
  v[1] = __builtin_bit_cast(unsigned, b);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, PP_STORE_AUX);
}
// Pair store with the cache policy picked per launch (a kernel template parameter): streaming (nt) when the iteration's working set
// is far larger than the 256 MB infinity cache -- the outputs are read next by the OTHER kernel, long after they have left
// the caches, and need not displace the halo rows and moving-image lines that neighbouring tiles share (512 x 512 x 256:
// -3 % per iteration) -- and cached when the next kernel can still find them there (340 x 341 x 171: nt costs 11 %).
template <bool STREAMING>
__device__ __forceinline__ void pp_bst2s(pp_rsrc r, unsigned byte_off, float a, float b) {
  pp_u2 v;
  v[0] = __builtin_bit_cast(unsigned, a);
  v[1] = __builtin_bit_cast(unsigned, b);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, STREAMING ? 2 : 0);
}

template <bool STREAMING>
__device__ __forceinline__ void pp_bst2ss(pp_rsrc r, unsigned byte_off, unsigned s_off, float a, float b) {
  pp_u2 v;
  v[0] = __builtin_bit_cast(unsigned, a);
  v[1] = __builtin_bit_cast(unsigned, b);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, s_off, STREAMING ? 2 : 0);
}
__device__ __forceinline__ void pp_bsts(pp_rsrc r, unsigned byte_off, unsigned s_off, float a) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), r, byte_off, s_off, PP_STORE_AUX);
}

// Sum of three doubles over the block: butterfly inside each wavefront (ds_bpermute shuffles, no LDS
// storage), one LDS hop for the per-wave results, fixed order -> deterministic.  Valid on thread 0.
template <int NTH>
__device__ __forceinline__ void pp_block_sum3_shfl(double& a, double& b, double& c, double* sm /* 3 * NTH / 64 */) {
  constexpr int NW = NTH / 64;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a += __shfl_xor(a, m, 64);
    b += __shfl_xor(b, m, 64);
    c += __shfl_xor(c, m, 64);
  }
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) {
    sm[3 * (t >> 6) + 0] = a;
    sm[3 * (t >> 6) + 1] = b;
    sm[3 * (t >> 6) + 2] = c;
  }
  __syncthreads();
  if (t == 0) {
    a = sm[0];
    b = sm[1];
    c = sm[2];
    for (int w = 1; w < NW; ++w) {
      a += sm[3 * w + 0];
      b += sm[3 * w + 1];
      c += sm[3 * w + 2];
    }
  }
}

// 16 bytes from a 4-byte-aligned position (global_load_dwordx4 v, v_off, s[base]; rows are 16-B aligned when nx % 4 == 0)
struct pp_f4u {
  float x, y, z, w;
} __attribute__((aligned(4)));
__device__ __forceinline__ float4 pp_gld4(const char* base, unsigned byte_off) {
  const pp_f4u v = *reinterpret_cast<const pp_f4u*>(base + (size_t)byte_off);
  return make_float4(v.x, v.y, v.z, v.w);
}

// Strip geometry of the generation-2 kernels.  The smoothing-input tile is fetched and published as STRIPS of four
// consecutive x voxels (one 16-byte load / LDS store per strip and array instead of four 4-byte ones): the tile's x
// halo is padded from R to RP = a whole number of strips, so its rows start on a strip boundary of the volume.
//   H  = extra halo rows/columns on top of R (0: smoothing input of kernel B;  1: image tile of kernel A)
template <int R, int SH, int H>
struct strip_geom {
  static constexpr int TX = tile_shape<SH>::TX, TY = tile_shape<SH>::TY;
  static constexpr int NTH = 512, LX = TX / 2;
  static constexpr int RP = (R + H + 3) / 4 * 4;   // padded x halo
  static constexpr int UW = TX + 2 * RP;           // tile width = row pitch (floats), a multiple of 4
  static constexpr int UH = TY + 2 * R;            // smoothing-input rows
  static constexpr int TH = UH + 2 * H;            // tile rows
  static constexpr int SPR = UW / 4;               // strips per row
  static constexpr int NS = SPR * TH;              // strips per plane
  static constexpr int NSL = (NS + NTH - 1) / NTH; // strips per thread (1 except at radius 4-5)
  // x pass: the four outputs at tile x = 4c .. 4c+3 read tile columns 4c + RP - R .. 4c + RP + R + 3
  static constexpr int XA0 = (RP - R) / 4 * 4;     // first 16-B group read, relative to 4c
  static constexpr int XOFF = (RP - R) - XA0;      // position of the first tap inside it
  static constexpr int XNR = (XOFF + 4 + 2 * R + 3) / 4;   // 16-B LDS reads per item
  static constexpr int XI = UH * (TX / 4);         // x-pass items per component
  static constexpr int NXI = (3 * XI + NTH - 1) / NTH;
  static constexpr int SZ_U = 3 * UH * UW;         // smoothing input (3 components)
  static constexpr int SZ_X = 3 * UH * TX;         // after the x pass
  static_assert(UH * UW < 32768, "LDS slots are 16 bit");
};

// Plane-invariant description of one strip of a thread.  A strip whose voxels leave the volume in x is loaded from
// the nearest 4 in-row voxels (xl) and re-mapped element-wise (jm: 2 bits per element = index into the loaded four)
// so that every position holds the value of its clamped voxel (ZeroFluxNeumann); rows are clamped by address.
struct pp_strip {
  unsigned goff;   // in-plane BYTE offset of the four voxels that are loaded
  int slot;        // float index of the strip in the LDS tile (< 0: this thread has no such strip)
  unsigned jm;     // element map; 0xE4 = identity
};
__device__ __forceinline__ pp_strip pp_strip_setup(int s, int ns, int spr, int uw, int x_first, int y_first, const pp_dims& d, int px) {
  pp_strip st;
  const int ss = s < ns ? s : 0;
  const int uy = ss / spr, sx = ss - uy * spr;
  const int xs = x_first + 4 * sx;
  const int yc = pp_clampi(y_first + uy, 0, d.ny - 1);
  const int xl = pp_clampi(xs, 0, d.nx - 4);
  unsigned jm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) jm |= (unsigned)(pp_clampi(xs + i, 0, d.nx - 1) - xl) << (2 * i);
  st.goff = ((unsigned)yc * (unsigned)px + (unsigned)xl) * 4u;   // (px: row pitch in voxels, >= d.nx)
  st.slot = s < ns ? uy * uw + 4 * sx : -1;
  st.jm = jm;
  return st;
}
// Element map of a strip that leaves the volume in x, applied in place on the thread's own freshly written LDS strip
// (NC components `cstride` floats apart): position i takes loaded element (jm >> 2i) & 3.  Same-thread LDS accesses
// are ordered, so no barrier is involved; only lanes of x-border tiles come here.
template <int NC>
__device__ __forceinline__ void pp_strip_remap_lds(float* strip, int cstride, unsigned jm) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float* p = strip + c * cstride;
    const float e0 = p[jm & 3u], e1 = p[(jm >> 2) & 3u], e2 = p[(jm >> 4) & 3u], e3 = p[(jm >> 6) & 3u];
    *reinterpret_cast<float4*>(p) = make_float4(e0, e1, e2, e3);
  }
}

// 16 bytes from a 16-byte-aligned LDS address as ONE ds_read_b128.  (Left to the compiler, the x pass's two float4 loads are
// re-cut by the SLP vectoriser into the operand pairs of its v_pk_fma_f32 -- ds_read2_b32 / ds_read_b64 at odd dword offsets
// and a 16-byte lane stride: 2- and 4-way bank conflicts, most of kernel A's 34 % LDS conflict cycles.)  The wait is in the
// statement because the compiler does not count loads it cannot see.
__device__ __forceinline__ void pp_lds_read2x128(const float* p, float4& a, float4& b) {
#if defined(__HIP_DEVICE_COMPILE__) && PP_A_XASM
  const unsigned addr = (unsigned)(size_t)p;   // (LDS pointers are 32-bit offsets in the low half of a flat address)
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
#else
  a = *reinterpret_cast<const float4*>(p);
  b = *reinterpret_cast<const float4*>(p + 4);
#endif
}

// x pass with per-thread precomputed LDS offsets (float indices; src < 0: no item).
template <int R, int NXI, int NITEMS>
__device__ __forceinline__ void fused2_xpass(const float* __restrict__ us, float* __restrict__ xs, const pp_taps_small& wx,
                                             const int (&xsrc)[NXI], const int (&xdst)[NXI]) {
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    if ((i + 1) * 512 > NITEMS && xsrc[i] < 0) continue;   // only the last round can be partial
    const float* src = us + xsrc[i];
    float in[4 + 2 * R];
    if constexpr ((4 + 2 * R) / 4 == 2) {   // radii 2 and 3: two whole 16-byte groups
      float4 v0, v1;
      pp_lds_read2x128(src, v0, v1);
      in[0] = v0.x; in[1] = v0.y; in[2] = v0.z; in[3] = v0.w;
      in[4] = v1.x; in[5] = v1.y; in[6] = v1.z; in[7] = v1.w;
    } else {
#pragma unroll
      for (int q = 0; q < (4 + 2 * R) / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
        in[4 * q + 0] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
      }
    }
    if constexpr ((4 + 2 * R) % 4 == 2) {
      const float2 v = *reinterpret_cast<const float2*>(src + (4 + 2 * R) / 4 * 4);
      in[(4 + 2 * R) / 4 * 4 + 0] = v.x;
      in[(4 + 2 * R) / 4 * 4 + 1] = v.y;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < 2 * R + 1; ++k) s = fmaf(wx.h[k < R ? R - k : k - R], in[j + k], s);
      o[j] = s;
    }
    *reinterpret_cast<float4*>(xs + xdst[i]) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Kernel A's x-pass tile s_x: row pitch.  With PP_A_XPERM the lanes of a wave are dealt to (row, column group) items so that
// every lane group the hardware services a ds_read_b128 in -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- reads ONE row's 16 consecutive 16-byte groups: 64 consecutive banks, no conflict, whatever
// the pitch of the source tile (in plain lane order a group straddles two rows 68 floats apart and one bank quartet is hit
// twice: every x-pass read ran at half rate, half of kernel A's 34 % LDS conflict cycles).  The 8-lane groups of the
// ds_write_b128 then hold quarter rows of TWO rows; the pitch of s_x moves them 16 banks apart (64 + 16 floats).
#ifndef PP_A_XPERM
#define PP_A_XPERM 1
#endif
template <int SH>
struct fused2_xtile {
  static constexpr bool PERM = (PP_A_XPERM != 0) && tile_shape<SH>::TX == 64;
  static constexpr int XP = tile_shape<SH>::TX + (PERM ? 16 : 0);   // row pitch of s_x (floats)
};
__device__ __forceinline__ int fused2_xperm_lane(int lane) {   // lane of a wave -> position in the wave's item order
  const int l = lane & 31, half = lane & 32;
  int g, pos;
  if (l < 4) { g = 0; pos = l; }
  else if (l < 12) { g = 1; pos = l - 4; }
  else if (l < 16) { g = 0; pos = l - 8; }
  else if (l < 20) { g = 1; pos = l - 8; }
  else if (l < 28) { g = 0; pos = l - 12; }
  else { g = 1; pos = l - 16; }
  return half + 16 * g + pos;
}
template <int R, int SH, int NXI>
__device__ __forceinline__ void fused2_xpass_setup(int t, int (&xsrc)[NXI], int (&xdst)[NXI]) {
  using G = fused_geom<R, 2, SH>;
  constexpr int XP = fused2_xtile<SH>::XP;
  const int tp = fused2_xtile<SH>::PERM ? ((t & ~63) | fused2_xperm_lane(t & 63)) : t;
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int it = tp + i * G::NTH;
    if (it < 3 * G::XI) {
      const int c = it / G::XI;
      const int rem = it - c * G::XI;
      const int uy = rem / (G::TX / 4);
      const int c4 = rem - uy * (G::TX / 4);
      xsrc[i] = (c * G::UH + uy) * G::UWP + 4 * c4;
      xdst[i] = (c * G::UH + uy) * XP + 4 * c4;
    } else {
      xsrc[i] = -1;
      xdst[i] = 0;
    }
  }
}

// y pass: this thread's two outputs of component c (s_x row pitch XP, `yb` = cy * XP + 2 cx).
template <int R, int SH>
__device__ __forceinline__ void fused2_ypass(const float* __restrict__ xs, int c, int yb, const pp_taps_small& wy, float v[2]) {
  using G = fused_geom<R, 2, SH>;
  constexpr int XP = fused2_xtile<SH>::XP;
  v[0] = 0.0f;
  v[1] = 0.0f;
#pragma unroll
  for (int k = 0; k < 2 * R + 1; ++k) {
    const float2 a = *reinterpret_cast<const float2*>(xs + (c * G::UH + k) * XP + yb);
    const float w = wy.h[k < R ? R - k : k - R];
    v[0] = fmaf(w, a.x, v[0]);
    v[1] = fmaf(w, a.y, v[1]);
  }
}

// x pass over a strip-layout tile: item i of this thread reads XNR aligned 16-B groups of `us` (pitch UW) starting at
// xsrc[i] and writes four outputs to xs at xdst[i].
template <int R, class SG>
__device__ __forceinline__ void fused2_xpass_strips(const float* __restrict__ us, float* __restrict__ xs, const pp_taps_small& wx,
                                                    const int (&xsrc)[SG::NXI], const int (&xdst)[SG::NXI]) {
#pragma unroll
  for (int i = 0; i < SG::NXI; ++i) {
    if ((i + 1) * SG::NTH > 3 * SG::XI && xsrc[i] < 0) continue;   // only the last round can be partial
    const float* src = us + xsrc[i];
    float in[4 * SG::XNR];
#pragma unroll
    for (int q = 0; q < SG::XNR; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      in[4 * q + 0] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.0f;
#pragma unroll
      for (int k = 0; k < 2 * R + 1; ++k) a = fmaf(wx.h[k < R ? R - k : k - R], in[SG::XOFF + j + k], a);
      o[j] = a;
    }
    *reinterpret_cast<float4*>(xs + xdst[i]) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
template <class SG>
__device__ __forceinline__ void fused2_xpass_strips_setup(int t, int (&xsrc)[SG::NXI], int (&xdst)[SG::NXI]) {
#pragma unroll
  for (int i = 0; i < SG::NXI; ++i) {
    const int it = t + i * SG::NTH;
    if (it < 3 * SG::XI) {
      const int c = it / SG::XI;
      const int rem = it - c * SG::XI;
      const int uy = rem / (SG::TX / 4);
      const int c4 = rem - uy * (SG::TX / 4);
      xsrc[i] = (c * SG::UH + uy) * SG::UW + 4 * c4 + SG::XA0;
      xdst[i] = (c * SG::UH + uy) * SG::TX + 4 * c4;
    } else {
      xsrc[i] = -1;
      xdst[i] = 0;
    }
  }
}
// y pass on the x-pass output (pitch TX), rows of the smoothing-input tile
template <int R, class SG>
__device__ __forceinline__ void fused2_ypass_strips(const float* __restrict__ xs, int c, int yb, const pp_taps_small& wy, float v[2]) {
  v[0] = 0.0f;
  v[1] = 0.0f;
#pragma unroll
  for (int k = 0; k < 2 * R + 1; ++k) {
    const float2 a = *reinterpret_cast<const float2*>(xs + (c * SG::UH + k) * SG::TX + yb);
    const float w = wy.h[k < R ? R - k : k - R];
    v[0] = fmaf(w, a.x, v[0]);
    v[1] = fmaf(w, a.y, v[1]);
  }
}

// x pass of kernel B in registers: a lane holds one strip (four consecutive x voxels, three components) and takes the R
// voxels either side from the neighbouring lanes with wavefront shuffles -- strips are laid out RPW whole rows per wave
// (strip_lanes below), so a strip's x neighbours are the adjacent lanes.  The x-passed strip goes straight to the
// y pass's LDS buffer: no staging of the raw tile, one LDS trip and one barrier per plane less than fused2_xpass_strips.
// Same products, same summation order.
template <class G>
struct strip_lanes {
  static constexpr int RPW = 64 / G::SPR;                      // whole rows per wave
  static_assert(G::RP == 4 && G::NSL == 1, "one halo strip either side, one strip per thread");
  static_assert(RPW >= 1 && (G::UH + RPW - 1) / RPW <= G::NTH / 64, "the tile's rows fit the block's waves");
};
// (pp_lane_prev / pp_lane_next -- the DPP whole-wave shifts -- live in pp_internal.h: pp_fir.hip's fused Gaussian uses them too)
__device__ __forceinline__ float pp_pick4(float ax, float ay, float az, float aw, unsigned i) {   // (selects on scalars: no vector indexing)
  const float lo = (i & 1u) ? ay : ax, hi = (i & 1u) ? aw : az;
  return (i & 2u) ? hi : lo;
}
template <int R, class G>
__device__ __forceinline__ void fused2_xpass_shfl(const float4 (&strip)[3], unsigned jm, bool has_out, int xoff, float* __restrict__ xs,
                                                  const pp_taps_small& wx) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sx_ = strip[c].x, sy_ = strip[c].y, sz_ = strip[c].z, sw_ = strip[c].w;
    float own[4] = {sx_, sy_, sz_, sw_};
    if (jm != 0xE4u) {   // strips that leave the volume in x (x-border tiles only): every position takes its clamped voxel
      const unsigned j = pp_opaque(jm);   // (the eight selects' predicates are formed here, not hoisted as 16 scalar registers)
      own[0] = pp_pick4(sx_, sy_, sz_, sw_, j & 3u);
      own[1] = pp_pick4(sx_, sy_, sz_, sw_, (j >> 2) & 3u);
      own[2] = pp_pick4(sx_, sy_, sz_, sw_, (j >> 4) & 3u);
      own[3] = pp_pick4(sx_, sy_, sz_, sw_, (j >> 6) & 3u);
    }
    float in[4 + 2 * R];
#pragma unroll
    for (int q = 0; q < R; ++q) in[q] = pp_lane_prev(own[4 - R + q]);          // left neighbour's last R voxels
#pragma unroll
    for (int q = 0; q < 4; ++q) in[R + q] = own[q];
#pragma unroll
    for (int q = 0; q < R; ++q) in[R + 4 + q] = pp_lane_next(own[q]);        // right neighbour's first R voxels
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 2 * R + 1; ++k) acc = fmaf(wx.h[k < R ? R - k : k - R], in[j + k], acc);
      o[j] = acc;
    }
    if (has_out) *reinterpret_cast<float4*>(xs + c * G::UH * G::TX + xoff) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// z window: slot P receives the newest plane; the dot runs oldest -> newest like zring::dot.
template <int R, int P>
__device__ __forceinline__ void fused2_ring_step(float (&rg)[3][2][2 * R + 1], const float (&v)[3][2], const pp_taps_small& wz,
                                                 float (&out)[3][2]) {
  constexpr int W = 2 * R + 1;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rg[c][j][P] = v[c][j];
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < W; ++k) s = fmaf(wz.h[k < R ? R - k : k - R], rg[c][j][(P + 1 + k) % W], s);
      out[c][j] = s;
    }
}
// P < 0: the window is shifted by register moves (one loop body for every plane; radii whose (2R+1)-fold unrolled body
// would not fit the instruction cache).
template <int R, int P>
__device__ __forceinline__ void fused2_ring(float (&rg)[3][2][2 * R + 1], const float (&v)[3][2], const pp_taps_small& wz,
                                            float (&out)[3][2]) {
  if constexpr (P >= 0) {
    fused2_ring_step<R, P>(rg, v, wz, out);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) rg[c][j][k] = rg[c][j][k + 1];
        rg[c][j][2 * R] = v[c][j];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) s = fmaf(wz.h[k < R ? R - k : k - R], rg[c][j][k], s);
        out[c][j] = s;
      }
  }
}
template <int P>
struct pp_phase { static constexpr int value = P; };
// Plane loop driver: step(n, pp_phase<P>) for n = 0 .. nsteps-1.  UNROLL: the body is instantiated once per
// window phase (2R+1 copies, the z window is renamed instead of moved); otherwise one copy with P = -1.
template <int P, int W>
struct pp_phase_unroll {
  template <class F>
  static __device__ __forceinline__ bool run(F& step, int n0, int nsteps) {
    if (n0 + P >= nsteps) return false;
    step(n0 + P, pp_phase<P>{});
    return pp_phase_unroll<P + 1, W>::run(step, n0, nsteps);
  }
};
template <int W>
struct pp_phase_unroll<W, W> {
  template <class F>
  static __device__ __forceinline__ bool run(F&, int, int) { return true; }
};
template <int R, bool UNROLL, class F>
__device__ __forceinline__ void fused2_plane_loop(F& step, int nsteps) {
  if constexpr (UNROLL) {
    for (int n0 = 0; n0 < nsteps; n0 += 2 * R + 1)
      if (!pp_phase_unroll<0, 2 * R + 1>::run(step, n0, nsteps)) break;
  } else {
    for (int n = 0; n < nsteps; ++n) step(n, pp_phase<-1>{});
  }
}
// Round 4: the plane loop in three parts.  Steps s_lo <= n < s_hi are STEADY: every wave-uniform condition of the step (a
// fresh plane either side, an output plane, a plane to prefetch, every lane inside the volume) holds, so `step(n, phase,
// pp_steady<true>)` is straight-line code around its memory instructions.  That matters for more than the branches: the
// hardware's vmcnt counts loads AND stores in issue order and the compiler's s_waitcnt pass merges the states of all paths
// into a step, so with the conditions dynamic every wait for an old load was emitted as the wait of the worst path --
// vmcnt(0) right behind the warp's gathers, between the field stores, at the top of a step -- and each wave sat out the
// acknowledgement of the stores and the HBM latency of the loads it had issued a moment earlier (47-61 % of wave cycles in
// SQ_WAIT_ANY, profiles/round3_pmc_counters.md).  The steady steps run in whole groups of 2R+1 window phases (renamed z
// window); the steps before and after run the general step with the shifting window (P = -1), whose window order equals the
// renamed one's at every group boundary.
template <bool B>
struct pp_steady { static constexpr bool value = B; };
template <int P, int W>
struct pp_steady_unroll {
  template <class F>
  static __device__ __forceinline__ void run(F& step, int n0) {
    step(n0 + P, pp_phase<P>{}, pp_steady<true>{});
    pp_steady_unroll<P + 1, W>::run(step, n0);
  }
};
template <int W>
struct pp_steady_unroll<W, W> {
  template <class F>
  static __device__ __forceinline__ void run(F&, int) {}
};
#ifndef PP_STEADY
#define PP_STEADY 0
#endif
#ifndef PP_A_SPLIT_LDS
#define PP_A_SPLIT_LDS 1
#endif
#ifndef PP_B_XLATE
#define PP_B_XLATE 1
#endif
#ifndef PP_B_VOTE
#define PP_B_VOTE 1
#endif
#ifndef PP_A_VOTE
#define PP_A_VOTE 1
#endif
#ifndef PP_A_ESM_PAIRS
#define PP_A_ESM_PAIRS 1
#endif
#ifndef PP_A_FLIP
#define PP_A_FLIP 0   // (measured, tools/r5/gpu6.sh: kernel B's role flip -1 % on B in three alternating rounds, kernel A's none)
#endif
#ifndef PP_B_FLIP
#define PP_B_FLIP 1
#endif
#ifndef PP_B_FINISH_AFTER_BARRIER
#define PP_B_FINISH_AFTER_BARRIER 1
#endif
#ifndef PP_A_STEADY
#define PP_A_STEADY 0
#endif
template <int R, bool UNROLL, class F>
__device__ __forceinline__ void fused2_plane_loop3(F& step, int nsteps, int s_lo, int s_hi) {
  constexpr int W = 2 * R + 1;
  int n = 0;
  if (s_hi - s_lo >= (UNROLL ? W : 1)) {
    for (; n < s_lo; ++n) step(n, pp_phase<-1>{}, pp_steady<false>{});
    if constexpr (UNROLL) {
      for (; n + W <= s_hi; n += W) pp_steady_unroll<0, W>::run(step, n);
    } else {
      for (; n < s_hi; ++n) step(n, pp_phase<-1>{}, pp_steady<true>{});
    }
  }
  for (; n < nsteps; ++n) step(n, pp_phase<-1>{}, pp_steady<false>{});
}


// ---- Round 5: soft synchronisation of the blocks that share an L2 -------------------------------------------------------
// The x/y halo of a tile (two 16-byte strips a row in x, 2R rows in y) lies in cache lines that belong to the NEIGHBOUR tiles:
// a block finds them in its XCD's L2 only while the neighbour's march is within a plane or two of its own (4 MB of L2 see
// ~1-2 MB of traffic per plane step of the 64 resident blocks).  Nothing keeps them there: measured with -DPP_DRIFT, the 64
// blocks of an XCD are spread over 45 (kernel A) to 70 (kernel B) plane steps of a 132-step march (the two blocks of a CU do
// not share it evenly), and kernel B's fetch swings between 1.13 and 1.6 x its compulsory reads with launch timing.
// With PP_SOFTSYNC each block of an XCD (block b runs on XCD b % 8, the assumption the tile order already makes -- speed
// only) publishes the number of plane steps it has finished in a word of its own, and starts step n + 1 only when EVERY
// block of its group has finished step n + 1 - lag: the leaders wait, nobody else.
//   * Plain stores and L1-bypassing loads (workgroup scope: `global_store / global_load sc0`), both served by the XCD's own
//     L2.  Not atomics: on this part a device atomic is executed beyond the L2 and takes microseconds, and since a wave's
//     memory instructions retire in issue order every younger load of the step waited behind it -- the first two versions
//     (one counter per XCD, agent- and workgroup-scope fetch_add) ran the kernels 2.7-4 x slower (profiles/round5_softsync.md).
//   * The wait is bounded: a group that is not co-resident, or a dispatcher that spreads a group over XCDs (whose L2s would
//     then each see their own blocks' words only), costs one time-out per block and switches that block's waiting off.
//     Slower, never wrong, and no launch can hang on it.
//   * One wave per block does all of it; the words it tests were requested a whole step earlier.
#ifndef PP_SOFTSYNC
#define PP_SOFTSYNC 0
#endif
#ifndef PP_SOFTSYNC_TRIES
#define PP_SOFTSYNC_TRIES 24
#endif
constexpr unsigned PP_SYNC_GROUP = 64;      // words per XCD: one per resident block (2 blocks x 32 CUs)
constexpr unsigned PP_SYNC_DONE = 0x40000000u;
struct pp_softsync {
  unsigned* words;   // this XCD's PP_SYNC_GROUP progress words
  unsigned* mine;    // this block's word
  const unsigned* peek_at;   // the word this LANE watches (wave 0: lane l watches block l of the group)
  unsigned seen;     // ... as last read
  int lag;           // <= 0: publish only
  bool wave0;        // this wave publishes and waits for its block (wave-uniform, held in scalar registers)
};
#ifndef PP_SOFTSYNC_LOAD_SCOPE
#define PP_SOFTSYNC_LOAD_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
#ifndef PP_SOFTSYNC_STORE_SCOPE
#define PP_SOFTSYNC_STORE_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
__device__ __forceinline__ unsigned pp_sync_peek(const unsigned* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, PP_SOFTSYNC_LOAD_SCOPE); }
__device__ __forceinline__ void pp_softsync_init(pp_softsync& y, const fused_args& a, unsigned* other_set) {
  const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const unsigned ntiles = ((unsigned)a.gx * a.gy + (unsigned)a.gx2 * a.gy2) * (unsigned)a.gz;
  const unsigned first = xcd * (unsigned)a.per_xcd;
  const unsigned group = ntiles > first ? (ntiles - first < (unsigned)a.per_xcd ? ntiles - first : (unsigned)a.per_xcd) : 0u;
  y.words = a.sync + xcd * PP_SYNC_GROUP;
  y.mine = y.words + (j < PP_SYNC_GROUP ? j : 0u);
  const unsigned lane = threadIdx.x & 63u;
  y.peek_at = (lane < group && lane < PP_SYNC_GROUP) ? y.words + lane : y.mine;
  y.seen = 0u;
  y.lag = (group <= PP_SYNC_GROUP && j < PP_SYNC_GROUP) ? a.sync_lag : 0;
  y.wave0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64;
  // the OTHER kernel's words are idle while this launch runs: clear this block's for that kernel's next launch
  if (threadIdx.x == 0 && j < PP_SYNC_GROUP) __hip_atomic_store(other_set + xcd * PP_SYNC_GROUP + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Start of a plane step: request the group's words (consumed at the end of the step).
__device__ __forceinline__ void pp_softsync_peek(pp_softsync& y) {
  if (y.wave0) y.seen = pp_sync_peek(y.peek_at);
}
// End of plane step n (0-based) of a block, ahead of the barrier that closes the step; every wave calls it, wave 0 acts.
__device__ __forceinline__ void pp_softsync_step(pp_softsync& y, int n) {
  if (y.wave0) {
    if (threadIdx.x == 0) __hip_atomic_store(y.mine, (unsigned)(n + 1), __ATOMIC_RELAXED, PP_SOFTSYNC_STORE_SCOPE);
    if (y.lag > 0 && n + 1 > y.lag) {
      const unsigned need = (unsigned)(n + 1 - y.lag);
      if (__any(y.seen < need)) {
        int tries = 0;
        bool behind;
        do {
          __builtin_amdgcn_s_sleep(4);
          behind = __any(pp_sync_peek(y.peek_at) < need);
        } while (behind && ++tries < PP_SOFTSYNC_TRIES);
        if (behind) y.lag = 0;   // timed out: this block stops waiting (it still publishes)
      }
    }
  }
}
// ---- Round 5: the two blocks of a CU, kept level by wave priority ------------------------------------------------------
// -DPP_DRIFT shows where the drift comes from: block b + 256 of a 512-block launch -- the SECOND block on the CU that block b
// got first (XCD-local indices j and j + 32) -- finishes 75-115 us (kernel A) / 125-175 us (kernel B) after block b, every pair,
// while the first blocks of an XCD finish within 20 us of each other: the SIMD arbiter serves the older wave first, so the older
// block runs as if alone and the younger one fills its gaps, and for the last fifth to third of the launch every CU runs ONE
// block (8 waves: half the latency hiding).  With PP_PAIRPRIO each block publishes its finished plane steps (the soft barrier's
// words), reads its partner's once a step, and every wave of the block that is BEHIND raises its issue priority (s_setprio) for
// the next step: bang-bang control that keeps the pair within a step or two of each other, so that both finish together and
// the CU runs two blocks to the end.  Wrong pairing (another dispatcher, another kernel's block as neighbour) only means a
// priority that helps nobody.
#ifndef PP_PAIRPRIO
#define PP_PAIRPRIO 1   // (measured, tools/kbench/ab.sh main prio1 ...: pair end difference 97 / 153 us -> 6 us, iteration -1.9 .. -3.5 % at
                        // 512 x 512 x 256, -3 % at 341 x 341 x 171, kernel B's fetch -11 %; profiles/round5_pair_priority.md)
#endif
#ifndef PP_PAIRPRIO_LEVEL
#define PP_PAIRPRIO_LEVEL 1
#endif
#ifndef PP_PAIRPRIO_ASYM
#define PP_PAIRPRIO_ASYM 0    // (measurement: the second block of a pair tolerates being this many more steps behind -- an anti-phase pair)
#endif
#ifndef PP_PAIRPRIO_SLACK
#define PP_PAIRPRIO_SLACK 1   // behind = the partner's count (a step old) exceeds this block's finished steps n + 1 - 1 + SLACK
#endif
struct pp_pairprio {
  unsigned* mine;            // this block's progress word
  const unsigned* partner;   // the progress word of the block that shares the CU
  unsigned seen;             // ... as last read (requested at the start of a step, used at its end)
  bool has;                  // a partner exists in this launch
  bool wave0;
};
__device__ __forceinline__ void pp_pairprio_init(pp_pairprio& y, const fused_args& a, unsigned* other_set) {
  const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const unsigned ntiles = ((unsigned)a.gx * a.gy + (unsigned)a.gx2 * a.gy2) * (unsigned)a.gz;
  const unsigned first = xcd * (unsigned)a.per_xcd;
  const unsigned group = ntiles > first ? (ntiles - first < (unsigned)a.per_xcd ? ntiles - first : (unsigned)a.per_xcd) : 0u;
  unsigned* const words = a.sync + xcd * PP_SYNC_GROUP;
  const unsigned pj = j ^ 32u;
  y.has = j < PP_SYNC_GROUP && pj < group && group <= PP_SYNC_GROUP;
  y.mine = words + (j < PP_SYNC_GROUP ? j : 0u);
  y.partner = words + (y.has ? pj : (j < PP_SYNC_GROUP ? j : 0u));
  y.seen = 0u;
  y.wave0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64;
  if (threadIdx.x == 0 && j < PP_SYNC_GROUP) __hip_atomic_store(other_set + xcd * PP_SYNC_GROUP + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pp_pairprio_peek(pp_pairprio& y) {   // every wave: one uniform word, beyond the (stale) vector cache AND L2 copy
  if (y.has) y.seen = __hip_atomic_load(y.partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pp_pairprio_step(pp_pairprio& y, int n) {
  if (y.wave0 && threadIdx.x == 0) __hip_atomic_store(y.mine, (unsigned)(n + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (y.has) {
    const unsigned theirs = (unsigned)__builtin_amdgcn_readfirstlane((int)y.seen);
    if (theirs > (unsigned)(n + PP_PAIRPRIO_SLACK + (PP_PAIRPRIO_ASYM ? (int)((blockIdx.x >> 8) & 1u) * PP_PAIRPRIO_ASYM : 0))) __builtin_amdgcn_s_setprio(PP_PAIRPRIO_LEVEL);
    else __builtin_amdgcn_s_setprio(0);
  }
}
__device__ __forceinline__ void pp_pairprio_finish(pp_pairprio& y) {
  if (threadIdx.x == 0) __hip_atomic_store(y.mine, PP_SYNC_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A block that has finished its march never holds anybody back.
__device__ __forceinline__ void pp_softsync_finish(pp_softsync& y) {
  if (threadIdx.x == 0) __hip_atomic_store(y.mine, PP_SYNC_DONE, __ATOMIC_RELAXED, PP_SOFTSYNC_STORE_SCOPE);
}
// Measurement builds (-DPP_DRIFT, tools/kbench): 100 MHz wall-clock stamps of EVERY block at the quarter points of its march.
#ifdef PP_DRIFT
__device__ unsigned long long pp_drift_buf[2][1024][4];
__device__ unsigned pp_drift_xcc[2][1024];   // HW_REG_XCC_ID of the block's first wave: which XCD did block b really run on?
#define PP_DRIFT_MARK(kern, n, nsteps)                                                                                  \
  do {                                                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                        \
      const int q_ = ((n) == (nsteps) / 4) ? 0 : ((n) == (nsteps) / 2) ? 1 : ((n) == 3 * (nsteps) / 4) ? 2 : ((n) == (nsteps)-1) ? 3 : -1; \
      if (q_ >= 0) pp_drift_buf[kern][blockIdx.x][q_] = wall_clock64();                                                 \
      if (q_ == 0) pp_drift_xcc[kern][blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));              \
    }                                                                                                                   \
  } while (0)
#else
#define PP_DRIFT_MARK(kern, n, nsteps) do { } while (0)
#endif

// ---- kernel B, generation 2: D' = G_d * (D + U), then the next iteration's warped moving image --------
// SUM: `Us` already holds D + U (kernel A<SUM> added D at its own output voxels, where it needs no halo), `D` is not read:
// three halo'd arrays instead of six.  The sum is the same fp32 add on the same operands, so the fields are bit-identical.
// MASK (round 4): every memory instruction of a plane step is issued by every wave on every step -- loads from clamped
// addresses, stores with the lane mask carried by the offset (PP_OOB) -- so that no branch surrounds one.  The s_waitcnt pass
// merges the counter states of all paths into a point, and vmcnt retires loads AND stores in issue order: with the stores
// and gathers inside `if (emit)` / `if (out_ok)` regions every wait for an old load was emitted as the wait of the worst
// path (vmcnt(0) right behind the warp's gathers, between the field stores, at the top of a step), and each wave sat out the
// acknowledgement of its own stores and the HBM latency of the loads it had just issued.  The host picks MASK when rows are
// even (8-byte buffer stores need 8-byte alignment) and a whole field spans < 2^31 bytes.
// LDS floats of kernel B for one tile shape
template <int R, int SH, bool SUM>
struct fused2_warp_lds {
  using G = strip_geom<R, SH, 0>;
  static constexpr bool XS = (PP_B_XSHFL != 0) && SUM;
  static constexpr int N = XS ? 2 * G::SZ_X : G::SZ_X + G::SZ_U;
};
// The kernel's body for ONE tile shape (SH 0 / 1); `smem` = the block's LDS (fused2_warp_lds<R, SH, SUM>::N floats), `region`:
// fused_tile.  The __global__ wrapper below owns the LDS and, in a mixed launch, picks the shape per block.
// BIG: the three-component field arrays span >= 2^32 bytes (e.g. 512 x 512 x 1400): their component / plane offsets no longer fit
// the scalar-offset operand of ONE resource per array, so those accesses rebuild a resource from a 64-bit base per component and
// plane (the PP_SOFF = 0 form); the scalar images (< 2^32 bytes each: checked on the host) keep theirs.
template <int R, int SH, bool UNROLL, bool SUM, bool NT, bool MASK, bool BIG = false>
__device__ __forceinline__ void fused2_warp_body(const float* __restrict__ D, const float* __restrict__ Us, const float* __restrict__ M,
                                                 float* __restrict__ Dn, float* __restrict__ Mw, const fused_args& a,
                                                 const pp_warp_scale& sc, const int* __restrict__ halt, float* const smem,
                                                 const int region) {
  using G = strip_geom<R, SH, 0>;
  constexpr int NTH = G::NTH, TX = G::TX, TY = G::TY, W = 2 * R + 1;
  constexpr bool XS = (PP_B_XSHFL != 0) && SUM;   // x pass in registers (wavefront shuffles), y-pass buffer double-buffered
  constexpr bool SOFF_F = (PP_SOFF != 0) && !BIG, SOFF_S = (PP_SOFF != 0);   // scalar-offset addressing of field / scalar arrays
  static_assert(!(BIG && MASK), "the MASK instances keep whole arrays under 2^31 bytes");
  float* const s_x = smem;
  float* const s_u = smem + G::SZ_X;   // (XS: the second y-pass buffer)
  if (halt && *halt) return;
  int tx0, ty0, z0;
  unsigned rank;
  if (!fused_tile(a, TX, TY, tx0, ty0, z0, rank, region)) return;

  const pp_dims d = a.d;
  const int t = threadIdx.x;
  const int cx = t % G::LX, cy = t / G::LX;
  const unsigned sy = (unsigned)a.px, sz = (unsigned)a.px * d.ny;   // rows are a.px voxels apart (>= d.nx: padded rows, see pp_demons.hip)
  const size_t N = (size_t)sz * d.nz;

  pp_strip st[G::NSL];
  int xsrc[G::NXI], xdst[G::NXI];
  bool xs_out = false;   // XS: this lane's strip lies in the tile's columns (not a halo strip) -> it stores an x-pass result
  int xs_off = 0;
  if constexpr (XS) {
    constexpr int RPW = strip_lanes<G>::RPW;
    // (strip rows by ROLE: three rows a wave for waves 0-5, two for wave 6, none for wave 7 -- the wave index is reversed in
    // every other block of a CU's pair, as in kernel A, so that a SIMD hosts a heavy wave of one block and a light one of the other)
    const int wrole = ((PP_B_FLIP != 0) && ((blockIdx.x >> 8) & 1u)) ? (NTH / 64 - 1) - (t >> 6) : (t >> 6);
    const int lane = t & 63, riw = lane / G::SPR, sx = lane - riw * G::SPR;
    const int uy = wrole * RPW + riw;
    const bool valid = riw < RPW && uy < G::UH;
    st[0] = pp_strip_setup(valid ? uy * G::SPR + sx : G::NS, G::NS, G::SPR, G::UW, tx0 - G::RP, ty0 - R, d, a.px);
    xs_out = valid && sx >= 1 && sx <= G::SPR - 2;
    xs_off = uy * TX + 4 * (sx - 1);
  } else {
#pragma unroll
    for (int i = 0; i < G::NSL; ++i) st[i] = pp_strip_setup(t + i * NTH, G::NS, G::SPR, G::UW, tx0 - G::RP, ty0 - R, d, a.px);
    fused2_xpass_strips_setup<G>(t, xsrc, xdst);
  }
  const int yb = cy * TX + 2 * cx;
  const int x = tx0 + 2 * cx, y = ty0 + cy;
  const bool out_ok = (y < d.ny) && (x < d.nx);
  const bool pair_ok = (a.px % 2) == 0;     // then the 8-B accesses are aligned and voxel x + 1 of a pair exists -- in the row or in its padding
  const unsigned o_xy = ((unsigned)y * sy + (unsigned)x) * 4u;
  const pp_warp_dims wd{d.nx, d.ny, d.nz, sy * 4u, sz * 4u};
  const char* const rm = reinterpret_cast<const char*>(M);
  const pp_rsrc r_dn = MASK ? pp_make_rsrc_masked(Dn) : pp_make_rsrc(Dn), r_mw = MASK ? pp_make_rsrc_masked(Mw) : pp_make_rsrc(Mw);   // one resource per output array (PP_SOFF)

  const int zs = z0 - R;
  const int zo_last = (z0 + a.zchunk - 1 < d.nz - 1) ? z0 + a.zchunk - 1 : d.nz - 1;
  const int ze = zo_last + R;
  const int zhi = pp_clampi(ze, 0, d.nz - 1);
  const int nsteps = ze - zs + 1;
  constexpr bool SYNC = (PP_SOFTSYNC != 0) && MASK;
  pp_softsync ysync{};
  if constexpr (SYNC) pp_softsync_init(ysync, a, a.sync_other);
  constexpr bool PRIO = (PP_PAIRPRIO != 0) && MASK;
  pp_pairprio yprio{};
  if constexpr (PRIO) pp_pairprio_init(yprio, a, a.sync_other);

  float4 dl[SUM ? 1 : 3][G::NSL], ul[3][G::NSL];   // raw D and U strips of the plane about to be published
  // (ALL: every lane loads -- a lane without a strip re-reads strip 0 of the tile, whose address pp_strip_setup gave it -- so
  // that no divergent branch surrounds the loads of a steady step: inside one, the compiler copies the loaded registers at
  // once and waits for the load it has just issued)
  auto load_plane = [&](int zc, auto all_tag) {
    constexpr bool ALL = decltype(all_tag)::value;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const char* const pd = SUM ? nullptr : reinterpret_cast<const char*>(D + c * N + (size_t)zc * sz);
      const char* const pu = reinterpret_cast<const char*>(Us + c * N + (size_t)zc * sz);
#pragma unroll
      for (int i = 0; i < G::NSL; ++i)
        if (ALL || (i + 1) * NTH <= G::NS || st[i].slot >= 0) {
#ifdef PP_ABL_NOLOAD
          if constexpr (!SUM) dl[c][i] = make_float4((float)st[i].goff * 1e-9f + (float)zc, 0.f, 1.f, 2.f);
          ul[c][i] = make_float4((float)st[i].goff * 1e-9f, 1.f, 0.f, 3.f);
#else
          if constexpr (!SUM) dl[c][i] = pp_gld4(pd, st[i].goff);
          ul[c][i] = pp_gld4(pu, st[i].goff);
#endif
        }
    }
  };
  auto publish = [&]() {
#pragma unroll
    for (int i = 0; i < G::NSL; ++i)
      if ((i + 1) * NTH <= G::NS || st[i].slot >= 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (SUM)
            *reinterpret_cast<float4*>(s_u + c * G::UH * G::UW + st[i].slot) = ul[c][i];
          else
            *reinterpret_cast<float4*>(s_u + c * G::UH * G::UW + st[i].slot) =
                make_float4(dl[c][i].x + ul[c][i].x, dl[c][i].y + ul[c][i].y, dl[c][i].z + ul[c][i].z, dl[c][i].w + ul[c][i].w);
        }
        if (st[i].jm != 0xE4u) pp_strip_remap_lds<3>(s_u + st[i].slot, G::UH * G::UW, st[i].jm);   // x-border tiles only
      }
  };

  float rg[3][2][W];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < W; ++k) rg[c][j][k] = 0.0f;
  float v[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};

  // prologue: first plane through publish + x pass, second plane in flight
  {
    const int zc0 = pp_clampi(zs, 0, d.nz - 1);
    load_plane(zc0, pp_steady<false>{});
    if constexpr (XS) {
      const float4 s3[3] = {ul[0][0], ul[1][0], ul[2][0]};
      fused2_xpass_shfl<R, G>(s3, st[0].jm, xs_out, xs_off, s_x, a.wx);
      if (zc0 < zhi) load_plane(zc0 + 1, pp_steady<false>{});
      __syncthreads();
    } else {
      publish();
      if (zc0 < zhi) load_plane(zc0 + 1, pp_steady<false>{});
      __syncthreads();
      fused2_xpass_strips<R, G>(s_u, s_x, a.wx, xsrc, xdst);
      __syncthreads();
    }
  }
  int ybuf = 0;   // XS: the buffer that holds the current plane's x-pass result

  const bool trace_on = (rank == (unsigned)(a.gx * (a.gy / 2) + a.gx / 2));   // (PP_TRACE builds: an interior tile, first z-chunk)
  (void)trace_on;
  // steady steps (fused2_plane_loop3): planes zi - 1, zi, zi + 1, zi + 2 exist, zi - R is an output plane of this chunk and
  // the tile lies inside the volume with even rows, so every lane stores pairs
  const bool tile_full = (tx0 + TX <= d.nx) && (ty0 + TY <= d.ny) && pair_ok;
  const int s_lo = (2 * R > 1 - zs) ? 2 * R : 1 - zs;
  const int s_hi = tile_full ? zhi - 1 - zs : 0;
  auto step = [&](int n, auto phase_tag, auto steady_tag) {
    constexpr int P = decltype(phase_tag)::value;
    constexpr bool ST = decltype(steady_tag)::value;
    const int zi = zs + n;
    const int cur = ST ? zi : pp_clampi(zi, 0, d.nz - 1);
    const bool fresh_cur = ST || (n == 0) || (cur != pp_clampi(zi - 1, 0, d.nz - 1));
    const int nxt = ST ? zi + 1 : pp_clampi(zi + 1, 0, d.nz - 1);
    const bool fresh_next = ST || ((n + 1 < nsteps) && (nxt != cur));
    PP_TRACE_MARK(trace_on, 1, n, 0);
    PP_DRIFT_MARK(1, n, nsteps);
    if constexpr (SYNC) pp_softsync_peek(ysync);
    if constexpr (PRIO) pp_pairprio_peek(yprio);
    // ---- interval 1: y pass of plane `cur` (reads s_x) | publish plane `nxt` (writes s_u) ----
    if (fresh_cur) {
#pragma unroll
      for (int c = 0; c < 3; ++c) fused2_ypass_strips<R, G>(XS ? smem + ybuf * G::SZ_X : s_x, c, yb, a.wy, v[c]);
    }
    // Round 4 (PP_B_XLATE): the x pass of plane `nxt` -- the consumer of the strips -- runs at the END of the interval, and the
    // next strips are requested right behind it: a strip load then has a whole plane step to arrive (it used to be issued
    // two thirds into a step and consumed at the top of the next one, ~1.4 us later: less than the loaded HBM latency).
    auto xpass_next = [&]() {
      if constexpr (XS) {
        if (fresh_next) {   // x pass of plane `nxt` in registers, into the other buffer
          const float4 s3[3] = {ul[0][0], ul[1][0], ul[2][0]};
          fused2_xpass_shfl<R, G>(s3, st[0].jm, xs_out, xs_off, smem + (ybuf ^ 1) * G::SZ_X, a.wx);
        }
      } else {
        if (fresh_next) publish();
      }
    };
    constexpr bool XLATE = XS && (PP_B_XLATE != 0);
    if constexpr (!XLATE) xpass_next();
    float dn[3][2];
    fused2_ring<R, P>(rg, v, a.wz, dn);
    const int zo = zi - R;
    const bool emit = ST || ((zo >= z0) && (zo <= zo_last));
    constexpr bool UNC = MASK || ST;   // memory instructions issued unconditionally (see MASK above the kernel)
    float mw0 = FLT_MAX, mw1 = FLT_MAX;
#ifdef PP_ABL_NOGATHER
    if (emit) {
      mw0 = dn[0][0] + dn[1][0] * dn[2][0];
      mw1 = dn[0][1] + dn[1][1] * dn[2][1];
    }
#else
    // The 8 corner loads of the two samples are issued here and consumed after the barrier (PP_B_DEFER): their latency hides
    // behind the strip loads, the field stores and the barrier instead of stalling the wave at once.  (UNC: also on steps
    // that emit nothing -- the displacement is clamped, so the addresses are valid, and the result is never stored.)
    pp_warp_pending g0, g1;
    bool wfast = false;   // (wave-uniform: the vote of fused2_warp_issue_pair)
    if (UNC || emit) {
#if PP_B_VOTE
      wfast = fused2_warp_issue_pair(rm, wd, x, y, zo, dn[0][0] * sc.ix, dn[1][0] * sc.iy, dn[2][0] * sc.iz, dn[0][1] * sc.ix,
                                     dn[1][1] * sc.iy, dn[2][1] * sc.iz, ST || out_ok, ST || (out_ok && (x + 1 < d.nx)), g0, g1);
#else
      fused2_warp_issue(rm, wd, x, dn[0][0] * sc.ix, y, dn[1][0] * sc.iy, zo, dn[2][0] * sc.iz, ST || out_ok, g0);
      fused2_warp_issue(rm, wd, x + 1, dn[0][1] * sc.ix, y, dn[1][1] * sc.iy, zo, dn[2][1] * sc.iz, ST || (out_ok && (x + 1 < d.nx)), g1);
#endif
#if !PP_B_DEFER
      if (wfast) {
        mw0 = fused2_warp_finish_interior(g0);
        mw1 = fused2_warp_finish_interior(g1);
      } else {
        mw0 = fused2_warp_finish(g0);
        mw1 = fused2_warp_finish(g1);
      }
#endif
    }
#endif
    // the plane after `nxt` goes in flight behind the gathers.  UNC pins the issue order gathers -> strip loads -> field
    // stores: vmcnt retires in issue order, so the wait for the gathers must not have the younger HBM loads ahead of it.
    auto load_next = [&]() {
      if constexpr (UNC) {
        __builtin_amdgcn_sched_barrier(0);
        // (steps that would not load re-read the plane the strips already hold: min(nxt + 1, zhi))
        const int zl = ST ? nxt + 1 : (nxt + 1 < zhi ? nxt + 1 : zhi);
        load_plane(zl, pp_steady<(G::NSL == 1)>{});
        __builtin_amdgcn_sched_barrier(0);
      } else {
        if (fresh_next && nxt < zhi) load_plane(nxt + 1, pp_steady<false>{});
      }
    };
    if constexpr (!XLATE) load_next();
    const int zoc = (UNC && !ST) ? pp_clampi(zo, z0, zo_last) : zo;   // (a plane of this chunk also when nothing is stored)
    const size_t po = (size_t)zoc * sz;
    const unsigned po4 = (unsigned)zoc * sz * 4u, N4 = (unsigned)N * 4u;   // (3 N * 4 < 2^32: checked on the host)
#ifdef PP_ABL_NOSTORE
    const bool do_store = emit && out_ok && dn[0][0] == 3.21e-29f && dn[1][1] == 1e-31f && dn[2][0] == 7e-33f && dn[0][1] == 2e-30f && dn[1][0] == 3e-30f && dn[2][1] == 4e-30f;
#else
    const bool do_store = ST || (emit && out_ok);
#endif
    // UNC: one store form (even rows), the lane mask in the offset
    const unsigned o_st = (MASK && !do_store) ? PP_OOB : o_xy;
    auto store_field = [&]() {
      if (UNC || pair_ok) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (SOFF_F) pp_bst2ss<NT>(r_dn, o_st, c * N4 + po4, dn[c][0], dn[c][1]);
          else pp_bst2s<NT>(pp_make_rsrc(Dn + c * N + po), o_st, dn[c][0], dn[c][1]);
        }
      } else if (x + 1 < d.nx) {   // odd row length: pairs at 4-byte alignment
#pragma unroll
        for (int c = 0; c < 3; ++c) pp_gst2(reinterpret_cast<char*>(Dn + c * N + po), o_xy, dn[c][0], dn[c][1]);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (SOFF_F) pp_bsts(r_dn, o_xy, c * N4 + po4, dn[c][0]);
          else pp_bst(pp_make_rsrc(Dn + c * N + po), o_xy, dn[c][0]);
        }
      }
    };
    auto store_image = [&]() {
      if (UNC || pair_ok) {
        if constexpr (SOFF_S) pp_bst2ss<NT>(r_mw, o_st, po4, mw0, mw1);
        else pp_bst2s<NT>(pp_make_rsrc(Mw + po), o_st, mw0, mw1);
      } else if (x + 1 < d.nx) {
        pp_gst2(reinterpret_cast<char*>(Mw + po), o_xy, mw0, mw1);
      } else {
        if constexpr (SOFF_S) pp_bsts(r_mw, o_xy, po4, mw0);
        else pp_bst(pp_make_rsrc(Mw + po), o_xy, mw0);
      }
    };
    if (UNC || do_store) {
      store_field();
#if !PP_B_DEFER || defined(PP_ABL_NOGATHER)
      store_image();
#endif
    }
    if constexpr (XLATE) {
      if constexpr (UNC) __builtin_amdgcn_sched_barrier(0);
      xpass_next();
      load_next();
    }
    if constexpr (UNC) __builtin_amdgcn_sched_barrier(0);
    // ---- interval 2: x pass of plane `nxt` (XS: already done above; one barrier hands the buffers over) ----
    PP_TRACE_MARK(trace_on, 1, n, 1);
    if constexpr (SYNC) pp_softsync_step(ysync, n);
    if constexpr (PRIO) pp_pairprio_step(yprio, n);
    if (fresh_next) {
      __syncthreads();
      PP_TRACE_MARK(trace_on, 1, n, 2);
      if constexpr (XS) {
        ybuf ^= 1;
      } else {
        fused2_xpass_strips<R, G>(s_u, s_x, a.wx, xsrc, xdst);
        __syncthreads();
      }
    }
#if PP_B_DEFER && !defined(PP_ABL_NOGATHER)
#if PP_B_FINISH_AFTER_BARRIER
    if constexpr (UNC) __builtin_amdgcn_sched_barrier(0);   // (keep the gathers' wait behind the barrier, where the source has it)
#endif
    if (UNC || emit) {
      if (wfast) {
        mw0 = fused2_warp_finish_interior(g0);
        mw1 = fused2_warp_finish_interior(g1);
      } else {
        mw0 = fused2_warp_finish(g0);
        mw1 = fused2_warp_finish(g1);
      }
    }
    if (UNC || do_store) store_image();
#endif
    PP_TRACE_MARK(trace_on, 1, n, 3);
  };
#if PP_STEADY
  fused2_plane_loop3<R, UNROLL>(step, nsteps, s_lo, s_hi);
#else
  (void)s_lo;
  (void)s_hi;
  auto step_general = [&](int n, auto phase_tag) { step(n, phase_tag, pp_steady<false>{}); };
  fused2_plane_loop<R, UNROLL>(step_general, nsteps);
#endif
  if constexpr (SYNC) pp_softsync_finish(ysync);
  if constexpr (PRIO) pp_pairprio_finish(yprio);
}

// SH 0 / 1: every tile of that shape.  SH 2: tiles of both shapes in one launch (fused_args: gx2 > 0) -- 64 x 16 wherever a
// whole 64-wide tile fits, 32 x 32 over the remaining columns -- with the LDS of the larger carve.
template <int R, int SH, bool UNROLL, bool SUM, bool NT, bool MASK, bool BIG = false>
__global__ void __launch_bounds__(512, PP_GEN2_WAVES) k_fused2_add_smooth_warp(const float* __restrict__ D, const float* __restrict__ Us,
                                                                            const float* __restrict__ M, float* __restrict__ Dn,
                                                                            float* __restrict__ Mw, fused_args a, pp_warp_scale sc,
                                                                            const int* __restrict__ halt) {
  if constexpr (SH == 2) {
    constexpr int N0 = fused2_warp_lds<R, 0, SUM>::N, N1 = fused2_warp_lds<R, 1, SUM>::N;
    __shared__ __attribute__((aligned(16))) float smem[N0 > N1 ? N0 : N1];
    if (fused_region(a) == 0) fused2_warp_body<R, 0, UNROLL, SUM, NT, MASK, BIG>(D, Us, M, Dn, Mw, a, sc, halt, smem, 0);
    else fused2_warp_body<R, 1, UNROLL, SUM, NT, MASK, BIG>(D, Us, M, Dn, Mw, a, sc, halt, smem, 1);
  } else {
    __shared__ __attribute__((aligned(16))) float smem[fused2_warp_lds<R, SH, SUM>::N];
    fused2_warp_body<R, SH, UNROLL, SUM, NT, MASK, BIG>(D, Us, M, Dn, Mw, a, sc, halt, smem, 0);
  }
}


// pp_esm_axis with the border rules carried by DATA instead of per-lane flags (the hoisted flag masks cost kernel A ~36
// scalar-register reloads per plane): a neighbour slot outside the volume holds the sentinel in its warped-image half, so
// "usable" needs no first/last-index test, and the fixed-image difference is scaled by hf = 0 on the first/last index
// (h elsewhere).  Same operations on the same operands as pp_esm_axis; on the border the fixed term is +-0 instead of +0.
__device__ __forceinline__ float pp_esm_axis_data(float fm, float fp, float mc, float mm, float mp, float hf, float inv_sp) {
  const float h = 0.5f * inv_sp;
  const bool up = (mp != FLT_MAX), um = (mm != FLT_MAX);
  const float hi_v = up ? mp : mc, lo_v = um ? mm : mc;
  const float su = up ? h : inv_sp;                 // (up && um) ? h : inv_sp as two selects on fresh compare masks (see pp_esm_voxel)
  const float wg = (hi_v - lo_v) * (um ? su : inv_sp);
  const float fg = (fp - fm) * hf;
  return fg + wg;
}

// The same on a (warped, fixed) PAIR as it lies in the packed image tile: both differences are one packed subtraction of the
// two ds_read_b64 results and both products one packed multiply -- registers that are pairs already.  Written as scalar
// statements the SLP vectoriser formed the same packed operations across the x and y axes instead, and paid eight v_mov_b32 per
// voxel to assemble their operands (a fifth of the voted path's instructions).  Same operations on the same operands as
// pp_esm_axis_plain -- two products, each rounded, then their sum -- so the same bits.
typedef float pp_v2f __attribute__((vector_size(8), may_alias));
__device__ __forceinline__ float pp_esm_axis_plain2(pp_v2f lo /* (m, f) below */, pp_v2f hi /* (m, f) above */, float inv_sp) {
  const float h = 0.5f * inv_sp;
  const pp_v2f hh = {h, h};
  const pp_v2f d = hi - lo;
  const pp_v2f p = d * hh;     // {wg, fg}
  const float wg = p[0], fg = p[1];
  return fg + wg;
}

// Kernel A's z window element: the (warped, fixed) values of one owned voxel on one plane -- a register PAIR (PP_A_ZPAIRS: the z
// gradient is then packed like the x and y ones and the window rotates by 64-bit moves) or two independent registers.
#ifndef PP_A_ZPAIRS
#define PP_A_ZPAIRS 0
#endif
#if PP_A_ZPAIRS
#define PP_ZWIN(name, n) pp_v2f name[n]
#else
#define PP_ZWIN(name, n) float name[n][2]
#endif

// One axis of the gradient where both neighbours exist and neither warped value is the sentinel: pp_esm_axis /
// pp_esm_axis_data with `up`, `um` true and the first / last-index factor h -- the same two statements, so the same roundings.
__device__ __forceinline__ float pp_esm_axis_plain(float fm, float fp, float mm, float mp, float inv_sp) {
  const float h = 0.5f * inv_sp;
  const float wg = (mp - mm) * h;
  const float fg = (fp - fm) * h;
  return fg + wg;
}

// ---- kernel A, generation 2: ESM update + 3-D Gaussian of the update -----------------------------------
// SUM: the stored volume is D + G_u * update (D read at the thread's own output voxels), what kernel B<SUM> smooths.
// MASK: as in kernel B -- every memory instruction of a plane step issued on every step, lane masks in the offsets.
// LDS floats of kernel A for one tile shape: the packed image tile, the smoothing input, the x-pass tile
template <int R, int SH>
struct fused2_force_lds {
  using G = fused_geom<R, 2, SH>;
  static constexpr int SZ_IMG2 = (2 * G::MH * G::MWP + 3) / 4 * 4;   // packed (moving, fixed) tile, floats
  static constexpr int SZ_U = G::SZ_U;
  static constexpr int SZ_XT = 3 * G::UH * fused2_xtile<SH>::XP;     // x-pass tile (>= G::SZ_X: the row pitch may be padded)
};
// The kernel's body for ONE tile shape; the three LDS objects come from the __global__ wrapper below (as for kernel B).
template <int R, int SH, bool UNROLL, bool SUM, bool NT, bool MASK, bool BIG = false>   // (BIG: as in fused2_warp_body)
__device__ __forceinline__ void fused2_force_body(const float* __restrict__ F, const float* __restrict__ Mw, const float* __restrict__ D,
                                                  float* __restrict__ Us, const fused_args& a, const pp_esm_consts& K,
                                                  double* __restrict__ partials, pp_dev_stats* __restrict__ st,
                                                  const double* __restrict__ prev, int nprev, double max_rms, float2* const s_mf,
                                                  float* const s_u, float* const s_x, const int region) {
  using G = fused_geom<R, 2, SH>;
  constexpr int NTH = G::NTH, TX = G::TX, TY = G::TY, W = 2 * R + 1;
  constexpr int NXI = (3 * G::XI + NTH - 1) / NTH;
  constexpr bool SOFF_F = (PP_SOFF != 0) && !BIG, SOFF_S = (PP_SOFF != 0);
  static_assert(!(BIG && MASK), "the MASK instances keep whole arrays under 2^31 bytes");
  float* const smem = s_u;   // (the reduction scratch of the prologue: 3 * 8 doubles)
  if (st->halt) return;   // (written by an earlier launch)
  // End of the PREVIOUS iteration, folded into this launch instead of a one-block kernel of its own (k_demons_finalize:
  // 5 us plus a launch gap per iteration, a fifth of an iteration on the coarse pyramid levels): every block adds the
  // previous launch's per-tile sums in the same fixed order, so all of them reach the same FiniteDifferenceImageFilter::Halt()
  // decision; block 0 publishes the statistics.  Kernel B of this iteration reads the flag after this launch has completed.
  if (nprev > 0) {
    __shared__ int s_halt;
    double fa = 0.0, fb = 0.0, fc = 0.0;
    for (int i = threadIdx.x; i < nprev; i += NTH) {
      fa += prev[3 * (size_t)i + 0];
      fb += prev[3 * (size_t)i + 1];
      fc += prev[3 * (size_t)i + 2];
    }
    pp_block_sum3_shfl<NTH>(fa, fb, fc, reinterpret_cast<double*>(smem));
    if (threadIdx.x == 0) {
      double rms = st->rms;   // (kept when nothing was counted; block 0 then leaves it untouched as well)
      if (fc > 0.0) rms = sqrt(fb / fc);
      const int h = max_rms > rms ? 1 : 0;   // Halt(): m_MaximumRMSError > m_RMSChange
      s_halt = h;
      if (blockIdx.x == 0) {
        st->ssd = fa;
        st->ssc = fb;
        st->npx = (long long)fc;
        if (fc > 0.0) {
          st->metric = fa / fc;
          st->rms = rms;
        }
        pp_stats_record(st);
        st->elapsed += 1;
        if (h) st->halt = 1;
      }
    }
    __syncthreads();
    if (s_halt) return;
    __syncthreads();   // smem is reused below
  }
  int tx0, ty0, z0;
  unsigned rank;
  if (!fused_tile(a, TX, TY, tx0, ty0, z0, rank, region)) return;

  const pp_dims d = a.d;
  const int t = threadIdx.x;
  // Work that is not tied to an output voxel -- the ESM update of the halo'd smoothing input (22 wave-rounds over 8 waves: 3 3 3
  // 3 3 3 2 2), the image tile's border ring (waves 0-2) and the x-pass items (2 2 2 2 2 2 2 1) -- is dealt by ROLE: the wave
  // index reversed in every other block of a CU's pair (blocks j and j + 32 of an XCD's run share a CU when the dispatcher
  // fills the CUs round-robin), so that the SIMD that hosts the heavy waves 0 / 4 of one block hosts the light waves 7 / 3 of
  // the other: 11 + 11 + 11 + 11 ESM rounds per SIMD and plane instead of 12 + 12 + 10 + 10.  The lane keeps its place (the
  // x pass's conflict-free order is by lane).  Same arithmetic by another thread: fields bit-identical.
  const int tr = ((PP_A_FLIP != 0) && ((blockIdx.x >> 8) & 1u)) ? (((NTH / 64 - 1) - (t >> 6)) << 6 | (t & 63)) : t;
  const int cx = t % G::LX, cy = t / G::LX;
  const unsigned sy = (unsigned)a.px, sz = (unsigned)a.px * d.ny;
  const size_t N = (size_t)sz * d.nz;
  const pp_rsrc r_f = pp_make_rsrc(F), r_mw = pp_make_rsrc(Mw);   // (PP_SOFF)
  const pp_rsrc r_d = pp_make_rsrc(D), r_us = MASK ? pp_make_rsrc_masked(Us) : pp_make_rsrc(Us);

  // Owned smoothing-input voxels.  Image values are fetched at the clamped position, so out-of-volume halo slots
  // replicate the edge update (ZeroFluxNeumann on the smoothing input).
  unsigned slots[G::KU];  // read slot of the clamped position, minus one row | write slot << 16   (in s_mf)
  unsigned uflag[G::KU];  // slot in s_u | flags << 16
  unsigned own_g[G::KU];  // in-plane BYTE offset of the clamped position
  // z window of the image pair at the owned voxels, kept as (warped, fixed) PAIRS: the z gradient is then the packed form of
  // the x and y ones (pp_esm_axis_plain2), and the window rotates by 64-bit moves
  PP_ZWIN(wprev, G::KU);
  PP_ZWIN(wcur, G::KU);
  PP_ZWIN(wnext, G::KU);
  // (border rules are carried by data -- a slot outside the volume publishes the sentinel in its warped-image half, the
  // fixed-gradient factor is 0 on a first/last index -- and both are derived from the flag bits where they are used: only
  // blocks on the volume's x/y border ever need them, and nine registers held them for every block.)
#pragma unroll
  for (int k = 0; k < G::KU; ++k) {
    const int e = tr + k * NTH;
    const int ee = e < G::NU ? e : 0;
    const int uy = ee / G::UW, ux = ee - uy * G::UW;
    const int xg = tx0 - R + ux, yg = ty0 - R + uy;
    const int xc = pp_clampi(xg, 0, d.nx - 1), yc = pp_clampi(yg, 0, d.ny - 1);
    own_g[k] = ((unsigned)yc * sy + (unsigned)xc) * 4u;
    const unsigned wslot = (unsigned)((uy + 1) * G::MWP + (ux + 1));
    // (read slot stored one tile row UP: the four neighbour reads of the ESM pass are then non-negative constant offsets from
    // one address register -- DS instructions take no negative immediate, and l - 1 / l - MWP each held a register per round)
    const unsigned rslot = (unsigned)((yc - (ty0 - R - 1) - 1) * G::MWP + (xc - (tx0 - R - 1)));
    slots[k] = rslot | (wslot << 16);
    unsigned fl = 0;
    if (e < G::NU) fl |= F_VALID;
    if (e < G::NU && xg >= tx0 && xg < tx0 + TX && xg < d.nx && yg >= ty0 && yg < ty0 + TY && yg < d.ny) fl |= F_CNT;
    if (xc == 0) fl |= F_XLO;
    if (xc == d.nx - 1) fl |= F_XHI;
    if (yc == 0) fl |= F_YLO;
    if (yc == d.ny - 1) fl |= F_YHI;
    if (xg != xc || yg != yc) fl |= F_OOV;
    uflag[k] = (unsigned)(uy * G::UWP + ux) | (fl << 16);
  }
  // Border ring of the image tile (needed only in-plane): one element per low thread.
  int brd_w = -1;
  unsigned brd_g = 0;
  float brd_oov = -FLT_MAX;
  if (tr < G::NB) {
    int my, mx;
    if (tr < G::MW) { my = 0; mx = tr; }
    else if (tr < 2 * G::MW) { my = G::MH - 1; mx = tr - G::MW; }
    else { const int q = tr - 2 * G::MW; my = 1 + q / 2; mx = (q & 1) ? G::MW - 1 : 0; }
    const int xc = pp_clampi(tx0 - R - 1 + mx, 0, d.nx - 1), yc = pp_clampi(ty0 - R - 1 + my, 0, d.ny - 1);
    if (xc != tx0 - R - 1 + mx || yc != ty0 - R - 1 + my) brd_oov = FLT_MAX;
    brd_w = my * G::MWP + mx;
    brd_g = ((unsigned)yc * sy + (unsigned)xc) * 4u;
  }
  int xsrc[NXI], xdst[NXI];
  fused2_xpass_setup<R, SH, NXI>(tr, xsrc, xdst);
  const int yb = cy * fused2_xtile<SH>::XP + 2 * cx;
  const int x = tx0 + 2 * cx, y = ty0 + cy;
  const bool out_ok = (y < d.ny) && (x < d.nx);
  const bool pair_ok = (a.px % 2) == 0;
  const unsigned o_xy = ((unsigned)y * sy + (unsigned)x) * 4u;

  const int zs = z0 - R;
  const int zo_last = (z0 + a.zchunk - 1 < d.nz - 1) ? z0 + a.zchunk - 1 : d.nz - 1;
  const int ze = zo_last + R;
  const int nsteps = ze - zs + 1;
  constexpr bool SYNC = (PP_SOFTSYNC != 0) && MASK;
  pp_softsync ysync{};
  if constexpr (SYNC) pp_softsync_init(ysync, a, a.sync_other);
  constexpr bool PRIO = (PP_PAIRPRIO != 0) && MASK;
  pp_pairprio yprio{};
  if constexpr (PRIO) pp_pairprio_init(yprio, a, a.sync_other);

  float bm = 0.0f, bf = 0.0f;          // border ring values of the plane about to be published
  PP_ZWIN(win_, G::KU);                // plane two ahead of the window centre, in flight
  float bm_n = 0.0f, bf_n = 0.0f;
  float a_ssd = 0.0f, a_ssc = 0.0f, a_n = 0.0f;   // <= ~40 terms per thread: fp32 is exact enough, folded in fp64 below

  auto publish = [&]() {   // window centre (mcur, fcur) + border ring -> packed image tile
#pragma unroll
    for (int k = 0; k < G::KU; ++k)
      if ((k + 1) * NTH <= G::NU || ((uflag[k] >> 16) & F_VALID)) {
        const float oov = ((pp_opaque(uflag[k]) >> 16) & F_OOV) ? FLT_MAX : -FLT_MAX;
        s_mf[slots[k] >> 16] = make_float2(fmaxf(wcur[k][0], oov), wcur[k][1]);
      }
    if (brd_w >= 0) s_mf[brd_w] = make_float2(fmaxf(bm, brd_oov), bf);
  };
  auto prefetch = [&](int zc) {   // own voxels of plane zc + 2, border ring of plane zc + 1
#ifdef PP_ABL_A_NOLOAD
    {   // (measurement builds: no image loads)
#pragma unroll
      for (int k = 0; k < G::KU; ++k) {
        win_[k][0] = (float)own_g[k] * 1e-3f + (float)zc;
        win_[k][1] = (float)own_g[k] * 2e-3f - (float)zc;
      }
      bm_n = 1.0f;
      bf_n = 2.0f;
      return;
    }
#endif
    const size_t p2 = (size_t)pp_clampi(zc + 2, 0, d.nz - 1) * sz, p1 = (size_t)pp_clampi(zc + 1, 0, d.nz - 1) * sz;
    if constexpr (SOFF_S) {
      const unsigned s2 = (unsigned)p2 * 4u, s1 = (unsigned)p1 * 4u;
#pragma unroll
      for (int k = 0; k < G::KU; ++k) {
        win_[k][0] = pp_blds(r_mw, own_g[k], s2);
        win_[k][1] = pp_blds(r_f, own_g[k], s2);
      }
      if (MASK || brd_w >= 0) {   // (MASK: lanes without a ring element read voxel 0 of the plane)
        bm_n = pp_blds(r_mw, brd_g, s1);
        bf_n = pp_blds(r_f, brd_g, s1);
      }
    } else {
      const pp_rsrc rm2 = pp_make_rsrc(Mw + p2), rf2 = pp_make_rsrc(F + p2);
#pragma unroll
      for (int k = 0; k < G::KU; ++k) {
        win_[k][0] = pp_bld(rm2, own_g[k]);
        win_[k][1] = pp_bld(rf2, own_g[k]);
      }
      if (MASK || brd_w >= 0) {
        bm_n = pp_bld(pp_make_rsrc(Mw + p1), brd_g);
        bf_n = pp_bld(pp_make_rsrc(F + p1), brd_g);
      }
    }
  };
  auto esm = [&](int zc, auto interior_tag) __attribute__((always_inline)) {   // update at every smoothing-input voxel of plane zc (the window centre) -> s_u, then rotate
    constexpr bool ZIN = decltype(interior_tag)::value;   // 0 < zc < nz - 1 known at compile time (steady steps)
    const bool count_plane = (zc >= z0 && zc <= zo_last);
    const bool zlo_b = ZIN ? false : (zc == 0), zhi_b = ZIN ? false : (zc == d.nz - 1);
    if constexpr ((PP_A_VOTE != 0) && MASK) {   // (the branchy instances have no registers left for it: 6 spills, +11 % measured)
    // Round 4: one wavefront vote per round.  Where no lane of the wavefront sits on a first / last index and none of its seven
    // warped-image values is the sentinel -- everywhere but next to the volume's border and to voxels the warp mapped outside
    // the moving image -- ITK's case analysis selects the central difference on every axis, and the three gradients are five
    // vector instructions each instead of ~13 (pp_esm_axis_plain: the same expressions with the selects resolved).  Every lane
    // of the wavefront reaches both votes: lanes without a voxel (the last round only) compute on slot 0 and store nothing.
    const bool z_inner = !(zlo_b || zhi_b);
#pragma unroll
    for (int k = 0; k < G::KU; ++k) {
      const unsigned fl = uflag[k] >> 16;
      const bool full_round = (k + 1) * NTH <= G::NU;
      const bool valid = full_round || (fl & F_VALID);
      if (full_round || __any(valid)) {
        const pp_v2f* const lp2 = reinterpret_cast<const pp_v2f*>(s_mf) + (slots[k] & 0xffffu);   // (the slot one row up, see the setup)
        const pp_v2f xm2 = lp2[G::MWP - 1], xp2 = lp2[G::MWP + 1], ym2 = lp2[0], yp2 = lp2[2 * G::MWP];
        const float2 xm = make_float2(xm2[0], xm2[1]), xp = make_float2(xp2[0], xp2[1]), ym = make_float2(ym2[0], ym2[1]), yp = make_float2(yp2[0], yp2[1]);
        const unsigned flb = pp_opaque(fl);   // (predicates formed here: hoisted, they would hold six scalar registers per voxel)
        const float mmax = fmaxf(fmaxf(fmaxf(xm.x, xp.x), fmaxf(ym.x, yp.x)), fmaxf(fmaxf(wprev[k][0], wnext[k][0]), wcur[k][0]));
        const bool plain = !valid || (((flb & (F_XLO | F_XHI | F_YLO | F_YHI)) == 0u) & (mmax < FLT_MAX));
        pp_esm_out o;
        if (z_inner && !__any(!plain)) {
#if PP_A_ESM_PAIRS
          const float gx = pp_esm_axis_plain2(xm2, xp2, K.ix);
          const float gy = pp_esm_axis_plain2(ym2, yp2, K.iy);
#else
          const float gx = pp_esm_axis_plain(xm.y, xp.y, xm.x, xp.x, K.ix);
          const float gy = pp_esm_axis_plain(ym.y, yp.y, ym.x, yp.x, K.iy);
#endif
#if PP_A_ESM_PAIRS && PP_A_ZPAIRS
          const float gz = pp_esm_axis_plain2(wprev[k], wnext[k], K.iz);
#else
          const float gz = pp_esm_axis_plain(wprev[k][1], wnext[k][1], wprev[k][0], wnext[k][0], K.iz);
#endif
          o = pp_esm_voxel<true>(K, wcur[k][1], wcur[k][0], gx, gy, gz);
        } else {
          const float hfx = (flb & (F_XLO | F_XHI)) ? 0.0f : 0.5f * K.ix, hfy = (flb & (F_YLO | F_YHI)) ? 0.0f : 0.5f * K.iy;
          const float gx = pp_esm_axis_data(xm.y, xp.y, wcur[k][0], xm.x, xp.x, hfx, K.ix);
          const float gy = pp_esm_axis_data(ym.y, yp.y, wcur[k][0], ym.x, yp.x, hfy, K.iy);
          const float gz = pp_esm_axis(wprev[k][1], wnext[k][1], wcur[k][0], wprev[k][0], wnext[k][0], zlo_b, zhi_b, K.iz);
          o = pp_esm_voxel(K, wcur[k][1], wcur[k][0], gx, gy, gz);
        }
        if (valid) {
          const int u = (int)(uflag[k] & 0xffffu);
          s_u[u] = o.ux;
          s_u[G::UH * G::UWP + u] = o.uy;
          s_u[2 * G::UH * G::UWP + u] = o.uz;
          if (count_plane && (fl & F_CNT)) {
            a_ssd += o.sq_speed;
            a_ssc += o.sq_update;
            a_n += (float)o.counted;
          }
        }
      }
    }
    } else {
#pragma unroll
    for (int k = 0; k < G::KU; ++k) {
      const unsigned fl = uflag[k] >> 16;
      if ((k + 1) * NTH <= G::NU || (fl & F_VALID)) {
        const float2* const lp = s_mf + (slots[k] & 0xffffu);   // (the slot one row up, see the setup)
        const float2 xm = lp[G::MWP - 1], xp = lp[G::MWP + 1], ym = lp[0], yp = lp[2 * G::MWP];
        const unsigned flb = pp_opaque(fl);   // (predicates formed here: hoisted, they would hold six scalar registers per voxel)
        const float hfx = (flb & (F_XLO | F_XHI)) ? 0.0f : 0.5f * K.ix, hfy = (flb & (F_YLO | F_YHI)) ? 0.0f : 0.5f * K.iy;
        const float gx = pp_esm_axis_data(xm.y, xp.y, wcur[k][0], xm.x, xp.x, hfx, K.ix);
        const float gy = pp_esm_axis_data(ym.y, yp.y, wcur[k][0], ym.x, yp.x, hfy, K.iy);
        const float gz = pp_esm_axis(wprev[k][1], wnext[k][1], wcur[k][0], wprev[k][0], wnext[k][0], zlo_b, zhi_b, K.iz);
        const pp_esm_out o = pp_esm_voxel(K, wcur[k][1], wcur[k][0], gx, gy, gz);
        const int u = (int)(uflag[k] & 0xffffu);
        s_u[u] = o.ux;
        s_u[G::UH * G::UWP + u] = o.uy;
        s_u[2 * G::UH * G::UWP + u] = o.uz;
        if (count_plane && (fl & F_CNT)) {
          a_ssd += o.sq_speed;
          a_ssc += o.sq_update;
          a_n += (float)o.counted;
        }
      }
    }
    }
#pragma unroll
    for (int k = 0; k < G::KU; ++k) {
#if PP_A_ZPAIRS
      wprev[k] = wcur[k];
      wcur[k] = wnext[k];
      wnext[k] = win_[k];
#else
      wprev[k][0] = wcur[k][0]; wcur[k][0] = wnext[k][0]; wnext[k][0] = win_[k][0];
      wprev[k][1] = wcur[k][1]; wcur[k][1] = wnext[k][1]; wnext[k][1] = win_[k][1];
#endif
    }
    bm = bm_n;
    bf = bf_n;
  };

  float rg[3][2][W];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < W; ++k) rg[c][j][k] = 0.0f;
  float v[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};

  float2 dsum[SUM ? 3 : 1];
  const bool dzero = (nprev == 0);   // (wave-uniform; see the adds in the plane step)
  const unsigned o_xy1 = (x + 1 < d.nx) ? o_xy + 4u : o_xy;
  auto load_dsum = [&](int zo, auto always_tag) {
    if constexpr (MASK) {
      // One 8-byte load per component (rows are even, so the pair is aligned) on every step: a plane of this chunk, lanes
      // outside the volume read voxel 0 of it.  (A global load through the array's base: hipcc 7.2 lowers
      // __builtin_amdgcn_raw_buffer_load_b64 to buffer_load_dword and broadcasts the first element -- tools/probes/unaligned.hip
      // met the same thing in round 2 and took it for an alignment rule.)
      const unsigned po = (unsigned)pp_clampi(zo, z0, zo_last) * sz;
      const unsigned o_ld = out_ok ? o_xy : 0u;
#pragma unroll
#ifdef PP_ABL_A_NOLOAD
      for (int c = 0; c < (SUM ? 3 : 1); ++c) dsum[c] = make_float2((float)(o_ld + po) * 1e-6f, (float)c);
#else
      for (int c = 0; c < (SUM ? 3 : 1); ++c) dsum[c] = pp_gld2(reinterpret_cast<const char*>(D), o_ld + ((unsigned)c * (unsigned)N + po) * 4u);
#endif
    } else if (decltype(always_tag)::value || (zo >= z0 && zo <= zo_last && out_ok)) {
      const size_t po = (size_t)zo * sz;
#pragma unroll
      for (int c = 0; c < (SUM ? 3 : 1); ++c) {   // two 4-byte buffer loads: no alignment case, no branch (x + 1 == nx re-reads x)
        if constexpr (SOFF_F) {
          const unsigned so = ((unsigned)c * (unsigned)N + (unsigned)po) * 4u;
          dsum[c].x = pp_blds(r_d, o_xy, so);
          dsum[c].y = pp_blds(r_d, o_xy1, so);
        } else {
          const pp_rsrc rd = pp_make_rsrc(D + c * N + po);
          dsum[c].x = pp_bld(rd, o_xy);
          dsum[c].y = pp_bld(rd, o_xy1);
        }
      }
    }
  };

  // prologue: z window and border ring at the first plane, published; its ESM update in s_u
  {
    const int zc0 = pp_clampi(zs, 0, d.nz - 1);
    const size_t pm = (size_t)pp_clampi(zc0 - 1, 0, d.nz - 1) * sz, pc = (size_t)zc0 * sz, pn = (size_t)pp_clampi(zc0 + 1, 0, d.nz - 1) * sz;
    const pp_rsrc rmm = pp_make_rsrc(Mw + pm), rfm = pp_make_rsrc(F + pm), rmc = pp_make_rsrc(Mw + pc), rfc = pp_make_rsrc(F + pc),
                  rmn = pp_make_rsrc(Mw + pn), rfn = pp_make_rsrc(F + pn);
#pragma unroll
    for (int k = 0; k < G::KU; ++k) {
      wprev[k][0] = pp_bld(rmm, own_g[k]); wprev[k][1] = pp_bld(rfm, own_g[k]);
      wcur[k][0] = pp_bld(rmc, own_g[k]);  wcur[k][1] = pp_bld(rfc, own_g[k]);
      wnext[k][0] = pp_bld(rmn, own_g[k]); wnext[k][1] = pp_bld(rfn, own_g[k]);
    }
    if (brd_w >= 0) {
      bm = pp_bld(rmc, brd_g);
      bf = pp_bld(rfc, brd_g);
    }
    publish();
    prefetch(zc0);
    __syncthreads();
    esm(zc0, pp_steady<false>{});
    {   // step 0's image loads (see the end of the plane step)
      const int n1 = pp_clampi(zs + 1, 0, d.nz - 1);
      if (nsteps > 1 && n1 != zc0) prefetch(n1);
    }
    if constexpr (SUM) load_dsum(zs - R, pp_steady<false>{});
    __syncthreads();
  }

  const bool trace_on = (rank == (unsigned)(a.gx * (a.gy / 2) + a.gx / 2));   // (PP_TRACE builds: an interior tile, first z-chunk)
  (void)trace_on;
  // steady steps (fused2_plane_loop3): planes zi - 1 .. zi + 2 exist and zi + 1 is not the last one (no z-border rule in the
  // update of plane zi + 1), zi - R and zi - R + 1 are output planes of this chunk, and the tile lies inside the volume with
  // even rows, so every lane loads D and stores pairs
  const bool tile_full = (tx0 + TX <= d.nx) && (ty0 + TY <= d.ny) && pair_ok;
  const int s_lo = (2 * R > 1 - zs) ? 2 * R : 1 - zs;
  const int s_hi = tile_full ? ((d.nz - 3 < ze - 2) ? d.nz - 3 : ze - 2) - zs + 1 : 0;
  auto step = [&](int n, auto phase_tag, auto steady_tag) {
    constexpr int P = decltype(phase_tag)::value;
    constexpr bool ST = decltype(steady_tag)::value;
    const int zi = zs + n;
    const int cur = ST ? zi : pp_clampi(zi, 0, d.nz - 1);
    const bool fresh_cur = ST || (n == 0) || (cur != pp_clampi(zi - 1, 0, d.nz - 1));
    const int nxt = ST ? zi + 1 : pp_clampi(zi + 1, 0, d.nz - 1);
    const bool fresh_next = ST || ((n + 1 < nsteps) && (nxt != cur));
    const int zo = zi - R;
    const bool emit = ST || ((zo >= z0) && (zo <= zo_last) && out_ok);
    PP_TRACE_MARK(trace_on, 0, n, 0);
    PP_DRIFT_MARK(0, n, nsteps);
    if constexpr (SYNC) pp_softsync_peek(ysync);
    if constexpr (PRIO) pp_pairprio_peek(yprio);
    // ---- interval 1: x pass of plane `cur` (s_u -> s_x) | publish the image tile of plane `nxt` ----
    if (fresh_next) publish();
    if (fresh_cur) fused2_xpass<R, NXI, 3 * G::XI>(s_u, s_x, a.wx, xsrc, xdst);
    PP_TRACE_MARK(trace_on, 0, n, 1);
    if (fresh_cur || fresh_next) __syncthreads();
    PP_TRACE_MARK(trace_on, 0, n, 2);
    // ---- interval 2: y pass of plane `cur` (s_x -> registers) | ESM update of plane `nxt` (s_mf -> s_u) ----
    if (fresh_cur) {
#pragma unroll
      for (int c = 0; c < 3; ++c) fused2_ypass<R, SH>(s_x, c, yb, a.wy, v[c]);
    }
    if (fresh_next) esm(nxt, steady_tag);
    float us[3][2];
    fused2_ring<R, P>(rg, v, a.wz, us);
    constexpr bool UNC = MASK || ST;   // memory instructions issued unconditionally
    if constexpr (SUM) {
      // First iteration of an Execute (nprev == 0): the field is zero by definition and its buffer has NOT been cleared (an
      // 805 MB memset at 512 x 512 x 256) -- whatever the loads above returned is replaced by the zeros it stands for.
      if (dzero) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dsum[c] = make_float2(0.0f, 0.0f);
      }
    }
    if constexpr (UNC) {
      // all three adds before the first store: the adds wait for the D loads of the previous step, and a wait placed between
      // two stores also waits for the first store's acknowledgement (vmcnt retires in issue order).  One store form (even
      // rows); the lane mask -- inside the volume, an output plane of this chunk -- travels in the offset.
      if constexpr (SUM) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          us[c][0] = dsum[c].x + us[c][0];
          us[c][1] = dsum[c].y + us[c][1];
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink each add to its store again)
      const unsigned po4 = (unsigned)(ST ? zo : pp_clampi(zo, z0, zo_last)) * sz * 4u, N4 = (unsigned)N * 4u;
#ifdef PP_ABL_A_NOSTORE
      const unsigned o_st = PP_OOB;   // (measurement builds: every store dropped by the range check)
#else
      const unsigned o_st = (ST || emit) ? o_xy : PP_OOB;
#endif
#pragma unroll
      for (int c = 0; c < 3; ++c) pp_bst2ss<NT>(r_us, o_st, (unsigned)c * N4 + po4, us[c][0], us[c][1]);
      __builtin_amdgcn_sched_barrier(0);   // (the next step's loads go behind the stores)
    } else if (emit) {
      const size_t po = (size_t)zo * sz;
      if constexpr (SUM) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          us[c][0] = dsum[c].x + us[c][0];
          us[c][1] = dsum[c].y + us[c][1];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if constexpr (SOFF_F) {
          const unsigned so = ((unsigned)c * (unsigned)N + (unsigned)po) * 4u;
          if (ST || pair_ok) {
            pp_bst2ss<NT>(r_us, o_xy, so, us[c][0], us[c][1]);
          } else if (x + 1 < d.nx) {
            pp_gst2(reinterpret_cast<char*>(Us + c * N + po), o_xy, us[c][0], us[c][1]);
          } else {
            pp_bsts(r_us, o_xy, so, us[c][0]);
          }
        } else {
          const pp_rsrc ro = pp_make_rsrc(Us + c * N + po);
          if (pair_ok) {
            pp_bst2s<NT>(ro, o_xy, us[c][0], us[c][1]);
          } else if (x + 1 < d.nx) {
            pp_gst2(reinterpret_cast<char*>(Us + c * N + po), o_xy, us[c][0], us[c][1]);
          } else {
            pp_bst(ro, o_xy, us[c][0]);
          }
        }
      }
    }
    // The next step's loads go in flight here, behind this step's stores, so that no wait of this step has them
    // pending: the image planes are consumed by the next ESM pass, D (SUM) by the next stores.
    {
      const int nxt2 = ST ? zi + 2 : pp_clampi(zi + 2, 0, d.nz - 1);
      if (UNC || ((n + 2 < nsteps) && (nxt2 != nxt))) prefetch(nxt2);   // (UNC: a step that would not load re-reads the planes it holds)
    }
    if constexpr (SUM) load_dsum(zo + 1, steady_tag);
    PP_TRACE_MARK(trace_on, 0, n, 3);
    if constexpr (SYNC) pp_softsync_step(ysync, n);
    if constexpr (PRIO) pp_pairprio_step(yprio, n);
    if (fresh_cur || fresh_next) __syncthreads();
    PP_TRACE_MARK(trace_on, 0, n, 4);
  };
#if PP_A_STEADY
  fused2_plane_loop3<R, UNROLL>(step, nsteps, s_lo, s_hi);
#else
  (void)s_lo;
  (void)s_hi;
  auto step_general = [&](int n, auto phase_tag) { step(n, phase_tag, pp_steady<false>{}); };
  fused2_plane_loop<R, UNROLL>(step_general, nsteps);
#endif
  if constexpr (SYNC) pp_softsync_finish(ysync);
  if constexpr (PRIO) pp_pairprio_finish(yprio);
  double r_ssd = (double)a_ssd, r_ssc = (double)a_ssc, r_n = (double)a_n;
  pp_block_sum3_shfl<NTH>(r_ssd, r_ssc, r_n, reinterpret_cast<double*>(s_u));
  if (t == 0) {
    partials[3 * (size_t)rank + 0] = r_ssd;
    partials[3 * (size_t)rank + 1] = r_ssc;
    partials[3 * (size_t)rank + 2] = r_n;
  }
}

// SH 0 / 1 / 2 as for kernel B.  (Three LDS objects, not one carved array: the compiler may then move the image-tile reads of
// one ESM round above the update stores of the previous one -- as slices of one array it must keep them in order, and each
// wave pays the LDS round trip once per round.)
template <int R, int SH, bool UNROLL, bool SUM, bool NT, bool MASK, bool BIG = false>
__global__ void __launch_bounds__(512, PP_GEN2_WAVES) k_fused2_force_smooth(const float* __restrict__ F, const float* __restrict__ Mw,
                                                                         const float* __restrict__ D, float* __restrict__ Us,
                                                                         fused_args a, pp_esm_consts K,
                                                                         double* __restrict__ partials, pp_dev_stats* __restrict__ st,
                                                                         const double* __restrict__ prev, int nprev, double max_rms) {
  if constexpr (SH == 2) {
    using L0 = fused2_force_lds<R, 0>;
    using L1 = fused2_force_lds<R, 1>;
    __shared__ __attribute__((aligned(16))) float2 s_mf_[(L0::SZ_IMG2 > L1::SZ_IMG2 ? L0::SZ_IMG2 : L1::SZ_IMG2) / 2];
    __shared__ __attribute__((aligned(16))) float s_u_[L0::SZ_U > L1::SZ_U ? L0::SZ_U : L1::SZ_U];
    __shared__ __attribute__((aligned(16))) float s_x_[L0::SZ_XT > L1::SZ_XT ? L0::SZ_XT : L1::SZ_XT];
    if (fused_region(a) == 0) fused2_force_body<R, 0, UNROLL, SUM, NT, MASK, BIG>(F, Mw, D, Us, a, K, partials, st, prev, nprev, max_rms, s_mf_, s_u_, s_x_, 0);
    else fused2_force_body<R, 1, UNROLL, SUM, NT, MASK, BIG>(F, Mw, D, Us, a, K, partials, st, prev, nprev, max_rms, s_mf_, s_u_, s_x_, 1);
  } else {
    using L = fused2_force_lds<R, SH>;
    __shared__ __attribute__((aligned(16))) float2 s_mf_[L::SZ_IMG2 / 2];
    __shared__ __attribute__((aligned(16))) float s_u_[L::SZ_U];
    __shared__ __attribute__((aligned(16))) float s_x_[L::SZ_XT];
    fused2_force_body<R, SH, UNROLL, SUM, NT, MASK, BIG>(F, Mw, D, Us, a, K, partials, st, prev, nprev, max_rms, s_mf_, s_u_, s_x_, 0);
  }
}
