// platipy_amd/csrc/pp_demons_fused3.h -- third generation of fused kernel A (included by pp_demons.hip after
// pp_demons_fused2.h, whose strip geometry, DPP lane shifts, register x pass, z ring and buffer helpers it reuses).
//
// S = D + G_u * ESM-update(F, M o D), as k_fused2_force_smooth<SUM> computes it -- the same operations on the same operands
// in the same order for every voxel, so the stored volume is bit-identical -- with the work laid out so that a plane costs
// ONE barrier instead of two and a fraction of the LDS round trips.  What the round-3 counters showed about generation 2
// (profiles/round3_*): its waves are parked at s_waitcnt / s_barrier for half of their cycles, the vector ALU is busy for a
// quarter, and removing vector instructions alone changes nothing; the plane time is the length of one wave's dependent
// chain  publish -> barrier -> LDS reads -> update -> LDS write -> barrier -> x pass -> LDS write -> barrier -> y pass.
//
// Here a lane owns a STRIP of four consecutive x voxels of the smoothing-input tile (rows of the tile laid out whole rows
// per wavefront, as kernel B lays out its strips):
//   * the two images arrive as 16-byte strips (two loads per lane and plane instead of five or six 4-byte ones), sit in a
//     three-plane register window for the z differences, and are published to a double-buffered LDS tile only for the
//     y neighbours (two 16-byte LDS stores, four 16-byte LDS loads per lane and plane);
//   * x neighbours of the images and of the update come from the adjacent lanes (DPP whole-wave shifts): the update never
//     visits LDS before its x pass, which runs in registers and writes the double-buffered y-pass tile directly;
//   * the y pass, the z ring, the + D and the store keep generation 2's two-voxels-per-thread layout (all 512 threads).
// One step of the plane loop =  publish image plane n+1 | update + x pass of plane n | y pass, z ring, store around plane
// n-1, on three different LDS buffers, then one barrier.
//
// Volume borders.  ITK's rules are carried by data exactly as in generation 2 (a voxel outside the volume shows the sentinel
// in the warped image, the fixed-image difference is scaled by 0 on a first/last index).  Generation 2 evaluated the update
// of an out-of-volume halo position at the clamped voxel (a duplicate computation); here the update is computed at
// in-volume positions only and copied outwards -- along x between lanes after the update, along y by clamping the row the
// y pass reads -- which yields the same values.  Blocks whose tile (with halo) lies inside the volume skip all of it.
#pragma once

#ifndef PP_A3_WAVES
#define PP_A3_WAVES 4
#endif
#ifndef PP_A3_PLAIN
#define PP_A3_PLAIN 1
#endif
#ifndef PP_A3_SPLIT
#define PP_A3_SPLIT 1
#endif

template <int R, int SH>
struct a3_geom {
  using G = strip_geom<R, SH, 0>;
  static constexpr int TX = G::TX, TY = G::TY, NTH = G::NTH, LX = G::LX;
  static constexpr int UW = G::UW, UH = G::UH, SPR = G::SPR;
  static constexpr int RPW = 64 / SPR;                 // smoothing-input rows per wavefront
  static constexpr int NEW = (UH + RPW - 1) / RPW;     // wavefronts that hold rows
  static constexpr int IDLE = 64 - RPW * SPR;          // their spare lanes fetch the two extra image rows (gradient halo in y)
  static constexpr int IH = UH + 2;                    // image tile rows
  static constexpr int SZ_IMG = IH * UW;               // floats per image array and buffer
  static constexpr int SZ_X = G::SZ_X;                 // y-pass tile, floats per buffer
  static constexpr int SMEM = 4 * SZ_IMG + 2 * SZ_X;   // two image buffers x (warped, fixed) + two y-pass buffers
  static_assert(G::RP == 4 && R + 1 <= G::RP, "one halo strip either side covers smoothing radius + gradient");
  static_assert(NEW <= NTH / 64, "the tile's rows fit the block's wavefronts");
  static_assert(NEW * IDLE >= 2 * SPR, "the two extra image rows fit the spare lanes");
  static_assert(SMEM * 4 >= 3 * 8 * 8 * 2, "the y-pass tile doubles as the reduction buffer");
};

// Element `e` (block-uniform) of a strip.
__device__ __forceinline__ float pp_strip_elem(const float4& v, int e) {
  const float lo = (e & 1) ? v.y : v.x, hi = (e & 1) ? v.w : v.z;
  return (e & 2) ? hi : lo;
}

template <int R, int SH, bool UNROLL, bool NT>
__global__ void __launch_bounds__(512, PP_A3_WAVES) k_fused3_force_smooth(const float* __restrict__ F, const float* __restrict__ Mw,
                                                                        const float* __restrict__ D, float* __restrict__ Us, fused_args a,
                                                                        pp_esm_consts K, double* __restrict__ partials,
                                                                        pp_dev_stats* __restrict__ st, const double* __restrict__ prev,
                                                                        int nprev, double max_rms) {
  using A = a3_geom<R, SH>;
  using G = typename A::G;
  constexpr int NTH = A::NTH, TX = A::TX, TY = A::TY, UW = A::UW, UH = A::UH, SPR = A::SPR, W = 2 * R + 1;
  __shared__ __attribute__((aligned(16))) float smem[A::SMEM];
  float* const s_img = smem;                     // [buffer][warped | fixed][IH][UW]
  float* const s_xs = smem + 4 * A::SZ_IMG;      // [buffer][3][UH][TX]
  if (st->halt) return;   // (written by an earlier launch)
  // End of the previous iteration (see k_fused2_force_smooth): every block folds the previous launch's per-tile sums in the
  // same fixed order and reaches the same Halt() decision; block 0 publishes the statistics.
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
      double rms = st->rms;
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
  if (!fused_tile(a, TX, TY, tx0, ty0, z0, rank)) return;

  const pp_dims d = a.d;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const unsigned sy = d.nx, sz = (unsigned)d.nx * d.ny;
  const size_t N = (size_t)sz * d.nz;
  // the tile with its halos (RP columns, R + 1 rows) lies inside the volume: no clamp, no sentinel, no first/last index
  const bool inner = (tx0 - G::RP >= 0) && (tx0 + TX + G::RP <= d.nx) && (ty0 - R - 1 >= 0) && (ty0 + TY + R + 1 <= d.ny);

  // ---- strip role of this lane: image-tile row ry (-1 .. UH; rows 0 .. UH-1 hold smoothing-input voxels) and strip sx ----
  int ry, sx;
  bool has_strip, is_esm;
  {
    const int riw = lane / SPR;
    if (riw < A::RPW) {
      sx = lane - riw * SPR;
      ry = wv * A::RPW + riw;
      has_strip = is_esm = (wv < A::NEW) && (ry < UH);
    } else {   // spare lanes of the row-holding wavefronts: the image rows above and below the smoothing-input tile
      const int e = wv * A::IDLE + (lane - A::RPW * SPR);
      has_strip = (wv < A::NEW) && (e < 2 * SPR);
      is_esm = false;
      ry = (e / SPR) ? UH : -1;
      sx = e % SPR;
    }
    if (!has_strip) {
      ry = 0;
      sx = 0;
    }
  }
  const bool esm_wave = wv < A::NEW;                          // this wavefront holds strips
  const int xs0 = tx0 - G::RP + 4 * sx, yg = ty0 - R + ry;   // first voxel of the strip, its row
  const int yc = pp_clampi(yg, 0, d.ny - 1), xl = pp_clampi(xs0, 0, d.nx - 4);
  const unsigned goff = ((unsigned)yc * sy + (unsigned)xl) * 4u;
  // per-element flags: bits 0-3 outside the volume, 4-7 first/last x index, 8-11 counted (an in-volume output voxel of this
  // tile), 12 first/last y index; element map of a strip that leaves the volume in x (0xE4 = identity) in bits 16-23
  unsigned flags = 0;
  {
    unsigned jm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = xs0 + i;
      jm |= (unsigned)(pp_clampi(x, 0, d.nx - 1) - xl) << (2 * i);
      if (x < 0 || x > d.nx - 1 || yg != yc) flags |= 1u << i;
      if (x == 0 || x == d.nx - 1) flags |= 16u << i;
      if (is_esm && x >= tx0 && x < tx0 + TX && x < d.nx && yg >= ty0 && yg < ty0 + TY && yg < d.ny) flags |= 256u << i;
    }
    if (yc == 0 || yc == d.ny - 1) flags |= 4096u;
    flags |= jm << 16;
  }
  const int islot = (ry + 1) * UW + 4 * sx;                                            // own strip in an image buffer
  const int iup = (ry < 0 ? 0 : ry) * UW + 4 * sx;                                      // rows above / below (kept in range for the spare lanes)
  const int idn = (ry + 2 > A::IH - 1 ? A::IH - 1 : ry + 2) * UW + 4 * sx;
  const bool xs_out = is_esm && sx >= 1 && sx <= SPR - 2;                               // the strip lies in the tile's columns
  const int xs_off = ry * TX + 4 * (sx - 1);
  // x fix-up of the update (border tiles): the left halo strip of the first tile takes u(x = 0), positions beyond nx - 1
  // take u(nx - 1), which sits in strip sL at element eL
  const int sL = (d.nx - 1 - (tx0 - G::RP)) >> 2, eL = (d.nx - 1 - (tx0 - G::RP)) & 3;
  const bool fix_left = (xs0 + 3 < 0);
  const int fix_right = (sx == sL) ? 1 : ((sx == sL + 1) ? 2 : 0);   // 1: own elements beyond eL, 2: the whole strip
  // y pass rows (border tiles): tile rows whose voxels exist
  const int rlo = (R - ty0 > 0) ? R - ty0 : 0, rhi = (d.ny - 1 - ty0 + R < UH - 1) ? d.ny - 1 - ty0 + R : UH - 1;

  // ---- output role (all threads): two voxels, generation 2's layout ----
  const int cx = t % A::LX, cy = t / A::LX;
  const int yb = cy * TX + 2 * cx;
  const int x = tx0 + 2 * cx, y = ty0 + cy;
  const bool out_ok = (y < d.ny) && (x < d.nx);
  const bool pair_ok = (d.nx % 2) == 0;
  const unsigned o_xy = ((unsigned)y * sy + (unsigned)x) * 4u;
  const unsigned o_xy1 = (x + 1 < d.nx) ? o_xy + 4u : o_xy;
  const pp_rsrc r_d = pp_make_rsrc(D), r_us = pp_make_rsrc(Us);
  const char* const pF = reinterpret_cast<const char*>(F);
  const char* const pM = reinterpret_cast<const char*>(Mw);

  const int zs = z0 - R;
  const int zo_last = (z0 + a.zchunk - 1 < d.nz - 1) ? z0 + a.zchunk - 1 : d.nz - 1;
  const int ze = zo_last + R;
  const int nsteps = ze - zs + 1;

  // register window of the two images at this lane's strip: planes c - 1, c, c + 1 around the update's plane c, plus the
  // plane in flight
  float4 mP, mC, mN, mL, fP, fC, fN, fL;
  auto fix_strip = [&](float4& m, float4& f) {   // border tiles: clamped voxels in x, the sentinel outside the volume
    const unsigned j = pp_opaque(flags);
    const unsigned jm = j >> 16;
    if ((jm & 0xffu) != 0xE4u) {
      const float4 m0 = m, f0 = f;
      m = make_float4(pp_pick4(m0.x, m0.y, m0.z, m0.w, jm & 3u), pp_pick4(m0.x, m0.y, m0.z, m0.w, (jm >> 2) & 3u),
                      pp_pick4(m0.x, m0.y, m0.z, m0.w, (jm >> 4) & 3u), pp_pick4(m0.x, m0.y, m0.z, m0.w, (jm >> 6) & 3u));
      f = make_float4(pp_pick4(f0.x, f0.y, f0.z, f0.w, jm & 3u), pp_pick4(f0.x, f0.y, f0.z, f0.w, (jm >> 2) & 3u),
                      pp_pick4(f0.x, f0.y, f0.z, f0.w, (jm >> 4) & 3u), pp_pick4(f0.x, f0.y, f0.z, f0.w, (jm >> 6) & 3u));
    }
    if (j & 1u) m.x = FLT_MAX;
    if (j & 2u) m.y = FLT_MAX;
    if (j & 4u) m.z = FLT_MAX;
    if (j & 8u) m.w = FLT_MAX;
  };
  auto load_plane = [&](int zc, float4& m, float4& f) {
    const size_t po = (size_t)zc * sz * 4u;
    m = pp_gld4(pM + po, goff);
    f = pp_gld4(pF + po, goff);
  };
  auto publish = [&](float* buf, const float4& m, const float4& f) {
    if (has_strip) {
      *reinterpret_cast<float4*>(buf + islot) = m;
      *reinterpret_cast<float4*>(buf + A::SZ_IMG + islot) = f;
    }
  };

  float a_ssd = 0.0f, a_ssc = 0.0f, a_n = 0.0f;
  // ESM update of plane zc (window centre) from the image buffer `img` (which holds plane zc) -> x pass -> y-pass tile `xs`
  auto esm_xpass = [&](int zc, const float* img, float* xs) __attribute__((always_inline)) {
    const bool count_plane = (zc >= z0 && zc <= zo_last);
    const bool zlo_b = (zc == 0), zhi_b = (zc == d.nz - 1);
    const float4 um = *reinterpret_cast<const float4*>(img + iup), uf = *reinterpret_cast<const float4*>(img + A::SZ_IMG + iup);
    const float4 dm = *reinterpret_cast<const float4*>(img + idn), df = *reinterpret_cast<const float4*>(img + A::SZ_IMG + idn);
    const float lm = pp_lane_prev(mC.w), lf = pp_lane_prev(fC.w), rm = pp_lane_next(mC.x), rf = pp_lane_next(fC.x);
    const float mcv[6] = {lm, mC.x, mC.y, mC.z, mC.w, rm}, fcv[6] = {lf, fC.x, fC.y, fC.z, fC.w, rf};
    const float umv[4] = {um.x, um.y, um.z, um.w}, ufv[4] = {uf.x, uf.y, uf.z, uf.w};
    const float dmv[4] = {dm.x, dm.y, dm.z, dm.w}, dfv[4] = {df.x, df.y, df.z, df.w};
    const float mpv[4] = {mP.x, mP.y, mP.z, mP.w}, fpv[4] = {fP.x, fP.y, fP.z, fP.w};
    const float mnv[4] = {mN.x, mN.y, mN.z, mN.w}, fnv[4] = {fN.x, fN.y, fN.z, fN.w};
    float u[3][4];
    // No case analysis can fire in this wavefront's strips -- no first/last index (an inner tile, not the first or last
    // plane) and none of the warped-image values any of its lanes reads is the sentinel: the plain arithmetic (same
    // operations, same operands, same order; pp_demons_fused2.h) replaces the selects, 40 instead of 70 instructions a voxel.
    bool plain = false;
    if constexpr (PP_A3_PLAIN != 0) {
      if (inner && !zlo_b && !zhi_b) {
        float mx = fmaxf(fmaxf(lm, rm), fmaxf(fmaxf(mC.x, mC.y), fmaxf(mC.z, mC.w)));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(um.x, um.y), fmaxf(um.z, um.w)), fmaxf(fmaxf(dm.x, dm.y), fmaxf(dm.z, dm.w))));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(mP.x, mP.y), fmaxf(mP.z, mP.w)), fmaxf(fmaxf(mN.x, mN.y), fmaxf(mN.z, mN.w))));
        plain = !__any(mx == FLT_MAX);
      }
    }
    if (plain) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float gx = pp_esm_axis_plain(fcv[i], fcv[i + 2], mcv[i], mcv[i + 2], K.ix);
        const float gy = pp_esm_axis_plain(ufv[i], dfv[i], umv[i], dmv[i], K.iy);
        const float gz = pp_esm_axis_plain(fpv[i], fnv[i], mpv[i], mnv[i], K.iz);
        const pp_esm_out o = pp_esm_voxel_plain(K, fcv[i + 1], mcv[i + 1], gx, gy, gz);
        u[0][i] = o.ux;
        u[1][i] = o.uy;
        u[2][i] = o.uz;
        if (count_plane && (flags & (256u << i))) {
          a_ssd += o.sq_speed;
          a_ssc += o.sq_update;
          a_n += 1.0f;
        }
      }
    } else {
      const unsigned fl = inner ? 0u : pp_opaque(flags);
      const float hfy = (fl & 4096u) ? 0.0f : 0.5f * K.iy;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float hfx = (fl & (16u << i)) ? 0.0f : 0.5f * K.ix;
        const float gx = pp_esm_axis_data(fcv[i], fcv[i + 2], mcv[i + 1], mcv[i], mcv[i + 2], hfx, K.ix);
        const float gy = pp_esm_axis_data(ufv[i], dfv[i], mcv[i + 1], umv[i], dmv[i], hfy, K.iy);
        const float gz = pp_esm_axis(fpv[i], fnv[i], mcv[i + 1], mpv[i], mnv[i], zlo_b, zhi_b, K.iz);
        const pp_esm_out o = pp_esm_voxel(K, fcv[i + 1], mcv[i + 1], gx, gy, gz);
        u[0][i] = o.ux;
        u[1][i] = o.uy;
        u[2][i] = o.uz;
        if (count_plane && (flags & (256u << i))) {
          a_ssd += o.sq_speed;
          a_ssc += o.sq_update;
          a_n += (float)o.counted;
        }
      }
    }
    float4 u4[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) u4[c] = make_float4(u[c][0], u[c][1], u[c][2], u[c][3]);
    if (!inner) {   // ZeroFluxNeumann on the smoothing input along x: positions outside the volume repeat the edge voxel's update
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float n0 = pp_lane_next(u4[c].x);
        const float own_e = pp_strip_elem(u4[c], eL);
        const float pe = pp_lane_prev(own_e);
        if (fix_left) u4[c] = make_float4(n0, n0, n0, n0);
        if (fix_right == 1) {
          if (eL < 1) u4[c].y = own_e;
          if (eL < 2) u4[c].z = own_e;
          if (eL < 3) u4[c].w = own_e;
        } else if (fix_right == 2) {
          u4[c] = make_float4(pe, pe, pe, pe);
        }
      }
    }
    fused2_xpass_shfl<R, G>(u4, 0xE4u, xs_out, xs_off, xs, a.wx);
  };
  // y pass of this thread's two voxels from the y-pass tile `xs` (rows clamped to the volume in border tiles)
  auto ypass = [&](const float* xs, float (&v)[3][2]) __attribute__((always_inline)) {
    if (inner) {
#pragma unroll
      for (int c = 0; c < 3; ++c) fused2_ypass_strips<R, G>(xs, c, yb, a.wy, v[c]);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c][0] = 0.0f;
        v[c][1] = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const int r = pp_clampi(cy + k, rlo, rhi);
          const float2 p = *reinterpret_cast<const float2*>(xs + (c * UH + r) * TX + 2 * cx);
          const float w = a.wy.h[k < R ? R - k : k - R];
          v[c][0] = fmaf(w, p.x, v[c][0]);
          v[c][1] = fmaf(w, p.y, v[c][1]);
        }
      }
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
  float2 dsum[3];
  auto load_dsum = [&](int zo) {
    if (zo >= z0 && zo <= zo_last && out_ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned so = ((unsigned)c * (unsigned)N + (unsigned)zo * sz) * 4u;
        dsum[c].x = pp_blds(r_d, o_xy, so);
        dsum[c].y = pp_blds(r_d, o_xy1, so);
      }
    }
  };

  // ---- prologue: window around the first plane; its update, x-passed, in y-pass buffer 0; the next image plane published ----
  int ibuf = 0, xbuf = 0;   // image buffer that holds the plane the next update is computed on / y-pass buffer of the current plane
  {
    const int zc0 = pp_clampi(zs, 0, d.nz - 1);
    const int zm = pp_clampi(zc0 - 1, 0, d.nz - 1), zp = pp_clampi(zc0 + 1, 0, d.nz - 1), zq = pp_clampi(zc0 + 2, 0, d.nz - 1);
    load_plane(zm, mP, fP);
    load_plane(zc0, mC, fC);
    load_plane(zp, mN, fN);
    load_plane(zq, mL, fL);
    if (!inner) {
      fix_strip(mP, fP);
      fix_strip(mC, fC);
      fix_strip(mN, fN);
    }
    publish(s_img, mC, fC);
    publish(s_img + 2 * A::SZ_IMG, mN, fN);
    __syncthreads();
    if (esm_wave) {   // (wavefronts without strips skip the update: the branch is wavefront-uniform)
      esm_xpass(zc0, s_img, s_xs);
      // window centre -> the plane after zc0 (the first plane the loop updates)
      mP = mC; fP = fC; mC = mN; fC = fN; mN = mL; fN = fL;
      if (!inner) fix_strip(mN, fN);
      const int zr = pp_clampi(zc0 + 3, 0, d.nz - 1);
      if (zr != zq) load_plane(zr, mL, fL);
    }
    load_dsum(zs - R);
    ibuf = 1;
    __syncthreads();
  }

  const bool trace_on = (rank == (unsigned)(a.gx * (a.gy / 2) + a.gx / 2));
  (void)trace_on;
  auto step = [&](int n, auto phase_tag) {
    constexpr int P = decltype(phase_tag)::value;
    const int zi = zs + n;
    const int cur = pp_clampi(zi, 0, d.nz - 1);
    const bool fresh_cur = (n == 0) || (cur != pp_clampi(zi - 1, 0, d.nz - 1));
    const int nxt = pp_clampi(zi + 1, 0, d.nz - 1);
    const bool fresh_next = (n + 1 < nsteps) && (nxt != cur);
    const int zo = zi - R;
    const bool emit = (zo >= z0) && (zo <= zo_last) && out_ok;
    PP_TRACE_MARK(trace_on, 0, n, 0);
    float* const img_cur = s_img + ibuf * 2 * A::SZ_IMG;
    float* const img_oth = s_img + (ibuf ^ 1) * 2 * A::SZ_IMG;
    // image plane nxt + 1 (the window's upper plane) -> the other image buffer, for the next step's update
    if (fresh_next && nxt < d.nz - 1) publish(img_oth, mN, fN);
    // y pass of plane cur
    if (fresh_cur) ypass(s_xs + xbuf * A::SZ_X, v);
    float us[3][2];
    fused2_ring<R, P>(rg, v, a.wz, us);
    if (emit) {
      const unsigned po4 = (unsigned)zo * sz * 4u, N4 = (unsigned)N * 4u;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        us[c][0] = dsum[c].x + us[c][0];
        us[c][1] = dsum[c].y + us[c][1];
        if (pair_ok) {
          pp_bst2ss<NT>(r_us, o_xy, c * N4 + po4, us[c][0], us[c][1]);
        } else if (x + 1 < d.nx) {
          pp_gst2(reinterpret_cast<char*>(Us + c * N + (size_t)zo * sz), o_xy, us[c][0], us[c][1]);
        } else {
          pp_bsts(r_us, o_xy, c * N4 + po4, us[c][0]);
        }
      }
    }
    load_dsum(zo + 1);
#if PP_A3_SPLIT
    // Keep the two halves of the step apart in the schedule: interleaved, their live registers (z ring + y-pass sums + D on
    // one side, image window + LDS rows + update on the other) exceed the 128 that four waves per SIMD leave.
    __builtin_amdgcn_sched_barrier(0);
#endif
    // update + x pass of plane nxt -> the other y-pass buffer; then the window moves on
    if (fresh_next && esm_wave) {
      esm_xpass(nxt, img_cur, s_xs + (xbuf ^ 1) * A::SZ_X);
      mP = mC; fP = fC; mC = mN; fC = fN; mN = mL; fN = fL;
      if (!inner) fix_strip(mN, fN);
      const int n3 = pp_clampi(nxt + 3, 0, d.nz - 1);
      if (n3 != pp_clampi(nxt + 2, 0, d.nz - 1) && n + 2 < nsteps) load_plane(n3, mL, fL);
    }
    PP_TRACE_MARK(trace_on, 0, n, 1);
    if (fresh_next) {
      __syncthreads();
      ibuf ^= 1;
      xbuf ^= 1;
    }
    PP_TRACE_MARK(trace_on, 0, n, 2);
  };
  fused2_plane_loop<R, UNROLL>(step, nsteps);
  double r_ssd = (double)a_ssd, r_ssc = (double)a_ssc, r_n = (double)a_n;
  __syncthreads();
  pp_block_sum3_shfl<NTH>(r_ssd, r_ssc, r_n, reinterpret_cast<double*>(s_xs));
  if (t == 0) {
    partials[3 * (size_t)rank + 0] = r_ssd;
    partials[3 * (size_t)rank + 1] = r_ssc;
    partials[3 * (size_t)rank + 2] = r_n;
  }
}
