// pp_demons_cube.h -- generation 3 of the fused demons iteration: the smallest pyramid levels (included by pp_demons.hip).
//
// The marching kernels of generations 1 and 2 walk a tile through its z-chunk plane by plane: 2R + 3 halo planes before the
// first output, so an iteration on a small level lasts 31 us however few voxels it has -- and the pipelines spend 150 of their
// 375 iterations on a level of ~50 k voxels (multiatlas/run.py:75-84: 6 mm).  Here a block owns a 16 x 8 x 6 BRICK of outputs and
// holds its whole 3-D halo in LDS: one round of loads, the ESM update on the brick + R, three separable passes through LDS per
// component, one round of stores -- nothing in a block waits for a previous plane: 20 us per iteration up to ~250 bricks.  The
// halo is recomputed (brick + R = 3.1 x the outputs at R = 2), which only latency-bound levels can afford: the launcher hands
// everything above 420 bricks to the marching kernels (pp_demons.hip, profiles/round6_cube_levels.txt).
//
// Same arithmetic as the other generations, voxel by voxel: pp_esm_axis / pp_esm_voxel on clamped positions (an out-of-volume
// halo slot holds its clamped neighbour's update: ZeroFluxNeumann on the smoothing input), each Gaussian pass a chain of fmaf
// from 0 over taps -R .. +R, S = D + G_u * update, D' = G_d * S, the warp of kernel B -- fields bit-identical
// (tests/test_kernels.py).  Same protocol as generation 2 in SUM mode: kernel A folds the previous launch's per-block sums and
// evaluates FiniteDifferenceImageFilter::Halt() on device; kernel B reads the flag.  Radii <= 2 for the update (sigma_u = 1
// voxel has 2) and <= 3 for the field (sigma_d <= 1.9 voxels: the pipelines' 1.5 mm on grids of >= 0.8 mm).

template <int R>
struct cube_geom {
  static constexpr int TX = 16, TY = 8, TZ = 6;
  static constexpr int LX = TX / 2;                      // a thread owns two x-neighbours of the brick
  static constexpr int NTH = LX * TY * TZ;               // 384
  static constexpr int UX = TX + 2 * R, UY = TY + 2 * R, UZ = TZ + 2 * R;   // smoothing input
  static constexpr int UXP = (UX + 3) / 4 * 4;
  static constexpr int IX = UX + 2, IY = UY + 2, IZ = UZ + 2;               // image values (one more voxel for the gradients)
  static constexpr int NI = IX * IY * IZ, NU = UX * UY * UZ;
  static constexpr int KI = (NI + NTH - 1) / NTH, KU = (NU + NTH - 1) / NTH;
  static constexpr int SZ_IMG = 2 * NI;
  // The Gaussian runs ONE component at a time (the other two wait in registers): a block's LDS is the image bricks -- reused
  // for the x and y pass outputs -- plus one component of smoothing input, 39 KB at R = 2, so four blocks share a CU's 160 KB
  // (with all three components resident: 58 KB, two blocks, and a level of 600 bricks already took three rounds).
  static constexpr int SZ_U = UZ * UY * UXP;             // smoothing input, one component
  static constexpr int SZ_X = UZ * UY * TX;              // x pass output
  static constexpr int SZ_Y = UZ * TY * TX;              // y pass output
  static constexpr int SZ_XY = (SZ_X + SZ_Y + 3) / 4 * 4;
  static constexpr int R1A = ((SZ_IMG > SZ_XY ? SZ_IMG : SZ_XY) + 3) / 4 * 4;   // kernel A: image bricks, then the passes
  static_assert(SZ_U * 4 >= 3 * (NTH / 64) * 8, "the fold's scratch");
};

struct cube_args {
  pp_dims d;
  pp_taps_small wx, wy, wz;
};

// x pass (s_u -> s_x) and y pass (s_x -> s_y) over the UZ planes of ONE component of a brick; the caller puts barriers around
// them.
template <int R>
__device__ __forceinline__ void cube_xpass(const float* __restrict__ s_u, float* __restrict__ s_x, const pp_taps_small& wx) {
  using G = cube_geom<R>;
  constexpr int Q = G::TX / 4, ITEMS = G::UZ * G::UY * Q;
  for (int it = threadIdx.x; it < ITEMS; it += G::NTH) {
    const int row = it / Q, q = it - row * Q;            // row = uz * UY + uy
    const float* src = s_u + row * G::UXP + 4 * q;
    float in[4 + 2 * R];
#pragma unroll
    for (int k = 0; k < (4 + 2 * R) / 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * k);
      in[4 * k + 0] = v.x; in[4 * k + 1] = v.y; in[4 * k + 2] = v.z; in[4 * k + 3] = v.w;
    }
    if ((4 + 2 * R) % 4 == 2) {
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
    *reinterpret_cast<float4*>(s_x + row * G::TX + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
template <int R>
__device__ __forceinline__ void cube_ypass(const float* __restrict__ s_x, float* __restrict__ s_y, const pp_taps_small& wy) {
  using G = cube_geom<R>;
  constexpr int ITEMS = G::UZ * G::TY * G::LX;
  for (int it = threadIdx.x; it < ITEMS; it += G::NTH) {
    const int xp = it % G::LX, r = it / G::LX, y = r % G::TY, cz = r / G::TY;   // cz = uz
    float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 2 * R + 1; ++k) {
      const float2 a = *reinterpret_cast<const float2*>(s_x + (cz * G::UY + y + k) * G::TX + 2 * xp);
      const float w = wy.h[k < R ? R - k : k - R];
      v0 = fmaf(w, a.x, v0);
      v1 = fmaf(w, a.y, v1);
    }
    *reinterpret_cast<float2*>(s_y + (cz * G::TY + y) * G::TX + 2 * xp) = make_float2(v0, v1);
  }
}
// z pass for this thread's two outputs
template <int R>
__device__ __forceinline__ void cube_zpass(const float* __restrict__ s_y, int cx, int cy, int cz, const pp_taps_small& wz, float o[2]) {
  using G = cube_geom<R>;
  float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
  for (int k = 0; k < 2 * R + 1; ++k) {
    const float2 a = *reinterpret_cast<const float2*>(s_y + ((cz + k) * G::TY + cy) * G::TX + 2 * cx);
    const float w = wz.h[k < R ? R - k : k - R];
    v0 = fmaf(w, a.x, v0);
    v1 = fmaf(w, a.y, v1);
  }
  o[0] = v0;
  o[1] = v1;
}

// ---- kernel A: ESM update, its 3-D Gaussian, + D -> S -----------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(cube_geom<R>::NTH) k_cube_force_smooth(const float* __restrict__ F, const float* __restrict__ Mw,
                                                                         const float* __restrict__ D, float* __restrict__ S, cube_args a,
                                                                         pp_esm_consts K, double* __restrict__ partials,
                                                                         pp_dev_stats* __restrict__ st, const double* __restrict__ prev,
                                                                         int nprev, double max_rms) {
  using G = cube_geom<R>;
  constexpr int NTH = G::NTH;
  static_assert((G::R1A + G::SZ_U) * 4 <= 40960, "four blocks per CU");
  __shared__ __attribute__((aligned(16))) float smem[G::R1A + G::SZ_U];
  float* const s_m = smem;
  float* const s_f = smem + G::NI;
  float* const s_x = smem;
  float* const s_y = smem + G::SZ_X;
  float* const s_u = smem + G::R1A;
  if (st->halt) return;   // (written by an earlier launch)
  const pp_dims d = a.d;
  const int t = threadIdx.x;
  const int tx0 = (int)blockIdx.x * G::TX, ty0 = (int)blockIdx.y * G::TY, tz0 = (int)blockIdx.z * G::TZ;
  const unsigned rank = ((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const size_t sy = (size_t)d.nx, sz = (size_t)d.nx * d.ny, N = sz * d.nz;

  // every global load of the block goes out first: the image bricks (clamped positions) and D at the outputs
  float fv[G::KI], mv[G::KI];
#pragma unroll
  for (int k = 0; k < G::KI; ++k) {
    const int e = t + k * NTH, ee = e < G::NI ? e : G::NI - 1;
    const int iz = ee / (G::IX * G::IY), r = ee - iz * (G::IX * G::IY), iy = r / G::IX, ix = r - iy * G::IX;
    const int xc = pp_clampi(tx0 - R - 1 + ix, 0, d.nx - 1), yc = pp_clampi(ty0 - R - 1 + iy, 0, d.ny - 1),
              zc = pp_clampi(tz0 - R - 1 + iz, 0, d.nz - 1);
    const size_t o = (size_t)zc * sz + (size_t)yc * sy + xc;
    fv[k] = F[o];
    mv[k] = Mw[o];
  }
  const int cx = t % G::LX, cy = (t / G::LX) % G::TY, cz = t / (G::LX * G::TY);
  const int x = tx0 + 2 * cx, y = ty0 + cy, z = tz0 + cz;
  const bool in0 = x < d.nx && y < d.ny && z < d.nz, in1 = in0 && x + 1 < d.nx;
  const size_t oo = (size_t)z * sz + (size_t)y * sy + x;
  float dsum[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // First iteration of an Execute (nprev == 0): the field is zero by definition (pp_demons_fused2.h does the same)
    dsum[c][0] = (in0 && nprev > 0) ? D[c * N + oo] : 0.0f;
    dsum[c][1] = (in1 && nprev > 0) ? D[c * N + oo + 1] : 0.0f;
  }

  // End of the PREVIOUS iteration (as in fused2_force_body): every block adds the previous launch's per-block sums in the
  // same fixed order and reaches the same Halt() decision; block 0 publishes the statistics.
  if (nprev > 0) {
    __shared__ int s_halt;
    double fa = 0.0, fb = 0.0, fc = 0.0;
    for (int i = t; i < nprev; i += NTH) {
      fa += prev[3 * (size_t)i + 0];
      fb += prev[3 * (size_t)i + 1];
      fc += prev[3 * (size_t)i + 2];
    }
    pp_block_sum3_shfl<NTH>(fa, fb, fc, reinterpret_cast<double*>(s_u));
    if (t == 0) {
      double rms = st->rms;
      if (fc > 0.0) rms = sqrt(fb / fc);
      const int h = max_rms > rms ? 1 : 0;   // Halt(): m_MaximumRMSError > m_RMSChange
      s_halt = h;
      if (rank == 0) {
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
  }
#pragma unroll
  for (int k = 0; k < G::KI; ++k) {
    const int e = t + k * NTH;
    if (e < G::NI) {
      s_m[e] = mv[k];
      s_f[e] = fv[k];
    }
  }
  __syncthreads();   // (also orders the fold's scratch in s_u before the updates written below)

  // ESM update at every smoothing-input voxel of the brick, taken at its clamped position
  double a_ssd = 0.0, a_ssc = 0.0, a_n = 0.0;
  float uv[3][G::KU];
  int uslot[G::KU];   // where the voxel goes in s_u; -1: none
#pragma unroll
  for (int k = 0; k < G::KU; ++k) {
    const int e = t + k * NTH;
    uv[0][k] = uv[1][k] = uv[2][k] = 0.0f;
    uslot[k] = -1;
    if (e < G::NU) {
      const int uz = e / (G::UX * G::UY), r = e - uz * (G::UX * G::UY), uy = r / G::UX, ux = r - uy * G::UX;
      const int xg = tx0 - R + ux, yg = ty0 - R + uy, zg = tz0 - R + uz;
      const int xc = pp_clampi(xg, 0, d.nx - 1), yc = pp_clampi(yg, 0, d.ny - 1), zc = pp_clampi(zg, 0, d.nz - 1);
      const int l = ((zc - (tz0 - R - 1)) * G::IY + (yc - (ty0 - R - 1))) * G::IX + (xc - (tx0 - R - 1));
      const float mc = s_m[l], fc = s_f[l];
      const float gx = pp_esm_axis(s_f[l - 1], s_f[l + 1], mc, s_m[l - 1], s_m[l + 1], xc == 0, xc == d.nx - 1, K.ix);
      const float gy = pp_esm_axis(s_f[l - G::IX], s_f[l + G::IX], mc, s_m[l - G::IX], s_m[l + G::IX], yc == 0, yc == d.ny - 1, K.iy);
      const float gz = pp_esm_axis(s_f[l - G::IX * G::IY], s_f[l + G::IX * G::IY], mc, s_m[l - G::IX * G::IY], s_m[l + G::IX * G::IY],
                                   zc == 0, zc == d.nz - 1, K.iz);
      const pp_esm_out o = pp_esm_voxel(K, fc, mc, gx, gy, gz);
      uv[0][k] = o.ux;
      uv[1][k] = o.uy;
      uv[2][k] = o.uz;
      uslot[k] = (uz * G::UY + uy) * G::UXP + ux;
      const bool own = xg >= tx0 && xg < tx0 + G::TX && xg < d.nx && yg >= ty0 && yg < ty0 + G::TY && yg < d.ny && zg >= tz0 &&
                       zg < tz0 + G::TZ && zg < d.nz;
      if (own) {
        a_ssd += (double)o.sq_speed;
        a_ssc += (double)o.sq_update;
        a_n += (double)o.counted;
      }
    }
  }
  // (the first component may enter s_u at once: the fold's scratch there was read before the barrier above)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int k = 0; k < G::KU; ++k)
      if (uslot[k] >= 0) s_u[uslot[k]] = uv[c][k];
    __syncthreads();   // s_u complete; for c == 0 also: the image bricks are dead from here
    cube_xpass<R>(s_u, s_x, a.wx);
    __syncthreads();
    cube_ypass<R>(s_x, s_y, a.wy);
    __syncthreads();
    float o[2];
    cube_zpass<R>(s_y, cx, cy, cz, a.wz, o);
    o[0] = dsum[c][0] + o[0];
    o[1] = dsum[c][1] + o[1];
    if (in1 && (d.nx & 1) == 0) {
      *reinterpret_cast<float2*>(S + c * N + oo) = make_float2(o[0], o[1]);
    } else if (in0) {
      S[c * N + oo] = o[0];
      if (in1) S[c * N + oo + 1] = o[1];
    }
    // (the next component's s_u is written at once -- the x pass that read it is two barriers back; its x pass overwrites s_x
    // behind the barrier that follows, after this component's y pass; its y pass overwrites s_y two barriers on, after this z pass)
  }
  __syncthreads();
  pp_block_sum3_shfl<NTH>(a_ssd, a_ssc, a_n, reinterpret_cast<double*>(s_u));
  if (t == 0) {
    partials[3 * (size_t)rank + 0] = a_ssd;
    partials[3 * (size_t)rank + 1] = a_ssc;
    partials[3 * (size_t)rank + 2] = a_n;
  }
}

// ---- kernel B: D' = G_d * S, then the next iteration's warped moving image ---------------------------------------------
template <int R>
__global__ void __launch_bounds__(cube_geom<R>::NTH) k_cube_smooth_warp(const float* __restrict__ S, const float* __restrict__ M,
                                                                        float* __restrict__ Dn, float* __restrict__ Mw, cube_args a,
                                                                        pp_warp_scale sc, const int* __restrict__ halt) {
  using G = cube_geom<R>;
  constexpr int NTH = G::NTH;
  __shared__ __attribute__((aligned(16))) float smem[G::SZ_XY + G::SZ_U];
  float* const s_x = smem;
  float* const s_y = smem + G::SZ_X;
  float* const s_u = smem + G::SZ_XY;
  if (halt && *halt) return;
  const pp_dims d = a.d;
  const int t = threadIdx.x;
  const int tx0 = (int)blockIdx.x * G::TX, ty0 = (int)blockIdx.y * G::TY, tz0 = (int)blockIdx.z * G::TZ;
  const size_t sy = (size_t)d.nx, sz = (size_t)d.nx * d.ny, N = sz * d.nz;
  float sv[3][G::KU];
  int uslot[G::KU];   // where the voxel goes in s_u; -1: none
#pragma unroll
  for (int k = 0; k < G::KU; ++k) {
    const int e = t + k * NTH, ee = e < G::NU ? e : G::NU - 1;
    const int uz = ee / (G::UX * G::UY), r = ee - uz * (G::UX * G::UY), uy = r / G::UX, ux = r - uy * G::UX;
    const int xc = pp_clampi(tx0 - R + ux, 0, d.nx - 1), yc = pp_clampi(ty0 - R + uy, 0, d.ny - 1), zc = pp_clampi(tz0 - R + uz, 0, d.nz - 1);
    const size_t o = (size_t)zc * sz + (size_t)yc * sy + xc;
    uslot[k] = e < G::NU ? (uz * G::UY + uy) * G::UXP + ux : -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) sv[c][k] = S[c * N + o];
  }
  const int cx = t % G::LX, cy = (t / G::LX) % G::TY, cz = t / (G::LX * G::TY);
  const int x = tx0 + 2 * cx, y = ty0 + cy, z = tz0 + cz;
  const bool in0 = x < d.nx && y < d.ny && z < d.nz, in1 = in0 && x + 1 < d.nx;
  float dn[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int k = 0; k < G::KU; ++k)
      if (uslot[k] >= 0) s_u[uslot[k]] = sv[c][k];
    __syncthreads();
    cube_xpass<R>(s_u, s_x, a.wx);
    __syncthreads();
    cube_ypass<R>(s_x, s_y, a.wy);
    __syncthreads();
    cube_zpass<R>(s_y, cx, cy, cz, a.wz, dn[c]);   // (buffer reuse across components: as in kernel A)
  }
  if (!in0) return;
  const size_t oo = (size_t)z * sz + (size_t)y * sy + x;
  float mw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int bx, by, bz;
    float fx, fy, fz;
    pp_split(x + j, dn[0][j] * sc.ix, bx, fx);
    pp_split(y, dn[1][j] * sc.iy, by, fy);
    pp_split(z, dn[2][j] * sc.iz, bz, fz);
    const bool inside = (x + j < d.nx) && pp_inside1(bx, fx, d.nx) && pp_inside1(by, fy, d.ny) && pp_inside1(bz, fz, d.nz);
    mw[j] = inside ? pp_trilinear(M, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz) : FLT_MAX;
  }
  if (in1 && (d.nx & 1) == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<float2*>(Dn + c * N + oo) = make_float2(dn[c][0], dn[c][1]);
    *reinterpret_cast<float2*>(Mw + oo) = make_float2(mw[0], mw[1]);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Dn[c * N + oo] = dn[c][0];
      if (in1) Dn[c * N + oo + 1] = dn[c][1];
    }
    Mw[oo] = mw[0];
    if (in1) Mw[oo + 1] = mw[1];
  }
}
