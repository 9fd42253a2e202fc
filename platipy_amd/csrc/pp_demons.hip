// platipy_amd/csrc/pp_demons.hip -- the demons inner loop.
//
// Replaces sitk.FastSymmetricForcesDemonsRegistrationFilter.Execute (reference:
// platipy/imaging/registration/deformable.py:149, configured at :244-257).  Per iteration ITK
// runs five whole-volume stages (warp, ESM update, smooth update, add, smooth field).  All
// are memory-bound stencils/gathers -- no MFMA.  Two schedules are built here:
//
//  STAGED  one launch per stage and per smoothing axis, the schedule the algorithmic byte
//          model (196 B/voxel/iteration) describes.  Any kernel radius.
//  FUSED   two launches per iteration.  Both march a 64x16 (x,y) tile through a z-chunk,
//          keeping the current plane (+halo) in LDS and the z-window in registers, so every
//          separable 3-D Gaussian costs one read and one write of the field:
//            A  force+smooth:  F, M.D  -> G_u * U            (+ metric / RMS partial sums)
//            B  add+smooth+warp:  D, U, M -> D' = G_d * (D+U),  M.D'
//          Real traffic ~64 B/voxel/iteration plus halo re-reads (served by L2/MALL).
//          Kernel radii <= 5 (sigma_d = 1.5 mm gives 1..5 for voxel spacings >= 0.5 mm).
//
// The early-halt rule (MaximumRMSError) is evaluated on device: the per-iteration finalize
// kernel raises a flag that turns every later launch into a no-op, so a whole Execute is
// enqueued without a host round trip.
#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr int NT = 256;

struct pp_dev_stats {
  double ssd, ssc;
  long long npx;
  double metric, rms;
  int elapsed;
  int halt;
  double* hist;   // {metric, RMS change} of iteration k at hist[2k], hist[2k + 1] (what an sitkIterationEvent observer reads)
  int hist_cap;
};

// One iteration's statistics into the history ring (called by the thread that publishes them, before elapsed is bumped).
__device__ __forceinline__ void pp_stats_record(pp_dev_stats* st) {
  if (!st->hist) return;
  if (st->elapsed < st->hist_cap) {   // hist[0] = entries recorded, entry k at hist[2 + 2k], hist[3 + 2k]
    st->hist[2 + 2 * st->elapsed] = st->metric;
    st->hist[3 + 2 * st->elapsed] = st->rms;
    st->hist[0] = (double)(st->elapsed + 1);
  }
  st->hist[1] = (double)(st->elapsed + 1);   // iterations that RAN (beyond the ring's capacity only this is kept)
}

__global__ void k_stats_init(pp_dev_stats* st, double* hist, int hist_cap) {
  pp_dev_stats z;
  z.ssd = 0.0; z.ssc = 0.0; z.npx = 0; z.metric = 0.0; z.rms = 0.0; z.elapsed = 0; z.halt = 0;
  z.hist = hist;
  z.hist_cap = hist_cap;
  if (hist) {
    hist[0] = 0.0;
    hist[1] = 0.0;
  }
  *st = z;
}

struct pp_esm_consts {
  float ix, iy, iz;     // 1 / spacing; the central-difference factor 0.5 / spacing is exactly half of it
  float inv_norm;       // 1 / normalizer, or 0 when MaximumUpdateStepLength <= 0 (then denom = |J|^2)
  float denom_thr;
  float intensity_thr;  // smallest float >= the fp64 threshold: |s| < thr decides exactly as in fp64
};

// One axis of the symmetric ESM gradient: itk::CentralDifferenceImageFunction on the fixed
// image (zero on the first/last index) plus the sentinel-aware difference of the warped moving
// image that ESMDemonsRegistrationFunction::ComputeUpdate builds "more or less by hand".
// lo / hi: the voxel is the first / last one along this axis (both set: the axis has one voxel).
__device__ __forceinline__ float pp_esm_axis(float fm, float fp, float mc, float mm, float mp, bool lo, bool hi, float inv_sp) {
  const float h = 0.5f * inv_sp;
  // Branch-free form of ITK's case analysis: a neighbour is usable when it exists and is not the sentinel;
  // both usable -> central difference, one usable -> one-sided, none -> 0.  Unused candidates may hold
  // inf (sentinel arithmetic) but are only ever selected away, never blended.
  const float SENT = FLT_MAX;
  const bool up = !hi && (mp != SENT);
  const bool um = !lo && (mm != SENT);
  // (mp - mm) h, (mp - mc) / sp, (mc - mm) / sp or 0, as one difference: an unusable side falls back to the centre value
  const float hi_v = up ? mp : mc, lo_v = um ? mm : mc;
  const float su = up ? h : inv_sp;                 // (up && um) ? h : inv_sp as two selects on fresh compare masks (see pp_esm_voxel)
  const float wg = (hi_v - lo_v) * (um ? su : inv_sp);
  const float fg = (lo || hi) ? 0.0f : (fp - fm) * h;
  return fg + wg;
}

struct pp_esm_out {
  float ux, uy, uz;
  float sq_speed, sq_update;
  int counted;
};

// MAPPED: the caller knows that mc is not the sentinel (the voted path of the generation-2 kernel A).
template <bool MAPPED = false>
__device__ __forceinline__ pp_esm_out pp_esm_voxel(const pp_esm_consts& K, float fc, float mc, float gx, float gy, float gz) {
  // Straight-line: every guard of ComputeUpdate becomes a select (a zero denominator only feeds a lane whose
  // result is discarded).  mc == sentinel: no update and the voxel is not counted.
  const bool mapped = MAPPED || (mc != FLT_MAX);
  const float speed = fc - mc;
  const float g2 = gx * gx + gy * gy + gz * gz;
  const float denom = g2 + speed * speed * K.inv_norm;
  // live = mapped && !(|speed| < intensity_thr) && !(denom < denom_thr), formed so that the mask the four selects below read
  // comes from ONE vector compare: a voxel that fails one of the first two tests carries -inf in place of its denominator
  // (-inf < denom_thr holds for any threshold, so the voxel is not live).  Written as three conditions joined by &&, the
  // compiler combined the compare masks on the scalar unit into VCC, and a v_cndmask reading a VCC that the SCALAR unit wrote
  // last costs ~22 cycles against ~3.5 behind a vector compare (tools/probes/valu_rate.hip, kinds 15 / 18 / 22): ~35 such
  // selects per wavefront and plane in kernel A (profiles/round4_vcc_selects.md).
  float gate = (fabsf(speed) < K.intensity_thr) ? -__builtin_inff() : denom;
  if constexpr (!MAPPED) gate = (mc != FLT_MAX) ? gate : -__builtin_inff();
  const bool live = !(gate < K.denom_thr);
  const float factor = live ? 2.0f * speed * __builtin_amdgcn_rcpf(denom) : 0.0f;   // v_rcp_f32: 1 ulp
  pp_esm_out o;
  o.ux = live ? factor * gx : 0.0f;
  o.uy = live ? factor * gy : 0.0f;
  o.uz = live ? factor * gz : 0.0f;
  o.sq_speed = mapped ? speed * speed : 0.0f;
  o.sq_update = o.ux * o.ux + o.uy * o.uy + o.uz * o.uz;
  o.counted = mapped ? 1 : 0;
  return o;
}

// ---------------------------------------------------------------------------------------
// STAGED: ESM update, one voxel per thread (neighbour loads are L1/L2 hits).

__global__ void __launch_bounds__(NT) k_demons_force(const float* __restrict__ F, const float* __restrict__ Mw,
                                                     float* __restrict__ U, pp_dims d, pp_esm_consts K,
                                                     double* __restrict__ partials, const int* __restrict__ halt) {
  __shared__ double red[3 * NT];
  if (halt && *halt) return;
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  const size_t sy = d.nx, sz = (size_t)d.nx * d.ny;
  double a_ssd = 0.0, a_ssc = 0.0, a_n = 0.0;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < N; i += (size_t)gridDim.x * NT) {
    const int x = (int)(i % d.nx);
    const int y = (int)((i / d.nx) % d.ny);
    const int z = (int)(i / sz);
    const size_t xm = x > 0 ? i - 1 : i, xp = x < d.nx - 1 ? i + 1 : i;
    const size_t ym = y > 0 ? i - sy : i, yp = y < d.ny - 1 ? i + sy : i;
    const size_t zm = z > 0 ? i - sz : i, zp = z < d.nz - 1 ? i + sz : i;
    const float fc = F[i], mc = Mw[i];
    const float gx = pp_esm_axis(F[xm], F[xp], mc, Mw[xm], Mw[xp], x == 0, x == d.nx - 1, K.ix);
    const float gy = pp_esm_axis(F[ym], F[yp], mc, Mw[ym], Mw[yp], y == 0, y == d.ny - 1, K.iy);
    const float gz = pp_esm_axis(F[zm], F[zp], mc, Mw[zm], Mw[zp], z == 0, z == d.nz - 1, K.iz);
    const pp_esm_out o = pp_esm_voxel(K, fc, mc, gx, gy, gz);
    U[i] = o.ux;
    U[N + i] = o.uy;
    U[2 * N + i] = o.uz;
    a_ssd += (double)o.sq_speed;
    a_ssc += (double)o.sq_update;
    a_n += (double)o.counted;
  }
  pp_block_sum3<NT>(a_ssd, a_ssc, a_n, red);
  if (threadIdx.x == 0) {
    partials[3 * (size_t)blockIdx.x + 0] = a_ssd;
    partials[3 * (size_t)blockIdx.x + 1] = a_ssc;
    partials[3 * (size_t)blockIdx.x + 2] = a_n;
  }
}

// The same update, four consecutive x voxels per thread (rows 16-byte aligned: nx % 4 == 0): the centre row and its four
// y / z neighbour rows arrive as one 16-byte load each, the two x neighbours beyond the quad as scalars -- 14 loads for
// four voxels instead of 56.  Same operations per voxel, so U is bit-identical; the partial sums group differently.
__global__ void __launch_bounds__(NT) k_demons_force4(const float* __restrict__ F, const float* __restrict__ Mw,
                                                      float* __restrict__ U, pp_dims d, pp_esm_consts K,
                                                      double* __restrict__ partials, const int* __restrict__ halt) {
  __shared__ double red[3 * NT];
  if (halt && *halt) return;
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  const size_t sy = d.nx, sz = (size_t)d.nx * d.ny;
  const size_t nxq = (size_t)d.nx / 4, NQ = N / 4;
  double a_ssd = 0.0, a_ssc = 0.0, a_n = 0.0;
  for (size_t q = (size_t)blockIdx.x * NT + threadIdx.x; q < NQ; q += (size_t)gridDim.x * NT) {
    const int x = (int)(q % nxq) * 4;
    const int y = (int)((q / nxq) % d.ny);
    const int z = (int)(q / (nxq * d.ny));
    const size_t i = ((size_t)z * d.ny + y) * d.nx + x;
    const size_t ym = y > 0 ? i - sy : i, yp = y < d.ny - 1 ? i + sy : i;
    const size_t zm = z > 0 ? i - sz : i, zp = z < d.nz - 1 ? i + sz : i;
    const float4 fc4 = *reinterpret_cast<const float4*>(F + i), mc4 = *reinterpret_cast<const float4*>(Mw + i);
    const float4 fym4 = *reinterpret_cast<const float4*>(F + ym), fyp4 = *reinterpret_cast<const float4*>(F + yp);
    const float4 fzm4 = *reinterpret_cast<const float4*>(F + zm), fzp4 = *reinterpret_cast<const float4*>(F + zp);
    const float4 mym4 = *reinterpret_cast<const float4*>(Mw + ym), myp4 = *reinterpret_cast<const float4*>(Mw + yp);
    const float4 mzm4 = *reinterpret_cast<const float4*>(Mw + zm), mzp4 = *reinterpret_cast<const float4*>(Mw + zp);
    const size_t ixm = x > 0 ? i - 1 : i, ixp = x + 4 < d.nx ? i + 4 : i + 3;
    const float fxm = F[ixm], fxp = F[ixp], mxm = Mw[ixm], mxp = Mw[ixp];
    const float fc[6] = {fxm, fc4.x, fc4.y, fc4.z, fc4.w, fxp}, mc[6] = {mxm, mc4.x, mc4.y, mc4.z, mc4.w, mxp};
    const float fym[4] = {fym4.x, fym4.y, fym4.z, fym4.w}, fyp[4] = {fyp4.x, fyp4.y, fyp4.z, fyp4.w};
    const float fzm[4] = {fzm4.x, fzm4.y, fzm4.z, fzm4.w}, fzp[4] = {fzp4.x, fzp4.y, fzp4.z, fzp4.w};
    const float mym[4] = {mym4.x, mym4.y, mym4.z, mym4.w}, myp[4] = {myp4.x, myp4.y, myp4.z, myp4.w};
    const float mzm[4] = {mzm4.x, mzm4.y, mzm4.z, mzm4.w}, mzp[4] = {mzp4.x, mzp4.y, mzp4.z, mzp4.w};
    float ux[4], uy[4], uz[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float gx = pp_esm_axis(fc[v], fc[v + 2], mc[v + 1], mc[v], mc[v + 2], x + v == 0, x + v == d.nx - 1, K.ix);
      const float gy = pp_esm_axis(fym[v], fyp[v], mc[v + 1], mym[v], myp[v], y == 0, y == d.ny - 1, K.iy);
      const float gz = pp_esm_axis(fzm[v], fzp[v], mc[v + 1], mzm[v], mzp[v], z == 0, z == d.nz - 1, K.iz);
      const pp_esm_out o = pp_esm_voxel(K, fc[v + 1], mc[v + 1], gx, gy, gz);
      ux[v] = o.ux;
      uy[v] = o.uy;
      uz[v] = o.uz;
      a_ssd += (double)o.sq_speed;
      a_ssc += (double)o.sq_update;
      a_n += (double)o.counted;
    }
    *reinterpret_cast<float4*>(U + i) = make_float4(ux[0], ux[1], ux[2], ux[3]);
    *reinterpret_cast<float4*>(U + N + i) = make_float4(uy[0], uy[1], uy[2], uy[3]);
    *reinterpret_cast<float4*>(U + 2 * N + i) = make_float4(uz[0], uz[1], uz[2], uz[3]);
  }
  pp_block_sum3<NT>(a_ssd, a_ssc, a_n, red);
  if (threadIdx.x == 0) {
    partials[3 * (size_t)blockIdx.x + 0] = a_ssd;
    partials[3 * (size_t)blockIdx.x + 1] = a_ssc;
    partials[3 * (size_t)blockIdx.x + 2] = a_n;
  }
}

// End of an iteration: fold the per-block partial sums (fixed order -> deterministic), publish
// metric / RMS change, count the iteration and apply FiniteDifferenceImageFilter::Halt().
__global__ void __launch_bounds__(NT) k_demons_finalize(const double* __restrict__ partials, int nblocks,
                                                        pp_dev_stats* __restrict__ st, double max_rms) {
  __shared__ double red[3 * NT];
  if (st->halt) return;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += NT) {
    a += partials[3 * (size_t)i + 0];
    b += partials[3 * (size_t)i + 1];
    c += partials[3 * (size_t)i + 2];
  }
  pp_block_sum3<NT>(a, b, c, red);
  if (threadIdx.x == 0) {
    st->ssd = a;
    st->ssc = b;
    st->npx = (long long)c;
    if (c > 0.0) {
      st->metric = a / c;
      st->rms = sqrt(b / c);
    }
    pp_stats_record(st);
    st->elapsed += 1;
    if (max_rms > st->rms) st->halt = 1;  // Halt(): m_MaximumRMSError > m_RMSChange
  }
}

__global__ void __launch_bounds__(NT) k_add_inplace(float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                    const int* __restrict__ halt) {
  if (halt && *halt) return;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) a[i] += b[i];
}

// After a fused Execute the newest field sits in `alt` when an odd number of iterations ran.
__global__ void __launch_bounds__(NT) k_copy_if_odd(float* __restrict__ dst, const float* __restrict__ alt, size_t n,
                                                    const pp_dev_stats* __restrict__ st) {
  if ((st->elapsed & 1) == 0) return;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) dst[i] = alt[i];
}

// Dense rows of nx voxels -> rows of px (a multiple of four) with the last voxel repeated into the padding (never read as
// data: every read clamps x to nx - 1; a finite value keeps the padding's arithmetic quiet).  One thread = one quad.
__global__ void __launch_bounds__(NT) k_pad_rows(const float* __restrict__ src, float* __restrict__ dst, int nx, int px, size_t rows) {
  const size_t qpr = (size_t)(px / 4), n = rows * qpr;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const size_t r = i / qpr;
    const int x = (int)(i - r * qpr) * 4;
    const float* s = src + r * (size_t)nx;
    float4 v;
    v.x = s[x < nx ? x : nx - 1];
    v.y = s[x + 1 < nx ? x + 1 : nx - 1];
    v.z = s[x + 2 < nx ? x + 2 : nx - 1];
    v.w = s[x + 3 < nx ? x + 3 : nx - 1];
    *reinterpret_cast<float4*>(dst + r * (size_t)px + x) = v;
  }
}
// The three components of the newest padded field (`even` after an even number of iterations, `odd` otherwise) -> dense rows.
__global__ void __launch_bounds__(NT) k_unpad_field(float* __restrict__ dst, const float* __restrict__ even, const float* __restrict__ odd,
                                                    int nx, int px, size_t rows, const pp_dev_stats* __restrict__ st) {
  const float* __restrict__ src = (st->elapsed & 1) ? odd : even;
  const size_t n = 3 * rows * (size_t)nx;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const size_t r = i / (size_t)nx;   // (row index over the three components: they are rows * px apart, i.e. contiguous rows)
    dst[i] = src[r * (size_t)px + (i - r * (size_t)nx)];
  }
}

// k_demons_force4's precondition: whole 16-byte quads per row, 16-byte aligned volumes
bool force_vec4(const pp_dims& d, const float* f, const float* m, const float* u) {
  const size_t N = (size_t)d.nx * d.ny * d.nz;
  return d.nx % 4 == 0 && N % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(u)) % 16 == 0);
}

unsigned grid_for(size_t work, unsigned cap = 65535u * 4u) {
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// ---------------------------------------------------------------------------------------
// FUSED kernels.  Tile TX x TY outputs per plane, NT = (TX/4) * TY threads; thread (cx, cy) owns
// the 4 consecutive x outputs x = tx0 + 4 cx .. +3 of row y = ty0 + cy on every plane of its chunk.
//
// Grid: one 1-D launch of 8 * per_xcd blocks.  The dispatcher places block b on XCD b % 8 (observed,
// used for speed only), so block b takes tile rank (b % 8) * per_xcd + b / 8 in (x fastest, y, z
// slowest) order: each XCD works through a contiguous run of tiles and the tiles resident on it at
// any moment are x/y neighbours.  Their 2-voxel halos -- whose 272-B rows straddle four 128-B lines
// instead of two -- then hit that XCD's L2 instead of being fetched once per XCD from the fabric.

// Two tile shapes of 1024 outputs each: 64 x 16 (rows of two cache lines; the default) and 32 x 32 (smaller
// halo, finer tiling of grids that 64-wide tiles overhang or whose tile count leaves block slots idle).
constexpr int TILE_OUT = 1024;
template <int SH> struct tile_shape;
template <> struct tile_shape<0> { static constexpr int TX = 64, TY = 16; };
template <> struct tile_shape<1> { static constexpr int TX = 32, TY = 32; };
// OPT = outputs per thread along x (4: 256 threads, float4 rows; 2: 512 threads, float2 rows -- half the
// registers per thread, twice the waves per CU for the same LDS tile).
template <int R, int OPT, int SH>
struct fused_geom {
  static constexpr int TX = tile_shape<SH>::TX, TY = tile_shape<SH>::TY;
  static constexpr int NTH = TX * TY / OPT;       // threads per block
  static constexpr int LX = TX / OPT;             // threads along x
  static constexpr int UW = TX + 2 * R;           // smoothing-input tile width  (x from tx0 - R)
  static constexpr int UH = TY + 2 * R;           // smoothing-input tile height (y from ty0 - R)
  static constexpr int UWP = (UW + 3) / 4 * 4;    // row pitch, keeps rows 16-B aligned
  static constexpr int NU = UW * UH;              // voxels per smoothing-input plane
  static constexpr int KU = (NU + NTH - 1) / NTH;   // of which one thread owns at most KU
  static constexpr int MW = UW + 2;               // image tile (1 more voxel each side for gradients)
  static constexpr int MH = UH + 2;
  // Row pitch of the packed image tile (float2).  OPT == 2 (the 512-thread kernels): padded to UW + 32, so that the ESM pass's
  // ds_read_b64 of a wave -- 64 consecutive update voxels, which wrap from one 68-voxel row into the next -- still touches 64
  // distinct banks after the wrap (2 (MWP - UW) = 64 dwords: the same bank phase); unpadded, every wrapped wave-read hit two
  // banks twice (+39 % cycles on kernel A's most frequent LDS read, MI355X_MICROARCH.md's LDS table).
#ifndef PP_A_MWP_PAD
#define PP_A_MWP_PAD 1
#endif
  static constexpr int MWP = (PP_A_MWP_PAD != 0 && OPT == 2) ? UW + 32 : MW;
  static constexpr int NB = MW * MH - UW * UH;    // border ring elements
  static constexpr int XI = UH * (TX / 4);        // x-pass work items per component
  // LDS carve (floats): region 1 holds the two image tiles during the force phase and is reused for
  // the x-pass output afterwards; region 2 holds the raw smoothing input.
  static constexpr int SZ_IMG = 2 * MH * MWP;
  static constexpr int SZ_X = 3 * UH * TX;
  static constexpr int R1 = ((SZ_IMG > SZ_X ? SZ_IMG : SZ_X) + 3) / 4 * 4;
  static constexpr int SZ_U = 3 * UH * UWP;
  static constexpr int SMEM = R1 + SZ_U;
  static_assert(SZ_U * 4 >= 3 * NTH * 8, "region 2 doubles as the reduction buffer");
  static_assert(NB <= NTH, "border ring must fit one pass");
  static_assert(MH * MWP < 65536 && UH * UWP < 65536, "packed LDS slots are 16 bit");
};

// x pass: item (row uy, group cx) reads 4 + 2R inputs of `us` and writes 4 outputs to `xs`.
template <int R, int OPT, int SH>
__device__ __forceinline__ void fused_xpass(const float* __restrict__ us /*[3][UH][UWP]*/,
                                            float* __restrict__ xs /*[3][UH][TX]*/, const pp_taps_small& wx) {
  using G = fused_geom<R, OPT, SH>;
  constexpr int TX = G::TX;
  for (int it = threadIdx.x; it < 3 * G::XI; it += G::NTH) {
    const int c = it / G::XI;
    const int rem = it - c * G::XI;
    const int uy = rem / (TX / 4);
    const int cx = rem - uy * (TX / 4);
    const float* src = us + (c * G::UH + uy) * G::UWP + 4 * cx;
    float in[4 + 2 * R];
#pragma unroll
    for (int q = 0; q < (4 + 2 * R) / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      in[4 * q + 0] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
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
    *reinterpret_cast<float4*>(xs + (c * G::UH + uy) * TX + 4 * cx) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// y pass for this thread's OPT outputs of component c.
template <int R, int OPT, int SH>
__device__ __forceinline__ void fused_ypass(const float* __restrict__ xs, int c, int cx, int cy, const pp_taps_small& wy,
                                            float v[OPT]) {
  using G = fused_geom<R, OPT, SH>;
  constexpr int TX = G::TX;
#pragma unroll
  for (int j = 0; j < OPT; ++j) v[j] = 0.0f;
#pragma unroll
  for (int k = 0; k < 2 * R + 1; ++k) {
    const float* p = xs + (c * G::UH + cy + k) * TX + OPT * cx;
    if (OPT == 4) {
      const float4 a = *reinterpret_cast<const float4*>(p);
      const float w = wy.h[k < R ? R - k : k - R];
      v[0] = fmaf(w, a.x, v[0]);
      v[1] = fmaf(w, a.y, v[1]);
      v[2] = fmaf(w, a.z, v[2]);
      v[3] = fmaf(w, a.w, v[3]);
    } else {
      const float2 a = *reinterpret_cast<const float2*>(p);
      const float w = wy.h[k < R ? R - k : k - R];
      v[0] = fmaf(w, a.x, v[0]);
      v[1] = fmaf(w, a.y, v[1]);
    }
  }
}

template <int R, int OPT>
struct zring {
  float r[3][OPT][2 * R + 1];
  __device__ __forceinline__ void push(const float v[3][OPT]) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < OPT; ++j) {
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) r[c][j][k] = r[c][j][k + 1];
        r[c][j][2 * R] = v[c][j];
      }
  }
  __device__ __forceinline__ float dot(int c, int j, const pp_taps_small& wz) const {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 2 * R + 1; ++k) s = fmaf(wz.h[k < R ? R - k : k - R], r[c][j][k], s);
    return s;
  }
};

struct fused_args {
  pp_dims d;
  int zchunk;
  int gx, gy, gz;   // tile grid
  int per_xcd;      // ceil(gx * gy * gz / 8)
  int streaming;    // generation 2: non-temporal output stores (volumes far beyond the infinity cache)
  int masked;       // generation 2: the MASK kernels (even rows, a whole field under 2^31 bytes; pp_demons_fused2.h)
  int big;          // generation 2: a 3-component field spans >= 2^32 bytes -- the BIG instances (64-bit bases for the field arrays)
  // generation 2, mixed tile shapes (SH == 2): region 0 = gx x gy tiles of 64 x 16 from x = 0, region 1 = gx2 x gy2 tiles of
  // 32 x 32 from x = x2_off (the columns a 64-wide tile would overhang by half or more); gx2 == 0: one shape only
  int gx2, gy2, x2_off;
  // generation 2: row pitch in voxels of EVERY volume the kernels touch (>= d.nx).  The fused Execute copies volumes whose
  // rows are not whole 16-byte quads into padded rows once: strips are then 16-byte aligned and pairs 8-byte aligned for any
  // row length, and the MASK instances (which need pairs) serve odd row lengths too.  Padding is written, never read as data.
  int px;
  // generation 2, PP_SOFTSYNC builds: progress words of this kernel's launch (8 XCDs x 64 resident blocks), the other kernel's
  // set (cleared by this launch) and the allowed lead in plane steps (0: publish only)
  unsigned* sync;
  unsigned* sync_other;
  int sync_lag;
  pp_taps_small wx, wy, wz;
};

// Tile of this block (see the grid note above); false for the few surplus blocks of the last XCD run.  `region`: which
// of the two tile regions of a mixed launch the caller's tile shape belongs to (0 when there is one shape).
// PP_PAIR_MIX (round 5): tile columns differ systematically in how long a block takes on them (x-border tiles re-read their
// clamped halo from lines they already hold; at 512 x 512 x 256 kernel A's columns 0, 3, 4, 7 end ~18 us and kernel B's
// columns 0, 7 ~33 us before the others), and the dispatcher gives a CU the XCD run's blocks j and j + 32 -- the SAME column
// twice when the run is a whole number of tile rows.  So the second block of a CU takes its x-NEIGHBOUR's tile instead
// (rank ^ 1, where both ranks lie in the second half of this XCD's run, in the 64 x 16 region and in one tile row: a bijection
// on the tiles): every CU then hosts a slow and a fast column, and the pair priority (pp_demons_fused2.h) levels them.
// Measured with it: an XCD's blocks end within 11 us (A) / 16 us (B) of each other instead of 27 / 40, iteration -2.0 .. -2.8 %.
#ifndef PP_PAIR_MIX
#define PP_PAIR_MIX 1
#endif
__device__ __forceinline__ unsigned fused_rank(const fused_args& a) {
  const unsigned b = blockIdx.x;
  const unsigned first = (b & 7u) * (unsigned)a.per_xcd, j = b >> 3;
  unsigned rank = first + j;
#if PP_PAIR_MIX
  if (j >= 32u) {
    const unsigned lo = rank & ~1u, hi = lo | 1u, n1 = (unsigned)a.gx * a.gy * a.gz;
    if (lo >= first + 32u && hi < first + (unsigned)a.per_xcd && hi < n1 && (hi % (unsigned)a.gx) != 0u) rank ^= 1u;
  }
#endif
  return rank;
}
__device__ __forceinline__ int fused_region(const fused_args& a) {   // (surplus blocks report region 0 and fail fused_tile there)
  const unsigned n1 = (unsigned)a.gx * a.gy * a.gz, n2 = (unsigned)a.gx2 * a.gy2 * a.gz, rank = fused_rank(a);
  return rank >= n1 && rank < n1 + n2 ? 1 : 0;
}
__device__ __forceinline__ bool fused_tile(const fused_args& a, int TX, int TY, int& tx0, int& ty0, int& z0, unsigned& rank, int region = 0) {
  rank = fused_rank(a);
  const unsigned n1 = (unsigned)a.gx * a.gy * a.gz, n2 = (unsigned)a.gx2 * a.gy2 * a.gz;
  if (rank >= n1 + n2) return false;
  if ((region != 0) != (rank >= n1)) return false;
  const unsigned r = region ? rank - n1 : rank, gx = region ? (unsigned)a.gx2 : (unsigned)a.gx, gy = region ? (unsigned)a.gy2 : (unsigned)a.gy;
  const unsigned tx = r % gx, ty = (r / gx) % gy, tz = r / (gx * gy);
  tx0 = (region ? a.x2_off : 0) + (int)tx * TX;
  ty0 = (int)ty * TY;
  z0 = (int)tz * a.zchunk;
  return true;
}

// Load through a wave-uniform base pointer plus a 32-bit per-lane BYTE offset: the address form
// global_load takes directly (scalar base + unsigned VGPR offset), no 64-bit VALU arithmetic.
__device__ __forceinline__ float ld_off(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}

// per-voxel flags packed beside the LDS slots
constexpr unsigned F_CNT = 1u, F_XLO = 2u, F_XHI = 4u, F_YLO = 8u, F_YHI = 16u, F_VALID = 32u, F_OOV = 64u;

// Measurement builds only (-DPP_TRACE, tools/kbench): shader-clock stamps of one block's waves at the barriers of the
// plane loop, read back through pp_debug_trace_read.  Product builds compile none of it.
#ifdef PP_TRACE
constexpr int PP_TRACE_STEPS = 140, PP_TRACE_SLOTS = 6;
__device__ unsigned pp_trace_buf[2][8][PP_TRACE_STEPS][PP_TRACE_SLOTS];
#define PP_TRACE_MARK(on, kern, step, slot)                                                     \
  do {                                                                                          \
    if ((on) && (threadIdx.x & 63u) == 0 && (step) >= 0 && (step) < PP_TRACE_STEPS)             \
      pp_trace_buf[kern][threadIdx.x >> 6][step][slot] = (unsigned)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PP_TRACE_MARK(on, kern, step, slot) \
  do {                                      \
  } while (0)
#endif

// ---- kernel A: ESM update + 3-D Gaussian of the update ---------------------------------
#ifndef PP_FUSED_DEFAULT_OPT
#define PP_FUSED_DEFAULT_OPT 2
#endif
#ifndef PP_A_WAVES
#define PP_A_WAVES 1
#endif
#ifndef PP_B_WAVES
#define PP_B_WAVES 1
#endif
template <int R, int OPT, int SH>
__global__ void __launch_bounds__(TILE_OUT / OPT, PP_A_WAVES) k_fused_force_smooth(const float* __restrict__ F, const float* __restrict__ Mw,
                                                           float* __restrict__ Us, fused_args a, pp_esm_consts K,
                                                           double* __restrict__ partials, const int* __restrict__ halt) {
  using G = fused_geom<R, OPT, SH>;
  constexpr int NTH = G::NTH, TX = G::TX, TY = G::TY;
  __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
  float* const s_m = smem;                      // warped moving, current plane   (force phase)
  float* const s_f = smem + G::MH * G::MWP;     // fixed, current plane           (force phase)
  float* const s_x = smem;                      // after the x pass               (smoothing phase)
  float* const s_u = smem + G::R1;              // raw update, current plane
  if (halt && *halt) return;
  int tx0, ty0, z0;
  unsigned rank;
  if (!fused_tile(a, TX, TY, tx0, ty0, z0, rank)) return;

  const pp_dims d = a.d;
  const int t = threadIdx.x;
  const int cx = t % G::LX, cy = t / G::LX;
  const unsigned sy = d.nx, sz = (unsigned)d.nx * d.ny;   // volumes hold < 2^31 voxels
  const size_t N = (size_t)sz * d.nz;

  // Owned smoothing-input voxels (ux, uy).  Image values are always fetched at the clamped position
  // so out-of-volume halo slots replicate the edge update (ZeroFluxNeumann on the smoothing input).
  unsigned slots[G::KU];  // read slot of the clamped position | write slot << 16   (in s_m / s_f)
  unsigned uflag[G::KU];  // slot in s_u | flags << 16
  unsigned own_g[G::KU];  // in-plane BYTE offset of the clamped position
  float mprev[G::KU], mcur[G::KU], mnext[G::KU], fprev[G::KU], fcur[G::KU], fnext[G::KU];
#pragma unroll
  for (int k = 0; k < G::KU; ++k) {
    const int e = t + k * NTH;
    const int ee = e < G::NU ? e : 0;
    const int uy = ee / G::UW, ux = ee - uy * G::UW;
    const int xg = tx0 - R + ux, yg = ty0 - R + uy;
    const int xc = pp_clampi(xg, 0, d.nx - 1), yc = pp_clampi(yg, 0, d.ny - 1);
    own_g[k] = ((unsigned)yc * sy + (unsigned)xc) * 4u;
    const unsigned wslot = (unsigned)((uy + 1) * G::MWP + (ux + 1));
    const unsigned rslot = (unsigned)((yc - (ty0 - R - 1)) * G::MWP + (xc - (tx0 - R - 1)));
    slots[k] = rslot | (wslot << 16);
    unsigned fl = 0;
    if (e < G::NU) fl |= F_VALID;
    if (e < G::NU && xg >= tx0 && xg < tx0 + TX && xg < d.nx && yg >= ty0 && yg < ty0 + TY && yg < d.ny) fl |= F_CNT;
    if (xc == 0) fl |= F_XLO;
    if (xc == d.nx - 1) fl |= F_XHI;
    if (yc == 0) fl |= F_YLO;
    if (yc == d.ny - 1) fl |= F_YHI;
    uflag[k] = (unsigned)(uy * G::UWP + ux) | (fl << 16);
  }
  // Border ring of the image tiles (needed only in-plane): one element per low thread.
  int brd_w = -1;
  unsigned brd_g = 0;
  if (t < G::NB) {
    int my, mx;
    if (t < G::MW) { my = 0; mx = t; }
    else if (t < 2 * G::MW) { my = G::MH - 1; mx = t - G::MW; }
    else { const int q = t - 2 * G::MW; my = 1 + q / 2; mx = (q & 1) ? G::MW - 1 : 0; }
    const int xc = pp_clampi(tx0 - R - 1 + mx, 0, d.nx - 1), yc = pp_clampi(ty0 - R - 1 + my, 0, d.ny - 1);
    brd_w = my * G::MWP + mx;
    brd_g = ((unsigned)yc * sy + (unsigned)xc) * 4u;
  }

  const int zs = z0 - R;                                            // first smoothing-input plane
  const int zo_last = (z0 + a.zchunk - 1 < d.nz - 1) ? z0 + a.zchunk - 1 : d.nz - 1;
  const int ze = zo_last + R;                                       // last smoothing-input plane
  const int zlo = pp_clampi(zs, 0, d.nz - 1);

  // prime the z window of the image values at plane zlo
  float bm = 0.0f, bf = 0.0f;
  {
    const size_t pm = (size_t)pp_clampi(zlo - 1, 0, d.nz - 1) * sz, pc = (size_t)zlo * sz,
                 pn = (size_t)pp_clampi(zlo + 1, 0, d.nz - 1) * sz;
#pragma unroll
    for (int k = 0; k < G::KU; ++k) {
      mprev[k] = ld_off(Mw + pm, own_g[k]); fprev[k] = ld_off(F + pm, own_g[k]);
      mcur[k] = ld_off(Mw + pc, own_g[k]);  fcur[k] = ld_off(F + pc, own_g[k]);
      mnext[k] = ld_off(Mw + pn, own_g[k]); fnext[k] = ld_off(F + pn, own_g[k]);
    }
    if (brd_w >= 0) {
      bm = ld_off(Mw + pc, brd_g);
      bf = ld_off(F + pc, brd_g);
    }
  }

  zring<R, OPT> ring;
  float v[3][OPT];
  float a_ssd = 0.0f, a_ssc = 0.0f, a_n = 0.0f;   // <= ~40 terms per thread: fp32 is exact enough, folded in fp64 below
  int zc_done = -1;

  for (int zi = zs; zi <= ze; ++zi) {
    const int zc = pp_clampi(zi, 0, d.nz - 1);
    if (zc != zc_done) {
      // (1) prefetch plane zc + 2 (and the border of zc + 1) into registers, publish plane zc to LDS
      float min_[G::KU], fin_[G::KU], bm_n = 0.0f, bf_n = 0.0f;
      {
        const size_t p2 = (size_t)pp_clampi(zc + 2, 0, d.nz - 1) * sz, p1 = (size_t)pp_clampi(zc + 1, 0, d.nz - 1) * sz;
#pragma unroll
        for (int k = 0; k < G::KU; ++k) {
          min_[k] = ld_off(Mw + p2, own_g[k]);
          fin_[k] = ld_off(F + p2, own_g[k]);
        }
        if (brd_w >= 0) {
          bm_n = ld_off(Mw + p1, brd_g);
          bf_n = ld_off(F + p1, brd_g);
        }
      }
      __syncthreads();  // the previous plane's y pass has finished reading region 1
#pragma unroll
      for (int k = 0; k < G::KU; ++k)
        if ((uflag[k] >> 16) & F_VALID) {
          s_m[slots[k] >> 16] = mcur[k];
          s_f[slots[k] >> 16] = fcur[k];
        }
      if (brd_w >= 0) {
        s_m[brd_w] = bm;
        s_f[brd_w] = bf;
      }
      __syncthreads();
      // (2) ESM update at every smoothing-input voxel of this plane
      const bool count_plane = (zc >= z0 && zc <= zo_last);
      const bool zlo_b = (zc == 0), zhi_b = (zc == d.nz - 1);
#pragma unroll
      for (int k = 0; k < G::KU; ++k) {
        const unsigned fl = uflag[k] >> 16;
        if (fl & F_VALID) {
          const int l = (int)(slots[k] & 0xffffu);
          const float gx = pp_esm_axis(s_f[l - 1], s_f[l + 1], mcur[k], s_m[l - 1], s_m[l + 1], (fl & F_XLO) != 0,
                                       (fl & F_XHI) != 0, K.ix);
          const float gy = pp_esm_axis(s_f[l - G::MWP], s_f[l + G::MWP], mcur[k], s_m[l - G::MWP], s_m[l + G::MWP],
                                       (fl & F_YLO) != 0, (fl & F_YHI) != 0, K.iy);
          const float gz = pp_esm_axis(fprev[k], fnext[k], mcur[k], mprev[k], mnext[k], zlo_b, zhi_b, K.iz);
          const pp_esm_out o = pp_esm_voxel(K, fcur[k], mcur[k], gx, gy, gz);
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
      __syncthreads();  // s_u complete; region 1 (image tiles) is dead from here
      // (3) x pass, (4) y pass
      fused_xpass<R, OPT, SH>(s_u, s_x, a.wx);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) fused_ypass<R, OPT, SH>(s_x, c, cx, cy, a.wy, v[c]);
      // rotate the z window of the image values
#pragma unroll
      for (int k = 0; k < G::KU; ++k) {
        mprev[k] = mcur[k]; mcur[k] = mnext[k]; mnext[k] = min_[k];
        fprev[k] = fcur[k]; fcur[k] = fnext[k]; fnext[k] = fin_[k];
      }
      bm = bm_n;
      bf = bf_n;
      zc_done = zc;
    }
    // (5) z pass out of the register window (repeated planes re-enter: clamped edge)
    ring.push(v);
    const int zo = zi - R;
    if (zo >= z0 && zo <= zo_last) {
      const int x = tx0 + OPT * cx, y = ty0 + cy;
      if (y < d.ny && x < d.nx) {
        const size_t o = (size_t)zo * sz + (size_t)y * sy + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float r[OPT];
#pragma unroll
          for (int j = 0; j < OPT; ++j) r[j] = ring.dot(c, j, a.wz);
          if ((d.nx % OPT) == 0) {
            if (OPT == 4) *reinterpret_cast<float4*>(Us + c * N + o) = make_float4(r[0], r[1], r[2], r[OPT - 1]);
            else *reinterpret_cast<float2*>(Us + c * N + o) = make_float2(r[0], r[1]);
          } else {
#pragma unroll
            for (int j = 0; j < OPT; ++j)
              if (x + j < d.nx) Us[c * N + o + j] = r[j];
          }
        }
      }
    }
  }
  __syncthreads();
  double r_ssd = (double)a_ssd, r_ssc = (double)a_ssc, r_n = (double)a_n;
  pp_block_sum3<NTH>(r_ssd, r_ssc, r_n, reinterpret_cast<double*>(s_u));
  if (t == 0) {
    partials[3 * (size_t)rank + 0] = r_ssd;
    partials[3 * (size_t)rank + 1] = r_ssc;
    partials[3 * (size_t)rank + 2] = r_n;
  }
}

// ---- kernel B: D' = G_d * (D + U), then the next iteration's warped moving image ------
template <int R, int OPT, int SH>
__global__ void __launch_bounds__(TILE_OUT / OPT, PP_B_WAVES) k_fused_add_smooth_warp(const float* __restrict__ D, const float* __restrict__ Us,
                                                              const float* __restrict__ M, float* __restrict__ Dn,
                                                              float* __restrict__ Mw, fused_args a, pp_warp_scale sc,
                                                              const int* __restrict__ halt) {
  using G = fused_geom<R, OPT, SH>;
  constexpr int NTH = G::NTH, TX = G::TX, TY = G::TY;
  __shared__ __attribute__((aligned(16))) float smem[G::SZ_X + G::SZ_U];
  float* const s_x = smem;
  float* const s_u = smem + G::SZ_X;
  if (halt && *halt) return;
  int tx0, ty0, z0;
  unsigned rank;
  if (!fused_tile(a, TX, TY, tx0, ty0, z0, rank)) return;

  const pp_dims d = a.d;
  const int t = threadIdx.x;
  const int cx = t % G::LX, cy = t / G::LX;
  const unsigned sy = d.nx, sz = (unsigned)d.nx * d.ny;
  const size_t N = (size_t)sz * d.nz;

  unsigned own_g[G::KU];   // in-plane BYTE offset of the clamped position
  unsigned own_u[G::KU];   // slot in s_u (0xffff....: not owned)
#pragma unroll
  for (int k = 0; k < G::KU; ++k) {
    const int e = t + k * NTH;
    const int ee = e < G::NU ? e : 0;
    const int uy = ee / G::UW, ux = ee - uy * G::UW;
    const int xc = pp_clampi(tx0 - R + ux, 0, d.nx - 1), yc = pp_clampi(ty0 - R + uy, 0, d.ny - 1);
    own_g[k] = ((unsigned)yc * sy + (unsigned)xc) * 4u;
    own_u[k] = e < G::NU ? (unsigned)(uy * G::UWP + ux) : 0xffffffffu;
  }

  const int zs = z0 - R;
  const int zo_last = (z0 + a.zchunk - 1 < d.nz - 1) ? z0 + a.zchunk - 1 : d.nz - 1;
  const int ze = zo_last + R;
  const int zlo = pp_clampi(zs, 0, d.nz - 1);

  // raw D and U of the plane about to be smoothed; summed when published
  float dl[3][G::KU], ul[3][G::KU];
  {
    const size_t pc = (size_t)zlo * sz;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < G::KU; ++k) {
        dl[c][k] = ld_off(D + c * N + pc, own_g[k]);
        ul[c][k] = ld_off(Us + c * N + pc, own_g[k]);
      }
  }

  zring<R, OPT> ring;
  float v[3][OPT];
  int zc_done = -1;

  for (int zi = zs; zi <= ze; ++zi) {
    const int zc = pp_clampi(zi, 0, d.nz - 1);
    if (zc != zc_done) {
      __syncthreads();  // the previous plane's x pass has finished reading s_u
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < G::KU; ++k)
          if (own_u[k] != 0xffffffffu) s_u[c * G::UH * G::UWP + own_u[k]] = dl[c][k] + ul[c][k];
      // issue the next plane's loads now; they are consumed at the top of the next step
      {
        const size_t pn = (size_t)pp_clampi(zc + 1, 0, d.nz - 1) * sz;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int k = 0; k < G::KU; ++k) {
            dl[c][k] = ld_off(D + c * N + pn, own_g[k]);
            ul[c][k] = ld_off(Us + c * N + pn, own_g[k]);
          }
      }
      __syncthreads();
      fused_xpass<R, OPT, SH>(s_u, s_x, a.wx);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) fused_ypass<R, OPT, SH>(s_x, c, cx, cy, a.wy, v[c]);
      zc_done = zc;
    }
    ring.push(v);
    const int zo = zi - R;
    if (zo >= z0 && zo <= zo_last) {
      const int x = tx0 + OPT * cx, y = ty0 + cy;
      if (y < d.ny && x < d.nx) {
        const size_t o = (size_t)zo * sz + (size_t)y * sy + x;
        float dn[3][OPT];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < OPT; ++j) dn[c][j] = ring.dot(c, j, a.wz);
        float mw[OPT];
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
          int bx, by, bz;
          float fx, fy, fz;
          pp_split(x + j, dn[0][j] * sc.ix, bx, fx);
          pp_split(y, dn[1][j] * sc.iy, by, fy);
          pp_split(zo, dn[2][j] * sc.iz, bz, fz);
          const bool inside = (x + j < d.nx) && pp_inside1(bx, fx, d.nx) && pp_inside1(by, fy, d.ny) && pp_inside1(bz, fz, d.nz);
          mw[j] = inside ? pp_trilinear(M, d.nx, d.ny, d.nz, bx, fx, by, fy, bz, fz) : FLT_MAX;
        }
        if ((d.nx % OPT) == 0) {
          if (OPT == 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
              *reinterpret_cast<float4*>(Dn + c * N + o) = make_float4(dn[c][0], dn[c][1], dn[c][2], dn[c][OPT - 1]);
            *reinterpret_cast<float4*>(Mw + o) = make_float4(mw[0], mw[1], mw[2], mw[OPT - 1]);
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<float2*>(Dn + c * N + o) = make_float2(dn[c][0], dn[c][1]);
            *reinterpret_cast<float2*>(Mw + o) = make_float2(mw[0], mw[1]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < OPT; ++j)
            if (x + j < d.nx) {
              Dn[o + j] = dn[0][j];
              Dn[N + o + j] = dn[1][j];
              Dn[2 * N + o + j] = dn[2][j];
              Mw[o + j] = mw[j];
            }
        }
      }
    }
  }
}

// z-window handling of the generation-2 kernels: plane loop unrolled 2R+1 times (window renamed) up to this radius,
// register moves above it (the unrolled body of radius 4/5 would be 9-11 copies of ~600 instructions)
#ifndef PP_RING_UNROLL_MAX_R
#define PP_RING_UNROLL_MAX_R 3
#endif
#include "pp_demons_fused2.h"
#include "pp_demons_cube.h"

// ---------------------------------------------------------------------------------------
// host side

void esm_consts(const pp_geom* g, const pp_demons_params* p, pp_esm_consts* K) {
  K->ix = (float)(1.0 / g->spacing[0]);
  K->iy = (float)(1.0 / g->spacing[1]);
  K->iz = (float)(1.0 / g->spacing[2]);
  if (p->max_step_length > 0.0) {
    double nrm = 0.0;
    for (int k = 0; k < 3; ++k) nrm += g->spacing[k] * g->spacing[k];
    nrm *= p->max_step_length * p->max_step_length / 3.0;
    K->inv_norm = (float)(1.0 / nrm);
  } else {
    K->inv_norm = 0.0f;
  }
  K->denom_thr = (float)p->denominator_threshold;
  float thr = (float)p->intensity_threshold;
  if ((double)thr < p->intensity_threshold) thr = nextafterf(thr, INFINITY);
  K->intensity_thr = thr;
}

void small_taps(const pp_taps& t, int R, pp_taps_small* s) {
  (void)R;
  for (int k = 0; k <= PP_FUSED_MAX_R; ++k) s->h[k] = k <= t.r ? t.w[t.r + k] : 0.0f;  // taps beyond the axis radius stay 0
}

// z-chunk length: long chunks amortise the 2R (+3 image) halo planes, but the launch should fill the
// chip a whole number of times.  `slots` = resident blocks of the slower kernel (256 CUs x blocks/CU).
int fused_zchunk(const pp_dims& d, int slots, int TX, int TY, double* cost_out = nullptr, char kernel = 0, int tiles_override = 0) {
  const char* e = kernel == 'A' ? pp_env("PP_FUSED_ZCHUNK_A") : (kernel == 'B' ? pp_env("PP_FUSED_ZCHUNK_B") : nullptr);
  if (!e) e = pp_env("PP_FUSED_ZCHUNK");
  if (e) {
    const int v = atoi(e);
    if (v >= 1) {
      if (cost_out) *cost_out = 0.0;
      return v < d.nz ? v : d.nz;
    }
  }
  const int tiles = tiles_override > 0 ? tiles_override : ((d.nx + TX - 1) / TX) * ((d.ny + TY - 1) / TY);
  // Small grids (the coarse pyramid levels) cannot fill the chip either way and are bound by the latency of one
  // block's march instead, ~2.5 us per plane: short chunks (down to 2 planes + halo) cut that chain.
  int best = d.nz < 32 ? d.nz : 32;
  double best_cost = 1e30;
  for (int zc = 2; zc <= 128 && zc <= d.nz; ++zc) {   // (one round of 128-plane chunks beats two rounds of 64: 512 x 512 x 256, kernel A -10 %)
    const int chunks = (d.nz + zc - 1) / zc;
    if ((chunks - 1) * zc >= d.nz) continue;
    const long blocks = (long)tiles * chunks;
    const long waves = (blocks + slots - 1) / slots;
    // time ~ (number of block waves) x (planes one block walks, halo included)
    const double cost = (double)waves * (zc + 7);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = zc;
    }
  }
  if (cost_out) *cost_out = best_cost;
  return best;
}

// Tile shape for a grid.  64 x 16 is ~4 % faster per voxel (rows of two whole cache lines), so it stays unless
// 32 x 32 needs clearly (>= 10 %) fewer plane-steps by the z-chunk model above -- grids whose 64 x 16 launch leaves
// block slots idle (341 x 341 x 171: 396 blocks on 512 slots, 11 % faster with 32 x 32) -- or, on a model tie,
// clearly fewer tiles (less overhang: 85 x 85 -> 9 tiles instead of 12, 18 % faster).
int fused_shape(const pp_dims& d, int slots0, int slots1) {
  if (const char* e = pp_env("PP_FUSED_TILE")) return atoi(e) == 1 ? 1 : 0;
  double c0 = 0.0, c1 = 0.0;
  fused_zchunk(d, slots0, tile_shape<0>::TX, tile_shape<0>::TY, &c0);
  fused_zchunk(d, slots1, tile_shape<1>::TX, tile_shape<1>::TY, &c1);
  if (c1 < 0.9 * c0) return 1;
  if (c0 < 0.9 * c1) return 0;
  const long t0 = (long)((d.nx + tile_shape<0>::TX - 1) / tile_shape<0>::TX) * ((d.ny + tile_shape<0>::TY - 1) / tile_shape<0>::TY);
  const long t1 = (long)((d.nx + tile_shape<1>::TX - 1) / tile_shape<1>::TX) * ((d.ny + tile_shape<1>::TY - 1) / tile_shape<1>::TY);
  return 10 * t1 < 9 * t0 ? 1 : 0;
}

// Per-kernel dispatch on (radius, outputs per thread).  The two kernels are independent: the update is smoothed
// with sigma_u (radius RA) and the field with sigma_d (radius RB), so each runs the narrowest template that fits.
template <int R, int OPT, int SH>
int occ_force() {
  static int cache = 0;
  if (cache) return cache;
  int a = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_fused_force_smooth<R, OPT, SH>, TILE_OUT / OPT, 0) != hipSuccess) a = 2;
  (void)hipGetLastError();
  return cache = (a < 1 ? 1 : a);
}
template <int R, int OPT, int SH>
int occ_warp() {
  static int cache = 0;
  if (cache) return cache;
  int a = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_fused_add_smooth_warp<R, OPT, SH>, TILE_OUT / OPT, 0) != hipSuccess) a = 2;
  (void)hipGetLastError();
  return cache = (a < 1 ? 1 : a);
}

// PP_MINI (measurement builds, tools/kbench/mini.sh): only the radius-2, 512-thread instances exist, so the file compiles in
// seconds while a kernel is being worked on.  Product builds never define it.
#ifdef PP_MINI
#define PP_BY_RADIUS(R, OPT, CALL) CALL(2, 2)
#else
#define PP_BY_RADIUS(R, OPT, CALL)                                                                   \
  ((OPT) == 4 ? ((R) == 1 ? CALL(1, 4) : ((R) == 2 ? CALL(2, 4) : CALL(3, 4)))                       \
              : ((R) == 1 ? CALL(1, 2) : ((R) == 2 ? CALL(2, 2) : ((R) == 3 ? CALL(3, 2) : ((R) == 4 ? CALL(4, 2) : CALL(5, 2))))))
#endif

// The 32 x 32 shape exists for the 512-thread layout only (OPT = 2).
template <int R, int OPT>
int occ_force_sh(int sh) {
  if constexpr (OPT == 2) {
    if (sh == 1) return occ_force<R, 2, 1>();
  }
  return occ_force<R, OPT, 0>();
}
template <int R, int OPT>
int occ_warp_sh(int sh) {
  if constexpr (OPT == 2) {
    if (sh == 1) return occ_warp<R, 2, 1>();
  }
  return occ_warp<R, OPT, 0>();
}

template <int R, int OPT>
int launch_force(pp_ctx* ctx, int sh, const float* F, const float* Mw_in, float* Us, const fused_args& fu, const pp_esm_consts& K,
                 double* partials, const int* halt) {
  pp_prof_scope ps(ctx, "k_fused_force_smooth");
  if constexpr (OPT == 2) {
    if (sh == 1) {
      hipLaunchKernelGGL((k_fused_force_smooth<R, 2, 1>), dim3(8u * (unsigned)fu.per_xcd), dim3(TILE_OUT / 2), 0, ctx->stream, F, Mw_in, Us,
                         fu, K, partials, halt);
      return PP_OK;
    }
  }
  hipLaunchKernelGGL((k_fused_force_smooth<R, OPT, 0>), dim3(8u * (unsigned)fu.per_xcd), dim3(TILE_OUT / OPT), 0, ctx->stream, F, Mw_in, Us,
                     fu, K, partials, halt);
  return PP_OK;
}
template <int R, int OPT>
int launch_warp(pp_ctx* ctx, int sh, const float* D, const float* Us, const float* M, float* Dn, float* Mw_out, const fused_args& fd,
                const pp_warp_scale& sc, const int* halt) {
  pp_prof_scope ps(ctx, "k_fused_add_smooth_warp");
  if constexpr (OPT == 2) {
    if (sh == 1) {
      hipLaunchKernelGGL((k_fused_add_smooth_warp<R, 2, 1>), dim3(8u * (unsigned)fd.per_xcd), dim3(TILE_OUT / 2), 0, ctx->stream, D, Us, M,
                         Dn, Mw_out, fd, sc, halt);
      return PP_OK;
    }
  }
  hipLaunchKernelGGL((k_fused_add_smooth_warp<R, OPT, 0>), dim3(8u * (unsigned)fd.per_xcd), dim3(TILE_OUT / OPT), 0, ctx->stream, D, Us, M,
                     Dn, Mw_out, fd, sc, halt);
  return PP_OK;
}

// ---- generation 2 (pp_demons_fused2.h): 512-thread layout only, both tile shapes; kernel A for radii 1..3, kernel B 1..4
// (beyond that the unrolled plane loop does not fit 128 registers and the first generation is faster).  SUM: see the header.
#ifdef PP_MINI
#define PP_BY_RADIUS_A2(R, CALL) CALL(2)
#define PP_BY_RADIUS_B2(R, CALL) CALL(2)
#else
#define PP_BY_RADIUS_A2(R, CALL) ((R) == 1 ? CALL(1) : ((R) == 2 ? CALL(2) : CALL(3)))
#define PP_BY_RADIUS_B2(R, CALL) ((R) == 1 ? CALL(1) : ((R) == 2 ? CALL(2) : ((R) == 3 ? CALL(3) : CALL(4))))
#endif
#define PP_A2_KERNEL(SHV, SUMV, NTV, MASKV) k_fused2_force_smooth<R, SHV, (R <= PP_RING_UNROLL_MAX_R), SUMV, NTV, MASKV>
#define PP_B2_KERNEL(SHV, SUMV, NTV, MASKV) k_fused2_add_smooth_warp<R, SHV, (R <= PP_RING_UNROLL_MAX_R), SUMV, NTV, MASKV>
// fields of >= 2^32 bytes (fused_args::big): SUM, streaming stores, branchy instances, 64-bit bases for the field arrays
#define PP_A2_KERNEL_BIG(SHV) k_fused2_force_smooth<R, SHV, (R <= PP_RING_UNROLL_MAX_R), true, true, false, true>
#define PP_B2_KERNEL_BIG(SHV) k_fused2_add_smooth_warp<R, SHV, (R <= PP_RING_UNROLL_MAX_R), true, true, false, true>
#ifndef PP_MINI_MASK
#define PP_MINI_MASK true
#endif

template <int R>
int occ_force2(int sh) {   // (cached: the answer depends on the kernel binary only, and the query costs ~10 us per call)
  static int cache[3] = {0, 0, 0};
  sh = sh < 0 ? 0 : (sh > 2 ? 2 : sh);
  if (cache[sh]) return cache[sh];
  int a = 0;
#ifdef PP_MINI
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_A2_KERNEL(0, true, true, PP_MINI_MASK), 512, 0);
#else
  const hipError_t e = sh == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_A2_KERNEL(2, true, false, true), 512, 0)
                       : sh  ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_A2_KERNEL(1, true, false, true), 512, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_A2_KERNEL(0, true, false, true), 512, 0);
#endif
  if (e != hipSuccess) a = 2;
  (void)hipGetLastError();
  return cache[sh] = (a < 1 ? 1 : a);
}
template <int R>
int occ_warp2(int sh) {   // (cached: the answer depends on the kernel binary only, and the query costs ~10 us per call)
  static int cache[3] = {0, 0, 0};
  sh = sh < 0 ? 0 : (sh > 2 ? 2 : sh);
  if (cache[sh]) return cache[sh];
  int a = 0;
#ifdef PP_MINI
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_B2_KERNEL(0, true, true, PP_MINI_MASK), 512, 0);
#else
  const hipError_t e = sh == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_B2_KERNEL(2, true, false, true), 512, 0)
                       : sh  ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_B2_KERNEL(1, true, false, true), 512, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, PP_B2_KERNEL(0, true, false, true), 512, 0);
#endif
  if (e != hipSuccess) a = 2;
  (void)hipGetLastError();
  return cache[sh] = (a < 1 ? 1 : a);
}
template <int R>
int launch_force2(pp_ctx* ctx, int sh, bool sum, const float* F, const float* Mw_in, const float* D, float* Us, const fused_args& fu,
                  const pp_esm_consts& K, double* partials, pp_dev_stats* st, const double* prev, int nprev, double max_rms) {
  pp_prof_scope ps(ctx, sum ? "k_fused2_force_smooth" : "k_fused2_force_smooth/sep");   // (bench.py keys its byte model on the name)
  const dim3 grid(8u * (unsigned)fu.per_xcd), block(512);
#define PP_GO(SHV, SUMV, NTV, MASKV) \
  hipLaunchKernelGGL((PP_A2_KERNEL(SHV, SUMV, NTV, MASKV)), grid, block, 0, ctx->stream, F, Mw_in, D, Us, fu, K, partials, st, prev, nprev, max_rms)
#define PP_GO_BIG(SHV) \
  hipLaunchKernelGGL((PP_A2_KERNEL_BIG(SHV)), grid, block, 0, ctx->stream, F, Mw_in, D, Us, fu, K, partials, st, prev, nprev, max_rms)
#ifdef PP_MINI
  (void)sh;
  PP_GO(0, true, true, PP_MINI_MASK);
#else
  if (fu.big) {
    if (sh == 2) PP_GO_BIG(2); else if (sh) PP_GO_BIG(1); else PP_GO_BIG(0);
  } else if (!sum) {   // (PP_FUSED_SUM=0, a measurement path: cached stores only)
    if (sh == 2) PP_GO(2, false, false, false); else if (sh) PP_GO(1, false, false, false); else PP_GO(0, false, false, false);
  } else if (fu.masked) {
    if (fu.streaming) { if (sh == 2) PP_GO(2, true, true, true); else if (sh) PP_GO(1, true, true, true); else PP_GO(0, true, true, true); }
    else { if (sh == 2) PP_GO(2, true, false, true); else if (sh) PP_GO(1, true, false, true); else PP_GO(0, true, false, true); }
  } else if (fu.streaming) {
    if (sh == 2) PP_GO(2, true, true, false); else if (sh) PP_GO(1, true, true, false); else PP_GO(0, true, true, false);
  } else {
    if (sh == 2) PP_GO(2, true, false, false); else if (sh) PP_GO(1, true, false, false); else PP_GO(0, true, false, false);
  }
#endif
#undef PP_GO
#undef PP_GO_BIG
  return PP_OK;
}
template <int R>
int launch_warp2(pp_ctx* ctx, int sh, bool sum, const float* D, const float* Us, const float* M, float* Dn, float* Mw_out,
                 const fused_args& fd, const pp_warp_scale& sc, const int* halt) {
  pp_prof_scope ps(ctx, sum ? "k_fused2_add_smooth_warp" : "k_fused2_add_smooth_warp/sep");
  const dim3 grid(8u * (unsigned)fd.per_xcd), block(512);
#define PP_GO(SHV, SUMV, NTV, MASKV) hipLaunchKernelGGL((PP_B2_KERNEL(SHV, SUMV, NTV, MASKV)), grid, block, 0, ctx->stream, D, Us, M, Dn, Mw_out, fd, sc, halt)
#define PP_GO_BIG(SHV) hipLaunchKernelGGL((PP_B2_KERNEL_BIG(SHV)), grid, block, 0, ctx->stream, D, Us, M, Dn, Mw_out, fd, sc, halt)
#ifdef PP_MINI
  (void)sh;
  PP_GO(0, true, true, PP_MINI_MASK);
#else
  if (fd.big) {
    if (sh == 2) PP_GO_BIG(2); else if (sh) PP_GO_BIG(1); else PP_GO_BIG(0);
  } else if (!sum) {
    if (sh == 2) PP_GO(2, false, false, false); else if (sh) PP_GO(1, false, false, false); else PP_GO(0, false, false, false);
  } else if (fd.masked) {
    if (fd.streaming) { if (sh == 2) PP_GO(2, true, true, true); else if (sh) PP_GO(1, true, true, true); else PP_GO(0, true, true, true); }
    else { if (sh == 2) PP_GO(2, true, false, true); else if (sh) PP_GO(1, true, false, true); else PP_GO(0, true, false, true); }
  } else if (fd.streaming) {
    if (sh == 2) PP_GO(2, true, true, false); else if (sh) PP_GO(1, true, true, false); else PP_GO(0, true, true, false);
  } else {
    if (sh == 2) PP_GO(2, true, false, false); else if (sh) PP_GO(1, true, false, false); else PP_GO(0, true, false, false);
  }
#endif
#undef PP_GO
#undef PP_GO_BIG
  return PP_OK;
}

// Generation 3 (pp_demons_cube.h): one block per 16 x 8 x 6 brick, plain 3-D grid (these levels live in the L2 / infinity cache).
template <int R>
int launch_cube_force(pp_ctx* ctx, const float* F, const float* Mw_in, const float* D, float* S, const cube_args& ca, const pp_esm_consts& K,
                      double* partials, pp_dev_stats* st, const double* prev, int nprev, double max_rms) {
  pp_prof_scope ps(ctx, "k_cube_force_smooth");
  using G = cube_geom<R>;
  const dim3 grid((unsigned)((ca.d.nx + G::TX - 1) / G::TX), (unsigned)((ca.d.ny + G::TY - 1) / G::TY), (unsigned)((ca.d.nz + G::TZ - 1) / G::TZ));
  hipLaunchKernelGGL((k_cube_force_smooth<R>), grid, dim3(G::NTH), 0, ctx->stream, F, Mw_in, D, S, ca, K, partials, st, prev, nprev, max_rms);
  return PP_OK;
}
template <int R>
int launch_cube_warp(pp_ctx* ctx, const float* S, const float* M, float* Dn, float* Mw_out, const cube_args& ca, const pp_warp_scale& sc,
                     const int* halt) {
  pp_prof_scope ps(ctx, "k_cube_smooth_warp");
  using G = cube_geom<R>;
  const dim3 grid((unsigned)((ca.d.nx + G::TX - 1) / G::TX), (unsigned)((ca.d.ny + G::TY - 1) / G::TY), (unsigned)((ca.d.nz + G::TZ - 1) / G::TZ));
  hipLaunchKernelGGL((k_cube_smooth_warp<R>), grid, dim3(G::NTH), 0, ctx->stream, S, M, Dn, Mw_out, ca, sc, halt);
  return PP_OK;
}
// Up to this many bricks the launch is one block per CU or little more and lasts ~20 us per iteration (both kernels) against the
// marching pair's 31 us floor; each further brick adds ~0.033 us (the halo's arithmetic: 3.1 x the update, 2.3 x the x pass), and
// from ~420 bricks (250-600 k voxels) the two generations trade places by shape -- 87 x 80 x 57: 37 against 44 us, 96 x 96 x 48: 35
// against 33 -- so the marching kernels keep everything beyond (profiles/round6_cube_levels.txt).  PP_FUSED_CUBE=1 / 0 forces /
// forbids the bricks wherever they apply.
constexpr size_t CUBE_MAX_BRICKS = 420;
size_t cube_bricks(const pp_dims& d) {
  using G = cube_geom<2>;   // (the brick shape is the same for every radius)
  return (size_t)((d.nx + G::TX - 1) / G::TX) * ((d.ny + G::TY - 1) / G::TY) * ((d.nz + G::TZ - 1) / G::TZ);
}

// Mixed tile shapes (generation 2, SH == 2): 64 x 16 tiles wherever a whole 64-wide tile fits and ONE column of 32 x 32 tiles
// over the rest, when that rest is at most 32 columns -- a 64-wide tile there would compute on >= 50 % overhang, and all
// 32 x 32 tiles run a 15-20 % longer plane step.  Measured against the launcher's single shape (tools/r4/run24.sh, bit-identical
// fields): 340 x 340 x 170 0.453 -> 0.414 ms per iteration, 405 x 405 x 200 0.745 -> 0.706, 341 x 341 x 171 (odd rows: the
// branchy instances) 0.466 -> 0.454, 288 x 288 x 160 0.273 -> 0.269; 85 x 85 x 43 0.033 -> 0.036 -- so only from 8 M voxels up
// (PP_FUSED_MIX=1 forces it where the shape allows, =0 switches it off).
bool fused_mix_ok(const pp_dims& d) {
  const int rest = d.nx % tile_shape<0>::TX;
  const bool shape_ok = d.nx >= tile_shape<0>::TX && rest != 0 && rest <= tile_shape<1>::TX;
  if (const char* e = pp_env("PP_FUSED_MIX")) return shape_ok && atoi(e) != 0;
  return shape_ok && (size_t)d.nx * d.ny * d.nz >= ((size_t)8 << 20);
}

void fused_grid(fused_args* f, const pp_dims& d, int occupancy, int sh, char kernel, int px = 0) {
  const int TX = sh == 1 ? tile_shape<1>::TX : tile_shape<0>::TX, TY = sh == 1 ? tile_shape<1>::TY : tile_shape<0>::TY;
  f->d = d;
  f->gx2 = f->gy2 = f->x2_off = 0;
  if (sh == 2) {
    f->gx = d.nx / tile_shape<0>::TX;
    f->gy = (d.ny + tile_shape<0>::TY - 1) / tile_shape<0>::TY;
    f->gx2 = 1;
    f->gy2 = (d.ny + tile_shape<1>::TY - 1) / tile_shape<1>::TY;
    f->x2_off = f->gx * tile_shape<0>::TX;
    f->zchunk = fused_zchunk(d, 256 * occupancy, TX, TY, nullptr, kernel, f->gx * f->gy + f->gx2 * f->gy2);
  } else {
    f->zchunk = fused_zchunk(d, 256 * occupancy, TX, TY, nullptr, kernel);
    f->gx = (d.nx + TX - 1) / TX;
    f->gy = (d.ny + TY - 1) / TY;
  }
  f->gz = (d.nz + f->zchunk - 1) / f->zchunk;
  f->per_xcd = (int)((((size_t)f->gx * f->gy + (size_t)f->gx2 * f->gy2) * f->gz + 7) / 8);
  // an iteration touches 13 volumes' worth of floats (F, M, two warped images, D, D', S: 3 each); stream the outputs once
  // that is several times the 256 MB infinity cache.  Measured (tools/kbench/run12.sh): 79 MB volumes +1.5 % slower with
  // nt stores, 113 MB -1.5 %, 180 MB -2 %, 268 MB -2.7 %.
  f->streaming = (size_t)d.nx * d.ny * d.nz * sizeof(float) > ((size_t)100 << 20);
  if (const char* e = pp_env("PP_FUSED_NT")) f->streaming = atoi(e) != 0;
  f->px = px > 0 ? px : d.nx;
  // MASK instances: pairs must exist (even pitch), offsets must fit the buffer-resource trick -- and the volume must be large
  // enough to be throughput-bound: on the small pyramid levels every lane loading on every step costs more than exact waits
  // give (128 x 128 x 64 and 64 x 64 x 32: +7 %, profiles/round4_kbench_mask.txt; 512 x 512 x 256: -2.6 %; equal at 340 x 340 x 170).
  f->big = 3 * (size_t)f->px * d.ny * d.nz * sizeof(float) >= ((size_t)1 << 32);   // (the BIG instances: pp_demons_fused2.h)
  f->masked = (f->px % 2 == 0) && 3 * (size_t)f->px * d.ny * d.nz * sizeof(float) < ((size_t)1 << 31) &&
              (size_t)d.nx * d.ny * d.nz >= ((size_t)8 << 20);
  if (const char* e = pp_env("PP_FUSED_MASK"))   // (0: the branchy kernels; 1: MASK wherever the shape allows -- A/B runs, tests)
    f->masked = atoi(e) != 0 && (f->px % 2 == 0) && 3 * (size_t)f->px * d.ny * d.nz * sizeof(float) < ((size_t)1 << 31);
  // soft synchronisation (PP_SOFTSYNC builds, MASK instances): only when every block of an XCD's run is resident at once --
  // 32 CUs x `occupancy` slots -- since a block waits for the MEAN progress of its group (pp_demons_fused2.h)
  f->sync = f->sync_other = nullptr;   // (set by the caller once the workspace is carved)
  f->sync_lag = (PP_SOFTSYNC > 0 && f->per_xcd <= 32 * occupancy) ? PP_SOFTSYNC : 0;
  if (const char* e = pp_env("PP_FUSED_SYNC")) f->sync_lag = (f->per_xcd <= 32 * occupancy) ? atoi(e) : 0;
}

int check_demons_args(pp_ctx* ctx, const pp_geom* g, const pp_demons_params* p) {
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  PP_REQUIRE(ctx, p, "demons: NULL parameters");
  if (!pp_geom_identity_dir(g))
    return pp_fail(ctx, PP_ERR_UNSUPPORTED, "demons: only identity direction cosines are supported");
  PP_REQUIRE(ctx, p->iterations >= 0, "demons: negative iteration count");
  return PP_OK;
}

int read_stats(pp_ctx* ctx, const pp_dev_stats* dst, pp_demons_stats* out) {
  pp_dev_stats h;
  const int rc = pp_read_back(ctx, dst, &h, sizeof(h));
  if (rc) return rc;
  out->metric = h.metric;
  out->rms_change = h.rms;
  out->sum_sq_diff = h.ssd;
  out->sum_sq_change = h.ssc;
  out->n_pixels = h.npx;
  out->elapsed_iterations = h.elapsed;
  out->halted = h.halt;
  return PP_OK;
}

}  // namespace

extern "C" {

int pp_demons_force_f32(pp_ctx* ctx, const float* fixed, const float* warped, const pp_geom* g, const pp_demons_params* p,
                        float* update, pp_demons_stats* stats) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && warped && update, "pp_demons_force_f32: NULL volume");
  int rc = check_demons_args(ctx, g, p);
  if (rc) return rc;
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  const size_t N = pp_nvox(g->size);
  const unsigned nb = grid_for(N, 8192);
  rc = pp_reserve(ctx, pp_align_up(3 * (size_t)nb * sizeof(double), 256) + 256);
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  double* partials = cv.take<double>(3 * (size_t)nb);
  pp_dev_stats* dst = cv.take<pp_dev_stats>(1);
  pp_esm_consts K;
  esm_consts(g, p, &K);
  PP_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(pp_dev_stats), ctx->stream));
  if (force_vec4(d, fixed, warped, update))
    hipLaunchKernelGGL(k_demons_force4, dim3(nb), dim3(NT), 0, ctx->stream, fixed, warped, update, d, K, partials, (const int*)nullptr);
  else
    hipLaunchKernelGGL(k_demons_force, dim3(nb), dim3(NT), 0, ctx->stream, fixed, warped, update, d, K, partials, (const int*)nullptr);
  PP_LAUNCH_CHECK(ctx, "k_demons_force");
  hipLaunchKernelGGL(k_demons_finalize, dim3(1), dim3(NT), 0, ctx->stream, (const double*)partials, (int)nb, dst, -1.0);
  PP_LAUNCH_CHECK(ctx, "k_demons_finalize");
  if (stats) return read_stats(ctx, dst, stats);
  return PP_OK;
}

int pp_demons_execute_f32(pp_ctx* ctx, const float* fixed, const float* moving, const pp_geom* g, const pp_demons_params* p,
                          float* field, pp_demons_stats* stats) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && moving && field, "pp_demons_execute_f32: NULL volume");
  int rc = check_demons_args(ctx, g, p);
  if (rc) return rc;
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  const size_t N = pp_nvox(g->size);

  pp_taps tu[3], td[3];
  for (int a = 0; a < 3; ++a) {
    rc = pp_make_taps(ctx, p->sigma_u_vox[a] * p->sigma_u_vox[a], p->max_error, p->max_kernel_width, &tu[a]);
    if (rc) return rc;
    rc = pp_make_taps(ctx, p->sigma_d_vox[a] * p->sigma_d_vox[a], p->max_error, p->max_kernel_width, &td[a]);
    if (rc) return rc;
  }
  int rmax = 0;
  for (int a = 0; a < 3; ++a) {
    if (tu[a].r > rmax) rmax = tu[a].r;
    if (td[a].r > rmax) rmax = td[a].r;
  }
  // the fused kernels address one z-plane with 32-bit byte offsets and store 8/16-B vectors
  const bool plane_ok = (size_t)d.nx * d.ny * sizeof(float) < ((size_t)1 << 32) && (reinterpret_cast<uintptr_t>(field) % 16 == 0);
  const bool fusable = p->smooth_update && p->smooth_displacement && rmax <= PP_FUSED_MAX_R && plane_ok;
  int variant = p->variant;
  if (variant == PP_DEMONS_AUTO) variant = fusable ? PP_DEMONS_FUSED : PP_DEMONS_STAGED;
  if (variant == PP_DEMONS_FUSED && !fusable)
    return pp_fail(ctx, PP_ERR_UNSUPPORTED,
                   "fused demons needs both smoothers on and kernel radii <= %d (got %d)", PP_FUSED_MAX_R, rmax);
  PP_REQUIRE(ctx, variant == PP_DEMONS_FUSED || variant == PP_DEMONS_STAGED, "demons: unknown variant");

  pp_esm_consts K;
  esm_consts(g, p, &K);
  const pp_warp_scale sc{(float)(1.0 / g->spacing[0]), (float)(1.0 / g->spacing[1]), (float)(1.0 / g->spacing[2])};
  const double max_rms = p->max_rms_error > 0.0 ? p->max_rms_error : -1.0;
  double* hist = nullptr;   // per-iteration {metric, RMS change}: read back by pp_demons_history
  rc = pp_history_buffer(ctx, &hist);
  if (rc) return rc;

  if (variant == PP_DEMONS_STAGED) {
    const unsigned nb = grid_for(N, 8192);
    const size_t need = pp_align_up(N * 4, 256) + 3 * pp_align_up(3 * N * 4, 256) + pp_align_up(3 * (size_t)nb * 8, 256) + 256;
    rc = pp_reserve(ctx, need);
    if (rc) return rc;
    pp_carver cv{ctx->ws, 0};
    float* Mw = cv.take<float>(N);
    float* U = cv.take<float>(3 * N);
    float* T1 = cv.take<float>(3 * N);
    float* T2 = cv.take<float>(3 * N);
    double* partials = cv.take<double>(3 * (size_t)nb);
    pp_dev_stats* dst = cv.take<pp_dev_stats>(1);
    const int* halt = &dst->halt;
    const int order[3] = {0, 1, 2};
    hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(1), 0, ctx->stream, dst, hist, ctx->hist_cap);
    PP_LAUNCH_CHECK(ctx, "k_stats_init");
    PP_HIP(ctx, hipMemsetAsync(field, 0, 3 * N * sizeof(float), ctx->stream));
    for (int it = 0; it < p->iterations; ++it) {
      {
        pp_prof_scope ps(ctx, "k_warp_same_grid");
        rc = pp_warp_same_grid(ctx, moving, field, d, sc, FLT_MAX, Mw, halt);
      }
      if (rc) return rc;
      {
        pp_prof_scope ps(ctx, "k_demons_force");
        if (force_vec4(d, fixed, Mw, U))
          hipLaunchKernelGGL(k_demons_force4, dim3(nb), dim3(NT), 0, ctx->stream, fixed, (const float*)Mw, U, d, K, partials, halt);
        else
          hipLaunchKernelGGL(k_demons_force, dim3(nb), dim3(NT), 0, ctx->stream, fixed, (const float*)Mw, U, d, K, partials, halt);
      }
      PP_LAUNCH_CHECK(ctx, "k_demons_force");
      if (p->smooth_update) {
        pp_prof_scope ps(ctx, "k_conv_axis x3 (update)");
        rc = pp_smooth3_staged(ctx, U, nullptr, U, T1, T2, d, 3, tu, order, halt);
        if (rc) return rc;
      }
      if (p->smooth_displacement) {
        pp_prof_scope ps(ctx, "k_conv_axis x3 (add+field)");
        rc = pp_smooth3_staged(ctx, field, U, field, T1, T2, d, 3, td, order, halt);  // add fused into pass 1
        if (rc) return rc;
      } else {
        hipLaunchKernelGGL(k_add_inplace, dim3(grid_for(3 * N)), dim3(NT), 0, ctx->stream, field, (const float*)U, 3 * N, halt);
        PP_LAUNCH_CHECK(ctx, "k_add_inplace");
      }
      hipLaunchKernelGGL(k_demons_finalize, dim3(1), dim3(NT), 0, ctx->stream, (const double*)partials, (int)nb, dst, max_rms);
      PP_LAUNCH_CHECK(ctx, "k_demons_finalize");
    }
    if (stats) return read_stats(ctx, dst, stats);
    return PP_OK;
  }

  // ---- fused schedule ----
  int ra = 1, rb = 1;
  for (int a = 0; a < 3; ++a) {
    if (tu[a].r > ra) ra = tu[a].r;
    if (td[a].r > rb) rb = td[a].r;
  }
  int opt = PP_FUSED_DEFAULT_OPT;
  if (const char* e = pp_env("PP_FUSED_OPT")) opt = atoi(e) == 2 ? 2 : 4;
  // kernel generation: 2 (pp_demons_fused2.h) unless the volume exceeds its 32-bit gather offsets / 24-bit row arithmetic,
  // rows are shorter than one strip, the 256-thread layout was asked for, or PP_FUSED_GEN=1 selects the first generation
  // (kept for A/B measurements).
  // (a scalar image under one buffer resource's 32-bit offsets; a 3-component field beyond that takes the BIG instances)
  const bool gen2_ok = N * sizeof(float) < ((size_t)1 << 32) &&
                       (size_t)d.ny * d.nz < ((size_t)1 << 24) && d.nx >= 4 && d.nx < (1 << 22) &&
                       d.ny < (1 << 22) && d.nz < (1 << 22);   // (every axis below the warp's 2^23-voxel displacement clamp)
  int gen = (gen2_ok && opt == 2) ? 2 : 1;
  if (const char* e = pp_env("PP_FUSED_GEN")) {
    if (atoi(e) == 1) gen = 1;
  }
  // (the second generation needs > 128 registers for kernel A from radius 4 -- sigma_u = 1 voxel gives radius 2 -- and for
  // kernel B at radius 5, i.e. voxels under 0.55 mm: measured slower than the first generation there)
  int gen_a = (gen == 2 && ra <= 3) ? 2 : 1, gen_b = (gen == 2 && rb <= 4) ? 2 : 1;
  // both kernels of generation 2: kernel A stores D + G_u * update and kernel B reads that one volume (three halo'd arrays
  // instead of six; the add itself is unchanged, so the fields are bit-identical).  PP_FUSED_SUM=0 keeps them apart.
  bool sum_mode = gen_a == 2 && gen_b == 2;
  if (const char* e = pp_env("PP_FUSED_SUM")) {
    if (atoi(e) == 0) sum_mode = false;
  }
  // fields of >= 2^32 bytes: generation 2 serves them through its BIG instances, which exist for the SUM pair only
  if (3 * N * sizeof(float) >= ((size_t)1 << 32) && !sum_mode) gen_a = gen_b = 1;
  const int opt_a = ra > 3 ? 2 : opt, opt_b = rb > 3 ? 2 : opt;   // radii 4 and 5 exist in the 512-thread layout only
  fused_args fu, fd;
  int sh_a = 0, sh_b = 0;
  // Padded rows (fused_args::px) when both kernels are generation 2 in SUM mode and the rows are not whole 16-byte quads.
  // Measured (tools/r4/run26.sh, bit-identical fields): 341 x 341 x 171 0.451 -> 0.420 ms per iteration, 405 x 405 x 200 0.704 ->
  // 0.655; 171 x 171 x 85 0.073 -> 0.082 and 85 x 85 x 43 0.033 -> 0.037 (latency-bound levels: the MASK instances' unconditional
  // loads and the two copies cost more than alignment gives) -- so from 8 M voxels up; PP_FUSED_PITCH=1 / 0 forces / forbids it.
  bool pitched = gen_a == 2 && gen_b == 2 && sum_mode && p->iterations > 0 && d.nx % 4 != 0;
  if (const char* e = pp_env("PP_FUSED_PITCH")) pitched = pitched && atoi(e) != 0;
  else pitched = pitched && N >= ((size_t)8 << 20);
  // (the padded pitch enlarges the component stride: the 32-bit offsets and 24-bit row products that gen2_ok checked on the
  // dense volume must also hold on the padded one -- e.g. 709 x 709 x 711 passes dense and wraps at px = 712 -- else dense rows)
  if (pitched) {
    const size_t pxc = (size_t)(d.nx + 3) / 4 * 4;
    if (!(pxc * d.ny * d.nz * sizeof(float) < ((size_t)1 << 32) && pxc * sizeof(float) < ((size_t)1 << 24))) pitched = false;
  }
  const int px = pitched ? (d.nx + 3) / 4 * 4 : d.nx;
  const size_t Np = (size_t)px * d.ny * d.nz;
  // generation 3 for the latency-bound levels: the SUM pair's protocol, radii <= 2 (update) / 3 (field), dense rows
  bool cube = gen_a == 2 && gen_b == 2 && sum_mode && !pitched && ra <= 2 && rb <= 3 && p->iterations > 0;
  if (const char* e = pp_env("PP_FUSED_CUBE")) {
    cube = cube && atoi(e) != 0;
  } else {
    // (a switch addressed to the marching kernels selects them: A/B runs and the tests of their variants keep their meaning)
    for (const char* name : {"PP_FUSED_GEN", "PP_FUSED_TILE", "PP_FUSED_MASK", "PP_FUSED_MIX", "PP_FUSED_NT", "PP_FUSED_OPT", "PP_FUSED_PITCH",
                             "PP_FUSED_SYNC", "PP_FUSED_ZCHUNK", "PP_FUSED_ZCHUNK_A", "PP_FUSED_ZCHUNK_B"})
      if (pp_env(name)) cube = false;
    cube = cube && cube_bricks(d) <= CUBE_MAX_BRICKS;
  }
  cube_args cu, cd;
  cu.d = cd.d = d;
  // tile shape per kernel: 32 x 32 where the z-chunk model says the 64 x 16 launch wastes >= 10 % (512-thread layouts only)
  if (gen_a == 2) {
#define PP_OCC_A20(RR) occ_force2<RR>(0)
#define PP_OCC_A21(RR) occ_force2<RR>(1)
    const int occ_a0 = PP_BY_RADIUS_A2(ra, PP_OCC_A20), occ_a1 = PP_BY_RADIUS_A2(ra, PP_OCC_A21);
#undef PP_OCC_A20
#undef PP_OCC_A21
    sh_a = fused_shape(d, 256 * occ_a0, 256 * occ_a1);
#define PP_OCC_A22(RR) occ_force2<RR>(2)
    if (fused_mix_ok(d) && !pp_env("PP_FUSED_TILE")) {
      sh_a = 2;
      fused_grid(&fu, d, PP_BY_RADIUS_A2(ra, PP_OCC_A22), 2, 'A', px);
    } else {
      fused_grid(&fu, d, sh_a ? occ_a1 : occ_a0, sh_a, 'A', px);
    }
#undef PP_OCC_A22
  } else {
#define PP_OCC_A0(RR, OO) occ_force_sh<RR, OO>(0)
#define PP_OCC_A1(RR, OO) occ_force_sh<RR, OO>(1)
    const int occ_a0 = PP_BY_RADIUS(ra, opt_a, PP_OCC_A0);
    const int occ_a1 = opt_a == 2 ? PP_BY_RADIUS(ra, opt_a, PP_OCC_A1) : occ_a0;
#undef PP_OCC_A0
#undef PP_OCC_A1
    sh_a = opt_a == 2 ? fused_shape(d, 256 * occ_a0, 256 * occ_a1) : 0;
    fused_grid(&fu, d, sh_a ? occ_a1 : occ_a0, sh_a, 'A');
  }
  if (gen_b == 2) {
#define PP_OCC_B20(RR) occ_warp2<RR>(0)
#define PP_OCC_B21(RR) occ_warp2<RR>(1)
    const int occ_b0 = PP_BY_RADIUS_B2(rb, PP_OCC_B20), occ_b1 = PP_BY_RADIUS_B2(rb, PP_OCC_B21);
#undef PP_OCC_B20
#undef PP_OCC_B21
    sh_b = fused_shape(d, 256 * occ_b0, 256 * occ_b1);
#define PP_OCC_B22(RR) occ_warp2<RR>(2)
    if (fused_mix_ok(d) && !pp_env("PP_FUSED_TILE")) {
      sh_b = 2;
      fused_grid(&fd, d, PP_BY_RADIUS_B2(rb, PP_OCC_B22), 2, 'B', px);
    } else {
      fused_grid(&fd, d, sh_b ? occ_b1 : occ_b0, sh_b, 'B', px);
    }
#undef PP_OCC_B22
  } else {
#define PP_OCC_B0(RR, OO) occ_warp_sh<RR, OO>(0)
#define PP_OCC_B1(RR, OO) occ_warp_sh<RR, OO>(1)
    const int occ_b0 = PP_BY_RADIUS(rb, opt_b, PP_OCC_B0);
    const int occ_b1 = opt_b == 2 ? PP_BY_RADIUS(rb, opt_b, PP_OCC_B1) : occ_b0;
#undef PP_OCC_B0
#undef PP_OCC_B1
    sh_b = opt_b == 2 ? fused_shape(d, 256 * occ_b0, 256 * occ_b1) : 0;
    fused_grid(&fd, d, sh_b ? occ_b1 : occ_b0, sh_b, 'B');
  }
  small_taps(tu[0], ra, &fu.wx);
  small_taps(tu[1], ra, &fu.wy);
  small_taps(tu[2], ra, &fu.wz);
  small_taps(td[0], rb, &fd.wx);
  small_taps(td[1], rb, &fd.wy);
  small_taps(td[2], rb, &fd.wz);
  cu.wx = fu.wx; cu.wy = fu.wy; cu.wz = fu.wz;
  cd.wx = fd.wx; cd.wy = fd.wy; cd.wz = fd.wz;
  const size_t nblk = cube ? cube_bricks(d) : ((size_t)fu.gx * fu.gy + (size_t)fu.gx2 * fu.gy2) * fu.gz;
  constexpr size_t SYNC_WORDS = 8 * 64;   // per kernel: one word per resident block of each XCD (PP_SYNC_GROUP)
  const size_t need = (pitched ? 4 : 2) * pp_align_up(Np * 4, 256) + (pitched ? 3 : 2) * pp_align_up(3 * Np * 4, 256) +
                      2 * pp_align_up(3 * nblk * 8, 256) + 256 + pp_align_up(2 * SYNC_WORDS * sizeof(unsigned), 256);
  rc = pp_reserve(ctx, need);
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* MwA = cv.take<float>(Np);
  float* MwB = cv.take<float>(Np);
  float* Us = cv.take<float>(3 * Np);
  float* D2 = cv.take<float>(3 * Np);
  float* D1 = field;   // the other field buffer: the caller's (dense rows) or a padded one of this call's
  if (pitched) {
    D1 = cv.take<float>(3 * Np);
    float* Fp = cv.take<float>(Np);
    float* Mp = cv.take<float>(Np);
    const size_t rows = (size_t)d.ny * d.nz;
    hipLaunchKernelGGL(k_pad_rows, dim3(grid_for(rows * (px / 4), 16384)), dim3(NT), 0, ctx->stream, fixed, Fp, d.nx, px, rows);
    hipLaunchKernelGGL(k_pad_rows, dim3(grid_for(rows * (px / 4), 16384)), dim3(NT), 0, ctx->stream, moving, Mp, d.nx, px, rows);
    PP_LAUNCH_CHECK(ctx, "k_pad_rows");
    fixed = Fp;
    moving = Mp;
  }
  double* partials = cv.take<double>(3 * nblk);
  double* partials2 = cv.take<double>(3 * nblk);   // generation-2 kernel A alternates: it folds the previous launch's sums itself
  pp_dev_stats* dst = cv.take<pp_dev_stats>(1);
  const int* halt = &dst->halt;
  {   // progress counters of the soft synchronisation: kernel A's set, kernel B's set (each launch clears the other's)
    unsigned* const sync_words = cv.take<unsigned>(2 * SYNC_WORDS);
    fu.sync = sync_words;
    fu.sync_other = sync_words + SYNC_WORDS;
    fd.sync = sync_words + SYNC_WORDS;
    fd.sync_other = sync_words;
    // (only the MASK instances publish progress: the latency-bound levels do not pay for the memset)
    if ((PP_SOFTSYNC > 0 || PP_PAIRPRIO > 0) && (fu.masked || fd.masked))
      PP_HIP(ctx, hipMemsetAsync(sync_words, 0, 2 * SYNC_WORDS * sizeof(unsigned), ctx->stream));
  }
  hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(1), 0, ctx->stream, dst, hist, ctx->hist_cap);
  PP_LAUNCH_CHECK(ctx, "k_stats_init");
  // D = 0 at the start.  With both generation-2 kernels in SUM mode nothing reads the field's buffer before it is written:
  // kernel A of iteration 0 replaces what it loads by zeros (nprev == 0), kernel B reads A's output only -- so the buffer
  // is cleared only when something else will read it (0 iterations, the first generation, PP_FUSED_SUM=0).
  // The skipped clear rests on one invariant: iteration 0's kernel B always runs (the halt flag can only be raised by the fold at
  // the head of iteration 1's kernel A), so the buffer is written before anything returns it.  It is worth keeping only where it
  // costs something -- an 805 MB memset at 512 x 512 x 256 -- so the latency-bound levels below 8 M voxels clear anyway (a few
  // microseconds) and cannot return uninitialised memory whatever a later change does to the halt logic.
  const bool skip_clear = gen_a == 2 && gen_b == 2 && sum_mode && p->iterations > 0 && N >= ((size_t)8 << 20);
  if (!skip_clear) PP_HIP(ctx, hipMemsetAsync(field, 0, 3 * N * sizeof(float), ctx->stream));
  for (int it = 0; it < p->iterations; ++it) {
    // D = 0 warps the moving image onto itself exactly, so iteration 0 reads it directly.
    const float* mw_in = it == 0 ? moving : ((it & 1) ? MwA : MwB);
    float* mw_out = (it & 1) ? MwB : MwA;
    const float* Dcur = (it & 1) ? D2 : D1;
    float* Dnext = (it & 1) ? D1 : D2;
    // a failed first launch must not be masked by the second one's status
    if (cube) {
      double* const pcur = (it & 1) ? partials2 : partials;
      const double* const pprev = (it & 1) ? partials : partials2;
      const int np = it > 0 ? (int)nblk : 0;
      rc = ra == 1 ? launch_cube_force<1>(ctx, fixed, mw_in, Dcur, Us, cu, K, pcur, dst, pprev, np, max_rms)
                   : launch_cube_force<2>(ctx, fixed, mw_in, Dcur, Us, cu, K, pcur, dst, pprev, np, max_rms);
    } else if (gen_a == 2) {
      double* const pcur = (it & 1) ? partials2 : partials;
      const double* const pprev = (it & 1) ? partials : partials2;
#define PP_CALL_A2(RR) launch_force2<RR>(ctx, sh_a, sum_mode, fixed, mw_in, Dcur, Us, fu, K, pcur, dst, pprev, it > 0 ? (int)nblk : 0, max_rms)
      rc = PP_BY_RADIUS_A2(ra, PP_CALL_A2);
#undef PP_CALL_A2
    } else {
#define PP_CALL_A(RR, OO) launch_force<RR, OO>(ctx, sh_a, fixed, mw_in, Us, fu, K, partials, halt)
      rc = PP_BY_RADIUS(ra, opt_a, PP_CALL_A);
#undef PP_CALL_A
    }
    PP_LAUNCH_CHECK(ctx, "k_fused_force_smooth");
    if (rc) return rc;
    if (cube) {
      rc = rb == 1   ? launch_cube_warp<1>(ctx, (const float*)Us, moving, Dnext, mw_out, cd, sc, halt)
           : rb == 2 ? launch_cube_warp<2>(ctx, (const float*)Us, moving, Dnext, mw_out, cd, sc, halt)
                     : launch_cube_warp<3>(ctx, (const float*)Us, moving, Dnext, mw_out, cd, sc, halt);
    } else if (gen_b == 2) {
#define PP_CALL_B2(RR) launch_warp2<RR>(ctx, sh_b, sum_mode, Dcur, (const float*)Us, moving, Dnext, mw_out, fd, sc, halt)
      rc = PP_BY_RADIUS_B2(rb, PP_CALL_B2);
#undef PP_CALL_B2
    } else {
#define PP_CALL_B(RR, OO) launch_warp<RR, OO>(ctx, sh_b, Dcur, (const float*)Us, moving, Dnext, mw_out, fd, sc, halt)
      rc = PP_BY_RADIUS(rb, opt_b, PP_CALL_B);
#undef PP_CALL_B
    }
    PP_LAUNCH_CHECK(ctx, "k_fused_add_smooth_warp");
    if (rc) return rc;
    // end of the iteration: generation-2 kernel A does it at the start of the next launch, the last iteration's here
    if (gen_a != 2 || it == p->iterations - 1) {
      const double* const pfin = (gen_a == 2 && (it & 1)) ? partials2 : partials;
      hipLaunchKernelGGL(k_demons_finalize, dim3(1), dim3(NT), 0, ctx->stream, pfin, (int)nblk, dst, max_rms);
      PP_LAUNCH_CHECK(ctx, "k_demons_finalize");
    }
  }
  // (a grid-stride copy: with an even iteration count -- the usual case -- every block returns at once, and a quarter of a
  // million blocks doing so took 22 us)
  if (pitched) {   // the newest field (D1 after an even number of iterations, D2 after an odd one) back into dense rows
    hipLaunchKernelGGL(k_unpad_field, dim3(grid_for(3 * N, 16384)), dim3(NT), 0, ctx->stream, field, (const float*)D1, (const float*)D2,
                       d.nx, px, (size_t)d.ny * d.nz, (const pp_dev_stats*)dst);
    PP_LAUNCH_CHECK(ctx, "k_unpad_field");
  } else {
    hipLaunchKernelGGL(k_copy_if_odd, dim3(grid_for(3 * N, 8192)), dim3(NT), 0, ctx->stream, field, (const float*)D2, 3 * N,
                       (const pp_dev_stats*)dst);
    PP_LAUNCH_CHECK(ctx, "k_copy_if_odd");
  }
  if (stats) return read_stats(ctx, dst, stats);
  return PP_OK;
}

int pp_demons_history(pp_ctx* ctx, double* metric, double* rms_change, int cap) {
  if (!ctx || cap < 0 || (cap > 0 && (!metric || !rms_change))) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  if (!ctx->hist) return 0;   // no Execute has run on this context
  double head[2];
  int rc = pp_read_back(ctx, ctx->hist, head, sizeof(head));
  if (rc) return rc;
  const int n = (int)head[0];                 // entries recorded (<= PP_HIST_CAP)
  const int ran = (int)head[1];               // iterations that ran: the return value, also when the ring was too short
  const int take = n < cap ? n : cap;
  constexpr int CH = 4096 / (2 * (int)sizeof(double));   // entries per staged read-back
  double buf[2 * CH];
  for (int k = 0; k < take; k += CH) {
    const int m = take - k < CH ? take - k : CH;
    rc = pp_read_back(ctx, ctx->hist + 2 + 2 * (size_t)k, buf, 2 * sizeof(double) * (size_t)m);
    if (rc) return rc;
    for (int i = 0; i < m; ++i) {
      metric[k + i] = buf[2 * i];
      rms_change[k + i] = buf[2 * i + 1];
    }
  }
  return ran > n ? ran : n;
}

#ifdef PP_DRIFT
int pp_debug_drift_read(unsigned long long* out, int cap) {
  const int n = 2 * 1024 * 4;
  if (cap < n) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_drift_buf), sizeof(unsigned long long) * n) != hipSuccess) return -2;
  return n;
}
int pp_debug_drift_xcc_read(unsigned* out, int cap) {
  const int n = 2 * 1024;
  if (cap < n) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_drift_xcc), sizeof(unsigned) * n) != hipSuccess) return -2;
  return n;
}
#endif

#ifdef PP_TRACE
int pp_debug_trace_read(unsigned* out, int cap) {
  const int n = 2 * 8 * PP_TRACE_STEPS * PP_TRACE_SLOTS;
  if (cap < n) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_trace_buf), sizeof(unsigned) * n) != hipSuccess) return -2;
  return n;
}
#endif

}  // extern "C"
