// platipy_amd/csrc/pp_dist.hip -- exact Euclidean distance map of a binary volume and label contours.
//
// Replaces sitk.SignedMaurerDistanceMap(mask, squaredDistance=False, useImageSpacing=True) and
// sitk.LabelContour for iterative atlas removal (reference: platipy/imaging/label/projection.py:80-90,
// called from label/iar.py:175).  Semantics of ITK's Maurer filter: the zero set is the object's border
// voxels -- object voxels with a background voxel in their 26-neighbourhood -- and every other voxel gets
// the Euclidean distance (mm) from its centre to the nearest border voxel centre, negative inside the
// object (insideIsPositive = False).
//
// Exact separable EDT: a two-sweep 1-D pass along z, then the lower envelope of parabolas (Felzenszwalb &
// Huttenlocher) along y and along x.  One thread owns one line; lanes sit on consecutive x so every access
// is coalesced -- for lines ALONG x the volume is transposed (x <-> y) through LDS tiles first.  The
// envelope's per-line stacks live in HBM scratch, interleaved across lanes.  Runs once per atlas, not in
// the registration loop.
#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr int NT = 256;
constexpr float BIG = 1.0e18f;  // "no border voxel on this line yet"

unsigned grid_for(size_t work, unsigned cap = 65535u) {
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// border[i] = 1 where mask != 0 and some voxel of the 26-neighbourhood (inside the volume) is background.
__global__ void __launch_bounds__(NT) k_border26(const uint8_t* __restrict__ mask, uint8_t* __restrict__ border, pp_dims d) {
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    uint8_t b = 0;
    if (mask[i]) {
      const int x = (int)(i % d.nx), y = (int)((i / d.nx) % d.ny), z = (int)(i / ((size_t)d.nx * d.ny));
      for (int dz = -1; dz <= 1 && !b; ++dz)
        for (int dy = -1; dy <= 1 && !b; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx, yy = y + dy, zz = z + dz;
            if (xx < 0 || yy < 0 || zz < 0 || xx >= d.nx || yy >= d.ny || zz >= d.nz) continue;
            if (!mask[((size_t)zz * d.ny + yy) * d.nx + xx]) { b = 1; break; }
          }
    }
    border[i] = b;
  }
}

// sitk.LabelContour (fullyConnected = False) on a binary label: object voxels with a face neighbour of a
// different value; voxels outside the image are not neighbours.
__global__ void __launch_bounds__(NT) k_contour6(const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, pp_dims d) {
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  const size_t sy = d.nx, sz = (size_t)d.nx * d.ny;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    uint8_t c = 0;
    if (mask[i]) {
      const int x = (int)(i % d.nx), y = (int)((i / d.nx) % d.ny), z = (int)(i / sz);
      c = (x > 0 && !mask[i - 1]) || (x < d.nx - 1 && !mask[i + 1]) || (y > 0 && !mask[i - sy]) ||
          (y < d.ny - 1 && !mask[i + sy]) || (z > 0 && !mask[i - sz]) || (z < d.nz - 1 && !mask[i + sz]);
    }
    out[i] = c;
  }
}

// Pass 1: squared distance (mm^2) to the nearest border voxel along z, per (x, y) column.
__global__ void __launch_bounds__(NT) k_edt_z(const uint8_t* __restrict__ border, float* __restrict__ d2, pp_dims d, float sz_mm) {
  const size_t ncol = (size_t)d.nx * d.ny;
  for (size_t c = (size_t)blockIdx.x * NT + threadIdx.x; c < ncol; c += (size_t)gridDim.x * NT) {
    float run = BIG;  // distance (mm) to the last border voxel seen
    for (int z = 0; z < d.nz; ++z) {
      const size_t i = (size_t)z * ncol + c;
      run = border[i] ? 0.0f : (run < BIG ? run + sz_mm : BIG);
      d2[i] = run;
    }
    run = BIG;
    for (int z = d.nz - 1; z >= 0; --z) {
      const size_t i = (size_t)z * ncol + c;
      run = border[i] ? 0.0f : (run < BIG ? run + sz_mm : BIG);
      const float m = fminf(d2[i], run);
      d2[i] = m < BIG ? m * m : BIG;
    }
  }
}

// Lower envelope of parabolas along the middle axis of a [nouter][len][ninner] array (lanes on `inner`):
// out(q) = min_p f(p) + ((q - p) * step)^2.  v / zb: per-line stacks, laid out like the data.
__global__ void __launch_bounds__(NT) k_edt_envelope(const float* __restrict__ f, float* __restrict__ out, int* __restrict__ v,
                                                     float* __restrict__ zb, int ninner, int len, int nouter, float step) {
  const size_t nlines = (size_t)ninner * nouter;
  const float s2 = step * step;
  for (size_t l = (size_t)blockIdx.x * NT + threadIdx.x; l < nlines; l += (size_t)gridDim.x * NT) {
    const size_t inner = l % ninner, outer = l / ninner;
    const size_t base = outer * (size_t)len * ninner + inner;
#define AT(q) (base + (size_t)(q) * ninner)
    int k = 0;
    v[AT(0)] = 0;
    zb[AT(0)] = -BIG;
    for (int q = 1; q < len; ++q) {
      const float fq = f[AT(q)];
      float s;
      for (;;) {
        const int p = v[AT(k)];
        // intersection (in index units) of the parabolas rooted at p and q
        s = (float)((((double)fq + (double)s2 * q * q) - ((double)f[AT(p)] + (double)s2 * p * p)) / (2.0 * (double)s2 * (double)(q - p)));
        if (k > 0 && s <= zb[AT(k)]) --k;
        else break;
      }
      ++k;
      v[AT(k)] = q;
      zb[AT(k)] = s;
    }
    int j = 0;
    for (int q = 0; q < len; ++q) {
      while (j < k && zb[AT(j + 1)] < (float)q) ++j;
      const int p = v[AT(j)];
      const float dq = (float)(q - p) * step;
      out[AT(q)] = fminf(f[AT(p)] + dq * dq, BIG);
    }
#undef AT
  }
}

// [Z][Y][X] -> [Z][X][Y] (and back with nx, ny swapped) through a padded LDS tile.
__global__ void __launch_bounds__(NT) k_transpose_xy(const float* __restrict__ in, float* __restrict__ out, int nx, int ny, int nz) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int z = blockIdx.z; z < nz; z += gridDim.z) {
    const float* src = in + (size_t)z * nx * ny;
    float* dst = out + (size_t)z * nx * ny;
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
      if (bx + tx < nx && by + j < ny) tile[j][tx] = src[(size_t)(by + j) * nx + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
      if (by + tx < ny && bx + j < nx) dst[(size_t)(bx + j) * ny + by + tx] = tile[tx][j];
  }
}

// sqrt, and the sign of ITK's signed map (negative inside the object unless inside_positive).
__global__ void __launch_bounds__(NT) k_edt_finish(const float* __restrict__ d2, const uint8_t* __restrict__ mask,
                                                   float* __restrict__ out, size_t n, int want_signed, int inside_positive) {
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    float r = sqrtf(d2[i]);
    if (want_signed && ((mask[i] != 0) != (inside_positive != 0))) r = -r;
    out[i] = r;
  }
}

}  // namespace

extern "C" {

int pp_label_contour_u8(pp_ctx* ctx, const uint8_t* mask, const int size[3], uint8_t* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, mask && size && out && mask != out, "pp_label_contour_u8: NULL or aliased argument");
  const pp_dims d{size[0], size[1], size[2]};
  hipLaunchKernelGGL(k_contour6, dim3(grid_for(pp_nvox(size))), dim3(NT), 0, ctx->stream, mask, out, d);
  PP_LAUNCH_CHECK(ctx, "k_contour6");
  return PP_OK;
}

int pp_distance_map_f32(pp_ctx* ctx, const uint8_t* mask, const pp_geom* g, int want_signed, int inside_positive, float* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, mask && out, "pp_distance_map_f32: NULL argument");
  int rc = pp_geom_check(ctx, g, "grid");
  if (rc) return rc;
  const pp_dims d{g->size[0], g->size[1], g->size[2]};
  const size_t n = pp_nvox(g->size);
  rc = pp_reserve(ctx, pp_align_up(n, 256) + 4 * pp_align_up(n * 4, 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  uint8_t* border = cv.take<uint8_t>(n);
  float* a = cv.take<float>(n);
  float* b = cv.take<float>(n);
  int* v = cv.take<int>(n);
  float* zb = cv.take<float>(n);
  const dim3 blk(NT);
  hipLaunchKernelGGL(k_border26, dim3(grid_for(n)), blk, 0, ctx->stream, mask, border, d);
  PP_LAUNCH_CHECK(ctx, "k_border26");
  hipLaunchKernelGGL(k_edt_z, dim3(grid_for((size_t)d.nx * d.ny)), blk, 0, ctx->stream, (const uint8_t*)border, a, d, (float)g->spacing[2]);
  PP_LAUNCH_CHECK(ctx, "k_edt_z");
  // along y in [Z][Y][X]: inner = x, len = ny, outer = z
  hipLaunchKernelGGL(k_edt_envelope, dim3(grid_for((size_t)d.nx * d.nz)), blk, 0, ctx->stream, (const float*)a, b, v, zb, d.nx, d.ny, d.nz,
                     (float)g->spacing[1]);
  PP_LAUNCH_CHECK(ctx, "k_edt_envelope(y)");
  // along x: transpose to [Z][X][Y], envelope along the middle axis (inner = y, len = nx), transpose back
  const dim3 tg((d.nx + 31) / 32, (d.ny + 31) / 32, d.nz < 1024 ? d.nz : 1024);
  hipLaunchKernelGGL(k_transpose_xy, tg, blk, 0, ctx->stream, (const float*)b, a, d.nx, d.ny, d.nz);
  PP_LAUNCH_CHECK(ctx, "k_transpose_xy");
  hipLaunchKernelGGL(k_edt_envelope, dim3(grid_for((size_t)d.ny * d.nz)), blk, 0, ctx->stream, (const float*)a, b, v, zb, d.ny, d.nx, d.nz,
                     (float)g->spacing[0]);
  PP_LAUNCH_CHECK(ctx, "k_edt_envelope(x)");
  const dim3 tg2((d.ny + 31) / 32, (d.nx + 31) / 32, d.nz < 1024 ? d.nz : 1024);
  hipLaunchKernelGGL(k_transpose_xy, tg2, blk, 0, ctx->stream, (const float*)b, a, d.ny, d.nx, d.nz);
  PP_LAUNCH_CHECK(ctx, "k_transpose_xy");
  hipLaunchKernelGGL(k_edt_finish, dim3(grid_for(n)), blk, 0, ctx->stream, (const float*)a, mask, out, n, want_signed, inside_positive);
  PP_LAUNCH_CHECK(ctx, "k_edt_finish");
  return PP_OK;
}

}  // extern "C"
