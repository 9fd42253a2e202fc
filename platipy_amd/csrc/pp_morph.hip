// platipy_amd/csrc/pp_morph.hip -- binary dilation / erosion / closing with ITK's "ball" structuring element.
//
// Replaces sitk.BinaryDilate(mask, radius) in convert_mask_to_reg_structure (reference:
// platipy/imaging/registration/utils.py:328-329) and sitk.BinaryMorphologicalClosing(mask, radius) in the
// pipelines' post-processing (platipy/imaging/projects/multiatlas/run.py:421-423,
// projects/cardiac/run.py:1127-1129), both with SimpleITK's defaults: kernelType = sitkBall,
// foregroundValue = 1, dilate boundaryToForeground = False, erode boundaryToForeground = True, closing
// safeBorder = True.
//
// The ball [ITK-upstream FlatStructuringElement::Ball, radiusIsParametric = false]: offset (dx, dy, dz) belongs
// to the element when sum_i (d_i / (r_i + 0.5))^2 <= 1 (an ellipsoid with axes 2 r_i + 1 centred on the centre
// voxel, pixel-centre inclusion).  Each (dy, dz) row of it is one contiguous span |dx| <= ex, so the element
// travels as a table of half-widths by value in the kernel arguments (<= 31 x 31 bytes) and a voxel's test is a
// handful of short byte-row scans with early exit: HBM/L2-bound byte work, nothing to tile.
#include <algorithm>

#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_R = 15;

struct morph_se {
  int rx, ry, rz;
  signed char ex[(2 * MAX_R + 1) * (2 * MAX_R + 1)];  // [(dz + rz) * (2 ry + 1) + (dy + ry)]; -1 = empty row
};

struct morph_off {
  int x, y, z;  // output voxel p reads input voxel p + off
};

// ERODE = 0: out = 1 where any element neighbour is foreground (outside the buffer = background).
// ERODE = 1: out = 1 where no element neighbour is background (outside the buffer = foreground).
template <int ERODE>
__global__ void __launch_bounds__(NT) k_morph_ball(const uint8_t* __restrict__ in, pp_dims din, uint8_t* __restrict__ out, pp_dims dout,
                                                   morph_off off, morph_se se) {
  const size_t n = (size_t)dout.nx * dout.ny * dout.nz;
  const int wy = 2 * se.ry + 1;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const int x = (int)(i % dout.nx) + off.x;
    const int y = (int)((i / dout.nx) % dout.ny) + off.y;
    const int z = (int)(i / ((size_t)dout.nx * dout.ny)) + off.z;
    bool hit = false;
    for (int dz = -se.rz; dz <= se.rz && !hit; ++dz) {
      const int zz = z + dz;
      if (zz < 0 || zz >= din.nz) continue;
      for (int dy = -se.ry; dy <= se.ry && !hit; ++dy) {
        const int e = se.ex[(dz + se.rz) * wy + (dy + se.ry)];
        const int yy = y + dy;
        if (e < 0 || yy < 0 || yy >= din.ny) continue;
        const uint8_t* row = in + ((size_t)zz * din.ny + yy) * din.nx;
        const int x0 = x - e < 0 ? 0 : x - e;
        const int x1 = x + e >= din.nx ? din.nx - 1 : x + e;
        for (int xx = x0; xx <= x1; ++xx) {
          const bool fg = row[xx] != 0;
          if (ERODE ? !fg : fg) {
            hit = true;
            break;
          }
        }
      }
    }
    out[i] = ERODE ? (hit ? 0 : 1) : (hit ? 1 : 0);
  }
}

// Bounding box of the voxels > 0 (label_to_roi, platipy/imaging/utils/crop.py:24-60): one streaming pass, a wave
// per (z, y) row, per-block LDS min/max, six global atomics per block.  box = {xmin, xmax, ymin, ymax, zmin, zmax}.
template <typename T>
__global__ void __launch_bounds__(NT) k_bounding_box(const T* __restrict__ v, pp_dims d, int* __restrict__ box) {
  __shared__ int sb[6];
  if (threadIdx.x < 6) sb[threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t rows = (size_t)d.ny * d.nz;
  int xlo = 0x7fffffff, xhi = -1, ylo = 0x7fffffff, yhi = -1, zlo = 0x7fffffff, zhi = -1;
  for (size_t r = (size_t)blockIdx.x * (NT / 64) + wave; r < rows; r += (size_t)gridDim.x * (NT / 64)) {
    const T* row = v + r * d.nx;
    const int y = (int)(r % d.ny), z = (int)(r / d.ny);
    bool any = false;
    for (int x = lane; x < d.nx; x += 64)
      if (row[x] > (T)0) {
        any = true;
        xlo = x < xlo ? x : xlo;
        xhi = x > xhi ? x : xhi;
      }
    if (any) {
      ylo = y < ylo ? y : ylo;
      yhi = y > yhi ? y : yhi;
      zlo = z < zlo ? z : zlo;
      zhi = z > zhi ? z : zhi;
    }
  }
  if (xhi >= 0) {
    atomicMin(&sb[0], xlo);
    atomicMax(&sb[1], xhi);
    atomicMin(&sb[2], ylo);
    atomicMax(&sb[3], yhi);
    atomicMin(&sb[4], zlo);
    atomicMax(&sb[5], zhi);
  }
  __syncthreads();
  if (threadIdx.x < 6 && sb[1] >= 0) {
    if (threadIdx.x & 1) atomicMax(&box[threadIdx.x], sb[threadIdx.x]);
    else atomicMin(&box[threadIdx.x], sb[threadIdx.x]);
  }
}

__global__ void k_bounding_box_init(int* box) {
  if (threadIdx.x < 6) box[threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
}

unsigned grid_for(size_t work) {
  const size_t b = (work + NT - 1) / NT;
  return (unsigned)(b < 1 ? 1 : (b > 65535u * 4u ? 65535u * 4u : b));
}

// Half-widths of the ITK ball, evaluated the way EllipsoidInteriorExteriorSpatialFunction does (fp64 sum of
// squared ratios, x then y then z, inside when <= 1).
void make_ball(const int r[3], morph_se* se) {
  se->rx = r[0];
  se->ry = r[1];
  se->rz = r[2];
  const double ax = r[0] + 0.5, ay = r[1] + 0.5, az = r[2] + 0.5;
  const int wy = 2 * r[1] + 1;
  for (int dz = -r[2]; dz <= r[2]; ++dz)
    for (int dy = -r[1]; dy <= r[1]; ++dy) {
      int e = -1;
      for (int dx = 0; dx <= r[0]; ++dx) {
        double s = (dx / ax) * (dx / ax);
        s += (dy / ay) * (dy / ay);
        s += (dz / az) * (dz / az);
        if (s <= 1.0) e = dx;
      }
      se->ex[(dz + r[2]) * wy + (dy + r[1])] = (signed char)e;
    }
}

template <int ERODE>
int launch(pp_ctx* ctx, const uint8_t* in, const pp_dims& din, uint8_t* out, const pp_dims& dout, const morph_off& off, const morph_se& se) {
  const size_t n = (size_t)dout.nx * dout.ny * dout.nz;
  hipLaunchKernelGGL(k_morph_ball<ERODE>, dim3(grid_for(n)), dim3(NT), 0, ctx->stream, in, din, out, dout, off, se);
  PP_LAUNCH_CHECK(ctx, ERODE ? "k_morph_ball<erode>" : "k_morph_ball<dilate>");
  return PP_OK;
}

}  // namespace

extern "C" int pp_binary_morph_ball_u8(pp_ctx* ctx, const uint8_t* in, const int size[3], const int radius[3], int op, uint8_t* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && size && radius && in != out, "pp_binary_morph_ball_u8: NULL or aliased argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_binary_morph_ball_u8: empty volume");
  PP_REQUIRE(ctx, op == PP_MORPH_DILATE || op == PP_MORPH_ERODE || op == PP_MORPH_CLOSE, "pp_binary_morph_ball_u8: unknown op");
  for (int a = 0; a < 3; ++a) PP_REQUIRE(ctx, radius[a] >= 0 && radius[a] <= MAX_R, "pp_binary_morph_ball_u8: radius outside [0, 15]");
  morph_se se;
  make_ball(radius, &se);
  const pp_dims d{size[0], size[1], size[2]};
  if (op == PP_MORPH_DILATE) return launch<0>(ctx, in, d, out, d, morph_off{0, 0, 0}, se);
  if (op == PP_MORPH_ERODE) return launch<1>(ctx, in, d, out, d, morph_off{0, 0, 0}, se);
  // closing with a safe border: dilate into a buffer padded by the radius, erode back onto the original grid
  const pp_dims dp{d.nx + 2 * radius[0], d.ny + 2 * radius[1], d.nz + 2 * radius[2]};
  const size_t np = (size_t)dp.nx * dp.ny * dp.nz;
  int rc = pp_reserve(ctx, pp_align_up(np, 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  uint8_t* padded = cv.take<uint8_t>(np);
  rc = launch<0>(ctx, in, d, padded, dp, morph_off{-radius[0], -radius[1], -radius[2]}, se);
  if (rc) return rc;
  return launch<1>(ctx, padded, dp, out, d, morph_off{radius[0], radius[1], radius[2]}, se);
}

extern "C" int pp_bounding_box(pp_ctx* ctx, const void* data, int dtype, const int size[3], int box[6]) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, data && size && box, "pp_bounding_box: NULL argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_bounding_box: empty volume");
  PP_REQUIRE(ctx, dtype == PP_DTYPE_U8 || dtype == PP_DTYPE_F32, "pp_bounding_box: dtype must be PP_DTYPE_U8 or PP_DTYPE_F32");
  const pp_dims d{size[0], size[1], size[2]};
  int rc = pp_reserve(ctx, 256);
  if (rc) return rc;
  int* dbox = reinterpret_cast<int*>(ctx->ws);
  hipLaunchKernelGGL(k_bounding_box_init, dim3(1), dim3(64), 0, ctx->stream, dbox);
  PP_LAUNCH_CHECK(ctx, "k_bounding_box_init");
  const size_t rows = (size_t)d.ny * d.nz;
  const unsigned nb = (unsigned)std::min<size_t>((rows + NT / 64 - 1) / (NT / 64), 4096);
  if (dtype == PP_DTYPE_U8)
    hipLaunchKernelGGL(k_bounding_box<uint8_t>, dim3(nb), dim3(NT), 0, ctx->stream, static_cast<const uint8_t*>(data), d, dbox);
  else
    hipLaunchKernelGGL(k_bounding_box<float>, dim3(nb), dim3(NT), 0, ctx->stream, static_cast<const float*>(data), d, dbox);
  PP_LAUNCH_CHECK(ctx, "k_bounding_box");
  return pp_read_back(ctx, dbox, box, 6 * sizeof(int));
}
