// platipy_amd/csrc/pp_fusion.hip -- locality-weighted label fusion and the linear-registration
// metric.
//
// Replaces the SimpleITK arithmetic of platipy/imaging/label/fusion.py: compute_weight_map
// (:56-202, vote types local/global), combine_labels (:239-292) and the threshold step of
// process_probability_image (:295-308); plus itk::MeanSquaresImageToImageMetricv4's value and
// derivative for registration/linear.py:141-148,238.  Everything is a streaming elementwise
// pass or a reduction (HBM-bound); reductions write per-block partials that a second, single
// block folds in a fixed order, so results do not depend on scheduling.
#include <vector>

#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr int NT = 256;

unsigned grid_for(size_t work, unsigned cap = 4096u) {
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// Elementwise passes: one 16-byte access per lane and array (four voxels), grid-stride, the n % 4 tail by the first
// lanes of block 0.  `vec` = every pointer is 16-byte aligned (torch allocations are; an offset view falls back to
// 4-byte accesses through the same kernel).
template <typename Op>
__global__ void __launch_bounds__(NT) k_map4(size_t n, int vec, Op op) {
  const size_t n4 = vec ? n / 4 : 0;
  for (size_t g = (size_t)blockIdx.x * NT + threadIdx.x; g < n4; g += (size_t)gridDim.x * NT) op.four(g);
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) op.one(i);
}

struct op_sqdiff {
  const float *a, *b;
  float* out;
  __device__ __forceinline__ static float f(float x, float y) { const float d = x - y; return d * d; }
  __device__ __forceinline__ void four(size_t g) const {
    const float4 x = reinterpret_cast<const float4*>(a)[g], y = reinterpret_cast<const float4*>(b)[g];
    reinterpret_cast<float4*>(out)[g] = make_float4(f(x.x, y.x), f(x.y, y.y), f(x.z, y.z), f(x.w, y.w));
  }
  __device__ __forceinline__ void one(size_t i) const { out[i] = f(a[i], b[i]); }
};

struct op_inv_eps {
  float* w;
  float eps;
  __device__ __forceinline__ void four(size_t g) const {
    float4 v = reinterpret_cast<float4*>(w)[g];
    v.x = 1.0f / (v.x + eps); v.y = 1.0f / (v.y + eps); v.z = 1.0f / (v.z + eps); v.w = 1.0f / (v.w + eps);
    reinterpret_cast<float4*>(w)[g] = v;
  }
  __device__ __forceinline__ void one(size_t i) const { w[i] = 1.0f / (w[i] + eps); }
};

// wsum += w (if wsum);  wlsum += w * label.  LabelT = uint8_t (binary masks, the pipelines' case) or float (probabilistic
// labels: the reference casts every label to float32 before weighting, label/fusion.py:269-272).
template <typename LabelT>
struct op_fuse_accumulate {
  const float* w;
  const LabelT* label;
  float *wsum, *wlsum;
  __device__ __forceinline__ void four(size_t g) const {
    const float4 wi = reinterpret_cast<const float4*>(w)[g];
    float l0, l1, l2, l3;
    if (sizeof(LabelT) == 1) {
      const uchar4 l = reinterpret_cast<const uchar4*>(label)[g];
      l0 = (float)l.x; l1 = (float)l.y; l2 = (float)l.z; l3 = (float)l.w;
    } else {
      const float4 l = reinterpret_cast<const float4*>(label)[g];
      l0 = l.x; l1 = l.y; l2 = l.z; l3 = l.w;
    }
    if (wsum) {
      float4 s = reinterpret_cast<float4*>(wsum)[g];
      s.x += wi.x; s.y += wi.y; s.z += wi.z; s.w += wi.w;
      reinterpret_cast<float4*>(wsum)[g] = s;
    }
    float4 a = reinterpret_cast<float4*>(wlsum)[g];
    a.x += wi.x * l0; a.y += wi.y * l1; a.z += wi.z * l2; a.w += wi.w * l3;
    reinterpret_cast<float4*>(wlsum)[g] = a;
  }
  __device__ __forceinline__ void one(size_t i) const {
    const float wi = w[i];
    if (wsum) wsum[i] += wi;
    wlsum[i] += wi * (float)label[i];
  }
};

// vote_type "block" (label/fusion.py:186-190): factor * box_mean^(-|gain / 2|), as sitk.Pow(raw, -1.0) ** abs(gain / 2) evaluates it
struct op_block_weight {
  float* w;
  float factor, power;
  __device__ __forceinline__ float f(float raw) const { return factor * powf(1.0f / raw, power); }
  __device__ __forceinline__ void four(size_t g) const {
    const float4 v = reinterpret_cast<float4*>(w)[g];
    reinterpret_cast<float4*>(w)[g] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
  }
  __device__ __forceinline__ void one(size_t i) const { w[i] = f(w[i]); }
};

struct op_fuse_divide {
  const float *wl, *ws;
  float* out;
  __device__ __forceinline__ static float f(float a, float s) { return a / (s == 0.0f ? 1.0f : s); }
  __device__ __forceinline__ void four(size_t g) const {
    const float4 a = reinterpret_cast<const float4*>(wl)[g], s = reinterpret_cast<const float4*>(ws)[g];
    reinterpret_cast<float4*>(out)[g] = make_float4(f(a.x, s.x), f(a.y, s.y), f(a.z, s.z), f(a.w, s.w));
  }
  __device__ __forceinline__ void one(size_t i) const { out[i] = f(wl[i], ws[i]); }
};

// itk::RescaleIntensityImageFilter (scale/shift in double, clamped to the output range) then
// itk::ThresholdImageFilter(lower, upper = 1, outside = 0).
struct op_rescale_threshold {
  float* data;
  double scale, shift;
  float lower;
  __device__ __forceinline__ float f(float x) const {
    double r = (double)x * scale + shift;
    r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
    const float v = (float)r;
    return (v < lower || v > 1.0f) ? 0.0f : v;
  }
  __device__ __forceinline__ void four(size_t g) const {
    const float4 v = reinterpret_cast<float4*>(data)[g];
    reinterpret_cast<float4*>(data)[g] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
  }
  __device__ __forceinline__ void one(size_t i) const { data[i] = f(data[i]); }
};

// sitk image / max (Div functor: fp64 quotient cast to the fp32 pixel) then BinaryThreshold(lower <= pixel).
struct op_binary_threshold {
  const float* prob;
  double max_value, threshold;
  uint8_t* out;
  __device__ __forceinline__ uint8_t f(float x) const {
    const float q = (float)((double)x / max_value);
    return ((double)q >= threshold) ? (uint8_t)1 : (uint8_t)0;
  }
  __device__ __forceinline__ void four(size_t g) const {
    const float4 v = reinterpret_cast<const float4*>(prob)[g];
    reinterpret_cast<uchar4*>(out)[g] = make_uchar4(f(v.x), f(v.y), f(v.z), f(v.w));
  }
  __device__ __forceinline__ void one(size_t i) const { out[i] = f(prob[i]); }
};

inline int all_aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) % 16) == 0;
}
// enough blocks to fill 256 CUs x 8 resident blocks; each thread then strides over its share
template <typename Op>
void launch_map4(pp_ctx* ctx, size_t n, int vec, const Op& op) {
  hipLaunchKernelGGL((k_map4<Op>), dim3(grid_for(vec ? (n + 3) / 4 : n, 2048u)), dim3(NT), 0, ctx->stream, n, vec, op);
}

__global__ void __launch_bounds__(NT) k_ssd_partial(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                    double* __restrict__ partials) {
  __shared__ double red[3 * NT];
  double s = 0.0, z0 = 0.0, z1 = 0.0;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const double d = (double)a[i] - (double)b[i];
    s += d * d;
  }
  pp_block_sum3<NT>(s, z0, z1, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void __launch_bounds__(NT) k_sum_final(const double* __restrict__ partials, int count, int stride, int nfields,
                                                  double* __restrict__ result) {
  __shared__ double red[3 * NT];
  for (int f = 0; f < nfields; ++f) {
    double s = 0.0, z0 = 0.0, z1 = 0.0;
    for (int i = threadIdx.x; i < count; i += NT) s += partials[(size_t)i * stride + f];
    pp_block_sum3<NT>(s, z0, z1, red);
    if (threadIdx.x == 0) result[f] = s;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(NT) k_minmax_partial(const float* __restrict__ in, size_t n, float* __restrict__ partials) {
  __shared__ float smin[NT], smax[NT];
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const float v = in[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = smin[0];
    partials[2 * blockIdx.x + 1] = smax[0];
  }
}

__global__ void __launch_bounds__(NT) k_minmax_final(const float* __restrict__ partials, int count, float* __restrict__ result) {
  __shared__ float smin[NT], smax[NT];
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int i = threadIdx.x; i < count; i += NT) {
    lo = fminf(lo, partials[2 * i]);
    hi = fmaxf(hi, partials[2 * i + 1]);
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    result[0] = smin[0];
    result[1] = smax[0];
  }
}

// ---------------------------------------------------------------------------------------
// Mean-squares metric and its gradient with respect to an affine map in INDEX space.
// Sample points are every `stride`-th voxel (raster order, first voxel included) of a virtual
// grid, as itk::ImageRegistrationMethodv4's REGULAR sampling walks its virtual domain.  For a
// sample with virtual index v:   f = trilinear(fixed,  Af v + bf),   m = trilinear(moving, Am v + bm),
// g = gradient of the moving interpolant in moving-index units; samples whose fixed or moving
// point leaves the buffer (or a mask) are skipped.  Accumulated in fp64:
//   [0] sum (f-m)^2   [1] count   [2..10] d/dAm[r][q] = sum -2 (f-m) g_r v_q   [11..13] d/dbm[r] = sum -2 (f-m) g_r
struct msq_args {
  double Af[9], bf[3], Am[9], bm[3];
  int vsize[3];
  int stride;
  const float* jit;   // ITK's per-sample jitter in virtual-index units, 3 floats per sample, or NULL (pp_linear_set_sample_jitter)
  const float* grad;  // ITK's filtered gradient image of the moving image, 3 volumes in moving-INDEX units, or NULL (pp_linear_set_moving_gradient)
  const float4* grad4;  // the same image PACKED with the intensity, (gx, gy, gz, m) per voxel, or NULL (pp_linear_set_moving_gradient_packed)
};

// itk::ImageRegistrationMethodv4::SetMetricSamplePoints (REGULAR): every sample point is the lattice voxel's physical point
// plus a seeded normal variate times a third of the virtual spacing per axis.  The host draws the variates (ITK's
// MersenneTwister sequence, platipy_amd/registration/linear.py) and hands them over in virtual-index units; sample e of the
// raster walk takes entries 3 e .. 3 e + 2.  NULL: the lattice itself (the default, a declared deviation from ITK).
__device__ __forceinline__ void msq_jitter(const float* __restrict__ jit, size_t e, double v[3]) {
  if (jit) {
    v[0] += (double)jit[3 * e + 0];
    v[1] += (double)jit[3 * e + 1];
    v[2] += (double)jit[3 * e + 2];
  }
}

__device__ __forceinline__ bool msq_locate(const double c[3], const pp_dims& n, int b[3], float f[3]) {
  if (!(c[0] >= -0.5 && c[0] < n.nx - 0.5 && c[1] >= -0.5 && c[1] < n.ny - 0.5 && c[2] >= -0.5 && c[2] < n.nz - 0.5)) return false;
  for (int k = 0; k < 3; ++k) {
    const double fl = floor(c[k]);
    b[k] = (int)fl;
    f[k] = (float)(c[k] - fl);
  }
  return true;
}

// itk::ImageToImageMetricv4 with UseMovingImageGradientFilter (the default): the moving-image gradient at a mapped point is the
// LINEAR interpolation of a gradient image (GradientRecursiveGaussianImageFilter, sigma = the moving image's largest spacing),
// not the derivative of the intensity interpolant.  `grad`: that image as three volumes of the moving image's size, already in
// moving-index units; same eight corners and weights as the intensity sample.
__device__ __forceinline__ void msq_gradient_image(const float* __restrict__ grad, const pp_dims& dm, int x0, int x1, int y0, int y1, int z0,
                                                   int z1, float wx, float wy, float wz, float g[3]) {
  const size_t sy = dm.nx, sz = (size_t)dm.nx * dm.ny, N = sz * dm.nz;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* G = grad + r * N;
    const float a000 = G[z0 * sz + y0 * sy + x0], a100 = G[z0 * sz + y0 * sy + x1];
    const float a010 = G[z0 * sz + y1 * sy + x0], a110 = G[z0 * sz + y1 * sy + x1];
    const float a001 = G[z1 * sz + y0 * sy + x0], a101 = G[z1 * sz + y0 * sy + x1];
    const float a011 = G[z1 * sz + y1 * sy + x0], a111 = G[z1 * sz + y1 * sy + x1];
    const float v00 = a000 + (a100 - a000) * wx, v10 = a010 + (a110 - a010) * wx;
    const float v01 = a001 + (a101 - a001) * wx, v11 = a011 + (a111 - a011) * wx;
    const float v0 = v00 + (v10 - v00) * wy, v1 = v01 + (v11 - v01) * wy;
    g[r] = v0 + (v1 - v0) * wz;
  }
}

// MODE 0: mean squares, NACC = 14 (layout above).
// MODE 1: raw moments for the correlation metric, NACC = 42:
//   [0] count [1] sum f [2] sum m [3] sum f^2 [4] sum m^2 [5] sum f m
//   [6..17] sum g (d/dAm row-major 9, d/dbm 3)   [18..29] sum f g   [30..41] sum m g      (g = interpolant gradient terms)
template <int MODE>
__global__ void __launch_bounds__(NT) k_metric_affine(const float* __restrict__ F, pp_dims df, const float* __restrict__ M,
                                                      pp_dims dm, const uint8_t* __restrict__ fmask,
                                                      const uint8_t* __restrict__ mmask, msq_args a,
                                                      double* __restrict__ partials /* [grid][NACC] */,
                                                      const float* __restrict__ fsamp /* pp_fixed_samples, or NULL */) {
  constexpr int NACC = MODE == 0 ? 14 : 42;
  __shared__ double red[3 * NT];
  double acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < nsamp; e += (size_t)gridDim.x * NT) {
    const size_t lin = e * (size_t)a.stride;
    double v[3] = {(double)(lin % a.vsize[0]), (double)((lin / a.vsize[0]) % a.vsize[1]),
                   (double)(lin / ((size_t)a.vsize[0] * a.vsize[1]))};
    msq_jitter(a.jit, e, v);
    double cf[3], cm[3];
    for (int r = 0; r < 3; ++r) {
      cf[r] = a.Af[r * 3 + 0] * v[0] + a.Af[r * 3 + 1] * v[1] + a.Af[r * 3 + 2] * v[2] + a.bf[r];
      cm[r] = a.Am[r * 3 + 0] * v[0] + a.Am[r * 3 + 1] * v[1] + a.Am[r * 3 + 2] * v[2] + a.bm[r];
    }
    int bf_[3], bm_[3];
    float ff[3], fm[3];
    float fval = 0.0f;
    if (fsamp) {   // the fixed side of this sample was evaluated once for the level (same arithmetic)
      fval = fsamp[e];
      if (__builtin_bit_cast(unsigned, fval) == PP_FSAMP_INVALID) continue;
      if (!msq_locate(cm, dm, bm_, fm)) continue;
    } else {
      if (!msq_locate(cf, df, bf_, ff) || !msq_locate(cm, dm, bm_, fm)) continue;
      if (fmask) {
        const int qx = (int)floor(cf[0] + 0.5), qy = (int)floor(cf[1] + 0.5), qz = (int)floor(cf[2] + 0.5);
        if (!fmask[((size_t)qz * df.ny + qy) * df.nx + qx]) continue;
      }
    }
    if (mmask) {
      const int qx = (int)floor(cm[0] + 0.5), qy = (int)floor(cm[1] + 0.5), qz = (int)floor(cm[2] + 0.5);
      if (!mmask[((size_t)qz * dm.ny + qy) * dm.nx + qx]) continue;
    }
    if (!fsamp) fval = pp_trilinear(F, df.nx, df.ny, df.nz, bf_[0], ff[0], bf_[1], ff[1], bf_[2], ff[2]);
    int x0, x1, y0, y1, z0, z1;
    float wx, wy, wz;
    pp_axis_setup(bm_[0], fm[0], dm.nx, x0, x1, wx);
    pp_axis_setup(bm_[1], fm[1], dm.ny, y0, y1, wy);
    pp_axis_setup(bm_[2], fm[2], dm.nz, z0, z1, wz);
    const size_t sy = dm.nx, sz = (size_t)dm.nx * dm.ny;
    const float a000 = M[z0 * sz + y0 * sy + x0], a100 = M[z0 * sz + y0 * sy + x1];
    const float a010 = M[z0 * sz + y1 * sy + x0], a110 = M[z0 * sz + y1 * sy + x1];
    const float a001 = M[z1 * sz + y0 * sy + x0], a101 = M[z1 * sz + y0 * sy + x1];
    const float a011 = M[z1 * sz + y1 * sy + x0], a111 = M[z1 * sz + y1 * sy + x1];
    const float v00 = a000 + (a100 - a000) * wx, v10 = a010 + (a110 - a010) * wx;
    const float v01 = a001 + (a101 - a001) * wx, v11 = a011 + (a111 - a011) * wx;
    const float v0 = v00 + (v10 - v00) * wy, v1 = v01 + (v11 - v01) * wy;
    const float m = v0 + (v1 - v0) * wz;
    // gradient of the trilinear interpolant, per moving voxel
    const float gx0 = (a100 - a000) + ((a110 - a010) - (a100 - a000)) * wy;
    const float gx1 = (a101 - a001) + ((a111 - a011) - (a101 - a001)) * wy;
    float gx = gx0 + (gx1 - gx0) * wz;
    float gy = (v10 - v00) + ((v11 - v01) - (v10 - v00)) * wz;
    float gz = v1 - v0;
    if (a.grad) {   // (uniform) ITK's filtered gradient image instead of the interpolant's derivative
      float gi[3];
      msq_gradient_image(a.grad, dm, x0, x1, y0, y1, z0, z1, wx, wy, wz, gi);
      gx = gi[0]; gy = gi[1]; gz = gi[2];
    }
    if (MODE == 0) {
      const double diff = (double)fval - (double)m;
      acc[0] += diff * diff;
      acc[1] += 1.0;
      const double s = -2.0 * diff;
      const double g[3] = {s * gx, s * gy, s * gz};
      for (int r = 0; r < 3; ++r) {
        acc[2 + r * 3 + 0] += g[r] * v[0];
        acc[2 + r * 3 + 1] += g[r] * v[1];
        acc[2 + r * 3 + 2] += g[r] * v[2];
        acc[11 + r] += g[r];
      }
    } else {
      const double fd = fval, md = m;
      acc[0] += 1.0;
      acc[1] += fd;
      acc[2] += md;
      acc[3] += fd * fd;
      acc[4] += md * md;
      acc[5] += fd * md;
      const double g[3] = {gx, gy, gz};
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 4; ++q) {
          const double t = g[r] * (q < 3 ? v[q] : 1.0);
          const int slot = q < 3 ? r * 3 + q : 9 + r;
          acc[6 + slot] += t;
          acc[18 + slot] += fd * t;
          acc[30 + slot] += md * t;
        }
    }
  }
  for (int k = 0; k < NACC; k += 3) {
    double p = acc[k], q = k + 1 < NACC ? acc[k + 1] : 0.0, r = k + 2 < NACC ? acc[k + 2] : 0.0;
    pp_block_sum3<NT>(p, q, r, red);
    if (threadIdx.x == 0) {
      partials[(size_t)blockIdx.x * NACC + k] = p;
      if (k + 1 < NACC) partials[(size_t)blockIdx.x * NACC + k + 1] = q;
      if (k + 2 < NACC) partials[(size_t)blockIdx.x * NACC + k + 2] = r;
    }
    __syncthreads();
  }
}

// Fold [count][14] partial rows into 14 sums with one tree (8 barrier steps for all fields at once); launched
// with one block per group of 14 fields (blockIdx.x selects the group of a wider row).
__global__ void __launch_bounds__(NT) k_sum14_final(const double* __restrict__ partials, int count, int row, double* result,
                                                    int to_mailbox, unsigned long long seq) {
  __shared__ double red[NT * 14];
  const int f0 = blockIdx.x * 14;
  const int nf = row - f0 < 14 ? row - f0 : 14;
  double acc[14];
  for (int f = 0; f < 14; ++f) acc[f] = 0.0;
  for (int i = threadIdx.x; i < count; i += NT)
    for (int f = 0; f < nf; ++f) acc[f] += partials[(size_t)i * row + f0 + f];
  for (int f = 0; f < 14; ++f) red[f * NT + threadIdx.x] = acc[f];
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int f = 0; f < 14; ++f) red[f * NT + threadIdx.x] += red[f * NT + threadIdx.x + s];
    __syncthreads();
  }
  if ((int)threadIdx.x < nf) {
    if (to_mailbox)   // `result` is the host mailbox: block b is writer b, field f its entry f (pp_internal.h)
      pp_mail_post(pp_mail_slot(result, (int)blockIdx.x) + threadIdx.x, red[threadIdx.x * NT], seq);
    else
      result[f0 + threadIdx.x] = red[threadIdx.x * NT];
  }
}

// ---- line-search evaluations: K candidate moving maps in ONE launch, values only --------------------------
// The golden-section line search of ITK's GradientDescentLineSearchOptimizerv4 needs ~13 metric VALUES per
// optimiser iteration, each depending on the previous comparison; one launch + one read-back per value is pure
// latency (the sample lattice is 16 K .. 1 M points).  The host speculates the next few levels of the search
// tree and evaluates all of their learning rates at once: blockIdx.y = candidate.  The last block of a candidate to
// finish (device ticket per candidate) folds that candidate's per-block partial sums with a fixed tree, so a candidate's value does not depend on
// which other candidates ride in the launch.  (ctx->ticket holds 16 counters 128 bytes apart; a chunk uses the one at chunk * TICKET_STRIDE.)
constexpr int PP_MAX_CAND = 16;
constexpr unsigned TICKET_STRIDE = 32;   // counters sit 128 bytes apart (ctx->ticket holds 16 of them)
struct mval_args {
  double Af[9], bf[3];
  double Am[PP_MAX_CAND][9], bm[PP_MAX_CAND][3];
  int vsize[3];
  int stride;
  const float* jit;   // (as msq_args)
};

// pp_trilinear with the two x-neighbours of each corner row fetched as ONE 8-byte access when they are adjacent
// in memory (they are, except on the clamped high edge): the probe kernel's lanes sit a full cache-line sector apart,
// so every access is its own L2 request and halving their number is what counts.  Same lerps, same order.
__device__ __forceinline__ float msq_trilinear_pairs(const float* __restrict__ im, int nx, int ny, int nz, int bx, float fx, int by,
                                                     float fy, int bz, float fz) {
  int x0, x1, y0, y1, z0, z1;
  float wx, wy, wz;
  pp_axis_setup(bx, fx, nx, x0, x1, wx);
  pp_axis_setup(by, fy, ny, y0, y1, wy);
  pp_axis_setup(bz, fz, nz, z0, z1, wz);
  const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
  const float* p00 = im + z0 * sz + y0 * sy + x0;
  const float* p10 = im + z0 * sz + y1 * sy + x0;
  const float* p01 = im + z1 * sz + y0 * sy + x0;
  const float* p11 = im + z1 * sz + y1 * sy + x0;
  float a000, a100, a010, a110, a001, a101, a011, a111;
  if (x1 == x0 + 1) {
    a000 = p00[0]; a100 = p00[1];
    a010 = p10[0]; a110 = p10[1];
    a001 = p01[0]; a101 = p01[1];
    a011 = p11[0]; a111 = p11[1];
  } else {
    a000 = a100 = p00[0];
    a010 = a110 = p10[0];
    a001 = a101 = p01[0];
    a011 = a111 = p11[0];
  }
  const float v00 = a000 + (a100 - a000) * wx;
  const float v10 = a010 + (a110 - a010) * wx;
  const float v01 = a001 + (a101 - a001) * wx;
  const float v11 = a011 + (a111 - a011) * wx;
  const float v0 = v00 + (v10 - v00) * wy;
  const float v1 = v01 + (v11 - v01) * wy;
  return v0 + (v1 - v0) * wz;
}

// MODE 0: [sum (f-m)^2, count].  MODE 1: [count, sum f, sum m, sum f^2, sum m^2, sum f m].
// One thread walks its samples and, per sample, ALL candidates of its chunk (CH per blockIdx.y): the lattice is
// sparse in the full-resolution images (every gather is its own cache line, so an evaluation is HBM traffic, not
// arithmetic), the candidates of a line search land within a voxel or two of each other, and probing them
// back-to-back turns 15 of 16 candidates' gathers into L1/L2 hits; the fixed sample is fetched once.
template <int MODE, int CH>
__global__ void __launch_bounds__(NT) k_metric_values(const float* __restrict__ F, pp_dims df, const float* __restrict__ M, pp_dims dm,
                                                      const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ mmask, mval_args a,
                                                      int ncand, double* partials /* [chunk][grid.x][CH*NV] */,
                                                      unsigned* __restrict__ ticket, void* mailbox /* host mailbox: chunk = writer */,
                                                      unsigned long long seq, const float* __restrict__ fsamp /* or NULL */) {
  constexpr int NV = MODE == 0 ? 2 : 6;
  constexpr int ROW = CH * NV;
  __shared__ double red[CH * (MODE == 0 ? 2 : 6) * NT];   // 64 KB / 48 KB: every accumulator of every thread
  __shared__ int is_last;
  const int c0 = blockIdx.y * CH;
  const int nc = ncand - c0 < CH ? ncand - c0 : CH;
  double acc[ROW];
  for (int k = 0; k < ROW; ++k) acc[k] = 0.0;
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < nsamp; e += (size_t)gridDim.x * NT) {
    const size_t lin = e * (size_t)a.stride;
    double v[3] = {(double)(lin % a.vsize[0]), (double)((lin / a.vsize[0]) % a.vsize[1]),
                   (double)(lin / ((size_t)a.vsize[0] * a.vsize[1]))};
    msq_jitter(a.jit, e, v);
    double fd;
    if (fsamp) {   // the fixed side of this sample was evaluated once for the level: a coalesced 4-byte read instead of four
                   // sparse cache lines per sample (at shrink 4 a probe dragged a quarter of the fixed image through HBM)
      const float fv = fsamp[e];
      if (__builtin_bit_cast(unsigned, fv) == PP_FSAMP_INVALID) continue;
      fd = fv;
    } else {
      double cf[3];
      for (int r = 0; r < 3; ++r) cf[r] = a.Af[r * 3 + 0] * v[0] + a.Af[r * 3 + 1] * v[1] + a.Af[r * 3 + 2] * v[2] + a.bf[r];
      int bf_[3];
      float ff[3];
      if (!msq_locate(cf, df, bf_, ff)) continue;
      if (fmask) {
        const int qx = (int)floor(cf[0] + 0.5), qy = (int)floor(cf[1] + 0.5), qz = (int)floor(cf[2] + 0.5);
        if (!fmask[((size_t)qz * df.ny + qy) * df.nx + qx]) continue;
      }
      fd = msq_trilinear_pairs(F, df.nx, df.ny, df.nz, bf_[0], ff[0], bf_[1], ff[1], bf_[2], ff[2]);
    }
    if constexpr (CH >= 16) {
    // Straight-line over the candidates: every candidate forms a valid (clamped) address and gathers unconditionally, and
    // "inside / masked" only selects what is accumulated -- so a group of four candidates has no branches and the gathers of the
    // next candidates are in flight while this one is interpolated (a thread's candidates used to be a serial chain of
    // dependent cache misses).  Adding 0.0 for a rejected candidate leaves the sums bit-identical.
#pragma unroll
    for (int j0 = 0; j0 < CH; j0 += 4) {
    if (j0 < nc) {   // (block-uniform: whole groups of four beyond the batch are skipped; inside a group no branches)
#pragma unroll
    for (int j = j0; j < j0 + 4; ++j) {
      double cm[3];
      for (int r = 0; r < 3; ++r)
        cm[r] = a.Am[c0 + j][r * 3 + 0] * v[0] + a.Am[c0 + j][r * 3 + 1] * v[1] + a.Am[c0 + j][r * 3 + 2] * v[2] + a.bm[c0 + j][r];
      bool ok = (j < nc) & (cm[0] >= -0.5) & (cm[0] < dm.nx - 0.5) & (cm[1] >= -0.5) & (cm[1] < dm.ny - 0.5) & (cm[2] >= -0.5) &
                (cm[2] < dm.nz - 0.5);
      int bm_[3];
      float fm[3];
      for (int r = 0; r < 3; ++r) {
        const double cc = ok ? cm[r] : 0.0;
        const double fl = floor(cc);
        bm_[r] = (int)fl;
        fm[r] = (float)(cc - fl);
      }
      if (mmask) {   // (uniform)
        const int qx = ok ? (int)floor(cm[0] + 0.5) : 0, qy = ok ? (int)floor(cm[1] + 0.5) : 0, qz = ok ? (int)floor(cm[2] + 0.5) : 0;
        ok = ok & (mmask[((size_t)qz * dm.ny + qy) * dm.nx + qx] != 0);
      }
      const double md = dm.nx >= 2 ? (double)pp_trilinear_pairs(M, dm.nx, dm.ny, dm.nz, bm_[0], fm[0], bm_[1], fm[1], bm_[2], fm[2])
                                   : (double)msq_trilinear_pairs(M, dm.nx, dm.ny, dm.nz, bm_[0], fm[0], bm_[1], fm[1], bm_[2], fm[2]);
      if (MODE == 0) {
        const double diff = fd - md;
        acc[j * NV + 0] += ok ? diff * diff : 0.0;
        acc[j * NV + 1] += ok ? 1.0 : 0.0;
      } else {
        acc[j * NV + 0] += ok ? 1.0 : 0.0;
        acc[j * NV + 1] += ok ? fd : 0.0;
        acc[j * NV + 2] += ok ? md : 0.0;
        acc[j * NV + 3] += ok ? fd * fd : 0.0;
        acc[j * NV + 4] += ok ? md * md : 0.0;
        acc[j * NV + 5] += ok ? fd * md : 0.0;
      }
    }
    }
    }
    } else {   // few candidates per thread: the branchy form measured faster (tools/bench_metric.py)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      double cm[3];
      for (int r = 0; r < 3; ++r)
        cm[r] = a.Am[c0 + j][r * 3 + 0] * v[0] + a.Am[c0 + j][r * 3 + 1] * v[1] + a.Am[c0 + j][r * 3 + 2] * v[2] + a.bm[c0 + j][r];
      int bm_[3] = {0, 0, 0};
      float fm[3] = {0.0f, 0.0f, 0.0f};
      bool ok = j < nc && msq_locate(cm, dm, bm_, fm);
      if (ok && mmask) {
        const int qx = (int)floor(cm[0] + 0.5), qy = (int)floor(cm[1] + 0.5), qz = (int)floor(cm[2] + 0.5);
        ok = mmask[((size_t)qz * dm.ny + qy) * dm.nx + qx] != 0;
      }
      if (ok) {
        const double md = msq_trilinear_pairs(M, dm.nx, dm.ny, dm.nz, bm_[0], fm[0], bm_[1], fm[1], bm_[2], fm[2]);
        if (MODE == 0) {
          const double diff = fd - md;
          acc[j * NV + 0] += diff * diff;
          acc[j * NV + 1] += 1.0;
        } else {
          acc[j * NV + 0] += 1.0;
          acc[j * NV + 1] += fd;
          acc[j * NV + 2] += md;
          acc[j * NV + 3] += fd * fd;
          acc[j * NV + 4] += md * md;
          acc[j * NV + 5] += fd * md;
        }
      }
    }
    }
  }
  // one wide tree for all ROW accumulators (8 barrier rounds instead of 8 per triple): red[f][thread]
  double* mine = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ROW;
#pragma unroll
  for (int k = 0; k < ROW; ++k) red[k * NT + threadIdx.x] = acc[k];
  __syncthreads();
  for (int st = NT / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
#pragma unroll
      for (int k = 0; k < ROW; ++k) red[k * NT + threadIdx.x] += red[k * NT + threadIdx.x + st];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < ROW) {
    // agent-scope atomic store / load for the rows (below): they cross XCDs, each behind its own L2, inside one launch
    __hip_atomic_store(mine + threadIdx.x, red[threadIdx.x * NT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();   // each writer publishes its own store before the block takes its ticket
  }
  __syncthreads();
  // the last block of this chunk to arrive folds the chunk's rows (fixed tree, independent of arrival order)
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket + blockIdx.y * TICKET_STRIDE, 1u);   // (one 128-byte line per chunk's counter)
    is_last = t == gridDim.x - 1u;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  // Every column of the chunk's rows at once: thread (part, col) adds rows part, part + NPARTS, ... of its column (a
  // coalesced read per row), then one thread per column adds the NPARTS partial sums in order.  Fixed tree, independent
  // of the arrival order; one barrier instead of a block-wide tree per three columns.
  const double* rows = partials + (size_t)blockIdx.y * gridDim.x * ROW;
  constexpr int NPARTS = NT / ROW;
  {
    const int col = (int)threadIdx.x % ROW, part = (int)threadIdx.x / ROW;
    double p = 0.0;
    if (part < NPARTS)
      for (unsigned i = (unsigned)part; i < gridDim.x; i += NPARTS)
        p += __hip_atomic_load(rows + (size_t)i * ROW + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();   // (red still holds the block's own tree)
    if (part < NPARTS) red[part * ROW + col] = p;
    __syncthreads();
    if ((int)threadIdx.x < nc * NV) {
      double tot = 0.0;
      for (int q = 0; q < NPARTS; ++q) tot += red[q * ROW + threadIdx.x];
      pp_mail_post(pp_mail_slot(mailbox, (int)blockIdx.y) + threadIdx.x, tot, seq);   // entry = local candidate * NV + field
    }
  }
  if (threadIdx.x == 0) atomicExch(ticket + blockIdx.y * TICKET_STRIDE, 0u);   // (where the increments go, not a cached plain store)
}

// ---- line-search evaluations, second generation: a candidate is a LANE ---------------------------------------
// k_metric_values walks a thread over its samples and, per sample, over the candidates: on the sparse lattices of the
// pipelines (every 8th voxel of the full-resolution moving image along x) the 64 lanes of a gather then address 16-64
// different cache lines, and the kernel runs at the vector cache's line-lookup rate (measured: 75 us for 16 candidates on
// 524 K samples, ~64 clocks per gather instruction).  Here lane = 16 * slot + candidate: a wavefront holds FOUR
// consecutive samples x 16 candidates; the four samples sit in neighbouring 32-byte sectors and, once the search bracket has
// shrunk, a sample's candidates within a voxel or two of each other, so a gather instruction touches a handful of lines; a thread keeps NV
// accumulators instead of 16 * NV (no 64 KB reduction tile: the block is bound by registers, not LDS), and a candidate's
// samples are spread over 16 x as many threads (shorter dependent chains on the small lattices).
// Sum order of a candidate: a thread adds its samples e0 + slot, e0 + 16 + slot, ... in increasing order; the four slots
// of a wavefront combine as (s0 + s1) + (s2 + s3), the four wavefronts as (w0 + w1) + (w2 + w3); the last block to finish folds the
// blocks' rows with the fixed tree of k_metric_values.  The layout does not depend on the number of candidates, so a
// candidate's value does not depend on its companions.
// pp_trilinear_pairs with 32-bit offsets built from 24-bit multiplies (full rate; a 64-bit offset costs four quarter-rate
// multiplies per row pair), split into "request the four row pairs" and "interpolate" so that a caller can have the pairs of
// several samples in flight before it touches the first.  Caller guarantees ny * nz < 2^24, nx < 2^24, nx >= 2 and fewer
// than 2^31 voxels.  Same lerps as pp_trilinear_pairs.
struct mv_pairs {
  pp_pair<float> p00, p10, p01, p11;
  float wx, wy, wz;
  bool xlast;
};
__device__ __forceinline__ void mv_pairs32_issue(const float* __restrict__ im, int nx, int ny, int nz, int bx, float fx, int by, float fy,
                                                 int bz, float fz, mv_pairs& g) {
  int x0, x1, y0, y1, z0, z1;
  pp_axis_setup(bx, fx, nx, x0, x1, g.wx);
  pp_axis_setup(by, fy, ny, y0, y1, g.wy);
  pp_axis_setup(bz, fz, nz, z0, z1, g.wz);
  g.xlast = x0 > nx - 2;
  const unsigned xs = (unsigned)(g.xlast ? nx - 2 : x0);
  const unsigned r00 = __umul24(__umul24((unsigned)z0, (unsigned)ny) + (unsigned)y0, (unsigned)nx) + xs;
  const unsigned dy = y1 > y0 ? (unsigned)nx : 0u, dz = z1 > z0 ? __umul24((unsigned)nx, (unsigned)ny) : 0u;
  g.p00 = *reinterpret_cast<const pp_pair<float>*>(im + r00);
  g.p10 = *reinterpret_cast<const pp_pair<float>*>(im + (r00 + dy));
  g.p01 = *reinterpret_cast<const pp_pair<float>*>(im + (r00 + dz));
  g.p11 = *reinterpret_cast<const pp_pair<float>*>(im + (r00 + dz + dy));
}
__device__ __forceinline__ float mv_pairs32_finish(const mv_pairs& g) {
  const float a000 = g.xlast ? g.p00.y : g.p00.x, a100 = g.p00.y;
  const float a010 = g.xlast ? g.p10.y : g.p10.x, a110 = g.p10.y;
  const float a001 = g.xlast ? g.p01.y : g.p01.x, a101 = g.p01.y;
  const float a011 = g.xlast ? g.p11.y : g.p11.x, a111 = g.p11.y;
  const float v00 = a000 + (a100 - a000) * g.wx;
  const float v10 = a010 + (a110 - a010) * g.wx;
  const float v01 = a001 + (a101 - a001) * g.wx;
  const float v11 = a011 + (a111 - a011) * g.wx;
  const float v0 = v00 + (v10 - v00) * g.wy;
  const float v1 = v01 + (v11 - v01) * g.wy;
  return v0 + (v1 - v0) * g.wz;
}
template <bool B> struct mv_flag { static constexpr bool value = B; };

constexpr int MV_CL = 16;              // lanes per sample
constexpr int MV_SLOTS = NT / MV_CL;   // samples a block handles side by side
template <int MODE>
#ifndef PP_MV_WAVES
#define PP_MV_WAVES 1
#endif
__global__ void __launch_bounds__(NT, PP_MV_WAVES) k_metric_values_lanes(const float* __restrict__ F, pp_dims df, const float* __restrict__ M, pp_dims dm,
                                                            const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ mmask,
                                                            mval_args a, int ncand, int spt /* samples per thread */,
                                                            double* partials /* [grid.x][16 * NV] */, unsigned* __restrict__ ticket,
                                                            void* mailbox, unsigned long long seq, const float* __restrict__ fsamp) {
  constexpr int NV = MODE == 0 ? 2 : 6;
  constexpr int ROW = MV_CL * NV;
  constexpr int CPW = MODE == 0 ? 16 : 8;   // candidates per mailbox writer (64 entries each)
  constexpr int NPARTS = NT / ROW;          // the last block folds the rows in NPARTS interleaved parts
  __shared__ double red[(NT / 64 > NPARTS ? NT / 64 : NPARTS) * ROW];
  __shared__ int is_last;
  const int c = (int)threadIdx.x % MV_CL, slot = (int)threadIdx.x / MV_CL;
  const bool live = c < ncand;
  double A[9], b[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = a.Am[live ? c : 0][k];
#pragma unroll
  for (int k = 0; k < 3; ++k) b[k] = a.bm[live ? c : 0][k];
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0;
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  const size_t e0 = (size_t)blockIdx.x * ((size_t)MV_SLOTS * spt) + slot;
  // Lattice coordinates of this thread's first sample by division, of the following ones by adding the decomposed step
  // (the samples of a thread are MV_SLOTS * stride raster positions apart): no division inside the loop.
  unsigned px, py, pz, sx_, sy_, sz_;
  {
    const size_t lin0 = (e0 < nsamp ? e0 : nsamp - 1) * (size_t)a.stride, step = (size_t)MV_SLOTS * a.stride;
    const size_t vx = (size_t)a.vsize[0], vy = (size_t)a.vsize[1];
    if (nvirt < ((size_t)1 << 31) && step < ((size_t)1 << 31)) {   // (uniform; 32-bit divisions)
      const unsigned l32 = (unsigned)lin0, s32 = (unsigned)step, vx32 = (unsigned)vx, vy32 = (unsigned)vy;
      unsigned q = l32 / vx32;
      px = l32 - q * vx32; py = q % vy32; pz = q / vy32;
      q = s32 / vx32;
      sx_ = s32 - q * vx32; sy_ = q % vy32; sz_ = q / vy32;
    } else {
      px = (unsigned)(lin0 % vx); py = (unsigned)((lin0 / vx) % vy); pz = (unsigned)(lin0 / (vx * vy));
      sx_ = (unsigned)(step % vx); sy_ = (unsigned)((step / vx) % vy); sz_ = (unsigned)(step / (vx * vy));
    }
  }
  const bool idx32 = dm.nx >= 2 && dm.nx < (1 << 24) && (size_t)dm.ny * dm.nz < ((size_t)1 << 24) &&
                     (size_t)dm.nx * dm.ny * dm.nz < ((size_t)1 << 31);   // (uniform)
  // Groups of four samples.  Every sample forms a valid (clamped) address and gathers unconditionally; "inside / masked /
  // beyond the lattice" only selects what is accumulated (adding 0.0 leaves the sums bit-identical).  FAST (fixed samples
  // cached for the level, 32-bit offsets -- what the optimiser runs): the body has no branch at all, the sixteen row pairs of
  // a group are requested before the first is interpolated.  (With the three uniform decisions INSIDE the body the compiler
  // waited for each sample's pairs before it requested the next: 32 dependent round trips per thread, 45 us of a 57 us probe.)
  // Lanes without a candidate skip the walk -- a launch costs what its candidates cost -- and join the reduction with zeros.
  auto walk = [&](auto fast_tag, auto mask_tag) {
    constexpr bool FAST = decltype(fast_tag)::value, MASKED = decltype(mask_tag)::value;
    for (int i0 = 0; i0 < spt; i0 += 4) {
      if (e0 + (size_t)i0 * MV_SLOTS - slot >= nsamp) break;   // (block-uniform: the whole group lies beyond the lattice)
      double fd[4], md[4];
      bool ok[4];
      mv_pairs g[4];
      float fv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      unsigned mk[4] = {1u, 1u, 1u, 1u};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t eq = e0 + (size_t)(i0 + j) * MV_SLOTS;
        ok[j] = (i0 + j < spt) && eq < nsamp;
        const size_t e = eq < nsamp ? eq : nsamp - 1;
        double v[3] = {(double)px, (double)py, (double)pz};
        msq_jitter(a.jit, e, v);
        {   // next sample of this thread (beyond the lattice the position is never used: ok[j] is false there)
          px += sx_;
          const bool cx = px >= (unsigned)a.vsize[0];
          px -= cx ? (unsigned)a.vsize[0] : 0u;
          py += sy_ + (cx ? 1u : 0u);
          const bool cy = py >= (unsigned)a.vsize[1];
          py -= cy ? (unsigned)a.vsize[1] : 0u;
          pz += sz_ + (cy ? 1u : 0u);
        }
        if (FAST) {            // the fixed side of this sample was evaluated once for the level (looked at with the pairs, below)
          fv[j] = fsamp[e];
        } else if (fsamp) {
          const float f1 = fsamp[e];
          ok[j] = ok[j] && __builtin_bit_cast(unsigned, f1) != PP_FSAMP_INVALID;
          fd[j] = f1;
        } else {
          double cf[3];
          for (int r = 0; r < 3; ++r) cf[r] = a.Af[r * 3 + 0] * v[0] + a.Af[r * 3 + 1] * v[1] + a.Af[r * 3 + 2] * v[2] + a.bf[r];
          int bf_[3] = {0, 0, 0};
          float ff[3] = {0.0f, 0.0f, 0.0f};
          bool okf = msq_locate(cf, df, bf_, ff);
          if (okf && fmask) {
            const int qx = (int)floor(cf[0] + 0.5), qy = (int)floor(cf[1] + 0.5), qz = (int)floor(cf[2] + 0.5);
            okf = fmask[((size_t)qz * df.ny + qy) * df.nx + qx] != 0;
          }
          fd[j] = okf ? (double)msq_trilinear_pairs(F, df.nx, df.ny, df.nz, bf_[0], ff[0], bf_[1], ff[1], bf_[2], ff[2]) : 0.0;
          ok[j] = ok[j] && okf;
        }
        double cm[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) cm[r] = A[r * 3 + 0] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2] + b[r];
        const bool in = (cm[0] >= -0.5) & (cm[0] < dm.nx - 0.5) & (cm[1] >= -0.5) & (cm[1] < dm.ny - 0.5) & (cm[2] >= -0.5) &
                        (cm[2] < dm.nz - 0.5);
        ok[j] = ok[j] & in;
        int bm_[3];
        float fm[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double cc = in ? cm[r] : 0.0;
          const double fl = floor(cc);
          bm_[r] = (int)fl;
          fm[r] = (float)(cc - fl);
        }
        if (MASKED) {
          const int qx = in ? (int)floor(cm[0] + 0.5) : 0, qy = in ? (int)floor(cm[1] + 0.5) : 0, qz = in ? (int)floor(cm[2] + 0.5) : 0;
          mk[j] = mmask[((size_t)qz * dm.ny + qy) * dm.nx + qx];
          if (!FAST) ok[j] = ok[j] & (mk[j] != 0);
        }
        if (FAST)
          mv_pairs32_issue(M, dm.nx, dm.ny, dm.nz, bm_[0], fm[0], bm_[1], fm[1], bm_[2], fm[2], g[j]);
        else
          md[j] = dm.nx >= 2 ? (double)pp_trilinear_pairs(M, dm.nx, dm.ny, dm.nz, bm_[0], fm[0], bm_[1], fm[1], bm_[2], fm[2])
                             : (double)msq_trilinear_pairs(M, dm.nx, dm.ny, dm.nz, bm_[0], fm[0], bm_[1], fm[1], bm_[2], fm[2]);
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if (FAST) {
        // Pin "all sixteen requests, then the interpolations": instruction selection is free to emit a sample's (pure)
        // interpolation arithmetic right under its own loads -- one sample in flight per thread -- and a scheduling barrier alone
        // does not order arithmetic.  The empty statements consume and re-define the loaded registers, so every
        // interpolation depends on a statement that follows every load.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j += 2)
          asm volatile("" : "+v"(g[j].p00.x), "+v"(g[j].p00.y), "+v"(g[j].p10.x), "+v"(g[j].p10.y), "+v"(g[j].p01.x), "+v"(g[j].p01.y),
                            "+v"(g[j].p11.x), "+v"(g[j].p11.y), "+v"(g[j + 1].p00.x), "+v"(g[j + 1].p00.y), "+v"(g[j + 1].p10.x),
                            "+v"(g[j + 1].p10.y), "+v"(g[j + 1].p01.x), "+v"(g[j + 1].p01.y), "+v"(g[j + 1].p11.x), "+v"(g[j + 1].p11.y),
                            "+v"(fv[j]), "+v"(fv[j + 1]), "+v"(mk[j]), "+v"(mk[j + 1]));
      }
#endif
      if (FAST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ok[j] = ok[j] && __builtin_bit_cast(unsigned, fv[j]) != PP_FSAMP_INVALID && mk[j] != 0;
          fd[j] = fv[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (FAST) md[j] = (double)mv_pairs32_finish(g[j]);
        if constexpr (MODE == 0) {
          const double diff = fd[j] - md[j];
          acc[0] += ok[j] ? diff * diff : 0.0;
          acc[1] += ok[j] ? 1.0 : 0.0;
        } else {
          acc[0] += ok[j] ? 1.0 : 0.0;
          acc[1] += ok[j] ? fd[j] : 0.0;
          acc[2] += ok[j] ? md[j] : 0.0;
          acc[3] += ok[j] ? fd[j] * fd[j] : 0.0;
          acc[4] += ok[j] ? md[j] * md[j] : 0.0;
          acc[5] += ok[j] ? fd[j] * md[j] : 0.0;
        }
      }
    }
  };
  if (live) {
    if (fsamp && idx32) {
      if (mmask) walk(mv_flag<true>{}, mv_flag<true>{});
      else walk(mv_flag<true>{}, mv_flag<false>{});
    } else {
      if (mmask) walk(mv_flag<false>{}, mv_flag<true>{});
      else walk(mv_flag<false>{}, mv_flag<false>{});
    }
  }
  // the four slots of a wavefront, then the four wavefronts
  const int wave = (int)threadIdx.x / 64, lane = (int)threadIdx.x % 64;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = acc[k];
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    if (lane < MV_CL) red[wave * ROW + lane * NV + k] = t;
  }
  __syncthreads();
  double* mine = partials + (size_t)blockIdx.x * ROW;
  if ((int)threadIdx.x < ROW) {
    const double t = (red[0 * ROW + threadIdx.x] + red[1 * ROW + threadIdx.x]) + (red[2 * ROW + threadIdx.x] + red[3 * ROW + threadIdx.x]);
    // Rows cross XCDs inside one launch: 8-byte agent-scope atomics on both sides (write-through store, L1-bypassing load) are
    // a complete hand-off on gfx950 -- no release / acquire fence, which would write back and invalidate the XCD's L2 once per
    // BLOCK (~3.5 us each, serialised per XCD: what made a probe slower the more blocks it had).  The store only has to have
    // left the wavefront before the block takes its ticket.
    __hip_atomic_store(mine + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__) && !defined(PP_ALLOW_FENCE_FREE_HANDOVER)
#error "the fence-free row hand-over above is a property of gfx950's L2 / memory-scope behaviour: re-derive it (or put the release / acquire fences back) before building for another target"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  __syncthreads();
  // Two-level ticket: block b counts on counter b % G (blocks are dealt to the XCDs round-robin, so a counter's increments
  // come from one XCD and the 1024 returning atomics of a launch do not queue on one word, ~11 ns each); the last block of
  // a group counts on the top counter, the last of those folds.  Counters sit 128 bytes apart.
  if (threadIdx.x == 0) {
    const unsigned G = gridDim.x < 8u ? gridDim.x : 8u, g = blockIdx.x % G;
    const unsigned members = (gridDim.x - g + G - 1u) / G;
    int last = 0;
    if (__hip_atomic_fetch_add(ticket + (1u + g) * TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
      __hip_atomic_store(ticket + (1u + g) * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1u;
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  {
    const int col = (int)threadIdx.x % ROW, part = (int)threadIdx.x / ROW;
    double p = 0.0;
    // (sixteen rows requested before the first is added: one dependent L2-miss load per row made the fold the longest part of a
    // probe -- ~0.24 us per row and part, 31 us of a 1024-block launch.  Same order of additions.)
    if (part < NPARTS)
      for (unsigned i = (unsigned)part; i < gridDim.x; i += NPARTS * 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned r = i + (unsigned)u * NPARTS;
          v[u] = __hip_atomic_load(partials + (size_t)(r < gridDim.x ? r : i) * ROW + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v[u] = r < gridDim.x ? v[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) p += v[u];
      }
    __syncthreads();   // (red still holds the block's own sums)
    if (part < NPARTS) red[part * ROW + col] = p;
    __syncthreads();
    if ((int)threadIdx.x < ncand * NV) {
      double tot = 0.0;
      for (int q = 0; q < NPARTS; ++q) tot += red[q * ROW + threadIdx.x];
      const int cand = (int)threadIdx.x / NV, f = (int)threadIdx.x % NV;
      pp_mail_post(pp_mail_slot(mailbox, cand / CPW) + (cand % CPW) * NV + f, tot, seq);
    }
  }
  if (threadIdx.x == 0) atomicExch(ticket, 0u);
}


// ---- metric value + gradient, second generation: one launch ---------------------------------------------------
// k_metric_affine + k_sum14_final with the lessons of the probe kernel: a thread's samples go four (two for the 42
// correlation sums) at a time with every corner request issued before the first is used, the block folds through wavefront shuffles
// instead of forty barrier rounds, and the last block to finish (two-level ticket, fence-free rows) folds the rows and
// posts the sums itself: no second launch.  Per-sample terms and accumulators are those of k_metric_affine; a thread adds
// its samples in increasing order, lanes combine by mg_wave_sum16's fixed tree, wavefronts as (w0 + w1) + (w2 + w3),
// block rows by the fixed interleaved fold of the probe kernel.
// Sum of sixteen per-lane doubles over the 64 lanes of a wavefront by "transpose and add": at xor distance 32 a lane keeps
// eight of its values and hands the other eight to its partner, at 16 four, at 8 two, at 4 one -- then two plain butterfly
// steps.  17 exchanges instead of 16 x 6; lanes 4 q .. 4 q + 3 end up holding the total of value q.  A fixed tree.
__device__ __forceinline__ double mg_wave_sum16(double* v, int lane) {
#pragma unroll
  for (int half = 8, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const double keep = up ? v[i + half] : v[i], send = up ? v[i] : v[i + half];
      v[i] = keep + __shfl_xor(send, bit);
    }
  }
  double t = v[0];
  t += __shfl_xor(t, 2);
  t += __shfl_xor(t, 1);
  return t;
}

struct mg_corners {
  float a000, a100, a010, a110, a001, a101, a011, a111;
  float wx, wy, wz;
};
// GI 1: the moving-image gradient is sampled from ITK's filtered gradient image (msq_args::grad) instead of being derived from the
// eight intensity corners -- its own instance, so that the default path keeps its registers.
// GI 2: the same from the PACKED image (msq_args::grad4: gradient and intensity of a voxel in one 16-byte element): eight gathers a
// sample instead of thirty-two, and a quarter of the cache sectors -- with ITK's jittered sample points every sample has its
// own rows, the planar form fetched ~0.75 KB of sectors per sample and ran at the cache-miss rate (profiles/round6_linear_pmc.md:
// 80 % L2 misses, 138 us at the 128 x 128 x 64 lattice).  Same interpolation arithmetic, so the same bits.
template <int MODE, int GI = 0>
__global__ void __launch_bounds__(NT) k_metric_grad(const float* __restrict__ F, pp_dims df, const float* __restrict__ M, pp_dims dm,
                                                    const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ mmask, msq_args a,
                                                    double* partials /* [grid][NACC] */, unsigned* __restrict__ ticket, void* mailbox,
                                                    unsigned long long seq, const float* __restrict__ fsamp) {
  constexpr int NACC = MODE == 0 ? 14 : 42;
  constexpr int G = (MODE == 0 && GI != 2) ? 4 : 2;   // samples in flight per thread (packed corners are four registers each)
  constexpr int NPARTS = NT / NACC;
  __shared__ double red[(NT / 64 > NPARTS ? NT / 64 : NPARTS) * NACC];
  __shared__ int is_last;
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  const size_t tid = (size_t)blockIdx.x * NT + threadIdx.x, nthr = (size_t)gridDim.x * NT;
  const bool small = nvirt < ((size_t)1 << 31);
  const size_t sy = dm.nx, sz = (size_t)dm.nx * dm.ny;
  for (size_t e0 = tid; e0 < nsamp; e0 += nthr * G) {
    mg_corners g[G];
    double v[G][3];
    float fval[G];
    bool ok[G];
    float gimg[GI ? G : 1][3];
    float4 q[GI == 2 ? G : 1][GI == 2 ? 8 : 1];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const size_t eq = e0 + (size_t)j * nthr;
      ok[j] = eq < nsamp;
      const size_t e = ok[j] ? eq : nsamp - 1;
      const size_t lin = e * (size_t)a.stride;
      if (small) {   // (uniform; 32-bit divisions)
        const unsigned l32 = (unsigned)lin, vx = (unsigned)a.vsize[0], vy = (unsigned)a.vsize[1];
        const unsigned q = l32 / vx;
        v[j][0] = (double)(l32 - q * vx);
        v[j][1] = (double)(q % vy);
        v[j][2] = (double)(q / vy);
      } else {
        v[j][0] = (double)(lin % a.vsize[0]);
        v[j][1] = (double)((lin / a.vsize[0]) % a.vsize[1]);
        v[j][2] = (double)(lin / ((size_t)a.vsize[0] * a.vsize[1]));
      }
      msq_jitter(a.jit, e, v[j]);
      double cf[3], cm[3];
      for (int r = 0; r < 3; ++r) {
        cf[r] = a.Af[r * 3 + 0] * v[j][0] + a.Af[r * 3 + 1] * v[j][1] + a.Af[r * 3 + 2] * v[j][2] + a.bf[r];
        cm[r] = a.Am[r * 3 + 0] * v[j][0] + a.Am[r * 3 + 1] * v[j][1] + a.Am[r * 3 + 2] * v[j][2] + a.bm[r];
      }
      int bf_[3] = {0, 0, 0}, bm_[3] = {0, 0, 0};
      float ff[3] = {0.0f, 0.0f, 0.0f}, fm[3] = {0.0f, 0.0f, 0.0f};
      fval[j] = 0.0f;
      if (fsamp) {   // (uniform) the fixed side of this sample was evaluated once for the level (same arithmetic)
        fval[j] = fsamp[e];
        ok[j] = ok[j] && __builtin_bit_cast(unsigned, fval[j]) != PP_FSAMP_INVALID;
      } else {
        bool okf = msq_locate(cf, df, bf_, ff);
        if (okf && fmask) {
          const int qx = (int)floor(cf[0] + 0.5), qy = (int)floor(cf[1] + 0.5), qz = (int)floor(cf[2] + 0.5);
          okf = fmask[((size_t)qz * df.ny + qy) * df.nx + qx] != 0;
        }
        if (okf) fval[j] = pp_trilinear(F, df.nx, df.ny, df.nz, bf_[0], ff[0], bf_[1], ff[1], bf_[2], ff[2]);
        ok[j] = ok[j] && okf;
      }
      const bool in = msq_locate(cm, dm, bm_, fm);   // (outside: base 0, fraction 0 -- a valid address, nothing accumulated)
      ok[j] = ok[j] && in;
      if (mmask) {   // (uniform)
        const int qx = in ? (int)floor(cm[0] + 0.5) : 0, qy = in ? (int)floor(cm[1] + 0.5) : 0, qz = in ? (int)floor(cm[2] + 0.5) : 0;
        ok[j] = ok[j] && mmask[((size_t)qz * dm.ny + qy) * dm.nx + qx] != 0;
      }
      int x0, x1, y0, y1, z0, z1;
      pp_axis_setup(bm_[0], fm[0], dm.nx, x0, x1, g[j].wx);
      pp_axis_setup(bm_[1], fm[1], dm.ny, y0, y1, g[j].wy);
      pp_axis_setup(bm_[2], fm[2], dm.nz, z0, z1, g[j].wz);
      if constexpr (GI == 2) {
        const float4* const P = a.grad4;
        q[j][0] = P[z0 * sz + y0 * sy + x0]; q[j][1] = P[z0 * sz + y0 * sy + x1];
        q[j][2] = P[z0 * sz + y1 * sy + x0]; q[j][3] = P[z0 * sz + y1 * sy + x1];
        q[j][4] = P[z1 * sz + y0 * sy + x0]; q[j][5] = P[z1 * sz + y0 * sy + x1];
        q[j][6] = P[z1 * sz + y1 * sy + x0]; q[j][7] = P[z1 * sz + y1 * sy + x1];
      } else {
        g[j].a000 = M[z0 * sz + y0 * sy + x0]; g[j].a100 = M[z0 * sz + y0 * sy + x1];
        g[j].a010 = M[z0 * sz + y1 * sy + x0]; g[j].a110 = M[z0 * sz + y1 * sy + x1];
        g[j].a001 = M[z1 * sz + y0 * sy + x0]; g[j].a101 = M[z1 * sz + y0 * sy + x1];
        g[j].a011 = M[z1 * sz + y1 * sy + x0]; g[j].a111 = M[z1 * sz + y1 * sy + x1];
      }
      if constexpr (GI == 1) msq_gradient_image(a.grad, dm, x0, x1, y0, y1, z0, z1, g[j].wx, g[j].wy, g[j].wz, gimg[j]);
    }
    if constexpr (GI == 2) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);   // every corner request before the first interpolation
#endif
#pragma unroll
      for (int j = 0; j < G; ++j) {
        g[j].a000 = q[j][0].w; g[j].a100 = q[j][1].w; g[j].a010 = q[j][2].w; g[j].a110 = q[j][3].w;
        g[j].a001 = q[j][4].w; g[j].a101 = q[j][5].w; g[j].a011 = q[j][6].w; g[j].a111 = q[j][7].w;
        const float wx = g[j].wx, wy = g[j].wy, wz = g[j].wz;
#pragma unroll
        for (int r = 0; r < 3; ++r) {   // msq_gradient_image's lerps on the packed corners
          const float a000 = r == 0 ? q[j][0].x : (r == 1 ? q[j][0].y : q[j][0].z), a100 = r == 0 ? q[j][1].x : (r == 1 ? q[j][1].y : q[j][1].z);
          const float a010 = r == 0 ? q[j][2].x : (r == 1 ? q[j][2].y : q[j][2].z), a110 = r == 0 ? q[j][3].x : (r == 1 ? q[j][3].y : q[j][3].z);
          const float a001 = r == 0 ? q[j][4].x : (r == 1 ? q[j][4].y : q[j][4].z), a101 = r == 0 ? q[j][5].x : (r == 1 ? q[j][5].y : q[j][5].z);
          const float a011 = r == 0 ? q[j][6].x : (r == 1 ? q[j][6].y : q[j][6].z), a111 = r == 0 ? q[j][7].x : (r == 1 ? q[j][7].y : q[j][7].z);
          const float v00 = a000 + (a100 - a000) * wx, v10 = a010 + (a110 - a010) * wx;
          const float v01 = a001 + (a101 - a001) * wx, v11 = a011 + (a111 - a011) * wx;
          const float v0 = v00 + (v10 - v00) * wy, v1 = v01 + (v11 - v01) * wy;
          gimg[j][r] = v0 + (v1 - v0) * wz;
        }
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // every corner request before the first interpolation (see k_metric_values_lanes)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < G; j += 2)
      asm volatile("" : "+v"(g[j].a000), "+v"(g[j].a100), "+v"(g[j].a010), "+v"(g[j].a110), "+v"(g[j].a001), "+v"(g[j].a101),
                        "+v"(g[j].a011), "+v"(g[j].a111), "+v"(g[j + 1].a000), "+v"(g[j + 1].a100), "+v"(g[j + 1].a010),
                        "+v"(g[j + 1].a110), "+v"(g[j + 1].a001), "+v"(g[j + 1].a101), "+v"(g[j + 1].a011), "+v"(g[j + 1].a111));
#endif
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const mg_corners& c = g[j];
      const float v00 = c.a000 + (c.a100 - c.a000) * c.wx, v10 = c.a010 + (c.a110 - c.a010) * c.wx;
      const float v01 = c.a001 + (c.a101 - c.a001) * c.wx, v11 = c.a011 + (c.a111 - c.a011) * c.wx;
      const float v0 = v00 + (v10 - v00) * c.wy, v1 = v01 + (v11 - v01) * c.wy;
      const float m = v0 + (v1 - v0) * c.wz;
      const float gx0 = (c.a100 - c.a000) + ((c.a110 - c.a010) - (c.a100 - c.a000)) * c.wy;
      const float gx1 = (c.a101 - c.a001) + ((c.a111 - c.a011) - (c.a101 - c.a001)) * c.wy;
      const float gx = GI ? gimg[GI ? j : 0][0] : gx0 + (gx1 - gx0) * c.wz;
      const float gy = GI ? gimg[GI ? j : 0][1] : (v10 - v00) + ((v11 - v01) - (v10 - v00)) * c.wz;
      const float gz = GI ? gimg[GI ? j : 0][2] : v1 - v0;
      const double w = ok[j] ? 1.0 : 0.0;   // (a rejected sample adds exact zeros -- also when the voxels under its clamped
      // position hold Inf / NaN padding: the gradient is SELECTED away, not multiplied by zero)
      const float gx_ = ok[j] ? gx : 0.0f, gy_ = ok[j] ? gy : 0.0f, gz_ = ok[j] ? gz : 0.0f;
      if constexpr (MODE == 0) {
        const double diff = ok[j] ? (double)fval[j] - (double)m : 0.0;
        acc[0] += diff * diff;
        acc[1] += w;
        const double sc = -2.0 * diff;
        const double gg[3] = {sc * gx_, sc * gy_, sc * gz_};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          acc[2 + r * 3 + 0] += gg[r] * v[j][0];
          acc[2 + r * 3 + 1] += gg[r] * v[j][1];
          acc[2 + r * 3 + 2] += gg[r] * v[j][2];
          acc[11 + r] += gg[r];
        }
      } else {
        const double fd = ok[j] ? (double)fval[j] : 0.0, md = ok[j] ? (double)m : 0.0;
        acc[0] += w;
        acc[1] += fd;
        acc[2] += md;
        acc[3] += fd * fd;
        acc[4] += md * md;
        acc[5] += fd * md;
        const double gg[3] = {w * gx_, w * gy_, w * gz_};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double t = gg[r] * (q < 3 ? v[j][q < 3 ? q : 0] : 1.0);
            const int slot = q < 3 ? r * 3 + q : 9 + r;
            acc[6 + slot] += t;
            acc[18 + slot] += fd * t;
            acc[30 + slot] += md * t;
          }
      }
    }
  }
  const int wave = (int)threadIdx.x / 64, lane = (int)threadIdx.x % 64;
#pragma unroll
  for (int k0 = 0; k0 < NACC; k0 += 16) {
    double v16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v16[i] = k0 + i < NACC ? acc[k0 + i] : 0.0;
    const double t = mg_wave_sum16(v16, lane);
    if ((lane & 3) == 0 && k0 + (lane >> 2) < NACC) red[wave * NACC + k0 + (lane >> 2)] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < NACC) {
    const double t = (red[0 * NACC + threadIdx.x] + red[1 * NACC + threadIdx.x]) + (red[2 * NACC + threadIdx.x] + red[3 * NACC + threadIdx.x]);
    __hip_atomic_store(partials + (size_t)blockIdx.x * NACC + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned GR = gridDim.x < 8u ? gridDim.x : 8u, gi = blockIdx.x % GR;
    const unsigned members = (gridDim.x - gi + GR - 1u) / GR;
    int last = 0;
    if (__hip_atomic_fetch_add(ticket + (1u + gi) * TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
      __hip_atomic_store(ticket + (1u + gi) * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GR - 1u;
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  {
    const int col = (int)threadIdx.x % NACC, part = (int)threadIdx.x / NACC;
    double p = 0.0;
    if (part < NPARTS)
      for (unsigned i = (unsigned)part; i < gridDim.x; i += NPARTS * 16) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned r = i + (unsigned)u * NPARTS;
          t[u] = __hip_atomic_load(partials + (size_t)(r < gridDim.x ? r : i) * NACC + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          t[u] = r < gridDim.x ? t[u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) p += t[u];
      }
    __syncthreads();
    if (part < NPARTS) red[part * NACC + col] = p;
    __syncthreads();
    if ((int)threadIdx.x < NACC) {
      double tot = 0.0;
      for (int q = 0; q < NPARTS; ++q) tot += red[q * NACC + threadIdx.x];
      pp_mail_post(pp_mail_slot(mailbox, 0) + threadIdx.x, tot, seq);
    }
  }
  if (threadIdx.x == 0) atomicExch(ticket, 0u);
}

// ---- mutual-information metrics (linear.py:145-148: mattes_mi, joint_hist_mi) -------------------------------
// Two passes over the same sample lattice as the metrics above.  Pass 1 builds the joint intensity histogram of the
// sample pairs: per block in LDS, in 64-bit fixed point (2^-32 units) so that the sum is associative and the result
// does not depend on scheduling, then one 64-bit atomic per non-empty bin into the global table.  The host turns it into
// PDFs, the metric value and a per-bin score table (log-ratio of the PDFs: host arithmetic on <= 64 x 64 numbers).  Pass 2
// weights every sample's interpolant gradient by its score derivative: sum_k dkernel_k(sample) * table[f_bin][k].
// PP_MI_MATTES: fixed intensity to its nearest bin, moving intensity spread over 4 bins with a cubic B-spline Parzen
// window (itk::MattesMutualInformationImageToImageMetricv4); PP_MI_JOINT: both to their nearest bin, score differenced
// between the two neighbouring moving-bin centres (joint-histogram MI; the PDF smoothing is the host's).
constexpr int MI_MAX_BINS = 64;
struct mi_args {
  double Af[9], bf[3], Am[9], bm[3];
  int vsize[3];
  int stride;
  int nbins, kernel;
  double f_bin, f_norm_min, m_bin, m_norm_min;
  const float* jit;   // (as msq_args)
  const float* grad;  // (as msq_args)
};

__device__ __forceinline__ double mi_bspline3(double u) {   // cubic B-spline, support (-2, 2)
  const double a = fabs(u);
  if (a < 1.0) return (4.0 - 6.0 * a * a + 3.0 * a * a * a) / 6.0;
  if (a < 2.0) { const double t = 2.0 - a; return t * t * t / 6.0; }
  return 0.0;
}
__device__ __forceinline__ double mi_bspline3_deriv(double u) {
  const double a = fabs(u), sg = u < 0.0 ? -1.0 : 1.0;
  if (a < 1.0) return sg * (-2.0 * a + 1.5 * a * a);
  if (a < 2.0) { const double t = 2.0 - a; return sg * (-0.5 * t * t); }
  return 0.0;
}

// -> valid; f / m values (trilinear), moving interpolant gradient (moving index units), virtual index v
__device__ __forceinline__ bool mi_sample(const float* __restrict__ F, const pp_dims& df, const float* __restrict__ M, const pp_dims& dm,
                                          const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ mmask, const mi_args& a, size_t e,
                                          double v[3], float& fval, float& mval, float g[3]) {
  const size_t lin = e * (size_t)a.stride;
  v[0] = (double)(lin % a.vsize[0]);
  v[1] = (double)((lin / a.vsize[0]) % a.vsize[1]);
  v[2] = (double)(lin / ((size_t)a.vsize[0] * a.vsize[1]));
  msq_jitter(a.jit, e, v);
  double cf[3], cm[3];
  for (int r = 0; r < 3; ++r) {
    cf[r] = a.Af[r * 3 + 0] * v[0] + a.Af[r * 3 + 1] * v[1] + a.Af[r * 3 + 2] * v[2] + a.bf[r];
    cm[r] = a.Am[r * 3 + 0] * v[0] + a.Am[r * 3 + 1] * v[1] + a.Am[r * 3 + 2] * v[2] + a.bm[r];
  }
  int bf_[3], bm_[3];
  float ff[3], fm[3];
  if (!msq_locate(cf, df, bf_, ff) || !msq_locate(cm, dm, bm_, fm)) return false;
  if (fmask) {
    const int qx = (int)floor(cf[0] + 0.5), qy = (int)floor(cf[1] + 0.5), qz = (int)floor(cf[2] + 0.5);
    if (!fmask[((size_t)qz * df.ny + qy) * df.nx + qx]) return false;
  }
  if (mmask) {
    const int qx = (int)floor(cm[0] + 0.5), qy = (int)floor(cm[1] + 0.5), qz = (int)floor(cm[2] + 0.5);
    if (!mmask[((size_t)qz * dm.ny + qy) * dm.nx + qx]) return false;
  }
  fval = pp_trilinear(F, df.nx, df.ny, df.nz, bf_[0], ff[0], bf_[1], ff[1], bf_[2], ff[2]);
  int x0, x1, y0, y1, z0, z1;
  float wx, wy, wz;
  pp_axis_setup(bm_[0], fm[0], dm.nx, x0, x1, wx);
  pp_axis_setup(bm_[1], fm[1], dm.ny, y0, y1, wy);
  pp_axis_setup(bm_[2], fm[2], dm.nz, z0, z1, wz);
  const size_t sy = dm.nx, sz = (size_t)dm.nx * dm.ny;
  const float a000 = M[z0 * sz + y0 * sy + x0], a100 = M[z0 * sz + y0 * sy + x1];
  const float a010 = M[z0 * sz + y1 * sy + x0], a110 = M[z0 * sz + y1 * sy + x1];
  const float a001 = M[z1 * sz + y0 * sy + x0], a101 = M[z1 * sz + y0 * sy + x1];
  const float a011 = M[z1 * sz + y1 * sy + x0], a111 = M[z1 * sz + y1 * sy + x1];
  const float v00 = a000 + (a100 - a000) * wx, v10 = a010 + (a110 - a010) * wx;
  const float v01 = a001 + (a101 - a001) * wx, v11 = a011 + (a111 - a011) * wx;
  const float v0 = v00 + (v10 - v00) * wy, v1 = v01 + (v11 - v01) * wy;
  mval = v0 + (v1 - v0) * wz;
  const float gx0 = (a100 - a000) + ((a110 - a010) - (a100 - a000)) * wy;
  const float gx1 = (a101 - a001) + ((a111 - a011) - (a101 - a001)) * wy;
  g[0] = gx0 + (gx1 - gx0) * wz;
  g[1] = (v10 - v00) + ((v11 - v01) - (v10 - v00)) * wz;
  g[2] = v1 - v0;
  if (a.grad) msq_gradient_image(a.grad, dm, x0, x1, y0, y1, z0, z1, wx, wy, wz, g);   // (uniform; see msq_gradient_image)
  return true;
}

// bin of an intensity: nearest (zero-order) index clamped to [lo, hi], and the continuous bin coordinate
__device__ __forceinline__ int mi_bin(double value, double bin, double norm_min, int lo, int hi, double& term) {
  term = value / bin - norm_min;
  int i = (int)floor(term);
  return i < lo ? lo : (i > hi ? hi : i);
}

__global__ void __launch_bounds__(NT) k_mi_histogram(const float* __restrict__ F, pp_dims df, const float* __restrict__ M, pp_dims dm,
                                                     const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ mmask, mi_args a,
                                                     unsigned long long* __restrict__ hist /* [nbins^2 + 1], last = sample count */) {
  __shared__ unsigned long long sh[MI_MAX_BINS * MI_MAX_BINS + 1];
  const int nb2 = a.nbins * a.nbins;
  for (int i = threadIdx.x; i <= nb2; i += NT) sh[i] = 0ull;
  __syncthreads();
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  const int pad = a.kernel == 0 ? 2 : 0;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < nsamp; e += (size_t)gridDim.x * NT) {
    double v[3];
    float fval, mval, g[3];
    if (!mi_sample(F, df, M, dm, fmask, mmask, a, e, v, fval, mval, g)) continue;
    double tf, tm;
    const int fb = mi_bin((double)fval, a.f_bin, a.f_norm_min, pad, a.nbins - 1 - pad, tf);
    const int mb = mi_bin((double)mval, a.m_bin, a.m_norm_min, pad, a.nbins - 1 - pad, tm);
    if (a.kernel == 0) {
      for (int k = mb - 1; k <= mb + 2; ++k) {
        const double w = mi_bspline3((double)k - tm);
        atomicAdd(&sh[fb * a.nbins + k], (unsigned long long)(w * 4294967296.0 + 0.5));
      }
    } else {
      atomicAdd(&sh[fb * a.nbins + mb], 4294967296ull);
    }
    atomicAdd(&sh[nb2], 1ull);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= nb2; i += NT)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

__global__ void __launch_bounds__(NT) k_mi_gradient(const float* __restrict__ F, pp_dims df, const float* __restrict__ M, pp_dims dm,
                                                    const uint8_t* __restrict__ fmask, const uint8_t* __restrict__ mmask, mi_args a,
                                                    const float* __restrict__ table /* device, [nbins^2] */,
                                                    double* __restrict__ partials /* [grid][14] */) {
  __shared__ double red[3 * NT];
  __shared__ float tab[MI_MAX_BINS * MI_MAX_BINS];
  for (int i = threadIdx.x; i < a.nbins * a.nbins; i += NT) tab[i] = table[i];
  __syncthreads();
  double acc[14];
  for (int k = 0; k < 14; ++k) acc[k] = 0.0;
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  const int pad = a.kernel == 0 ? 2 : 0;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < nsamp; e += (size_t)gridDim.x * NT) {
    double v[3];
    float fval, mval, g[3];
    if (!mi_sample(F, df, M, dm, fmask, mmask, a, e, v, fval, mval, g)) continue;
    double tf, tm;
    const int fb = mi_bin((double)fval, a.f_bin, a.f_norm_min, pad, a.nbins - 1 - pad, tf);
    const int mb = mi_bin((double)mval, a.m_bin, a.m_norm_min, pad, a.nbins - 1 - pad, tm);
    double w = 0.0;
    if (a.kernel == 0) {
      for (int k = mb - 1; k <= mb + 2; ++k) w += mi_bspline3_deriv((double)k - tm) * (double)tab[fb * a.nbins + k];
    } else {
      int k0 = (int)floor(tm - 0.5);
      k0 = k0 < 0 ? 0 : (k0 > a.nbins - 2 ? a.nbins - 2 : k0);
      w = (double)tab[fb * a.nbins + k0 + 1] - (double)tab[fb * a.nbins + k0];
    }
    acc[1] += 1.0;
    for (int r = 0; r < 3; ++r) {
      const double gr = w * (double)g[r];
      acc[2 + r * 3 + 0] += gr * v[0];
      acc[2 + r * 3 + 1] += gr * v[1];
      acc[2 + r * 3 + 2] += gr * v[2];
      acc[11 + r] += gr;
    }
  }
  for (int k = 0; k < 14; k += 3) {
    double p = acc[k], q = k + 1 < 14 ? acc[k + 1] : 0.0, r = k + 2 < 14 ? acc[k + 2] : 0.0;
    pp_block_sum3<NT>(p, q, r, red);
    if (threadIdx.x == 0) {
      partials[(size_t)blockIdx.x * 14 + k] = p;
      if (k + 1 < 14) partials[(size_t)blockIdx.x * 14 + k + 1] = q;
      if (k + 2 < 14) partials[(size_t)blockIdx.x * 14 + k + 2] = r;
    }
    __syncthreads();
  }
}

// The context's jitter array for a lattice of `nsamp` samples (NULL when none is set); an array too short for the lattice is
// an error, never a silent read past its end.
static int pp_jitter_for(pp_ctx* ctx, size_t nsamp, const float** out) {
  *out = nullptr;
  if (!ctx->jitter) return PP_OK;
  PP_REQUIRE(ctx, ctx->jitter_samples >= nsamp, "metric: the sample-jitter array set by pp_linear_set_sample_jitter is shorter than the sampling lattice");
  *out = ctx->jitter;
  return PP_OK;
}

// The context's gradient image for a moving image of `msize` voxels (NULL when none is set); one set for another size is an error.
static int pp_gradient_for(pp_ctx* ctx, const int msize[3], const float** out) {
  *out = nullptr;
  if (!ctx->mgrad) return PP_OK;
  PP_REQUIRE(ctx, ctx->mgrad_size[0] == msize[0] && ctx->mgrad_size[1] == msize[1] && ctx->mgrad_size[2] == msize[2],
             "metric: the gradient image set by pp_linear_set_moving_gradient does not have the moving image's size");
  *out = ctx->mgrad;
  return PP_OK;
}

int mi_fill_args(pp_ctx* ctx, mi_args* a, const int fsize[3], const int msize[3], const double Af[9], const double bf[3], const double Am[9],
                 const double bm[3], const int vsize[3], int stride, const pp_mi_bins* bins) {
  PP_REQUIRE(ctx, fsize && msize && Af && bf && Am && bm && vsize && bins, "mutual information: NULL argument");
  PP_REQUIRE(ctx, stride >= 1 && vsize[0] >= 1 && vsize[1] >= 1 && vsize[2] >= 1, "mutual information: bad sampling lattice");
  PP_REQUIRE(ctx, bins->kernel == PP_MI_MATTES || bins->kernel == PP_MI_JOINT, "mutual information: unknown Parzen kernel");
  PP_REQUIRE(ctx, bins->nbins >= (bins->kernel == PP_MI_MATTES ? 5 : 2) && bins->nbins <= MI_MAX_BINS, "mutual information: 5..64 bins (2..64 without padding)");
  PP_REQUIRE(ctx, bins->f_bin > 0.0 && bins->m_bin > 0.0, "mutual information: bin widths must be positive");
  memcpy(a->Af, Af, sizeof(a->Af));
  memcpy(a->bf, bf, sizeof(a->bf));
  memcpy(a->Am, Am, sizeof(a->Am));
  memcpy(a->bm, bm, sizeof(a->bm));
  for (int k = 0; k < 3; ++k) a->vsize[k] = vsize[k];
  a->stride = stride;
  a->nbins = bins->nbins;
  a->kernel = bins->kernel;
  a->f_bin = bins->f_bin;
  a->f_norm_min = bins->f_norm_min;
  a->m_bin = bins->m_bin;
  a->m_norm_min = bins->m_norm_min;
  {
    const size_t nsamp = ((size_t)vsize[0] * vsize[1] * vsize[2] + stride - 1) / stride;
    const int jrc = pp_jitter_for(ctx, nsamp, &a->jit);
    if (jrc) return jrc;
    const int grc = pp_gradient_for(ctx, msize, &a->grad);
    if (grc) return grc;
  }
  return PP_OK;
}

}  // namespace

extern "C" {

int pp_weight_map_local_f32(pp_ctx* ctx, const float* target, const float* moving, const int size[3],
                            const double spacing[3], double sigma, double epsilon, float* weight) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, target && moving && weight && size && spacing, "pp_weight_map_local_f32: NULL argument");
  const size_t N = pp_nvox(size);
  PP_REQUIRE(ctx, N > 0, "pp_weight_map_local_f32: empty volume");
  launch_map4(ctx, N, all_aligned16(target, moving, weight), op_sqdiff{target, moving, weight});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_sqdiff>");
  const double var[3] = {sigma * sigma, sigma * sigma, sigma * sigma};
  // sitk.DiscreteGaussian defaults: maximumKernelWidth 32, maximumError 0.01, useImageSpacing True
  int rc = pp_discrete_gaussian_f32(ctx, weight, weight, size, spacing, var, 0.01, 32, 1);
  if (rc) return rc;
  launch_map4(ctx, N, all_aligned16(weight), op_inv_eps{weight, (float)epsilon});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_inv_eps>");
  return PP_OK;
}

int pp_weight_map_block_f32(pp_ctx* ctx, const float* target, const float* moving, const int size[3], const int radius[3],
                            double factor, double gain, float* weight) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, target && moving && weight && size && radius, "pp_weight_map_block_f32: NULL argument");
  const size_t N = pp_nvox(size);
  PP_REQUIRE(ctx, N > 0, "pp_weight_map_block_f32: empty volume");
  launch_map4(ctx, N, all_aligned16(target, moving, weight), op_sqdiff{target, moving, weight});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_sqdiff>");
  // sitk.BoxMean(square, radius): the mean over a (2r+1)^3 box with ZeroFluxNeumann edges is three 1-D means
  pp_taps taps[3];
  for (int a = 0; a < 3; ++a) {
    PP_REQUIRE(ctx, radius[a] >= 0 && radius[a] <= PP_MAX_RADIUS, "pp_weight_map_block_f32: box radius out of range");
    taps[a].r = radius[a];
    for (int k = 0; k < 2 * radius[a] + 1; ++k) taps[a].w[k] = 1.0f / (float)(2 * radius[a] + 1);
  }
  int rc = pp_reserve(ctx, 2 * pp_align_up(N * sizeof(float), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* t1 = cv.take<float>(N);
  float* t2 = cv.take<float>(N);
  const pp_dims d{size[0], size[1], size[2]};
  const int order[3] = {0, 1, 2};
  rc = pp_smooth3_staged(ctx, weight, nullptr, weight, t1, t2, d, 1, taps, order, nullptr);
  if (rc) return rc;
  launch_map4(ctx, N, all_aligned16(weight), op_block_weight{weight, (float)factor, (float)fabs(gain / 2.0)});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_block_weight>");
  return PP_OK;
}

int pp_sum_sq_diff_f32(pp_ctx* ctx, const float* a, const float* b, size_t n, double* result) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, a && b && result, "pp_sum_sq_diff_f32: NULL argument");
  const unsigned nb = grid_for(n, 2048u);
  int rc = pp_reserve(ctx, pp_align_up((nb + 1) * sizeof(double), 256));
  if (rc) return rc;
  double* partials = reinterpret_cast<double*>(ctx->ws);
  hipLaunchKernelGGL(k_ssd_partial, dim3(nb), dim3(NT), 0, ctx->stream, a, b, n, partials);
  PP_LAUNCH_CHECK(ctx, "k_ssd_partial");
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(NT), 0, ctx->stream, (const double*)partials, (int)nb, 1, 1, partials + nb);
  PP_LAUNCH_CHECK(ctx, "k_sum_final");
  return pp_read_back(ctx, partials + nb, result, sizeof(double));
}

int pp_fuse_accumulate_u8(pp_ctx* ctx, const float* weight, const uint8_t* label, float* wsum, float* wlsum, size_t n) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, weight && label && wlsum, "pp_fuse_accumulate_u8: NULL argument");
  // uint8 labels: 4 of them are one 4-byte access (alignment 4 suffices for that array)
  launch_map4(ctx, n, all_aligned16(weight, wsum, wlsum) && reinterpret_cast<uintptr_t>(label) % 4 == 0,
              op_fuse_accumulate<uint8_t>{weight, label, wsum, wlsum});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_fuse_accumulate<u8>>");
  return PP_OK;
}

int pp_fuse_accumulate_f32(pp_ctx* ctx, const float* weight, const float* label, float* wsum, float* wlsum, size_t n) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, weight && label && wlsum, "pp_fuse_accumulate_f32: NULL argument");
  launch_map4(ctx, n, all_aligned16(weight, wsum, wlsum, label), op_fuse_accumulate<float>{weight, label, wsum, wlsum});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_fuse_accumulate<f32>>");
  return PP_OK;
}

int pp_fuse_divide_f32(pp_ctx* ctx, const float* wlsum, const float* wsum, float* out, size_t n) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, wlsum && wsum && out, "pp_fuse_divide_f32: NULL argument");
  launch_map4(ctx, n, all_aligned16(wlsum, wsum, out), op_fuse_divide{wlsum, wsum, out});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_fuse_divide>");
  return PP_OK;
}

int pp_minmax_f32(pp_ctx* ctx, const float* in, size_t n, float* min_out, float* max_out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && min_out && max_out && n > 0, "pp_minmax_f32: NULL or empty argument");
  const unsigned nb = grid_for(n, 2048u);
  int rc = pp_reserve(ctx, pp_align_up((2 * nb + 2) * sizeof(float), 256));
  if (rc) return rc;
  float* partials = reinterpret_cast<float*>(ctx->ws);
  hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(NT), 0, ctx->stream, in, n, partials);
  PP_LAUNCH_CHECK(ctx, "k_minmax_partial");
  hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(NT), 0, ctx->stream, (const float*)partials, (int)nb, partials + 2 * nb);
  PP_LAUNCH_CHECK(ctx, "k_minmax_final");
  float h[2];
  rc = pp_read_back(ctx, partials + 2 * nb, h, sizeof(h));
  if (rc) return rc;
  *min_out = h[0];
  *max_out = h[1];
  return PP_OK;
}

int pp_rescale_threshold_f32(pp_ctx* ctx, float* data, size_t n, float in_min, float in_max, float lower) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, data, "pp_rescale_threshold_f32: NULL argument");
  // RescaleIntensityImageFilter::BeforeThreadedGenerateData, output range [0, 1]
  double scale;
  if (in_min != in_max) scale = 1.0 / ((double)in_max - (double)in_min);
  else if (in_max != 0.0f) scale = 1.0 / (double)in_max;
  else scale = 0.0;
  const double shift = 0.0 - (double)in_min * scale;
  launch_map4(ctx, n, all_aligned16(data), op_rescale_threshold{data, scale, shift, lower});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_rescale_threshold>");
  return PP_OK;
}

int pp_binary_threshold_f32(pp_ctx* ctx, const float* prob, size_t n, double max_value, double threshold, uint8_t* out) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, prob && out, "pp_binary_threshold_f32: NULL argument");
  launch_map4(ctx, n, all_aligned16(prob) && reinterpret_cast<uintptr_t>(out) % 4 == 0, op_binary_threshold{prob, max_value, threshold, out});
  PP_LAUNCH_CHECK(ctx, "k_map4<op_binary_threshold>");
  return PP_OK;
}

// The fixed side of every sample of a metric lattice: value of the trilinear interpolant, or PP_FSAMP_INVALID where the
// point leaves the fixed buffer or its mask (the tests the metric kernels make, in their order).
__global__ void __launch_bounds__(NT) k_fixed_samples(const float* __restrict__ F, pp_dims df, const uint8_t* __restrict__ fmask, msq_args a,
                                                      float* __restrict__ fsamp) {
  const size_t nvirt = (size_t)a.vsize[0] * a.vsize[1] * a.vsize[2];
  const size_t nsamp = (nvirt + a.stride - 1) / a.stride;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < nsamp; e += (size_t)gridDim.x * NT) {
    const size_t lin = e * (size_t)a.stride;
    double v[3] = {(double)(lin % a.vsize[0]), (double)((lin / a.vsize[0]) % a.vsize[1]),
                   (double)(lin / ((size_t)a.vsize[0] * a.vsize[1]))};
    msq_jitter(a.jit, e, v);
    double cf[3];
    for (int r = 0; r < 3; ++r) cf[r] = a.Af[r * 3 + 0] * v[0] + a.Af[r * 3 + 1] * v[1] + a.Af[r * 3 + 2] * v[2] + a.bf[r];
    int bf_[3];
    float ff[3];
    bool ok = msq_locate(cf, df, bf_, ff);
    if (ok && fmask) {
      const int qx = (int)floor(cf[0] + 0.5), qy = (int)floor(cf[1] + 0.5), qz = (int)floor(cf[2] + 0.5);
      ok = fmask[((size_t)qz * df.ny + qy) * df.nx + qx] != 0;
    }
    fsamp[e] = ok ? msq_trilinear_pairs(F, df.nx, df.ny, df.nz, bf_[0], ff[0], bf_[1], ff[1], bf_[2], ff[2]) : __builtin_bit_cast(float, PP_FSAMP_INVALID);
  }
}

static int pp_fixed_samples(pp_ctx* ctx, const float* fixed, const int fsize[3], const double Af[9], const double bf[3], const int vsize[3],
                            int stride, const unsigned char* fmask, const float** out) {
  *out = nullptr;
  if (ctx->fsamp_scope <= 0 || pp_env("PP_NO_FIXED_SAMPLES")) return PP_OK;
  auto& k = ctx->fsamp_key;
  const bool same = ctx->fsamp_valid && k.fixed == fixed && k.fmask == fmask && k.stride == stride &&
                    memcmp(k.fsize, fsize, sizeof(k.fsize)) == 0 && memcmp(k.vsize, vsize, sizeof(k.vsize)) == 0 &&
                    memcmp(k.Af, Af, sizeof(k.Af)) == 0 && memcmp(k.bf, bf, sizeof(k.bf)) == 0 && k.jitter == ctx->jitter &&
                    k.jitter_gen == ctx->jitter_gen;
  if (!same) {
    const size_t nsamp = ((size_t)vsize[0] * vsize[1] * vsize[2] + stride - 1) / stride;
    const float* jit = nullptr;
    const int jrc = pp_jitter_for(ctx, nsamp, &jit);
    if (jrc) return jrc;
    if (nsamp > ctx->fsamp_cap) {
      if (ctx->fsamp) (void)hipFree(ctx->fsamp);
      ctx->fsamp = nullptr;
      ctx->fsamp_cap = 0;
      void* p = nullptr;
      if (hipMalloc(&p, nsamp * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        return PP_OK;   // (no cache: the metric kernels sample the fixed image themselves)
      }
      ctx->fsamp = static_cast<float*>(p);
      ctx->fsamp_cap = nsamp;
    }
    msq_args a;
    memset(&a, 0, sizeof(a));
    memcpy(a.Af, Af, sizeof(a.Af));
    memcpy(a.bf, bf, sizeof(a.bf));
    for (int i = 0; i < 3; ++i) a.vsize[i] = vsize[i];
    a.stride = stride;
    a.jit = jit;
    const pp_dims df{fsize[0], fsize[1], fsize[2]};
    hipLaunchKernelGGL(k_fixed_samples, dim3(grid_for(nsamp, 2048u)), dim3(NT), 0, ctx->stream, fixed, df, fmask, a, ctx->fsamp);
    PP_LAUNCH_CHECK(ctx, "k_fixed_samples");
    k.fixed = fixed;
    k.fmask = fmask;
    k.stride = stride;
    memcpy(k.fsize, fsize, sizeof(k.fsize));
    memcpy(k.vsize, vsize, sizeof(k.vsize));
    memcpy(k.Af, Af, sizeof(k.Af));
    memcpy(k.bf, bf, sizeof(k.bf));
    k.jitter = ctx->jitter;
    k.jitter_gen = ctx->jitter_gen;
    ctx->fsamp_valid = 1;
  }
  *out = ctx->fsamp;
  return PP_OK;
}

static int metric_affine(pp_ctx* ctx, int mode, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                         const double Af[9], const double bf[3], const double Am[9], const double bm[3], const int vsize[3],
                         int stride, const uint8_t* fixed_mask, const uint8_t* moving_mask, double* result) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && fsize && moving && msize && Af && bf && Am && bm && vsize && result, "metric: NULL argument");
  PP_REQUIRE(ctx, stride >= 1 && vsize[0] >= 1 && vsize[1] >= 1 && vsize[2] >= 1, "metric: bad sampling lattice");
  const int nacc = mode == 0 ? 14 : 42;
  msq_args a;
  memcpy(a.Af, Af, sizeof(a.Af));
  memcpy(a.bf, bf, sizeof(a.bf));
  memcpy(a.Am, Am, sizeof(a.Am));
  memcpy(a.bm, bm, sizeof(a.bm));
  for (int k = 0; k < 3; ++k) a.vsize[k] = vsize[k];
  a.stride = stride;
  const size_t nsamp = ((size_t)vsize[0] * vsize[1] * vsize[2] + stride - 1) / stride;
  {
    const int jrc = pp_jitter_for(ctx, nsamp, &a.jit);
    if (jrc) return jrc;
    const int grc = pp_gradient_for(ctx, msize, &a.grad);
    if (grc) return grc;
    // (the packed companion is set together with the planar image, for the same moving image: pp_gradient_for has checked the size)
    a.grad4 = (a.grad && ctx->mgrad4 && !pp_env("PP_METRIC_GRAD_PLANAR")) ? reinterpret_cast<const float4*>(ctx->mgrad4) : nullptr;
  }
  const pp_dims df{fsize[0], fsize[1], fsize[2]}, dm{msize[0], msize[1], msize[2]};
  const unsigned nb = grid_for(nsamp, 512u);   // (256 measures the same, 128 slower: profiles/round3_metric_probe_latency.txt)
  int rc = pp_reserve(ctx, pp_align_up((size_t)nb * nacc * sizeof(double), 256));
  if (rc) return rc;
  double* partials = reinterpret_cast<double*>(ctx->ws);
  const float* fsamp = nullptr;
  rc = pp_fixed_samples(ctx, fixed, fsize, Af, bf, vsize, stride, fixed_mask, &fsamp);
  if (rc) return rc;
  const char* one_env = pp_env("PP_METRIC_GRAD_ONE_LAUNCH");   // (0: k_metric_affine + k_sum14_final, for A/B runs and the equality test)
  if (!one_env || atoi(one_env) != 0) {
    unsigned* ticket = nullptr;
    rc = pp_ticket(ctx, &ticket);
    if (rc) return rc;
    char* mail1 = nullptr;
    unsigned long long seq1 = 0;
    rc = pp_mailbox(ctx, &mail1, &seq1);
    if (rc) return rc;
#define PP_MG_GO(MODEV, GIV)                                                                                                              \
  hipLaunchKernelGGL((k_metric_grad<MODEV, GIV>), dim3(nb), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a, \
                     partials, ticket, mail1, seq1, fsamp)
    if (a.grad4) { if (mode == 0) PP_MG_GO(0, 2); else PP_MG_GO(1, 2); }
    else if (a.grad) { if (mode == 0) PP_MG_GO(0, 1); else PP_MG_GO(1, 1); }
    else { if (mode == 0) PP_MG_GO(0, 0); else PP_MG_GO(1, 0); }
#undef PP_MG_GO
    PP_LAUNCH_CHECK(ctx, "k_metric_grad");
    return pp_mail_take(ctx, 0, nacc, seq1, result);
  }
  if (mode == 0)
    hipLaunchKernelGGL((k_metric_affine<0>), dim3(nb), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a, partials,
                       fsamp);
  else
    hipLaunchKernelGGL((k_metric_affine<1>), dim3(nb), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a, partials,
                       fsamp);
  PP_LAUNCH_CHECK(ctx, "k_metric_affine");
  char* mail = nullptr;
  unsigned long long seq = 0;
  rc = pp_mailbox(ctx, &mail, &seq);
  if (rc) return rc;
  const int nfold = (nacc + 13) / 14;
  hipLaunchKernelGGL(k_sum14_final, dim3(nfold), dim3(NT), 0, ctx->stream, (const double*)partials, (int)nb, nacc,
                     reinterpret_cast<double*>(mail), 1, seq);
  PP_LAUNCH_CHECK(ctx, "k_sum14_final");
  for (int b = 0; b < nfold; ++b) {
    rc = pp_mail_take(ctx, b, nacc - 14 * b < 14 ? nacc - 14 * b : 14, seq, result + 14 * b);
    if (rc) return rc;
  }
  return PP_OK;
}

int pp_meansq_affine_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                         const double Af[9], const double bf[3], const double Am[9], const double bm[3],
                         const int vsize[3], int stride, const uint8_t* fixed_mask, const uint8_t* moving_mask,
                         double* result) {
  pp_device_guard dev_guard_(ctx);
  return metric_affine(ctx, 0, fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask, result);
}

int pp_corr_moments_affine_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                               const double Af[9], const double bf[3], const double Am[9], const double bm[3],
                               const int vsize[3], int stride, const uint8_t* fixed_mask, const uint8_t* moving_mask,
                               double* result) {
  pp_device_guard dev_guard_(ctx);
  return metric_affine(ctx, 1, fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask, result);
}

int pp_metric_values_affine_f32(pp_ctx* ctx, int metric, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                                const double Af[9], const double bf[3], int ncand, const double* Am, const double* bm,
                                const int vsize[3], int stride, const uint8_t* fixed_mask, const uint8_t* moving_mask, double* result) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && fsize && moving && msize && Af && bf && Am && bm && vsize && result, "metric values: NULL argument");
  PP_REQUIRE(ctx, metric == 0 || metric == 1, "metric values: metric must be 0 (mean squares) or 1 (correlation moments)");
  PP_REQUIRE(ctx, ncand >= 1 && ncand <= PP_MAX_CAND, "metric values: 1..16 candidates per call");
  PP_REQUIRE(ctx, stride >= 1 && vsize[0] >= 1 && vsize[1] >= 1 && vsize[2] >= 1, "metric values: bad sampling lattice");
  mval_args a;
  memset(&a, 0, sizeof(a));
  memcpy(a.Af, Af, sizeof(a.Af));
  memcpy(a.bf, bf, sizeof(a.bf));
  memcpy(a.Am, Am, (size_t)ncand * 9 * sizeof(double));
  memcpy(a.bm, bm, (size_t)ncand * 3 * sizeof(double));
  for (int k = 0; k < 3; ++k) a.vsize[k] = vsize[k];
  a.stride = stride;
  const size_t nsamp = ((size_t)vsize[0] * vsize[1] * vsize[2] + stride - 1) / stride;
  {
    const int jrc = pp_jitter_for(ctx, nsamp, &a.jit);
    if (jrc) return jrc;
  }
  const pp_dims df{fsize[0], fsize[1], fsize[2]}, dm{msize[0], msize[1], msize[2]};
  // Candidates per thread: 16 on big lattices (HBM/L2-bound: the candidates of a sample share cache lines; the straight-line
  // form of the kernel keeps several candidates' gathers in flight), 4 for batches of up to four and on small lattices
  // (latency-bound: spread the candidates over blocks).
  const char* lanes_env = pp_env("PP_METRIC_LANES");   // (0: the candidate-loop kernels, for A/B runs and the equality test)
  if (!lanes_env || atoi(lanes_env) != 0) {
    // Blocks: the small lattices are bound by the latency of one probe (a block's dependent gather rounds, then one ticket
    // atomic per block, ~12 ns each, serialised), the large ones by throughput.  The cap depends on the lattice only.
    unsigned nb_cap = nsamp < 20000 ? 128u : 1024u;   // (tools/r4/run16.sh: 128 x 128 x 64 lattice 24.2 / 15.4 / 11.3 / 11.9 / 13.7 ms per level at 256 .. 4096)
    if (const char* e = pp_env("PP_METRIC_BLOCKS")) nb_cap = (unsigned)atoi(e) > 0 ? (unsigned)atoi(e) : nb_cap;
    size_t spt = (nsamp + (size_t)MV_SLOTS * nb_cap - 1) / ((size_t)MV_SLOTS * nb_cap);
    spt = (spt + 3) / 4 * 4;
    PP_REQUIRE(ctx, spt < ((size_t)1 << 30), "metric values: sampling lattice too large");
    const unsigned nb = (unsigned)((nsamp + (size_t)MV_SLOTS * spt - 1) / ((size_t)MV_SLOTS * spt));
    const int nv = metric == 0 ? 2 : 6, cpw = metric == 0 ? 16 : 8;
    int rc = pp_reserve(ctx, pp_align_up((size_t)nb * MV_CL * nv * sizeof(double), 256));
    if (rc) return rc;
    unsigned* ticket = nullptr;
    rc = pp_ticket(ctx, &ticket);
    if (rc) return rc;
    double* partials = reinterpret_cast<double*>(ctx->ws);
    char* mail = nullptr;
    unsigned long long seq = 0;
    rc = pp_mailbox(ctx, &mail, &seq);
    if (rc) return rc;
    const float* fsamp = nullptr;
    rc = pp_fixed_samples(ctx, fixed, fsize, Af, bf, vsize, stride, fixed_mask, &fsamp);
    if (rc) return rc;
    if (metric == 0)
      hipLaunchKernelGGL((k_metric_values_lanes<0>), dim3(nb), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a,
                         ncand, (int)spt, partials, ticket, mail, seq, fsamp);
    else
      hipLaunchKernelGGL((k_metric_values_lanes<1>), dim3(nb), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a,
                         ncand, (int)spt, partials, ticket, mail, seq, fsamp);
    PP_LAUNCH_CHECK(ctx, "k_metric_values_lanes");
    memset(result, 0, (size_t)ncand * 6 * sizeof(double));
    for (int w = 0; w * cpw < ncand; ++w) {
      const int k = ncand - w * cpw < cpw ? ncand - w * cpw : cpw;
      double vals[PP_MAIL_ENTRIES];
      rc = pp_mail_take(ctx, w, k * nv, seq, vals);
      if (rc) return rc;
      for (int j = 0; j < k; ++j)
        for (int f = 0; f < nv; ++f) result[((size_t)w * cpw + j) * 6 + f] = vals[j * nv + f];
    }
    return PP_OK;
  }
  constexpr int CH0 = 16, CH0S = 4, CH1 = 4;
  const bool small = metric == 0 && nsamp < 150000;
  const int ch = metric == 0 ? (small || ncand <= 4 ? CH0S : CH0) : CH1, nv = metric == 0 ? 2 : 6;
  const int nchunk = (ncand + ch - 1) / ch;
  PP_REQUIRE(ctx, nchunk <= PP_MAIL_WRITERS, "metric values: more chunks than mailbox writers");
  // Blocks per chunk: one sample per thread up to ~512 blocks in the whole launch (512 on big lattices, where 16 candidates
  // are one chunk; 128 on small ones, where they are four).  Measured on MI355X (profiles/round3_metric_probe_latency.txt):
  // beyond that a probe gets SLOWER with more blocks -- 1024 blocks of two samples per thread cost 76 / 150 us (1 / 16
  // candidates, 128 x 128 x 64 lattice), 512 blocks of four 54 / 125 us; the 64 x 64 x 32 lattice with 16 candidates 78 -> 57 us
  // at 128 blocks per chunk.  (Not the ticket: a two-level ticket changed nothing.)  The cap depends on the lattice only, so
  // a candidate's partial sums do not depend on how many companions ride in the launch.  PP_METRIC_BLOCKS overrides it.
  unsigned nb_cap = (metric == 0 && small) || metric != 0 ? 128u : 512u;
  if (const char* e = pp_env("PP_METRIC_BLOCKS")) nb_cap = (unsigned)atoi(e);
  const unsigned nb = grid_for(nsamp, nb_cap);
  const size_t row = (size_t)ch * nv;
  int rc = pp_reserve(ctx, pp_align_up((size_t)nb * nchunk * row * sizeof(double), 256));
  if (rc) return rc;
  unsigned* ticket = nullptr;
  rc = pp_ticket(ctx, &ticket);
  if (rc) return rc;
  double* partials = reinterpret_cast<double*>(ctx->ws);
  char* mail = nullptr;
  unsigned long long seq = 0;
  rc = pp_mailbox(ctx, &mail, &seq);
  if (rc) return rc;
  const float* fsamp = nullptr;
  rc = pp_fixed_samples(ctx, fixed, fsize, Af, bf, vsize, stride, fixed_mask, &fsamp);
  if (rc) return rc;
  if (metric == 0 && ch == CH0S)
    hipLaunchKernelGGL((k_metric_values<0, CH0S>), dim3(nb, nchunk), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask,
                       moving_mask, a, ncand, partials, ticket, mail, seq, fsamp);
  else if (metric == 0)
    hipLaunchKernelGGL((k_metric_values<0, CH0>), dim3(nb, nchunk), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask,
                       moving_mask, a, ncand, partials, ticket, mail, seq, fsamp);
  else
    hipLaunchKernelGGL((k_metric_values<1, CH1>), dim3(nb, nchunk), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask,
                       moving_mask, a, ncand, partials, ticket, mail, seq, fsamp);
  PP_LAUNCH_CHECK(ctx, "k_metric_values");
  memset(result, 0, (size_t)ncand * 6 * sizeof(double));
  for (int c = 0; c < nchunk; ++c) {
    const int k = ncand - c * ch < ch ? ncand - c * ch : ch;
    double vals[PP_MAIL_ENTRIES];
    rc = pp_mail_take(ctx, c, k * nv, seq, vals);
    if (rc) return rc;
    for (int j = 0; j < k; ++j)
      for (int f = 0; f < nv; ++f) result[((size_t)c * ch + j) * 6 + f] = vals[j * nv + f];
  }
  return PP_OK;
}

int pp_mi_histogram_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3], const double Af[9],
                        const double bf[3], const double Am[9], const double bm[3], const int vsize[3], int stride,
                        const uint8_t* fixed_mask, const uint8_t* moving_mask, const pp_mi_bins* bins, double* hist, double* count) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && moving && hist && count, "pp_mi_histogram_f32: NULL argument");
  mi_args a;
  int rc = mi_fill_args(ctx, &a, fsize, msize, Af, bf, Am, bm, vsize, stride, bins);
  if (rc) return rc;
  const int nb2 = a.nbins * a.nbins;
  const size_t nsamp = ((size_t)vsize[0] * vsize[1] * vsize[2] + stride - 1) / stride;
  rc = pp_reserve(ctx, pp_align_up((size_t)(nb2 + 1) * sizeof(unsigned long long), 256));
  if (rc) return rc;
  unsigned long long* dh = reinterpret_cast<unsigned long long*>(ctx->ws);
  PP_HIP(ctx, hipMemsetAsync(dh, 0, (size_t)(nb2 + 1) * sizeof(unsigned long long), ctx->stream));
  const pp_dims df{fsize[0], fsize[1], fsize[2]}, dm{msize[0], msize[1], msize[2]};
  hipLaunchKernelGGL(k_mi_histogram, dim3(grid_for(nsamp, 512u)), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a, dh);
  PP_LAUNCH_CHECK(ctx, "k_mi_histogram");
  std::vector<unsigned long long> h((size_t)nb2 + 1);
  PP_HIP(ctx, hipMemcpyAsync(h.data(), dh, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < nb2; ++i) hist[i] = (double)h[i] / 4294967296.0;
  *count = (double)h[nb2];
  return PP_OK;
}

int pp_mi_gradient_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3], const double Af[9],
                       const double bf[3], const double Am[9], const double bm[3], const int vsize[3], int stride,
                       const uint8_t* fixed_mask, const uint8_t* moving_mask, const pp_mi_bins* bins, const double* table, double* result) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && moving && table && result, "pp_mi_gradient_f32: NULL argument");
  mi_args a;
  int rc = mi_fill_args(ctx, &a, fsize, msize, Af, bf, Am, bm, vsize, stride, bins);
  if (rc) return rc;
  const int nb2 = a.nbins * a.nbins;
  const size_t nsamp = ((size_t)vsize[0] * vsize[1] * vsize[2] + stride - 1) / stride;
  const unsigned nb = grid_for(nsamp, 512u);
  const size_t tab_bytes = pp_align_up((size_t)nb2 * sizeof(float), 256);
  rc = pp_reserve(ctx, tab_bytes + pp_align_up((size_t)nb * 14 * sizeof(double), 256) + 256);
  if (rc) return rc;
  float* dtab = reinterpret_cast<float*>(ctx->ws);
  double* partials = reinterpret_cast<double*>(ctx->ws + tab_bytes);
  std::vector<float> ht((size_t)nb2);
  for (int i = 0; i < nb2; ++i) ht[i] = (float)table[i];
  PP_HIP(ctx, hipMemcpyAsync(dtab, ht.data(), (size_t)nb2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // ht is a local: the copy must have left it
  const pp_dims df{fsize[0], fsize[1], fsize[2]}, dm{msize[0], msize[1], msize[2]};
  hipLaunchKernelGGL(k_mi_gradient, dim3(nb), dim3(NT), 0, ctx->stream, fixed, df, moving, dm, fixed_mask, moving_mask, a, (const float*)dtab,
                     partials);
  PP_LAUNCH_CHECK(ctx, "k_mi_gradient");
  char* mail = nullptr;
  unsigned long long seq = 0;
  rc = pp_mailbox(ctx, &mail, &seq);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sum14_final, dim3(1), dim3(NT), 0, ctx->stream, (const double*)partials, (int)nb, 14, reinterpret_cast<double*>(mail), 1,
                     seq);
  PP_LAUNCH_CHECK(ctx, "k_sum14_final");
  double sums[14];
  rc = pp_mail_take(ctx, 0, 14, seq, sums);
  if (rc) return rc;
  memcpy(result, sums + 2, 12 * sizeof(double));
  return PP_OK;
}

int pp_linear_set_moving_gradient(pp_ctx* ctx, const float* gradient, const int msize[3]) {
  if (!ctx) return PP_ERR_ARG;
  PP_REQUIRE(ctx, gradient == nullptr || msize != nullptr, "pp_linear_set_moving_gradient: a gradient image needs the moving image's size");
  ctx->mgrad = gradient;
  ctx->mgrad4 = nullptr;   // (a packed companion belongs to ONE planar image: set it after this call)
  for (int k = 0; k < 3; ++k) ctx->mgrad_size[k] = gradient ? msize[k] : 0;
  return PP_OK;
}

int pp_linear_set_moving_gradient_packed(pp_ctx* ctx, const float* packed) {
  if (!ctx) return PP_ERR_ARG;
  PP_REQUIRE(ctx, packed == nullptr || ctx->mgrad != nullptr, "pp_linear_set_moving_gradient_packed: set the planar gradient image first");
  PP_REQUIRE(ctx, reinterpret_cast<uintptr_t>(packed) % 16 == 0, "pp_linear_set_moving_gradient_packed: the packed image must be 16-byte aligned");
  ctx->mgrad4 = packed;
  return PP_OK;
}

int pp_linear_set_sample_jitter(pp_ctx* ctx, const float* jitter, size_t nsamples) {
  if (!ctx) return PP_ERR_ARG;
  PP_REQUIRE(ctx, (jitter == nullptr) == (nsamples == 0), "pp_linear_set_sample_jitter: a jitter array needs its sample count, NULL needs 0");
  ctx->jitter = jitter;
  ctx->jitter_samples = nsamples;
  ctx->jitter_gen += 1;
  return PP_OK;
}

}  // extern "C"
