// platipy_amd/csrc/pp_cc.hip -- hole filling and largest connected component of a binary volume.
//
// Replaces sitk.BinaryFillhole -> sitk.ConnectedComponent -> LabelShapeStatistics -> "keep the largest"
// in process_probability_image (reference: platipy/imaging/label/fusion.py:310-328), all with face
// connectivity (SimpleITK's fullyConnected=False defaults).  The host version of this step was 85 % of one
// atlas chain's wall time at 512x512x256.
//
// Connected components by lock-free union-find over the voxel lattice (label-equivalence with atomicMin:
// a component's root is its first voxel in raster order, which is also the order ITK numbers components
// in, so "first largest" ties resolve identically).  One sweep labels foreground AND background
// components (neighbours are united when their binary values agree); background components that own no
// border voxel are holes.  Two stages: x-runs first (an LDS scan per row, no atomics), then unions only where a
// run starts against the rows above -- the background of a CT-sized mask is one 60-Mvoxel component, and uniting
// it voxel by voxel serialises on its root (12 ms at 512x512x256; 10x less in stages).
#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ int cc_find(const int* L, int i) {
  int r = i;
  for (;;) {
    const int p = L[r];
    if (p == r) return r;
    r = p;
  }
}

// Unite the sets of a and b; the smaller index becomes the root.  A stale (cached) read of L can only cost
// extra rounds: links never disappear and a merge only counts once its atomicMin saw the expected root.
__device__ __forceinline__ void cc_unite(int* L, int a, int b) {
  for (;;) {
    a = cc_find(L, a);
    b = cc_find(L, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&L[b], a);  // a < b
    if (old == b) return;
    b = old;
  }
}

// Stage 1, rows: every voxel points at the first voxel of its x-run of equal values.  A thread walks CC_E
// consecutive voxels; the position of the last run start at or before each thread's span comes from a max-scan over
// the threads' last starts through LDS (NT x CC_E voxels per pass, with a carry for longer rows).
// fg_only: background voxels keep their own index.
constexpr int CC_E = 8;
__global__ void __launch_bounds__(NT) k_cc_rows(const uint8_t* __restrict__ mask, int* __restrict__ L, pp_dims d, int fg_only) {
  __shared__ int s[NT];
  __shared__ int carry_s;
  const size_t rows = (size_t)d.ny * d.nz;
  const int span = NT * CC_E;
  for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const uint8_t* m = mask + row * d.nx;
    int carry = 0;
    for (int x0 = 0; x0 < d.nx; x0 += span) {
      const int left = d.nx - x0;
      const int nact = left >= span ? NT : (left + CC_E - 1) / CC_E;   // threads with at least one voxel
      const int xb = x0 + (int)threadIdx.x * CC_E;
      bool v[CC_E];
      int last = -1;                       // last run start inside my span
      bool prev = xb > 0 && xb <= d.nx ? m[xb - 1] != 0 : false;
      for (int e = 0; e < CC_E; ++e) {
        const int x = xb + e;
        v[e] = x < d.nx ? m[x] != 0 : false;
        if (x < d.nx && (x == 0 || v[e] != prev)) last = x;
        prev = v[e];
      }
      s[threadIdx.x] = last;
      __syncthreads();
      for (int off = 1; off < nact; off <<= 1) {
        const int t = (int)threadIdx.x >= off ? s[threadIdx.x - off] : -1;
        __syncthreads();
        if (t > s[threadIdx.x]) s[threadIdx.x] = t;
        __syncthreads();
      }
      // run start in force when my span begins: the scan over the threads to my left, else the carry
      int cur = threadIdx.x > 0 ? s[threadIdx.x - 1] : -1;
      if (cur < 0) cur = carry;
      prev = xb > 0 && xb <= d.nx ? m[xb - 1] != 0 : false;
      for (int e = 0; e < CC_E; ++e) {
        const int x = xb + e;
        if (x >= d.nx) break;
        if (x == 0 || v[e] != prev) cur = x;
        prev = v[e];
        L[row * d.nx + x] = (fg_only && !v[e]) ? (int)(row * d.nx + x) : (int)(row * d.nx + cur);
      }
      if ((int)threadIdx.x == nact - 1) carry_s = s[threadIdx.x] >= 0 ? s[threadIdx.x] : carry;
      __syncthreads();
      carry = carry_s;
      __syncthreads();
    }
  }
}

// The same labelling with one WAVEFRONT per row pass (rows up to 512 voxels in one pass, longer ones with a carry): a lane
// walks CC_E consecutive voxels, the run start in force at its span comes from a max-scan over the lanes by wavefront
// shuffles -- no LDS, no block barrier, four rows per block at a time.  (A block per row kept 38 of 256 threads busy on the
// pipelines' ~300-voxel rows and paid twelve block barriers per row.)
__global__ void __launch_bounds__(NT) k_cc_rows_wave(const uint8_t* __restrict__ mask, int* __restrict__ L, pp_dims d, int fg_only) {
  const size_t rows = (size_t)d.ny * d.nz;
  const int lane = (int)threadIdx.x & 63, wib = (int)threadIdx.x >> 6;
  constexpr int WPB = NT / 64, SPAN = 64 * CC_E;
  for (size_t row0 = (size_t)blockIdx.x * WPB; row0 < rows; row0 += (size_t)gridDim.x * WPB) {
    const size_t row = row0 + wib;
    if (row >= rows) continue;   // (shuffles are per wavefront: one without a row takes part in nothing)
    const uint8_t* m = mask + row * d.nx;
    int carry = 0;
    for (int x0 = 0; x0 < d.nx; x0 += SPAN) {
      const int xb = x0 + lane * CC_E;
      bool v[CC_E];
      int last = -1;                       // last run start inside my span
      const bool before = xb > 0 && xb <= d.nx ? m[xb - 1] != 0 : false;
      bool prev = before;
#pragma unroll
      for (int e = 0; e < CC_E; ++e) {
        const int x = xb + e;
        v[e] = x < d.nx ? m[x] != 0 : false;
        if (x < d.nx && (x == 0 || v[e] != prev)) last = x;
        prev = v[e];
      }
      int s = last;                        // inclusive max-scan over the lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(s, (unsigned)off);
        if (lane >= off && t > s) s = t;
      }
      int cur = __shfl_up(s, 1u);          // run start in force when my span begins
      if (lane == 0 || cur < 0) cur = carry;   // (lane 0's shuffle returns its own value; -1: no run start to the left in this pass)
      prev = before;
#pragma unroll
      for (int e = 0; e < CC_E; ++e) {
        const int x = xb + e;
        if (x < d.nx) {
          if (x == 0 || v[e] != prev) cur = x;
          prev = v[e];
          L[row * d.nx + x] = (fg_only && !v[e]) ? (int)(row * d.nx + x) : (int)(row * d.nx + cur);
        }
      }
      const int tail = __shfl(s, 63);
      carry = tail >= 0 ? tail : carry;
    }
  }
}

// Stage 2, across rows: two equal-valued runs in adjacent rows (y - 1 or z - 1) overlap somewhere, and the later
// of their two starts lies inside the overlap -- one union there joins them, so only positions where either run
// starts need to try.  That is a few unions per run instead of three per voxel, and finds are one or two hops.
__global__ void __launch_bounds__(NT) k_cc_merge(const uint8_t* __restrict__ mask, int* __restrict__ L, pp_dims d, int fg_only) {
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  const int sy = d.nx, sz = d.nx * d.ny;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const bool v = mask[i] != 0;
    if (fg_only && !v) continue;
    const unsigned i32 = (unsigned)i, row = i32 / (unsigned)d.nx;   // (n < 2^31: 32-bit divisions)
    const int x = (int)(i32 - row * (unsigned)d.nx), y = (int)(row % (unsigned)d.ny), z = (int)(row / (unsigned)d.ny);
    const bool start = x == 0 || (mask[i - 1] != 0) != v;
    if (y > 0 && (mask[i - sy] != 0) == v && (start || (mask[i - sy - 1] != 0) != v)) cc_unite(L, (int)i, (int)i - sy);
    if (z > 0 && (mask[i - sz] != 0) == v && (start || (mask[i - sz - 1] != 0) != v)) cc_unite(L, (int)i, (int)i - sz);
  }
}

__global__ void __launch_bounds__(NT) k_cc_compress(int* __restrict__ L, size_t n) {
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) L[i] = cc_find(L, (int)i);
}

// Background components that reach the volume border are not holes: flag their roots.
__global__ void __launch_bounds__(NT) k_cc_flag_border(const uint8_t* __restrict__ mask, const int* __restrict__ L,
                                                       int* __restrict__ flag, pp_dims d) {
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    if (mask[i]) continue;
    const unsigned i32 = (unsigned)i, row = i32 / (unsigned)d.nx;
    const int x = (int)(i32 - row * (unsigned)d.nx), y = (int)(row % (unsigned)d.ny), z = (int)(row / (unsigned)d.ny);
    if (x == 0 || y == 0 || z == 0 || x == d.nx - 1 || y == d.ny - 1 || z == d.nz - 1) flag[L[i]] = 1;
  }
}

__global__ void __launch_bounds__(NT) k_cc_fill(const uint8_t* __restrict__ mask, const int* __restrict__ L,
                                                const int* __restrict__ flag, uint8_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT)
    out[i] = (mask[i] || !flag[L[i]]) ? (uint8_t)1 : (uint8_t)0;
}

// Component sizes: one atomicAdd per run of equal roots inside a block (components are long runs along x).
__global__ void __launch_bounds__(NT) k_cc_count(const uint8_t* __restrict__ mask, const int* __restrict__ L,
                                                 int* __restrict__ count, size_t n) {
  __shared__ int roots[NT];
  for (size_t base = (size_t)blockIdx.x * NT; base < n; base += (size_t)gridDim.x * NT) {
    const size_t i = base + threadIdx.x;
    const int r = (i < n && mask[i]) ? L[i] : -1;
    __syncthreads();
    roots[threadIdx.x] = r;
    __syncthreads();
    if (r >= 0 && (threadIdx.x == 0 || roots[threadIdx.x - 1] != r)) {
      int len = 1;
      while (threadIdx.x + len < NT && roots[threadIdx.x + len] == r) ++len;
      atomicAdd(&count[r], len);
    }
  }
}

// arg max over roots of (count, -root): partials [grid][2] then a single-block fold.
__global__ void __launch_bounds__(NT) k_cc_argmax(const uint8_t* __restrict__ mask, const int* __restrict__ L,
                                                  const int* __restrict__ count, size_t n, int* __restrict__ partials) {
  __shared__ int bc[NT], br[NT];
  int c = 0, r = 0x7fffffff;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT)
    if (mask[i] && L[i] == (int)i) {
      const int ci = count[i];
      if (ci > c || (ci == c && (int)i < r)) {
        c = ci;
        r = (int)i;
      }
    }
  bc[threadIdx.x] = c;
  br[threadIdx.x] = r;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const int c2 = bc[threadIdx.x + s], r2 = br[threadIdx.x + s];
      if (c2 > bc[threadIdx.x] || (c2 == bc[threadIdx.x] && r2 < br[threadIdx.x])) {
        bc[threadIdx.x] = c2;
        br[threadIdx.x] = r2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = bc[0];
    partials[2 * blockIdx.x + 1] = br[0];
  }
}

__global__ void __launch_bounds__(NT) k_cc_argmax_final(const int* __restrict__ partials, int nb, int* __restrict__ result) {
  __shared__ int bc[NT], br[NT];
  int c = 0, r = 0x7fffffff;
  for (int i = threadIdx.x; i < nb; i += NT) {
    const int ci = partials[2 * i], ri = partials[2 * i + 1];
    if (ci > c || (ci == c && ri < r)) {
      c = ci;
      r = ri;
    }
  }
  bc[threadIdx.x] = c;
  br[threadIdx.x] = r;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const int c2 = bc[threadIdx.x + s], r2 = br[threadIdx.x + s];
      if (c2 > bc[threadIdx.x] || (c2 == bc[threadIdx.x] && r2 < br[threadIdx.x])) {
        bc[threadIdx.x] = c2;
        br[threadIdx.x] = r2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    result[0] = bc[0];
    result[1] = br[0];
  }
}

__global__ void __launch_bounds__(NT) k_cc_select(const uint8_t* __restrict__ mask, const int* __restrict__ L,
                                                  const int* __restrict__ result, uint8_t* __restrict__ out, size_t n) {
  const int best = result[1], cnt = result[0];
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT)
    out[i] = cnt > 0 ? ((mask[i] && L[i] == best) ? (uint8_t)1 : (uint8_t)0) : mask[i];  // no component: input back
}

unsigned grid_for(size_t work, unsigned cap = 16384u) {
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

int cc_label(pp_ctx* ctx, const uint8_t* mask, int* L, const pp_dims& d, size_t n, int fg_only) {
  const dim3 g(grid_for(n)), b(NT);
  const size_t rows = (size_t)d.ny * d.nz;
  if (pp_env("PP_CC_ROWS_BLOCK")) {   // (the block-per-row form, for A/B runs)
    hipLaunchKernelGGL(k_cc_rows, dim3((unsigned)(rows < 65535 ? rows : 65535)), b, 0, ctx->stream, mask, L, d, fg_only);
  } else {
    const size_t nbr = (rows + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL(k_cc_rows_wave, dim3((unsigned)(nbr < 16384 ? nbr : 16384)), b, 0, ctx->stream, mask, L, d, fg_only);
  }
  PP_LAUNCH_CHECK(ctx, "k_cc_rows");
  hipLaunchKernelGGL(k_cc_merge, g, b, 0, ctx->stream, mask, L, d, fg_only);
  PP_LAUNCH_CHECK(ctx, "k_cc_merge");
  hipLaunchKernelGGL(k_cc_compress, g, b, 0, ctx->stream, L, n);
  PP_LAUNCH_CHECK(ctx, "k_cc_compress");
  return PP_OK;
}

}  // namespace

extern "C" int pp_fillhole_largest_component_u8(pp_ctx* ctx, const uint8_t* in, const int size[3], int fill_holes,
                                                 uint8_t* out, int64_t* component_voxels) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && size, "pp_fillhole_largest_component_u8: NULL argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_fillhole_largest_component_u8: empty volume");
  const size_t n = pp_nvox(size);
  PP_REQUIRE(ctx, n < 2147483647u, "pp_fillhole_largest_component_u8: more than 2^31-1 voxels");
  const pp_dims d{size[0], size[1], size[2]};
  const unsigned nb = grid_for(n, 2048u);
  int rc = pp_reserve(ctx, 2 * pp_align_up(n * sizeof(int), 256) + pp_align_up(n, 256) + pp_align_up((2 * (size_t)nb + 2) * sizeof(int), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  int* L = cv.take<int>(n);
  int* aux = cv.take<int>(n);          // border flags, then component sizes
  uint8_t* filled = cv.take<uint8_t>(n);
  int* partials = cv.take<int>(2 * (size_t)nb + 2);
  int* result = partials + 2 * (size_t)nb;
  const dim3 g(grid_for(n)), b(NT);
  const uint8_t* mask = in;
  if (fill_holes) {
    rc = cc_label(ctx, in, L, d, n, 0);
    if (rc) return rc;
    PP_HIP(ctx, hipMemsetAsync(aux, 0, n * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_cc_flag_border, g, b, 0, ctx->stream, in, (const int*)L, aux, d);
    PP_LAUNCH_CHECK(ctx, "k_cc_flag_border");
    hipLaunchKernelGGL(k_cc_fill, g, b, 0, ctx->stream, in, (const int*)L, (const int*)aux, filled, n);
    PP_LAUNCH_CHECK(ctx, "k_cc_fill");
    mask = filled;
  }
  rc = cc_label(ctx, mask, L, d, n, 1);
  if (rc) return rc;
  PP_HIP(ctx, hipMemsetAsync(aux, 0, n * sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_cc_count, g, b, 0, ctx->stream, mask, (const int*)L, aux, n);
  PP_LAUNCH_CHECK(ctx, "k_cc_count");
  hipLaunchKernelGGL(k_cc_argmax, dim3(nb), b, 0, ctx->stream, mask, (const int*)L, (const int*)aux, n, partials);
  PP_LAUNCH_CHECK(ctx, "k_cc_argmax");
  hipLaunchKernelGGL(k_cc_argmax_final, dim3(1), b, 0, ctx->stream, (const int*)partials, (int)nb, result);
  PP_LAUNCH_CHECK(ctx, "k_cc_argmax_final");
  hipLaunchKernelGGL(k_cc_select, g, b, 0, ctx->stream, mask, (const int*)L, (const int*)result, out, n);
  PP_LAUNCH_CHECK(ctx, "k_cc_select");
  if (component_voxels) {
    int h[2];
    rc = pp_read_back(ctx, result, h, sizeof(h));
    if (rc) return rc;
    *component_voxels = h[0];
  }
  return PP_OK;
}
