// platipy_amd/csrc/pp_fir.hip -- separable Gaussian FIR passes (any radius).
//
// Replaces itk::DiscreteGaussianImageFilter (reference: registration/utils.py:226,
// label/fusion.py:168,279) and the PDE demons field smoothers (deformable.py:248-257) in
// their staged, one-axis-per-launch form.  Memory-bound: each pass reads the volume once
// through the L1/L2 (neighbour taps are cache hits) and writes it once, 8 B/voxel; lanes run
// along x so every tap is a coalesced row segment.  The 3-axis fused kernels of the demons
// inner loop live in pp_demons.hip.
#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr int NT = 256;

// One FIR pass along AXIS with clamped (ZeroFluxNeumann) edges.  VEC voxels per thread along x
// (VEC = 4 needs nx % 4 == 0 so rows keep 16-B alignment; AXIS 0 always runs VEC = 1).
// blockIdx.y selects the field component (plane stride `cstride`).
template <int AXIS, int VEC, bool ADD>
__global__ void __launch_bounds__(NT) k_conv_axis(const float* __restrict__ in, const float* __restrict__ add,
                                                  float* __restrict__ out, pp_dims d, size_t cstride, pp_taps taps,
                                                  const int* __restrict__ halt, const int* __restrict__ rows, int use_y, int use_z) {
  if (halt && *halt) return;
  const size_t comp = (size_t)blockIdx.y * cstride;
  in += comp;
  out += comp;
  if (ADD) add += comp;
  const int nxv = d.nx / VEC;
  // sparse mode (`rows`, see k_compact_rows): only the listed y / z are produced -- the work items are enumerated over the
  // LISTS, so no thread is launched for a row that is never read
  const int nys = (rows && use_y) ? rows[0] : d.ny, nzs = (rows && use_z) ? rows[1] : d.nz;
  const size_t total = (size_t)nxv * nys * nzs;
  const int r = taps.r;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
    const int xv = (int)(e % nxv);
    const int yi = (int)((e / nxv) % nys), zi = (int)(e / ((size_t)nxv * nys));
    const int y = (rows && use_y) ? rows[2 + yi] : yi, z = (rows && use_z) ? rows[2 + d.ny + zi] : zi;
    const size_t row = ((size_t)z * d.ny + y) * d.nx;
    if (AXIS == 0) {
      float s = 0.0f;
      for (int k = -r; k <= r; ++k) {
        const int q = pp_clampi(xv + k, 0, d.nx - 1);
        float v = in[row + q];
        if (ADD) v += add[row + q];
        s = fmaf(taps.w[k + r], v, s);
      }
      out[row + xv] = s;
    } else {
      const int pos = AXIS == 1 ? y : z;
      const int len = AXIS == 1 ? d.ny : d.nz;
      const size_t stride = AXIS == 1 ? (size_t)d.nx : (size_t)d.nx * d.ny;
      const size_t base = row + (size_t)xv * VEC;
      float s[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) s[v] = 0.0f;
      for (int k = -r; k <= r; ++k) {
        const int q = pp_clampi(pos + k, 0, len - 1);
        const size_t o = (size_t)((long long)base + (long long)(q - pos) * (long long)stride);
        const float w = taps.w[k + r];
        if (VEC == 4) {
          float4 v = *reinterpret_cast<const float4*>(in + o);
          if (ADD) {
            const float4 a = *reinterpret_cast<const float4*>(add + o);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          s[0] = fmaf(w, v.x, s[0]);
          s[1] = fmaf(w, v.y, s[1]);
          s[2] = fmaf(w, v.z, s[2]);
          s[3] = fmaf(w, v.w, s[3]);
        } else {
          float v = in[o];
          if (ADD) v += add[o];
          s[0] = fmaf(w, v, s[0]);
        }
      }
      if (VEC == 4)
        *reinterpret_cast<float4*>(out + base) = make_float4(s[0], s[1], s[2], s[3]);
      else
        out[base] = s[0];
    }
  }
}

// x pass, 4 consecutive outputs per thread: the row is read as aligned float4 chunks (each input value is
// loaded once per thread instead of once per tap) and every chunk is scattered into the 4 running sums.
// Needs nx % 4 == 0 and 16-B aligned rows; edge chunks clamp per element.
template <bool ADD>
__global__ void __launch_bounds__(NT) k_conv_x4(const float* __restrict__ in, const float* __restrict__ add,
                                                float* __restrict__ out, pp_dims d, size_t cstride, pp_taps taps,
                                                const int* __restrict__ halt, const int* __restrict__ rows, int use_y, int use_z) {
  if (halt && *halt) return;
  const size_t comp = (size_t)blockIdx.y * cstride;
  in += comp;
  out += comp;
  if (ADD) add += comp;
  const int nxv = d.nx / 4;
  const int nys = (rows && use_y) ? rows[0] : d.ny, nzs = (rows && use_z) ? rows[1] : d.nz;   // (sparse mode: see k_conv_axis)
  const size_t total = (size_t)nxv * nys * nzs;
  const int r = taps.r;
  const int r4 = (r + 3) / 4 * 4;
  for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
    const int x0 = (int)(e % nxv) * 4;
    const int yi = (int)((e / nxv) % nys), zi = (int)(e / ((size_t)nxv * nys));
    const int y = (rows && use_y) ? rows[2 + yi] : yi, z = (rows && use_z) ? rows[2 + d.ny + zi] : zi;
    const size_t row = ((size_t)z * d.ny + y) * (size_t)d.nx;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int o = -r4; o < 4 + r4; o += 4) {           // chunk covers inputs x0 + o .. x0 + o + 3
      float v[4];
      const int xa = x0 + o;
      if (xa >= 0 && xa + 3 < d.nx) {
        float4 q = *reinterpret_cast<const float4*>(in + row + xa);
        if (ADD) {
          const float4 a = *reinterpret_cast<const float4*>(add + row + xa);
          q.x += a.x; q.y += a.y; q.z += a.z; q.w += a.w;
        }
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int xc = pp_clampi(xa + i, 0, d.nx - 1);
          v[i] = in[row + xc];
          if (ADD) v[i] += add[row + xc];
        }
      }
      // input at relative position p = o + i feeds output j through tap k = p - j + r
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k0 = o + i + r;                     // tap index for output 0
        const float w0 = (k0 >= 0 && k0 <= 2 * r) ? taps.w[k0] : 0.0f;
        const float w1 = (k0 - 1 >= 0 && k0 - 1 <= 2 * r) ? taps.w[k0 - 1] : 0.0f;
        const float w2 = (k0 - 2 >= 0 && k0 - 2 <= 2 * r) ? taps.w[k0 - 2] : 0.0f;
        const float w3 = (k0 - 3 >= 0 && k0 - 3 <= 2 * r) ? taps.w[k0 - 3] : 0.0f;
        s0 = fmaf(w0, v[i], s0);
        s1 = fmaf(w1, v[i], s1);
        s2 = fmaf(w2, v[i], s2);
        s3 = fmaf(w3, v[i], s3);
      }
    }
    *reinterpret_cast<float4*>(out + row + x0) = make_float4(s0, s1, s2, s3);
  }
}


// rows[0] = number of needed y, rows[1] = number of needed z, rows[2 ..] = the needed y in order, rows[2 + ny ..] = the
// needed z.  One block: the flags of a list go to LDS and every listed entry counts the flags before it (a few hundred
// entries; the serial form spent 50 us chasing one dependent load per entry).
constexpr int COMPACT_MAX = 4096;
__device__ void compact_list(const uint8_t* __restrict__ need, int n, int* __restrict__ dst, int* __restrict__ count, uint8_t* flags) {
  if (n > COMPACT_MAX) {   // (not a size this path sees: one thread walks the list)
    if (threadIdx.x == 0) {
      int c = 0;
      for (int i = 0; i < n; ++i)
        if (need[i]) dst[c++] = i;
      *count = c;
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) flags[i] = need[i] ? 1 : 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (!flags[i]) continue;
    int before = 0;
    for (int j = 0; j < i; ++j) before += flags[j];
    dst[before] = i;
  }
  if (threadIdx.x == 0) {
    int c = 0;
    for (int j = 0; j < n; ++j) c += flags[j];
    *count = c;
  }
  __syncthreads();
}
__global__ void __launch_bounds__(NT) k_compact_rows(const uint8_t* __restrict__ need_y, int ny, const uint8_t* __restrict__ need_z, int nz,
                                                     int* __restrict__ rows) {
  __shared__ uint8_t flags[COMPACT_MAX];
  compact_list(need_y, ny, rows + 2, rows + 0, flags);
  compact_list(need_z, nz, rows + 2 + ny, rows + 1, flags);
}

// ---------------------------------------------------------------------------------------
// Passes for the pyramid's blur -- sparse outputs (a quarter of the planes and rows are ever read by the resample that
// follows; radius ~11 / ~21 at sigma 4 / 8 voxels) -- and for dense passes of radius 9 .. 24 (y, z) and 17 .. 32 (x).
//
// y / z (k_fir_march_sp, defined after k_fir_march below): that register-window march with three changes.  The window is longer than
// the filter by K slots and the load issued at a step lands K steps later, so K loads per thread are in flight (the march
// is a chain of dependent loads otherwise; an LDS ring per thread was tried first and capped the occupancy at 6 waves per
// CU: 1.4 TB/s).  The march runs over the EXTENDED line (in[clamp(p)]), so ITK's ZeroFluxNeumann edge needs no case
// analysis.  And a step forms its output only if the list names it (wavefront-uniform branch): every input is loaded
// once and only the outputs that are read cost arithmetic.  k_conv_axis, which this replaces here, re-reads 2r + 1 planes
// per output: with sparse outputs (no reuse between neighbouring outputs in L1) ~6 reads of every plane from the
// infinity cache.  Taps are padded with zeros to the bucket RB on both sides: fma(0, v, s) = s for finite v and the
// leading zero taps give +0, the value the sum starts from -- bit-identical to k_conv_axis.
// x (k_fir_x_row): one extended row per block round in LDS, two adjacent outputs per thread.
template <int WB>
__global__ void __launch_bounds__(NT) k_fir_x_row(const float* __restrict__ in, float* __restrict__ out, pp_dims d, size_t cstride, pp_taps taps,
                                                  const int* __restrict__ halt, const int* __restrict__ rows, int use_y, int use_z) {
  // The row is staged EXTENDED (r clamped voxels either side, then WB finite pad values), so an output reads 2r + 1
  // consecutive LDS words at immediate offsets; a thread forms two adjacent outputs from one set of reads.
  constexpr int ROW_MAX = 4096;
  __shared__ __attribute__((aligned(16))) float srow[ROW_MAX + 3 * WB + 3];
  if (halt && *halt) return;
  in += (size_t)blockIdx.y * cstride;
  out += (size_t)blockIdx.y * cstride;
  const int nys = (rows && use_y) ? rows[0] : d.ny, nzs = (rows && use_z) ? rows[1] : d.nz;
  const int r = taps.r, W = 2 * r + 1;
  float tw[WB];
#pragma unroll
  for (int k = 0; k < WB; ++k) tw[k] = k < W ? taps.w[k] : 0.0f;
  const int next = d.nx + 2 * r;   // extended row: word e holds in[clamp(e - r)]
  for (size_t e = blockIdx.x; e < (size_t)nys * nzs; e += gridDim.x) {
    const int yi = (int)(e % nys), zi = (int)(e / nys);
    const int y = (rows && use_y) ? rows[2 + yi] : yi, z = (rows && use_z) ? rows[2 + d.ny + zi] : zi;
    const size_t row = ((size_t)z * d.ny + y) * (size_t)d.nx;
    for (int q = threadIdx.x; q < next + WB + 2; q += NT) srow[q] = q < next ? in[row + pp_clampi(q - r, 0, d.nx - 1)] : 0.0f;
    __syncthreads();
    for (int x = 2 * threadIdx.x; x < d.nx; x += 2 * NT) {
      const float* const win = srow + x;   // words x .. x + 2r feed output x, x + 1 .. x + 2r + 1 feed output x + 1
      float a[WB + 1];
#pragma unroll
      for (int k = 0; k <= WB; ++k) a[k] = win[k];
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int k = 0; k < WB; ++k) {
        s0 = fmaf(tw[k], a[k], s0);
        s1 = fmaf(tw[k], a[k + 1], s1);
      }
      out[row + x] = s0;
      if (x + 1 < d.nx) out[row + x + 1] = s1;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Register-window passes along y / z (radius <= 16) and a wavefront-shuffle pass along x (radius <= 16).
//
// y / z: a thread owns a column of VEC consecutive x voxels and MARCHES along the pass axis over a segment of SEG outputs,
// keeping the last 2 RB + 1 inputs in registers: every input is loaded once (coalesced along x) and every output stored
// once -- 8 B/voxel through HBM instead of (2r + 1) L2 re-reads.  The loop is unrolled 2 RB + 1 times so that the window
// is renamed, not moved.  RB is a radius BUCKET >= r: taps beyond r are zero, which leaves every sum bit-identical
// (fma(0, v, s) = s for finite v; the leading zero taps give +0, the value the sum starts from).  VEC = 4 up to bucket 8,
// scalar columns above (the window would not fit the register file otherwise).
template <int RB>
struct taps_bucket {
  float w[2 * RB + 1];
};

template <int AXIS, int RB, int VEC, bool ADD>
__global__ void __launch_bounds__(NT) k_fir_march(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out, pp_dims d,
                                                  size_t cstride, taps_bucket<RB> taps, int seg, const int* __restrict__ halt) {
  if (halt && *halt) return;
  constexpr int W = 2 * RB + 1;
  const size_t comp = (size_t)blockIdx.z * cstride;
  in += comp;
  out += comp;
  if (ADD) add += comp;
  const int nxv = d.nx / VEC;
  const int len = AXIS == 1 ? d.ny : d.nz;          // pass axis
  const int other = AXIS == 1 ? d.nz : d.ny;        // the remaining axis
  const size_t stride = AXIS == 1 ? (size_t)d.nx : (size_t)d.nx * d.ny;
  const size_t ostride = AXIS == 1 ? (size_t)d.nx * d.ny : (size_t)d.nx;
  const size_t ncol = (size_t)nxv * other;
  const size_t col = (size_t)blockIdx.x * NT + threadIdx.x;
  if (col >= ncol) return;
  const int xv = (int)(col % nxv), o = (int)(col / nxv);
  const size_t base = (size_t)o * ostride + (size_t)xv * VEC;
  const int p0 = blockIdx.y * seg;
  const int p1 = p0 + seg < len ? p0 + seg : len;
  auto load = [&](int q, float (&v)[VEC]) {
    q = q < 0 ? 0 : (q > len - 1 ? len - 1 : q);
    const size_t a = base + (size_t)q * stride;
    if constexpr (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(in + a);
      if (ADD) {
        const float4 u = *reinterpret_cast<const float4*>(add + a);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      v[0] = in[a];
      if (ADD) v[0] += add[a];
    }
  };
  float win[W][VEC];
#pragma unroll
  for (int k = 0; k < W - 1; ++k) load(p0 - RB + k, win[(k + 1) % W]);   // slots 1 .. W-1 hold inputs p0-RB .. p0+RB-1
  for (int pb = p0; pb < p1; pb += W) {
#pragma unroll
    for (int u = 0; u < W; ++u) {
      const int pos = pb + u;
      if (pos < p1) {
        // the newest input (pos + RB) goes to slot u; the oldest (pos - RB) sits in slot u + 1
        load(pos + RB, win[u % W]);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fmaf(taps.w[k], win[(u + 1 + k) % W][v], acc[v]);
        const size_t a = base + (size_t)pos * stride;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(out + a) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else out[a] = acc[0];
      }
    }
  }
}

// x pass: a lane holds four consecutive voxels of a row, a wavefront 256; the (up to 16) neighbours on either side come
// from the adjacent lanes by wavefront shuffles, from memory only at the ends of a wavefront's span and of the row
// (clamped: ZeroFluxNeumann).  One 16-byte load and one 16-byte store per lane.
template <int RB, bool ADD>
__global__ void __launch_bounds__(NT) k_fir_x_shfl(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out, pp_dims d,
                                                   size_t cstride, taps_bucket<RB> taps, const int* __restrict__ halt) {
  if (halt && *halt) return;
  constexpr int NB4 = (RB + 3) / 4;                 // neighbour strips per side
  const size_t comp = (size_t)blockIdx.z * cstride;
  in += comp;
  out += comp;
  if (ADD) add += comp;
  const int nxv = d.nx / 4;
  const int spans = (nxv + 63) / 64;                // wavefront spans per row
  const size_t nrows = (size_t)d.ny * d.nz;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // block-uniform trip count: every thread runs every round (shuffles need the whole wavefront)
  const size_t nwork = nrows * spans;
  for (size_t w0 = (size_t)blockIdx.x * (NT / 64); w0 < nwork; w0 += (size_t)gridDim.x * (NT / 64)) {
    const size_t wk = w0 + wave;
    const bool live_w = wk < nwork;
    const size_t row = live_w ? wk / spans : 0;
    const int span = live_w ? (int)(wk % spans) : 0;
    const int xv = span * 64 + lane;
    const bool live = live_w && xv < nxv;
    const size_t rb = row * (size_t)d.nx;
    auto strip = [&](int sv, float (&v)[4]) {       // strip sv of the row, clamped element-wise at the row ends
      if (sv >= 0 && sv < nxv) {
        float4 t = *reinterpret_cast<const float4*>(in + rb + 4 * (size_t)sv);
        if (ADD) {
          const float4 u = *reinterpret_cast<const float4*>(add + rb + 4 * (size_t)sv);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        const size_t e = rb + (sv < 0 ? 0 : (size_t)d.nx - 1);
        float t = in[e];
        if (ADD) t += add[e];
        v[0] = v[1] = v[2] = v[3] = t;
      }
    };
    float c[4 * (2 * NB4 + 1)];                      // [left strips .. own .. right strips]
    float own[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) strip(xv, own);
#pragma unroll
    for (int j = 0; j < 4; ++j) c[4 * NB4 + j] = own[j];
#pragma unroll
    for (int s_ = 1; s_ <= NB4; ++s_) {
      float l[4], r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        l[j] = __shfl_up(own[j], s_, 64);
        r[j] = __shfl_down(own[j], s_, 64);
      }
      if (live && (lane < s_)) strip(xv - s_, l);                        // no lane to the left in this wavefront
      if (live && (lane + s_ > 63 || xv + s_ >= nxv)) strip(xv + s_, r); // ... or to the right / beyond the row
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[4 * (NB4 - s_) + j] = l[j];
        c[4 * (NB4 + s_) + j] = r[j];
      }
    }
    if (live) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 2 * RB + 1; ++k) a = fmaf(taps.w[k], c[4 * NB4 + j - RB + k], a);
        o[j] = a;
      }
      *reinterpret_cast<float4*>(out + rb + 4 * (size_t)xv) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

template <int P>
struct march_phase { static constexpr int value = P; };
template <int P, int L>
struct march_phases {   // step(pb + P, phase P) for P = 0 .. L - 1 while the position stays <= last
  template <class F>
  static __device__ __forceinline__ bool run(F& step, int pb, int last) {
    if (pb + P > last) return false;
    step(pb + P, march_phase<P>{});
    return march_phases<P + 1, L>::run(step, pb, last);
  }
};
template <int L>
struct march_phases<L, L> {
  template <class F>
  static __device__ __forceinline__ bool run(F&, int, int) { return true; }
};

template <int AXIS, int RB, int K>
__global__ void __launch_bounds__(NT) k_fir_march_sp(const float* __restrict__ in, float* __restrict__ out, pp_dims d, size_t cstride,
                                                     taps_bucket<RB> taps, const int* __restrict__ halt, const int* __restrict__ rows, int use_y,
                                                     int use_z, int nseg) {
  constexpr int W = 2 * RB + 1, L = W + K;
  if (halt && *halt) return;
  in += (size_t)blockIdx.z * cstride;
  out += (size_t)blockIdx.z * cstride;
  const int len = AXIS == 1 ? d.ny : d.nz;
  const int other = AXIS == 1 ? d.nz : d.ny;
  // outputs along the axis and lines across the other one: everything, or the lists of k_compact_rows
  const int* out_list = nullptr;
  const int* line_list = nullptr;
  int nout = len, nlines = other;
  if (rows) {
    if (AXIS == 1 ? use_y : use_z) {
      out_list = AXIS == 1 ? rows + 2 : rows + 2 + d.ny;
      nout = AXIS == 1 ? rows[0] : rows[1];
    }
    if (AXIS == 1 ? use_z : use_y) {
      line_list = AXIS == 1 ? rows + 2 + d.ny : rows + 2;
      nlines = AXIS == 1 ? rows[1] : rows[0];
    }
  }
  const size_t total = (size_t)d.nx * nlines;
  if ((size_t)blockIdx.x * NT >= total) return;
  const size_t col = (size_t)blockIdx.x * NT + threadIdx.x;
  const bool valid = col < total;
  const size_t colc = valid ? col : total - 1;
  const int x = (int)(colc % d.nx), li = (int)(colc / d.nx);
  const int line = line_list ? line_list[li] : li;
  const size_t stride = AXIS == 1 ? (size_t)d.nx : (size_t)d.nx * d.ny;
  const size_t base = (AXIS == 1 ? (size_t)line * d.nx * d.ny : (size_t)line * d.nx) + x;
  // this block's share of the outputs (consecutive list entries), marched from its first to its last
  const int per = (nout + nseg - 1) / nseg;
  int oi = (int)blockIdx.y * per;
  const int o_end = oi + per < nout ? oi + per : nout;
  if (oi >= o_end) return;
  const int p_first = out_list ? out_list[oi] : oi, p_last = out_list ? out_list[o_end - 1] : o_end - 1;
  auto ld = [&](int q) { return in[base + (size_t)pp_clampi(q, 0, len - 1) * stride]; };
  // slot (q - (p_first - RB)) % L holds extended position q; L - 1 of them loaded ahead
  float win[L];
#pragma unroll
  for (int j = 0; j < L - 1; ++j) win[j] = ld(p_first - RB + j);
  int o_next = p_first;
  // one step per position; instantiated once per ring phase u (L copies: the slots are renamed, never moved or indexed)
  auto step = [&](int pos, auto phase) {
    constexpr int u = decltype(phase)::value;
    win[(u + L - 1) % L] = ld(pos + RB + K);   // replaces position pos - RB - 1; read K steps from now
    if (pos == o_next) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < W; ++k) acc = fmaf(taps.w[k], win[(u + k) % L], acc);
      if (valid) out[base + (size_t)pos * stride] = acc;
      ++oi;
      o_next = oi < o_end ? (out_list ? out_list[oi] : oi) : -1;
    }
  };
  for (int pb = p_first; pb <= p_last; pb += L)
    if (!march_phases<0, L>::run(step, pb, p_last)) break;
}

template <int RB>
taps_bucket<RB> bucket_taps(const pp_taps& t) {
  taps_bucket<RB> b;
  for (int k = 0; k < 2 * RB + 1; ++k) {
    const int kk = k - RB + t.r;                     // index into the true taps
    b.w[k] = (kk >= 0 && kk <= 2 * t.r) ? t.w[kk] : 0.0f;
  }
  return b;
}

template <int AXIS, int RB, int VEC, bool ADD>
void launch_march(pp_ctx* ctx, const float* in, const float* add, float* out, const pp_dims& d, int ncomp, const pp_taps& taps, const int* halt) {
  const size_t cstride = (size_t)d.nx * d.ny * d.nz;
  const int len = AXIS == 1 ? d.ny : d.nz, other = AXIS == 1 ? d.nz : d.ny;
  const size_t ncol = (size_t)(d.nx / VEC) * other;
  const unsigned bx = (unsigned)((ncol + NT - 1) / NT);
  // segments: enough blocks to fill 256 CUs x 8, but not shorter than 8 windows (the 2 RB halo loads amortise)
  int nseg = (int)((2048 + bx - 1) / bx);
  const int min_seg = 8 * (2 * RB + 1) < 32 ? 32 : 8 * (2 * RB + 1);
  if (nseg > (len + min_seg - 1) / min_seg) nseg = (len + min_seg - 1) / min_seg;
  if (nseg < 1) nseg = 1;
  const int seg = (len + nseg - 1) / nseg;
  nseg = (len + seg - 1) / seg;
  hipLaunchKernelGGL((k_fir_march<AXIS, RB, VEC, ADD>), dim3(bx, (unsigned)nseg, (unsigned)ncomp), dim3(NT), 0, ctx->stream, in, add, out, d,
                     cstride, bucket_taps<RB>(taps), seg, halt);
}

template <int RB, bool ADD>
void launch_x_shfl(pp_ctx* ctx, const float* in, const float* add, float* out, const pp_dims& d, int ncomp, const pp_taps& taps, const int* halt) {
  const size_t cstride = (size_t)d.nx * d.ny * d.nz;
  const size_t nwork = (size_t)d.ny * d.nz * (((size_t)d.nx / 4 + 63) / 64);
  size_t blocks = (nwork + NT / 64 - 1) / (NT / 64);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((k_fir_x_shfl<RB, ADD>), dim3((unsigned)blocks, 1, (unsigned)ncomp), dim3(NT), 0, ctx->stream, in, add, out, d, cstride,
                     bucket_taps<RB>(taps), halt);
}

template <int AXIS, bool ADD>
int launch_axis(pp_ctx* ctx, const float* in, const float* add, float* out, const pp_dims& d, int ncomp,
                const pp_taps& taps, const int* halt, const int* rows = nullptr, int use_y = 0, int use_z = 0) {
  const bool al16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | (ADD ? reinterpret_cast<uintptr_t>(add) : 0)) % 16 == 0);
  if (!rows && !pp_env("PP_FIR_LEGACY")) {
    const int r = taps.r;
    if (AXIS != 0) {
      const bool v4 = al16 && (d.nx % 4 == 0);
#define PP_MARCH(RB, VEC)                                                                    \
  do {                                                                                       \
    launch_march<(AXIS == 0 ? 1 : AXIS), RB, VEC, ADD>(ctx, in, add, out, d, ncomp, taps, halt); \
    PP_LAUNCH_CHECK(ctx, "k_fir_march");                                                    \
    return PP_OK;                                                                            \
  } while (0)
      if (v4 && r <= 2) PP_MARCH(2, 4);
      if (v4 && r <= 4) PP_MARCH(4, 4);
      if (v4 && r <= 8) PP_MARCH(8, 4);
      // scalar columns: from radius 9 k_fir_march_sp (loads K steps ahead of their use) is the faster march -- measured
      // 0.64 -> 0.50 ms for the three passes of a variance-16 blur at 512 x 512 x 256 (r = 13); it has no ADD form
      if (ADD || r <= 8) {
        if (r <= 8) PP_MARCH(8, 1);
        if (r <= 16) PP_MARCH(16, 1);
      }
#undef PP_MARCH
    } else if (al16 && (d.nx % 4 == 0) && r <= 16) {
      if (r <= 4) launch_x_shfl<4, ADD>(ctx, in, add, out, d, ncomp, taps, halt);
      else if (r <= 8) launch_x_shfl<8, ADD>(ctx, in, add, out, d, ncomp, taps, halt);
      else launch_x_shfl<16, ADD>(ctx, in, add, out, d, ncomp, taps, halt);
      PP_LAUNCH_CHECK(ctx, "k_fir_x_shfl");
      return PP_OK;
    }
  }
  const size_t cstride = (size_t)d.nx * d.ny * d.nz;
  // sparse outputs, or a radius beyond the dense register-window buckets (PP_FIR_MARCH_SP=0 keeps the one-output-per-thread kernels)
  const char* sp_env = pp_env("PP_FIR_MARCH_SP");   // (once per pass of a once-per-level filter)
  if (!ADD && !(sp_env && atoi(sp_env) == 0) && !pp_env("PP_FIR_LEGACY") && taps.r <= 32) {
    if (AXIS != 0 && taps.r <= 24) {
      const int len = AXIS == 1 ? d.ny : d.nz, other = AXIS == 1 ? d.nz : d.ny;
      const int W = 2 * taps.r + 1;
      const unsigned bx = (unsigned)(((size_t)d.nx * other + NT - 1) / NT);
      // segments of the outputs: enough blocks to fill the chip a few times over, each with >= 4 windows of outputs' inputs
      int nseg = (int)((1536 + bx - 1) / bx);
      const int max_seg = len / (4 * W) > 1 ? len / (4 * W) : 1;
      if (nseg > max_seg) nseg = max_seg;
      if (nseg < 1) nseg = 1;
      const dim3 grid(bx, (unsigned)nseg, (unsigned)ncomp);
#define PP_MSP(RB, KK) hipLaunchKernelGGL((k_fir_march_sp<(AXIS == 0 ? 1 : AXIS), RB, KK>), grid, dim3(NT), 0, ctx->stream, in, out, d, cstride, bucket_taps<RB>(taps), halt, rows, use_y, use_z, nseg)
      if (taps.r <= 12) PP_MSP(12, 24);
      else PP_MSP(24, 24);   // (a bucket of 32 doubles this file's compile time for sigmas between 9 and 12 voxels: left to k_conv_axis)
#undef PP_MSP
      PP_LAUNCH_CHECK(ctx, "k_fir_march_sp");
      return PP_OK;
    } else if (AXIS == 0 && d.nx <= 4096) {   // (dense x passes of radius <= 16 went to k_fir_x_shfl above)
      const int W = 2 * taps.r + 1;
#define PP_XROW(WB) hipLaunchKernelGGL((k_fir_x_row<WB>), dim3(8192, (unsigned)ncomp, 1), dim3(NT), 0, ctx->stream, in, out, d, cstride, taps, halt, rows, use_y, use_z)
      if (W <= 15) PP_XROW(15);
      else if (W <= 23) PP_XROW(23);
      else if (W <= 31) PP_XROW(31);
      else if (W <= 47) PP_XROW(47);
      else PP_XROW(65);
#undef PP_XROW
      PP_LAUNCH_CHECK(ctx, "k_fir_x_row");
      return PP_OK;
    }
  }
  if (AXIS == 0 && (d.nx % 4 == 0) && taps.r >= 3 &&
      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | (ADD ? reinterpret_cast<uintptr_t>(add) : 0)) % 16 == 0)) {
    size_t blocks = (cstride / 4 + NT - 1) / NT;
    if (blocks > 65535u * 4u) blocks = 65535u * 4u;
    if (rows && blocks > 4096) blocks = 4096;   // the number of listed rows is known on the device only: grid-stride over them
    hipLaunchKernelGGL((k_conv_x4<ADD>), dim3((unsigned)blocks, (unsigned)ncomp, 1), dim3(NT, 1, 1), 0, ctx->stream, in, add, out, d,
                       cstride, taps, halt, rows, use_y, use_z);
    PP_LAUNCH_CHECK(ctx, "k_conv_x4");
    return PP_OK;
  }
  const bool vec4 = AXIS != 0 && (d.nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) |
                                                       (ADD ? reinterpret_cast<uintptr_t>(add) : 0)) % 16 == 0);
  const size_t work = cstride / (vec4 ? 4 : 1);
  size_t blocks = (work + NT - 1) / NT;
  if (blocks > 65535u * 4u) blocks = 65535u * 4u;
  if (rows && blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks, (unsigned)ncomp, 1), block(NT, 1, 1);
  if (vec4)
    hipLaunchKernelGGL((k_conv_axis<AXIS, 4, ADD>), grid, block, 0, ctx->stream, in, add, out, d, cstride, taps, halt, rows, use_y, use_z);
  else
    hipLaunchKernelGGL((k_conv_axis<AXIS, 1, ADD>), grid, block, 0, ctx->stream, in, add, out, d, cstride, taps, halt, rows, use_y, use_z);
  PP_LAUNCH_CHECK(ctx, "k_conv_axis");
  return PP_OK;
}

}  // namespace

int pp_conv_axis(pp_ctx* ctx, int axis, const float* in, const float* add, float* out, const pp_dims& d, int ncomp,
                 const pp_taps& taps, const int* halt) {
  if (add) {
    switch (axis) {
      case 0: return launch_axis<0, true>(ctx, in, add, out, d, ncomp, taps, halt);
      case 1: return launch_axis<1, true>(ctx, in, add, out, d, ncomp, taps, halt);
      default: return launch_axis<2, true>(ctx, in, add, out, d, ncomp, taps, halt);
    }
  }
  switch (axis) {
    case 0: return launch_axis<0, false>(ctx, in, nullptr, out, d, ncomp, taps, halt);
    case 1: return launch_axis<1, false>(ctx, in, nullptr, out, d, ncomp, taps, halt);
    default: return launch_axis<2, false>(ctx, in, nullptr, out, d, ncomp, taps, halt);
  }
}

// ---------------------------------------------------------------------------------------
// DiscreteGaussian of a scalar volume with small radii (<= 4 on every axis) as ONE kernel (round 4): z, y, x in one sweep.
//
// The three-launch form moves 24 bytes per voxel (each pass reads and writes the volume); this one reads it once and writes
// it once.  A 64 x 16 tile (plus an R-voxel rim in x and y, fetched as 16-byte strips) marches along z:
//   z pass   a thread owns one strip of four x voxels of the rimmed tile and keeps its last 2R + 1 planes in registers;
//            the z-filtered strip goes to an LDS tile (two buffers: one barrier per plane);
//   y pass   a thread owns one strip of one OUTPUT row: 2R + 1 ds_read_b128 down the tile;
//   x pass   in registers: the 18 strips of a row sit in adjacent lanes (three rows per wavefront), the R voxels either side
//            come from the neighbouring lanes by DPP whole-wave shifts; the sixteen interior strips store 16 bytes.
// Arithmetic: each pass is the chain acc = fmaf(w[k], v[k], acc) from the lowest index up, in fp32, on fp32 intermediates --
// k_fir_march / k_fir_x_shfl operation for operation, taps beyond an axis's own radius being zero (they add +0) -- so the
// output is bit-identical to the three passes (tests/test_kernels.py).  Rim voxels outside the volume repeat the clamped
// INPUT voxel; the z pass acts on each (x, y) column alone, so that is the z-filtered edge value the staged y pass would
// have clamped to, and likewise for x after y (ZeroFluxNeumann on every intermediate).
// Resource over the first 2^31 bytes only: a lane whose per-lane offset is FIR_OOB (= 2^31) is outside it and the hardware
// drops its store -- the lane mask of a store without a branch around it (same device as the fused demons kernels' PP_OOB).
constexpr unsigned FIR_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fir_rsrc_first_2g(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)FIR_OOB, 0x00020000);
}

template <int R>
struct gauss3_taps {
  float wz[2 * R + 1], wy[2 * R + 1], wx[2 * R + 1];
};
template <int R>
__global__ void __launch_bounds__(512) k_gauss3_zyx(const float* __restrict__ in, float* __restrict__ out, pp_dims d, gauss3_taps<R> taps,
                                                     int zchunk, int gx, int gy, int per_xcd) {
  constexpr int TX = 64, TY = 16, W = 2 * R + 1;
  constexpr int SPR = TX / 4 + 2;            // strips per row of the rimmed tile (one rim strip either side: R <= 4)
  constexpr int UW = 4 * SPR, UH = TY + 2 * R;
  constexpr int NS = SPR * UH;               // strips per plane (<= 432)
  constexpr int RPW = 64 / SPR;              // rows per wavefront in the y / x pass (3)
  static_assert(NS <= 512 && (TY + RPW - 1) / RPW <= 8, "one strip per thread, the output rows fit the block's waves");
  __shared__ __attribute__((aligned(16))) float tile[2][UH * UW];
  // XCD-aware tile order as the fused demons kernels: block b runs on XCD b % 8 and takes tile rank (b % 8) * per_xcd + b / 8
  const unsigned b = blockIdx.x, rank = (b & 7u) * (unsigned)per_xcd + (b >> 3);
  const unsigned gz = (unsigned)((d.nz + zchunk - 1) / zchunk);
  if (rank >= (unsigned)gx * gy * gz) return;
  const int tx0 = (int)(rank % gx) * TX, ty0 = (int)((rank / gx) % gy) * TY, z0 = (int)(rank / ((unsigned)gx * gy)) * zchunk;
  const int z1 = z0 + zchunk < d.nz ? z0 + zchunk : d.nz;
  const int t = threadIdx.x;
  const size_t sz = (size_t)d.nx * d.ny;

  // z pass ownership: strip t of the rimmed tile
  const bool zs_live = t < NS;
  const int zs = zs_live ? t : 0;
  const int uy = zs / SPR, sx = zs - uy * SPR;
  const int xs = tx0 - 4 + 4 * sx, yc = pp_clampi(ty0 - R + uy, 0, d.ny - 1);
  const int xl = pp_clampi(xs, 0, d.nx - 4);
  unsigned jm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) jm |= (unsigned)(pp_clampi(xs + i, 0, d.nx - 1) - xl) << (2 * i);
  const size_t g_in = (size_t)yc * d.nx + xl;
  auto fetch = [&](int q) -> float4 {   // the raw strip of plane q (clamped in z): every lane loads, lanes without a strip re-read strip 0
    q = q < 0 ? 0 : (q > d.nz - 1 ? d.nz - 1 : q);
    return *reinterpret_cast<const float4*>(in + (size_t)q * sz + g_in);
  };
  auto place = [&](const float4& r, float (&v)[4]) {
    if (jm == 0xE4u) {
      v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    } else {   // strips that leave the volume in x: every position takes its clamped voxel
      const float e[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned s_ = (jm >> (2 * i)) & 3u;
        v[i] = s_ == 0 ? e[0] : (s_ == 1 ? e[1] : (s_ == 2 ? e[2] : e[3]));
      }
    }
  };
  // y / x pass ownership: strip sxo of output row oy, rows dealt RPW per wavefront
  const int lane = t & 63, riw = lane / SPR, sxo = lane - riw * SPR;
  const int oy = (t >> 6) * RPW + riw;
  const bool ys_live = riw < RPW && oy < TY;
  const int xo = tx0 + 4 * (sxo - 1), yo = ty0 + oy;
  const bool st_live = ys_live && sxo >= 1 && sxo <= SPR - 2 && xo < d.nx && yo < d.ny;   // (nx % 4 == 0: whole strips)
  // the store's lane mask travels in its offset (FIR_OOB: dropped by the hardware), so that every wave issues it on every
  // step: with the store inside `if (st_live)` the wait for the plane requested a step ahead is emitted as vmcnt(0) and also
  // sits out the store's acknowledgement (the demons kernels' MASK instances, pp_demons_fused2.h)
  const __amdgpu_buffer_rsrc_t r_out = fir_rsrc_first_2g(out);
  const unsigned g_out4 = st_live ? (unsigned)(((size_t)yo * d.nx + xo) * 4u) : FIR_OOB;
  const unsigned sz4 = (unsigned)(sz * 4u);

  float win[W][4];
#pragma unroll
  for (int k = 0; k < W; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) win[k][i] = 0.0f;
  {
    float4 raw[W - 1];
#pragma unroll
    for (int k = 1; k < W; ++k) raw[k - 1] = fetch(z0 - R + k - 1);   // all in flight together
#pragma unroll
    for (int k = 1; k < W; ++k) place(raw[k - 1], win[k]);            // slots 1 .. W-1: planes z0-R .. z0+R-1
  }
  float4 ahead = fetch(z0 + R);   // the plane the first step appends; every step requests the next one before it computes
  {
    // A store that stores nothing (every lane out of range), issued behind that load: the loop body is [wait for `ahead`,
    // request the next plane, ..., store], so on the way round exactly one store is younger than the load waited for -- with
    // the same picture on the way in, the compiler's wait is vmcnt(1) at both and never drains the output store.
    typedef unsigned pp_u4 __attribute__((vector_size(16)));
    const pp_u4 nothing = {0u, 0u, 0u, 0u};
    __builtin_amdgcn_raw_buffer_store_b128(nothing, r_out, FIR_OOB, 0u, 0);
  }
  for (int z = z0; z < z1; ++z) {
    float* const tb = tile[(z - z0) & 1];
    {
#pragma unroll
      for (int k = 0; k < W - 1; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) win[k][i] = win[k + 1][i];
      float4 cur = ahead;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w));   // (the wait for the strip sits HERE, in straight-line code with an exact count -- not inside place()'s lane-divergent branches)
#endif
      ahead = fetch(z + 1 + R);   // (no branch around the load either)
      place(cur, win[W - 1]);
    }
    if (zs_live) {
      float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < W; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fmaf(taps.wz[k], win[k][i], a[i]);
      *reinterpret_cast<float4*>(tb + uy * UW + 4 * sx) = make_float4(a[0], a[1], a[2], a[3]);
    }
    __syncthreads();
    // (the other buffer is written in the next step, behind this barrier: every read of it finished before the barrier of
    // the step after)
    float own[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ys_live) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const float4 r = *reinterpret_cast<const float4*>(tb + (oy + k) * UW + 4 * sxo);
        own[0] = fmaf(taps.wy[k], r.x, own[0]);
        own[1] = fmaf(taps.wy[k], r.y, own[1]);
        own[2] = fmaf(taps.wy[k], r.z, own[2]);
        own[3] = fmaf(taps.wy[k], r.w, own[3]);
      }
    }
    // x pass: [prev lane's strip | own | next lane's strip], every lane of the wavefront takes part in the shifts
    float c[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c[i] = pp_lane_prev(own[i]);
      c[4 + i] = own[i];
      c[8 + i] = pp_lane_next(own[i]);
    }
    {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k) acc = fmaf(taps.wx[k], c[4 + j - R + k], acc);
        o[j] = acc;
      }
      typedef unsigned pp_u4 __attribute__((vector_size(16)));
      pp_u4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = __builtin_bit_cast(unsigned, o[j]);
      __builtin_amdgcn_raw_buffer_store_b128(w, r_out, g_out4, (unsigned)z * sz4, 0);
    }
  }
}

template <int R>
int launch_gauss3(pp_ctx* ctx, const float* in, float* out, const pp_dims& d, const pp_taps taps[3]) {
  gauss3_taps<R> g;
  for (int k = 0; k < 2 * R + 1; ++k) {   // an axis's taps centred in the bucket, zeros beyond its own radius
    const int o = k - R;
    g.wz[k] = (o < -taps[2].r || o > taps[2].r) ? 0.0f : taps[2].w[o + taps[2].r];
    g.wy[k] = (o < -taps[1].r || o > taps[1].r) ? 0.0f : taps[1].w[o + taps[1].r];
    g.wx[k] = (o < -taps[0].r || o > taps[0].r) ? 0.0f : taps[0].w[o + taps[0].r];
  }
  const int gx = (d.nx + 63) / 64, gy = (d.ny + 15) / 16;
  // z-chunks: enough tiles for ~4 blocks on each of the 256 CUs, chunks no shorter than 8 planes (2R planes are re-read per chunk)
  int zchunk = d.nz;
  while (zchunk > 8 && (long)gx * gy * ((d.nz + zchunk - 1) / zchunk) < 1024) zchunk = (zchunk + 1) / 2;
  const int gz = (d.nz + zchunk - 1) / zchunk;
  const int per_xcd = (int)(((long)gx * gy * gz + 7) / 8);
  hipLaunchKernelGGL((k_gauss3_zyx<R>), dim3(8u * (unsigned)per_xcd), dim3(512), 0, ctx->stream, in, out, d, g, zchunk, gx, gy, per_xcd);
  PP_LAUNCH_CHECK(ctx, "k_gauss3_zyx");
  return PP_OK;
}

// dst = G_order[2] G_order[1] G_order[0] (src [+ add]); tmp1/tmp2 are ncomp-plane scratch.
// dst may alias src (and add): the last pass reads only tmp2.
int pp_smooth3_staged(pp_ctx* ctx, const float* src, const float* add, float* dst, float* tmp1, float* tmp2,
                      const pp_dims& d, int ncomp, const pp_taps taps[3], const int order[3], const int* halt) {
  int rc = pp_conv_axis(ctx, order[0], src, add, tmp1, d, ncomp, taps[order[0]], halt);
  if (rc) return rc;
  rc = pp_conv_axis(ctx, order[1], tmp1, nullptr, tmp2, d, ncomp, taps[order[1]], halt);
  if (rc) return rc;
  return pp_conv_axis(ctx, order[2], tmp2, nullptr, dst, d, ncomp, taps[order[2]], halt);
}

extern "C" {

int pp_discrete_gaussian_f32(pp_ctx* ctx, const float* in, float* out, const int size[3], const double spacing[3],
                             const double variance[3], double max_error, int max_kernel_width, int use_image_spacing) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && size && spacing && variance, "pp_discrete_gaussian_f32: NULL argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_discrete_gaussian_f32: empty volume");
  const pp_dims d{size[0], size[1], size[2]};
  pp_taps taps[3];
  for (int a = 0; a < 3; ++a) {
    double var = variance[a];
    if (use_image_spacing) {
      PP_REQUIRE(ctx, spacing[a] > 0.0, "pp_discrete_gaussian_f32: spacing must be positive");
      var /= spacing[a] * spacing[a];
    }
    const int rc = pp_make_taps(ctx, var, max_error, max_kernel_width, &taps[a]);
    if (rc) return rc;
  }
  const size_t N = pp_nvox(size);
  // small radii on a volume with whole 16-byte strips per row: one kernel, z, y, x in one sweep (k_gauss3_zyx; in != out there)
  int rmax = taps[0].r > taps[1].r ? taps[0].r : taps[1].r;
  if (taps[2].r > rmax) rmax = taps[2].r;
  bool fused = rmax <= 4 && d.nx % 4 == 0 && d.nx >= 8 && in != out && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
               N * sizeof(float) < ((size_t)1 << 31);   // (the output goes through a 2^31-byte buffer resource)
  if (const char* e = pp_env("PP_GAUSS3")) fused = fused && atoi(e) != 0;   // (0: the three separable launches, for A/B runs)
  if (fused) {
    pp_prof_scope ps(ctx, "k_gauss3_zyx");
    switch (rmax < 1 ? 1 : rmax) {
      case 1: return launch_gauss3<1>(ctx, in, out, d, taps);
      case 2: return launch_gauss3<2>(ctx, in, out, d, taps);
      case 3: return launch_gauss3<3>(ctx, in, out, d, taps);
      default: return launch_gauss3<4>(ctx, in, out, d, taps);
    }
  }
  int rc = pp_reserve(ctx, 2 * pp_align_up(N * sizeof(float), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* t1 = cv.take<float>(N);
  float* t2 = cv.take<float>(N);
  // The ITK mini-pipeline convolves the last axis first: z, y, x.
  const int order[3] = {2, 1, 0};
  return pp_smooth3_staged(ctx, in, nullptr, out, t1, t2, d, 1, taps, order, nullptr);
}

// DiscreteGaussian whose result will only be read at rows (y, z) with need_y[y] && need_z[z] (masks in device
// memory) -- the pyramid's blur feeds a resample onto a several-times coarser grid (registration/utils.py:226 then
// :257-267), which touches a fraction of the rows.  Pass order z, y, x as above: the z pass runs for the needed z
// (all y: the y pass reads along y), the y and x passes for the needed (y, z).  Every value that is produced is the
// value the dense filter produces; the other entries of `out` are unspecified.
int pp_discrete_gaussian_rows_f32(pp_ctx* ctx, const float* in, float* out, const int size[3], const double spacing[3],
                                  const double variance[3], double max_error, int max_kernel_width, int use_image_spacing,
                                  const uint8_t* need_y, const uint8_t* need_z) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, in && out && size && spacing && variance && need_y && need_z, "pp_discrete_gaussian_rows_f32: NULL argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_discrete_gaussian_rows_f32: empty volume");
  const pp_dims d{size[0], size[1], size[2]};
  pp_taps taps[3];
  for (int a = 0; a < 3; ++a) {
    double var = variance[a];
    if (use_image_spacing) {
      PP_REQUIRE(ctx, spacing[a] > 0.0, "pp_discrete_gaussian_rows_f32: spacing must be positive");
      var /= spacing[a] * spacing[a];
    }
    const int rc = pp_make_taps(ctx, var, max_error, max_kernel_width, &taps[a]);
    if (rc) return rc;
  }
  const size_t N = pp_nvox(size);
  int rc = pp_reserve(ctx, 2 * pp_align_up(N * sizeof(float), 256) + pp_align_up((2 + (size_t)d.ny + d.nz) * sizeof(int), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* t1 = cv.take<float>(N);
  float* t2 = cv.take<float>(N);
  int* rows = cv.take<int>(2 + (size_t)d.ny + d.nz);
  hipLaunchKernelGGL(k_compact_rows, dim3(1), dim3(NT), 0, ctx->stream, need_y, d.ny, need_z, d.nz, rows);
  PP_LAUNCH_CHECK(ctx, "k_compact_rows");
  rc = launch_axis<2, false>(ctx, in, nullptr, t1, d, 1, taps[2], nullptr, rows, 0, 1);
  if (rc) return rc;
  rc = launch_axis<1, false>(ctx, t1, nullptr, t2, d, 1, taps[1], nullptr, rows, 1, 1);
  if (rc) return rc;
  return launch_axis<0, false>(ctx, t2, nullptr, out, d, 1, taps[0], nullptr, rows, 1, 1);
}

int pp_smooth_field_f32(pp_ctx* ctx, float* field, const int size[3], const double sigma_vox[3], double max_error,
                        int max_kernel_width) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, field && size && sigma_vox, "pp_smooth_field_f32: NULL argument");
  PP_REQUIRE(ctx, size[0] > 0 && size[1] > 0 && size[2] > 0, "pp_smooth_field_f32: empty volume");
  const pp_dims d{size[0], size[1], size[2]};
  pp_taps taps[3];
  for (int a = 0; a < 3; ++a) {
    const int rc = pp_make_taps(ctx, sigma_vox[a] * sigma_vox[a], max_error, max_kernel_width, &taps[a]);
    if (rc) return rc;
  }
  const size_t N = pp_nvox(size);
  int rc = pp_reserve(ctx, 2 * pp_align_up(3 * N * sizeof(float), 256));
  if (rc) return rc;
  pp_carver cv{ctx->ws, 0};
  float* t1 = cv.take<float>(3 * N);
  float* t2 = cv.take<float>(3 * N);
  const int order[3] = {0, 1, 2};  // SmoothDisplacementField: x, y, z
  return pp_smooth3_staged(ctx, field, nullptr, field, t1, t2, d, 3, taps, order, nullptr);
}

}  // extern "C"
