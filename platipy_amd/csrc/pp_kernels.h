// platipy_amd/csrc/pp_kernels.h -- internal entry points shared between the .hip files.
#pragma once
#include "pp_internal.h"

// pp_fir.hip
int pp_conv_axis(pp_ctx* ctx, int axis, const float* in, const float* add, float* out, const pp_dims& d, int ncomp,
                 const pp_taps& taps, const int* halt);
int pp_smooth3_staged(pp_ctx* ctx, const float* src, const float* add, float* dst, float* tmp1, float* tmp2,
                      const pp_dims& d, int ncomp, const pp_taps taps[3], const int order[3], const int* halt);

// pp_resample.hip
struct pp_warp_scale {
  float ix, iy, iz;  // 1 / spacing: mm -> voxels on an axis-aligned grid
};
int pp_warp_same_grid(pp_ctx* ctx, const float* moving, const float* field, const pp_dims& d, const pp_warp_scale& sc,
                      float edge_value, float* out, const int* halt_flag);
