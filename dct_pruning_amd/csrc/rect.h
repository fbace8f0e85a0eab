// rect.h - what dct_kernels.hip needs to know of rect.hip (non-square / non-dense-row tiles through the 1-D codelets)
#pragma once
#include <hip/hip_runtime.h>

namespace dctsi {
struct RectGeom {
  const float* x;       // first sample, channel 0
  long long nmaps;      // N * c_count
  long long strideN, strideC, strideH;  // elements
  int c_count, c_begin;
  int H, W;             // data dims (before the odd front pad)
  int HP, WP, pad;      // transformed dims (H + pad, W + pad), pad = 0 | 1
  int G, G1, G2, S, map_lds;  // maps per wave iteration, per pass-1 / pass-2 step; LDS row stride and floats per map
  int contiguous;       // map m starts at x + (c_begin + m) * strideC
  float scale_e, scale_c;  // 4 / (HP * WP) for the energy, 2 / sqrt(HP * WP) for coefficients (host, from double)
};
int has_rect(int HP, int WP);
int dispatch_rect(const RectGeom& g, float* out, int store_coeff, hipStream_t st);
}  // namespace dctsi
