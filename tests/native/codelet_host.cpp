// Host build of dct_codelets.hpp for the CPU test-suite (test infrastructure only:
// nothing in the product path loads this). Exposes the unnormalised 1-D codelets and a
// 2-D orthonormal transform + energy built from them the same way the HIP kernel does.
#include <cstdint>
#include <vector>
#include "../../dct_pruning_amd/csrc/dct_codelets.hpp"
#include "../../dct_pruning_amd/csrc/codelet_sizes.h"

using namespace dcts;

template <int N>
static void run1d(const float* x, float* X) {
  float a[N], b[N];
  for (int i = 0; i < N; ++i) a[i] = x[i];
  Dct2<N>::run(a, b);
  for (int i = 0; i < N; ++i) X[i] = b[i];
}

template <int N>
static void run4(const float* x, float* X) {
  float a[N], b[N];
  for (int i = 0; i < N; ++i) a[i] = x[i];
  Dct4<N>::run(a, b);
  for (int i = 0; i < N; ++i) X[i] = b[i];
}

// 2-D: columns first (length H), then rows (length W), as in k_energy_codelet.
template <int H, int W>
static float energy2d(const float* x, float* coeff) {
  std::vector<float> t(H * W);
  for (int c = 0; c < W; ++c) {
    float col[H], out[H];
    for (int r = 0; r < H; ++r) col[r] = x[r * W + c];
    Dct2<H>::run(col, out);
    out[0] *= kInvSqrt2;
    for (int k = 0; k < H; ++k) t[k * W + c] = out[k];
  }
  float e = 0.f;
  const float s = 2.0f / float(cx_sqrt(double(H) * double(W)));
  for (int k = 0; k < H; ++k) {
    float row[W], out[W];
    for (int c = 0; c < W; ++c) row[c] = t[k * W + c];
    Dct2<W>::run(row, out);
    out[0] *= kInvSqrt2;
    for (int l = 0; l < W; ++l) {
      e += out[l] * out[l];
      if (coeff) coeff[k * W + l] = out[l] * s;
    }
  }
  return e * (4.0f / float(H * W));
}

extern "C" {
int codelet_dct2_1d(int n, const float* x, float* X) {
  switch (n) {
#define DCTS_CASE(N) case N: run1d<N>(x, X); return 0;
    DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  }
  return -1;
}
int codelet_dct4_1d(int n, const float* x, float* X) {
  switch (n) {
#define DCTS_CASE(N) case N: run4<N>(x, X); return 0;
    DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  }
  return -1;
}
int codelet_energy_2d(int n, const float* x, float* coeff, float* energy) {
  switch (n) {
#define DCTS_CASE(N) case N: *energy = energy2d<N, N>(x, coeff); return 0;
    DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  }
  return -1;
}
}
