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

// every length 1 ... 64: what rect.hip instantiates (its odd and odd-part lengths have no square kernel of their own)
#define DCTS_ALL_SIZES(X) \
  X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64)

extern "C" {
int codelet_dct2_1d_any(int n, const float* x, float* X) {
  switch (n) {
#define DCTS_CASE(N) case N: run1d<N>(x, X); return 0;
    DCTS_ALL_SIZES(DCTS_CASE)
#undef DCTS_CASE
  }
  return -1;
}
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
