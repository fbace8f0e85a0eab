// dct_codelets.hpp — straight-line, register-resident 1-D DCT-II codelets.
//
// One lane transforms a whole length-N vector held in its own VGPRs: no cross-lane
// traffic, no LDS, every twiddle an instruction literal. The kernels in
// dct_kernels.hip run one such codelet per lane along H, transpose the tile through
// LDS, and run a second one along W.
//
// What is computed (the unnormalised DCT-II; the orthonormal scaling of
// torch_dct.dct(norm='ortho') — SURVEY.md Appendix B step 4, reference call site
// utils/common.py:267 — is applied by the caller):
//
//     X[k] = sum_{n<N} x[n] * cos(pi * (2n+1) * k / (2N)),     k = 0..N-1
//
// Factorisation (any N = 2^s * m, m odd):
//   even N : u[n] = x[n] + x[N-1-n], v[n] = x[n] - x[N-1-n]  (n < N/2)
//            X[2k]   = DCT-II_{N/2}(u)[k]
//            X[2k+1] = DCT-IV_{N/2}(v)[k]
//   DCT-IV_M, M even, M1 = M/2, beta_n = (2n+1)pi/(4M):
//            a[n] =          v[n] cos(beta_n) + v[M-1-n] sin(beta_n)
//            b[n] = (-1)^n (-v[n] sin(beta_n) + v[M-1-n] cos(beta_n))
//            A = DCT-II_{M1}(a), B = DCT-II_{M1}(b)
//            X[0] = A[0], X[M-1] = -B[0],
//            X[2j] = A[j] + B[M1-j], X[2j-1] = A[j] - B[M1-j]   (0 < j < M1)
//            (rotations are orthogonal and the tail is add/sub only: numerically stable)
//   odd N  : direct, using the x[n] <-> x[N-1-n] symmetry (about N^2/2 FMAs)
//   DCT-IV odd M : direct M x M.
//
// This header is plain C++17 and is also compiled for the host by the CPU test-suite
// (tests/native/codelet_host.cpp) so the algebra is checked without a GPU.
#pragma once

#include <utility>

#if defined(__HIPCC__)
#define DCTS_HD __host__ __device__ __forceinline__
#else
#define DCTS_HD inline __attribute__((always_inline))
#endif

namespace dcts {

// ---------------------------------------------------------------------------------
// compile-time trigonometry: cos(pi * p / q) for integers p, q > 0, evaluated in double
// ---------------------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_taylor(double x) {  // |x| <= pi/4
  const double x2 = x * x;
  double term = x, sum = x;
  for (int i = 1; i <= 12; ++i) {
    term *= -x2 / double((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double cx_cos_taylor(double x) {  // |x| <= pi/4
  const double x2 = x * x;
  double term = 1.0, sum = 1.0;
  for (int i = 1; i <= 12; ++i) {
    term *= -x2 / double((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
// cos(pi * p / q), exact integer range reduction to [0, pi/4]
constexpr double cospi_frac(long long p, long long q) {
  long long r = p % (2 * q);
  if (r < 0) r += 2 * q;
  if (r > q) r = 2 * q - r;  // cos(2pi - t) = cos t          -> t in [0, pi]
  double sign = 1.0;
  if (2 * r > q) {           // cos(pi - t) = -cos t          -> t in [0, pi/2]
    r = q - r;
    sign = -1.0;
  }
  if (4 * r > q) {           // cos t = sin(pi/2 - t), pi/2 - t = pi (q - 2r) / (2q)
    return sign * cx_sin_taylor(kPi * double(q - 2 * r) / double(2 * q));
  }
  return sign * cx_cos_taylor(kPi * double(r) / double(q));
}
constexpr double sinpi_frac(long long p, long long q) {  // sin(pi p/q) = cos(pi (q - 2p) / (2q))
  return cospi_frac(q - 2 * p, 2 * q);
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N)
template <class F, int... I>
DCTS_HD void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
DCTS_HD void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

#define DCTS_LAMBDA_INLINE __attribute__((always_inline))

// Two independent values carried through the same arithmetic: on gfx950 every +, -, * on this type
// is one packed-f32 instruction (v_pk_add/mul/fma_f32, two results per issue slot). A DCT-IV of
// even length M spawns two DCT-II of length M/2 on the rotated halves a, b: they run as ONE
// transform on pairs (a[n], b[n]): 17 % fewer VALU instructions at 56x56 (1385 -> 1151). MEASURED
// SLOWER on MI355X (56x56: 4.9 -> 4.4 TB/s): the packed-f32 instructions issue at half rate here
// (the timings fit 8 cycles per v_pk_* against 4 per scalar op), so two results per instruction buy
// nothing and the pair/unpair moves cost extra. Kept behind DCTS_PAIRS for other parts.
typedef float f2 __attribute__((vector_size(8)));

template <class T>
struct is_pair {
  static constexpr bool value = false;
};
template <>
struct is_pair<f2> {
  static constexpr bool value = true;
};

template <int N>
struct Dct2;
template <int M>
struct Dct4;

// ---------------------------------------------------------------------------------
// DCT-IV:  X[k] = sum_n v[n] cos(pi (2n+1)(2k+1) / (4M))
// ---------------------------------------------------------------------------------
template <int M>
struct Dct4 {
  template <class T>
  static DCTS_HD void run(const T (&v)[M], T (&X)[M]) {
    if constexpr (M == 1) {
      constexpr float c = float(cospi_frac(1, 4));
      X[0] = v[0] * c;
    } else if constexpr (M % 2 == 0) {
      constexpr int H = M / 2;
#ifdef DCTS_PAIRS
      if constexpr (!is_pair<T>::value) {
        // a[n], b[n] as the halves of one pair; the two half-length DCT-II run as one
        f2 ab[H];
        static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
          constexpr int n = decltype(i)::value;
          constexpr float c = float(cospi_frac(2 * n + 1, 4 * M));
          constexpr float s = float(sinpi_frac(2 * n + 1, 4 * M));
          constexpr float sg = (n % 2 == 0) ? 1.0f : -1.0f;
          const f2 p = {v[n], v[M - 1 - n]}, q = {v[M - 1 - n], v[n]};
          const f2 cc = {c, sg * c}, ss = {s, -sg * s};
          ab[n] = p * cc + q * ss;
        });
        f2 AB[H];
        Dct2<H>::run(ab, AB);
        X[0] = AB[0][0];
        X[M - 1] = -AB[0][1];
        static_for<H - 1>([&](auto i) DCTS_LAMBDA_INLINE {
          constexpr int j = decltype(i)::value + 1;
          X[2 * j] = AB[j][0] + AB[H - j][1];
          X[2 * j - 1] = AB[j][0] - AB[H - j][1];
        });
        return;
      }
#endif
      T a[H], b[H];
      static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int n = decltype(i)::value;
        constexpr float c = float(cospi_frac(2 * n + 1, 4 * M));
        constexpr float s = float(sinpi_frac(2 * n + 1, 4 * M));
        constexpr float sg = (n % 2 == 0) ? 1.0f : -1.0f;
        a[n] = v[n] * c + v[M - 1 - n] * s;
        b[n] = v[M - 1 - n] * (sg * c) - v[n] * (sg * s);
      });
      T A[H], B[H];
      Dct2<H>::run(a, A);
      Dct2<H>::run(b, B);
      X[0] = A[0];
      X[M - 1] = -B[0];
      static_for<H - 1>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int j = decltype(i)::value + 1;
        X[2 * j] = A[j] + B[H - j];
        X[2 * j - 1] = A[j] - B[H - j];
      });
    } else {
      static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
        constexpr int k = decltype(ik)::value;
        T acc{};
        static_for<M>([&](auto in) DCTS_LAMBDA_INLINE {
          constexpr int n = decltype(in)::value;
          constexpr float c = float(cospi_frac((2 * n + 1) * (2 * k + 1), 4 * M));
          if constexpr (n == 0)
            acc = v[0] * c;
          else
            acc += v[n] * c;
        });
        X[k] = acc;
      });
    }
  }
};

// ---------------------------------------------------------------------------------
// DCT-II:  X[k] = sum_n x[n] cos(pi (2n+1) k / (2N))
// ---------------------------------------------------------------------------------
template <int N>
struct Dct2 {
  template <class T>
  static DCTS_HD void run(const T (&x)[N], T (&X)[N]) {
    if constexpr (N == 1) {
      X[0] = x[0];
    } else if constexpr (N % 2 == 0) {
      constexpr int H = N / 2;
      T u[H], v[H];
      static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int n = decltype(i)::value;
        u[n] = x[n] + x[N - 1 - n];
        v[n] = x[n] - x[N - 1 - n];
      });
      T E[H], O[H];
      Dct2<H>::run(u, E);
      Dct4<H>::run(v, O);
      static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int k = decltype(i)::value;
        X[2 * k] = E[k];
        X[2 * k + 1] = O[k];
      });
    } else {
      constexpr int m = (N - 1) / 2;
      T u[m], v[m];
      static_for<m>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int n = decltype(i)::value;
        u[n] = x[n] + x[N - 1 - n];
        v[n] = x[n] - x[N - 1 - n];
      });
      const T mid = x[m];
      static_for<N>([&](auto ik) DCTS_LAMBDA_INLINE {
        constexpr int k = decltype(ik)::value;
        if constexpr (k == 0) {
          T acc = mid;
          static_for<m>([&](auto in) DCTS_LAMBDA_INLINE { acc += u[decltype(in)::value]; });
          X[0] = acc;
        } else if constexpr (k % 2 == 0) {
          // middle sample: cos(pi N k / (2N)) = cos(pi k / 2) = (-1)^(k/2)
          T acc = ((k / 2) % 2 == 0) ? mid : -mid;
          static_for<m>([&](auto in) DCTS_LAMBDA_INLINE {
            constexpr int n = decltype(in)::value;
            constexpr float c = float(cospi_frac((2 * n + 1) * k, 2 * N));
            acc += u[n] * c;
          });
          X[k] = acc;
        } else {
          T acc{};
          static_for<m>([&](auto in) DCTS_LAMBDA_INLINE {
            constexpr int n = decltype(in)::value;
            constexpr float c = float(cospi_frac((2 * n + 1) * k, 2 * N));
            if constexpr (n == 0)
              acc = v[0] * c;
            else
              acc += v[n] * c;
          });
          X[k] = acc;
        }
      });
    }
  }
};

// Orthonormal scale of output k of a length-N transform:
// sqrt(1/N) for k == 0, sqrt(2/N) otherwise. cx_sqrt: Newton iteration in double.
constexpr double cx_sqrt(double a) {
  if (a <= 0.0) return 0.0;
  double x = a > 1.0 ? a : 1.0;
  for (int i = 0; i < 200; ++i) x = 0.5 * (x + a / x);
  return x;
}
template <int N>
constexpr float ortho_scale(int k) {
  return float(k == 0 ? cx_sqrt(1.0 / double(N)) : cx_sqrt(2.0 / double(N)));
}
constexpr float kInvSqrt2 = float(0.70710678118654752440084436210484903928);

}  // namespace dcts
