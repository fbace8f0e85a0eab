// rect.hip - tiles whose two edges both have a codelet (codelet_sizes.h) but that the square codelet kernels of
// dct_kernels.hip do not take: NON-SQUARE maps (56 x 28, 14 x 20 ...: an --input_size that is not square) and maps
// whose rows are not dense (strideH > W: a spatial crop of a wider tensor).
//
// Replaces, for these shapes, the per-map loop of the reference hooks
// (utils/common.py:262-277: dct.dct_2d(output[i,j,:,:], norm='ortho') then sum(coeff^2); torch_dct.dct_2d and cv2.dct
// take any (H, W) - :267, :237), which until round 3 fell to the cosine-matrix kernel (k_energy_direct: O(N) flops per
// point, a few % of the HBM peak).
//
// ONE kernel for every (HP, WP) pair instead of 22 x 22 instantiations: the tile edges are kernel arguments (wave-
// uniform, in SGPRs), and each of the two passes is a switch over the codelet sizes - a scalar branch to the straight-
// line codelet of that length, so a wave executes exactly two of the 44 codelet bodies the kernel contains. Everything
// else is the codelet kernel's scheme (dct_kernels.hip, codelet_group): a wave takes G = floor(64 / min(HP, WP)) maps per
// iteration (each pass in as many sub-steps as its lane count needs); pass 1: lane = column, HP strided loads (consecutive lanes = consecutive addresses), column codelet,
// results into the wave's LDS slab; pass 2: lane = row, WP LDS reads, row codelet, squares, segmented shuffle sum over
// the HP rows of a map. No workgroup barrier (the slab is private to the wave). Registers are those of the largest
// codelet (64 points) whatever the shape.
//
// HBM traffic: the input once. Roofline: HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dctscore.h"
#include "codelet_sizes.h"
#include "dct_codelets.hpp"

#include "rect.h"

// every edge up to 64: the codelet template factorises any N = 2^s * m (m odd: direct, about m^2 / 2 FMAs with literal
// constants), so edges without a square kernel of their own (3, 5, 11, 13, 15, 22, 26, 34 ...) are served here too
#ifdef DCTS_DEV_FAST
#define DCTS_RECT_SIZES(X) DCTS_CODELET_SIZES(X)
#else
#define DCTS_RECT_SIZES(X) \
  X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64)
#endif

namespace {

using dctsi::RectGeom;
constexpr int kRectWaves = 4;

__device__ __forceinline__ const float* rect_map_base(const RectGeom& g, long long m) {
  if (g.contiguous) return g.x + ((long long)g.c_begin + m) * g.strideC;
  const long long n = m / g.c_count;
  const long long j = m - n * g.c_count;
  return g.x + n * g.strideN + (g.c_begin + j) * g.strideC;
}

// pass 1 for a column of HP samples: rows r >= pad come from p + (r - pad) * rs, the padded row 0 and the padded column are zeros
template <int HP>
__device__ __forceinline__ void rect_cols(const float* p, long long rs, int pad, bool zero_col, float* dst, int S, bool act) {
  float xr[HP], y[HP];
  const float* q = p - (long long)pad * rs;  // row r of the padded tile at q + r * rs (row 0 is never read when pad == 1)
  xr[0] = pad ? 0.f : q[0];
  dcts::static_for<HP - 1>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int r = decltype(i)::value + 1;
    xr[r] = q[r * rs];
  });
  if (pad) {  // wave-uniform
    dcts::static_for<HP>([&](auto i) DCTS_LAMBDA_INLINE { xr[decltype(i)::value] = zero_col ? 0.f : xr[decltype(i)::value]; });
  }
  dcts::Dct2<HP>::run(xr, y);
  y[0] *= dcts::kInvSqrt2;
  if (act) {
    dcts::static_for<HP>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int kk = decltype(i)::value;
      dst[kk * S] = y[kk];
    });
  }
}

// pass 2 for a row of WP intermediate values: the lane's energy, or (STORE) its WP coefficients scaled by sc
template <int WP, bool STORE>
__device__ __forceinline__ float rect_rows(const float* src, float* o, float sc, bool store_ok) {
  float z[WP], w[WP];
  dcts::static_for<WP>([&](auto i) DCTS_LAMBDA_INLINE { z[decltype(i)::value] = src[decltype(i)::value]; });
  dcts::Dct2<WP>::run(z, w);
  w[0] *= dcts::kInvSqrt2;
  float e = 0.f;
  if constexpr (STORE) {
    if (store_ok) dcts::static_for<WP>([&](auto i) DCTS_LAMBDA_INLINE { o[decltype(i)::value] = w[decltype(i)::value] * sc; });
  } else {
    dcts::static_for<WP>([&](auto i) DCTS_LAMBDA_INLINE { e = fmaf(w[decltype(i)::value], w[decltype(i)::value], e); });
  }
  return e;
}

// MAXE: the longest edge this instantiation serves (16 / 32 / 64): its register count is that of the MAXE-point codelet, so
// small tiles keep the occupancy their HBM-latency-bound loops need (64: 151 VGPRs, 3 waves per SIMD for every shape)
// ALL: every edge 1 ... 64 (the odd-length direct codelets need more registers: 209 instead of 151 in the 64 class), else the
// tabulated codelet sizes only
template <bool STORE, int MAXE, bool ONESTEP, bool ALL>
__global__ __launch_bounds__(64 * kRectWaves) void k_energy_rect(RectGeom g, float* __restrict__ out) {
  extern __shared__ float rect_slab[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HP = g.HP, WP = g.WP, G = g.G, S = g.S, pad = g.pad;
  float* my = rect_slab + wave * (G * g.map_lds);
  // pass-1 role: (map g1, column c); pass-2 role: (map g2, row k)
  const int g1 = lane / WP, c = lane - g1 * WP;
  const int g2 = lane / HP, k = lane - g2 * HP;
  const int G1 = ONESTEP ? G : g.G1, G2 = ONESTEP ? G : g.G2;  // maps a pass takes at once: floor(64 / WP) columns-as-lanes, floor(64 / HP) rows-as-lanes
  const long long ngroups = (g.nmaps + G - 1) / G;
  const long long nwaves = (long long)gridDim.x * kRectWaves;
  const float scale_e = g.scale_e, scale_c = g.scale_c;

  for (long long grp = (long long)blockIdx.x * kRectWaves + wave; grp < ngroups; grp += nwaves) {
    const long long mg = grp * G;  // first map of the group
    // ---- pass 1, G1 maps at a time: lanes without a map (or in the padded column) read a valid address; their values are
    // zeroed or unused
    for (int s = 0; ONESTEP ? s < 1 : (s < G && mg + s < g.nmaps); s += G1) {  // ONESTEP: G1 == G2 == G, no loop
      const int gi = s + g1;
      const bool a1 = g1 < G1 && gi < G;
      const long long m1 = mg + gi;
      const bool has = a1 && m1 < g.nmaps;
      const bool data_col = has && c >= pad;
      const float* p = rect_map_base(g, has ? m1 : g.nmaps - 1) + (data_col ? c - pad : 0);
      float* dst = my + (a1 ? gi : 0) * g.map_lds + (a1 ? c : 0);
#define DCTS_CASE(N) \
  case N: if constexpr (N <= MAXE) rect_cols<N>(p, g.strideH, pad, !data_col, dst, S, a1); break;
      if constexpr (ALL) {
        switch (HP) {
          DCTS_RECT_SIZES(DCTS_CASE)
          default: break;
        }
      } else {
        switch (HP) {
          DCTS_CODELET_SIZES(DCTS_CASE)
          default: break;
        }
      }
#undef DCTS_CASE
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- pass 2, G2 maps at a time
    for (int s = 0; ONESTEP ? s < 1 : (s < G && mg + s < g.nmaps); s += G2) {
      const int gi = s + g2;
      const bool a2 = g2 < G2 && gi < G;
      const long long m2 = mg + gi;
      const bool ok2 = a2 && m2 < g.nmaps;
      const float* src = my + (a2 ? gi : 0) * g.map_lds + (a2 ? k : 0) * S;
      float* o = STORE ? out + ((ok2 ? m2 : 0) * HP + k) * WP : nullptr;
      float e = 0.f;
#define DCTS_CASE(N) \
  case N: if constexpr (N <= MAXE) e = rect_rows<N, STORE>(src, o, scale_c, ok2); break;
      if constexpr (ALL) {
        switch (WP) {
          DCTS_RECT_SIZES(DCTS_CASE)
          default: break;
        }
      } else {
        switch (WP) {
          DCTS_CODELET_SIZES(DCTS_CASE)
          default: break;
        }
      }
#undef DCTS_CASE
      if constexpr (!STORE) {
        if (!a2) e = 0.f;
        // segmented sum over the HP lanes of a map (lane k == 0 ends with it): the codelet kernel's order
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          if (off < HP) {
            const float t = __shfl_down(e, off, 64);
            if (k + off < HP) e += t;
          }
        }
        if (ok2 && k == 0) out[m2] = e * scale_e;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

int rect_num_cus() {
  static int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}

bool rect_tab_1d(int n) {
#define DCTS_CASE(N) \
  if (n == N) return true;
  DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

bool rect_has_1d(int n) {
#define DCTS_CASE(N) \
  if (n == N) return true;
  DCTS_RECT_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

}  // namespace

namespace dctsi {

int has_rect(int HP, int WP) { return (rect_has_1d(HP) && rect_has_1d(WP)) ? 1 : 0; }

// fills in G, S, map_lds; launches
int dispatch_rect(const RectGeom& g_in, float* out, int store_coeff, hipStream_t st) {
  if (!has_rect(g_in.HP, g_in.WP)) return DCTS_E_UNSUPPORTED;
  RectGeom g = g_in;
  const int edge = g.HP > g.WP ? g.HP : g.WP;
  // Maps per pass step: floor(64 / WP) with columns as lanes, floor(64 / HP) with rows as lanes. A group is G maps, each pass
  // taking them in ceil(G / G1) resp. ceil(G / G2) steps: G is chosen to minimise the codelet runs per map (56 x 28: G = 2,
  // pass 1 once, pass 2 twice; 14 x 20: G = 12, four steps of three and three of four), within the slab a wave may have.
  g.G1 = 64 / g.WP;
  g.G2 = 64 / g.HP;
  g.S = g.WP | 1;             // odd row stride: the row-wise reads of pass 2 hit distinct banks within a map
  g.map_lds = g.HP * g.S + ((g.HP * g.S) % 2 == 0 ? 1 : 0);
  // slab per wave: what leaves the LDS room for as many waves as the registers of the size class allow (14 x 20 with a 14 KB
  // slab of twelve maps ran two workgroups per CU: 28 % of the HBM peak against 45 % with three maps)
  const int slab_cap = edge <= 16 ? 1536 : (edge <= 32 ? 2304 : 3400);
  const int gmax = slab_cap / g.map_lds > 0 ? slab_cap / g.map_lds : 1;
  int best = g.G1 < g.G2 ? g.G1 : g.G2;
  if (best > gmax) best = gmax;
  auto runs = [&](int G) { return (G + g.G1 - 1) / g.G1 + (G + g.G2 - 1) / g.G2; };
  if (!store_coeff)
    for (int G = best + 1; G <= gmax; ++G)
      if ((long long)runs(G) * best < (long long)runs(best) * G) best = G;  // strictly fewer runs per map
  g.G = best;
  if (g.G1 > g.G) g.G1 = g.G;
  if (g.G2 > g.G) g.G2 = g.G;
  const bool onestep = g.G1 == g.G && g.G2 == g.G;
  const bool tabulated = rect_tab_1d(g.HP) && rect_tab_1d(g.WP);
  g.scale_e = float(4.0 / (double(g.HP) * double(g.WP)));
  g.scale_c = float(2.0 / dcts::cx_sqrt(double(g.HP) * double(g.WP)));
  const size_t lds = (size_t)kRectWaves * g.G * g.map_lds * sizeof(float);
  const long long ngroups = (g.nmaps + g.G - 1) / g.G;
  long long blocks = (ngroups + kRectWaves - 1) / kRectWaves;
  const long long cap = (long long)rect_num_cus() * 64;  // a grid several times the residency (dct_kernels.hip, GRID_WAVES_PER_CU)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  static const hipError_t attr_rc = [] {  // four 64 x 65 slabs are 66.6 KB: above the 64 KB a kernel gets without asking
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_energy_rect<true, 64, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_energy_rect<false, 64, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_energy_rect<false, 64, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    return rc;
  }();
  if (attr_rc != hipSuccess) return (int)attr_rc;  // a HIP error (positive), as for a failed launch
  const dim3 grid((unsigned)blocks), wg(64 * kRectWaves);
  auto launch = [&](auto maxe) {
    constexpr int E = decltype(maxe)::value;
    if (store_coeff)
      hipLaunchKernelGGL((k_energy_rect<true, E, true, true>), grid, wg, lds, st, g, out);  // coefficients: parity path, one step per pass
    else if (onestep && tabulated)
      hipLaunchKernelGGL((k_energy_rect<false, E, true, false>), grid, wg, lds, st, g, out);
    else if (onestep)
      hipLaunchKernelGGL((k_energy_rect<false, E, true, true>), grid, wg, lds, st, g, out);
    else if (tabulated)
      hipLaunchKernelGGL((k_energy_rect<false, E, false, false>), grid, wg, lds, st, g, out);
    else
      hipLaunchKernelGGL((k_energy_rect<false, E, false, true>), grid, wg, lds, st, g, out);
  };
  if (edge <= 16)
    launch(std::integral_constant<int, 16>{});
  else if (edge <= 32)
    launch(std::integral_constant<int, 32>{});
  else
    launch(std::integral_constant<int, 64>{});
  return (int)hipGetLastError();
}

}  // namespace dctsi
