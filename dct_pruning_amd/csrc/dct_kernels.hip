// dct_kernels.hip — gfx950 kernels behind include/dctscore.h.
//
// Replaces the per-map Python loop of the reference hooks (utils/common.py:262-309):
//   c = [dct.dct_2d(output[i,j,:,:], norm='ortho') ...]; torch.sum(dct.mul(dct)).item()
// with one launch per hooked tensor: every (sample, channel) map gets its orthonormal
// 2-D DCT-II and the squared coefficients are reduced to one fp32 energy per map.
//
// Two kernel families:
//   k_energy_codelet  maps with both edges <= 64 that have a codelet (codelet_sizes.h).
//                     One wave owns floor(64/edge) maps. Pass 1: lane = column, the lane
//                     holds the whole column in VGPRs (coalesced dword loads straight
//                     from HBM, row r of a map is one contiguous segment across lanes) and
//                     runs a straight-line factorised DCT-II (dct_codelets.hpp). The
//                     tile is transposed through a per-wave LDS slab (odd row stride ->
//                     conflict-free both ways). Pass 2: lane = row, second codelet, the
//                     squares are summed in-lane and then across the map's lanes with a
//                     segmented wave shuffle reduction. HBM traffic = the algorithmic
//                     4*H*W + 4 bytes per map; LDS traffic = one write + one read per
//                     element.
//   k_energy_direct   any (H, W) <= DCTS_MAX_EDGE: separable cosine-matrix transform with
//                     the basis block staged in LDS; intermediate tile in a caller-provided
//                     workspace (L2-resident). O(H*W*(H+W)) flops per map: the correct
//                     fallback, compute-bound for large tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dctscore.h"
#include "codelet_sizes.h"
#include "dct_codelets.hpp"

namespace {

struct MapGeom {
  const float* x;
  long long nmaps;    // N * c_count
  long long strideN;  // elements
  long long strideC;  // elements
  long long strideH;  // elements (direct kernel only; codelet kernels require == W)
  int c_count;
  int c_begin;
  int H, W;           // data dims (before the odd front pad)
  int contiguous;     // 1: map m starts at x + c_begin*strideC + m*strideC (no div needed)
};

__device__ __forceinline__ const float* map_base(const MapGeom& g, long long m) {
  if (g.contiguous) return g.x + (long long)g.c_begin * g.strideC + m * g.strideC;
  const long long n = m / g.c_count;
  const long long j = m - n * g.c_count;
  return g.x + n * g.strideN + (g.c_begin + j) * g.strideC;
}

// ---------------------------------------------------------------------------------------
// codelet family
// ---------------------------------------------------------------------------------------
template <int HP, int WP>
struct CodeletCfg {
  static constexpr int EDGE = HP > WP ? HP : WP;
  static constexpr int G = 64 / EDGE;           // maps per wave per iteration
  // LDS row stride S and per-map stride: odd S is conflict-free inside one map; when several maps
  // share a wave the pair (S, MAP_LDS) below keeps the G*edge lanes of a half-wave on distinct
  // banks for both the column-wise store and the row-wise load (brute-force search over paddings,
  // SQ_LDS_BANK_CONFLICT was 18-47 % of LDS cycles before for these edges)
  static constexpr int S = (HP == WP && WP == 7) ? 8 : (HP == WP && (WP == 10 || WP == 14)) ? 17
                         : (HP == WP && WP == 20) ? 25 : (HP == WP && WP == 28) ? 33 : (WP | 1);
  static constexpr int MAP_LDS = (HP == WP && WP == 7) ? 71 : HP * S;  // floats per map in the transpose slab
  static constexpr int WAVE_LDS = G * MAP_LDS;  // floats per wave
  // waves per workgroup: keep a workgroup's slab <= 48 KiB so >= 3 workgroups fit a CU
  static constexpr int WAVES = (WAVE_LDS * 4 * 4 <= 49152) ? 4 : ((WAVE_LDS * 4 * 2 <= 49152) ? 2 : 1);
};

template <int HP, int WP, int PAD, bool STORE_COEFF>
__global__ __launch_bounds__((64 * CodeletCfg<HP, WP>::WAVES)) void k_energy_codelet(
    MapGeom g, float* __restrict__ out) {
  using Cfg = CodeletCfg<HP, WP>;
  constexpr int G = Cfg::G, S = Cfg::S, MAP_LDS = Cfg::MAP_LDS, WAVES = Cfg::WAVES;
  constexpr int W = WP - PAD;  // data row length == row stride (dense rows)
  __shared__ float slab[WAVES][Cfg::WAVE_LDS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* my = slab[wave];

  // pass-1 role: (map g1, column c); pass-2 role: (map g2, row k)
  const int g1 = lane / WP, c = lane - g1 * WP;
  const int g2 = lane / HP, k = lane - g2 * HP;
  const bool act1 = g1 < G, act2 = g2 < G;

  const long long ngroups = (g.nmaps + G - 1) / G;
  const long long wave_gid = (long long)blockIdx.x * WAVES + wave;
  const long long nwaves = (long long)gridDim.x * WAVES;

  for (long long grp = wave_gid; grp < ngroups; grp += nwaves) {
    // ---- pass 1: column DCT-II of length HP, lane = column -------------------------
    const long long m1 = grp * G + g1;
    float xr[HP];
    const bool ld = act1 && m1 < g.nmaps && c >= PAD;
    if (ld) {
      const float* p = map_base(g, m1) + (c - PAD);
      dcts::static_for<HP>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int r = decltype(i)::value;
        if constexpr (r < PAD)
          xr[r] = 0.f;
        else
          xr[r] = p[(r - PAD) * W];
      });
    } else {
      dcts::static_for<HP>([&](auto i) DCTS_LAMBDA_INLINE { xr[decltype(i)::value] = 0.f; });
    }
    float y[HP];
    dcts::Dct2<HP>::run(xr, y);
    y[0] *= dcts::kInvSqrt2;
    if (act1) {
      float* dst = my + g1 * MAP_LDS + c;
      dcts::static_for<HP>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int kk = decltype(i)::value;
        dst[kk * S] = y[kk];
      });
    }
    // the wave's own LDS traffic is in order; only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- pass 2: row DCT-II of length WP, lane = row --------------------------------
    float z[WP], w[WP];
    {
      const float* src = my + (act2 ? g2 : 0) * MAP_LDS + (act2 ? k : 0) * S;
      dcts::static_for<WP>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int cc = decltype(i)::value;
        z[cc] = src[cc];
      });
    }
    dcts::Dct2<WP>::run(z, w);
    w[0] *= dcts::kInvSqrt2;
    const long long m2 = grp * G + g2;
    if constexpr (STORE_COEFF) {
      // debug/parity path: out is [nmaps][HP][WP] orthonormal coefficients
      if (act2 && m2 < g.nmaps) {
        constexpr float sc = float(2.0 / dcts::cx_sqrt(double(HP) * double(WP)));
        float* o = out + (m2 * HP + k) * WP;
        dcts::static_for<WP>([&](auto i) DCTS_LAMBDA_INLINE {
          constexpr int l = decltype(i)::value;
          o[l] = w[l] * sc;
        });
      }
    } else {
      float e = 0.f;
      dcts::static_for<WP>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int l = decltype(i)::value;
        e = fmaf(w[l], w[l], e);
      });
      if (!act2) e = 0.f;
      // segmented reduction over the HP lanes of each map (lane k == 0 ends with the sum)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        if (off < HP) {
          const float t = __shfl_down(e, off, 64);
          if (k + off < HP) e += t;
        }
      }
      if (act2 && k == 0 && m2 < g.nmaps) {
        constexpr float sc = float(4.0 / (double(HP) * double(WP)));
        out[m2] = e * sc;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Prefetching variant for dense square even-edge tiles (the common case: every hooked tensor of
// the reference nets except 7x7 / 9x9). Same two passes and the same LDS slab, but the NEXT
// group of maps is streamed into the slab with direct-to-LDS loads (global_load_lds_dwordx4, no
// VGPRs) as soon as pass 2 has read the transposed tile out of it, so the HBM latency of group
// i+1 hides under the pass-2 codelet of group i instead of stalling the wave (s_waitcnt was
// 28 % of the wave's cycles in k_energy_codelet). Pass 1 then reads its column from the linear
// LDS image (lane = column: consecutive addresses, conflict-free).
template <int N>
__global__ __launch_bounds__((64 * CodeletCfg<N, N>::WAVES)) void k_energy_codelet_dma(
    MapGeom g, float* __restrict__ out) {
  using Cfg = CodeletCfg<N, N>;
  constexpr int G = Cfg::G, S = Cfg::S, MAP_LDS = Cfg::MAP_LDS, WAVES = Cfg::WAVES;
  constexpr int NN = N * N;
  constexpr int QPG = G * NN / 4;                // 16-byte quads per full group
  constexpr int DMA_IT = (QPG + 63) / 64;        // direct-to-LDS instructions per group
  constexpr int SLAB = ((Cfg::WAVE_LDS > DMA_IT * 256 ? Cfg::WAVE_LDS : DMA_IT * 256) + 3) / 4 * 4;
  static_assert((G * NN) % 4 == 0, "group must be a whole number of 16-byte quads");
  __shared__ __attribute__((aligned(16))) float slab[WAVES][SLAB];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* my = slab[wave];
  const int g1 = lane / N, c = lane - g1 * N;  // square tile: pass-1 and pass-2 roles coincide
  const bool act = g1 < G;

  const long long ngroups = (g.nmaps + G - 1) / G;
  const long long wave_gid = (long long)blockIdx.x * WAVES + wave;
  const long long nwaves = (long long)gridDim.x * WAVES;
  const float* x0 = g.x + (long long)g.c_begin * g.strideC;  // dense: map m starts at x0 + m*NN

  auto prefetch = [&](long long grp) DCTS_LAMBDA_INLINE {
    const long long m0 = grp * G;
    const long long left = g.nmaps - m0;
    const int nq = (int)((left < G ? left : G) * (NN / 4));
    const float* src = x0 + m0 * NN;
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) {
      const int q = it * 64 + lane;
      if (q < nq)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * q),
                                         (__attribute__((address_space(3))) void*)(my + it * 256), 16, 0, 0);
    }
  };

  if (wave_gid < ngroups) prefetch(wave_gid);
  for (long long grp = wave_gid; grp < ngroups; grp += nwaves) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group's tiles have landed in LDS
    const long long m1 = grp * G + g1;
    const bool valid = act && m1 < g.nmaps;
    // ---- pass 1: column DCT-II, lane = column, input from the linear LDS image --------
    float xr[N], y[N];
    {
      const float* src = my + (valid ? g1 * NN + c : 0);
      dcts::static_for<N>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int r = decltype(i)::value;
        xr[r] = src[r * N];
      });
      if (!valid) dcts::static_for<N>([&](auto i) DCTS_LAMBDA_INLINE { xr[decltype(i)::value] = 0.f; });
    }
    dcts::Dct2<N>::run(xr, y);
    y[0] *= dcts::kInvSqrt2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (act) {
      float* dst = my + g1 * MAP_LDS + c;
      dcts::static_for<N>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int kk = decltype(i)::value;
        dst[kk * S] = y[kk];
      });
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- pass 2: row DCT-II, lane = row -----------------------------------------------
    float z[N], w[N];
    {
      const float* src = my + (act ? g1 : 0) * MAP_LDS + (act ? c : 0) * S;
      dcts::static_for<N>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int cc = decltype(i)::value;
        z[cc] = src[cc];
      });
    }
    // the slab is free once these reads have returned: stream the next group into it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (grp + nwaves < ngroups) prefetch(grp + nwaves);
    dcts::Dct2<N>::run(z, w);
    w[0] *= dcts::kInvSqrt2;
    float e = 0.f;
    dcts::static_for<N>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int l = decltype(i)::value;
      e = fmaf(w[l], w[l], e);
    });
    if (!act) e = 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      if (off < N) {
        const float t = __shfl_down(e, off, 64);
        if (c + off < N) e += t;
      }
    }
    if (valid && c == 0) {
      constexpr float sc = float(4.0 / (double(N) * double(N)));
      out[m1] = e * sc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// split-4 family: tiles whose edge N = 4*M is too long for one lane's registers
// ---------------------------------------------------------------------------------------
// A length-N DCT-II is cut by two radix-2 levels into four length-M problems ("roles"),
// each run by one lane with an M-point codelet (dct_codelets.hpp recursion, top two levels
// unrolled across waves instead of inside a lane). With x3 = x[N-1-p], x1 = x[2M-1-p],
// x2 = x[2M+p], beta_p = (2p+1) pi / (8M):
//   role 0: in[p] = (x[p] + x3) + (x1 + x2)                  DCT-II_M  -> X[4k]
//   role 1: in[p] = (x[p] + x3) - (x1 + x2)                  DCT-IV_M  -> X[4k+2]
//   role 2: in[p] =  (x[p] - x3) cos(beta) + (x1 - x2) sin(beta)            DCT-II_M -> A[k]
//   role 3: in[p] = (-1)^p ((x1 - x2) cos(beta) - (x[p] - x3) sin(beta))    DCT-II_M -> B[k]
// and the odd outputs are X[2(2j)+1] = A[j] + B[M-j], X[2(2j-1)+1] = A[j] - B[M-j] (0<j<M),
// X[1] = A[0], X[2N-1...] = -B[0]. That last add/sub layer is a 45-degree rotation of each
// pair scaled by sqrt(2); the energy kernels fuse it into the reduction,
// (a+b)^2 + (a-b)^2 = 2a^2 + 2b^2, i.e. A[k], B[k] (k>0) are carried with weight sqrt(2).
// Every other butterfly, rotation and twiddle of the transform is computed.
//
// k_pass1d transforms the row axis of In[b][n][line] (lines contiguous), one wave per
// (64-line strip, role), the four role waves of a strip in one workgroup so they share the
// strip's cache lines. Non-final pass: the result goes to T[b][line][role*M + k] through a
// per-wave LDS transpose (coalesced stores), so the second launch of the same kernel
// transforms the other axis. Final pass: squares are reduced per wave into partial sums.
template <int M, int ROLE>
__device__ __forceinline__ void split4_inputs(const float* col, int rs, float (&in)[M]) {
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int p = decltype(i)::value;
    const float x0 = col[p * rs];
    const float x1 = col[(2 * M - 1 - p) * rs];
    const float x2 = col[(2 * M + p) * rs];
    const float x3 = col[(4 * M - 1 - p) * rs];
    if constexpr (ROLE == 0) {
      in[p] = (x0 + x3) + (x1 + x2);
    } else if constexpr (ROLE == 1) {
      in[p] = (x0 + x3) - (x1 + x2);
    } else {
      constexpr float c = float(dcts::cospi_frac(2 * p + 1, 8 * M));
      constexpr float sn = float(dcts::sinpi_frac(2 * p + 1, 8 * M));
      const float d1 = x0 - x3, d2 = x1 - x2;
      if constexpr (ROLE == 2) {
        in[p] = d1 * c + d2 * sn;
      } else {
        constexpr float sg = (p % 2 == 0) ? 1.f : -1.f;
        in[p] = d2 * (sg * c) - d1 * (sg * sn);
      }
    }
    // keep the scheduler from hoisting every LDS read to the top (register pressure -> spills);
    // ALU work may still move across (mask: 1 ALU | 2 VALU | 4 SALU)
    if constexpr (p % 8 == 7) __builtin_amdgcn_sched_barrier(7);
  });
}

template <int M, int ROLE>
__device__ __forceinline__ void split4_transform(const float (&in)[M], float (&out)[M]) {
  if constexpr (ROLE == 1)
    dcts::Dct4<M>::run(in, out);
  else
    dcts::Dct2<M>::run(in, out);
  if constexpr (ROLE == 0) out[0] *= dcts::kInvSqrt2;  // DC of the whole axis
  if constexpr (ROLE >= 2) {
    constexpr float r2 = float(1.41421356237309504880168872420969808);
    dcts::static_for<M - 1>([&](auto i) DCTS_LAMBDA_INLINE { out[decltype(i)::value + 1] *= r2; });
  }
}

template <int M>
struct SplitCfg {
  static constexpr int N = 4 * M;
  static constexpr int STRIPS = (N + 63) / 64;
  static constexpr int SW = (((N + STRIPS - 1) / STRIPS) + 3) / 4 * 4;  // lines per strip, multiple of 4
  static constexpr int SWP = SW | 1;                                    // odd LDS stride for the transpose
  static constexpr int IN_LDS = N * SW;                                 // floats: the staged input strip
  static constexpr int TR_LDS = 4 * M * SWP;                            // floats: 4 per-wave transpose slabs
  static constexpr int LDS_NONFINAL = IN_LDS > TR_LDS ? IN_LDS : TR_LDS;
};

// register budget (waves/SIMD) chosen so that no role spills
template <int M>
constexpr int split_waves_per_simd() { return M <= 32 ? 4 : (M <= 40 ? 3 : (M <= 56 ? 2 : 1)); }

template <int M, int ROLE, bool FINAL>
__device__ __forceinline__ void split4_wave(const float* lds_in, float* __restrict__ t_b, float* lds_tr,
                                            int strip, int lane, float* part) {
  using Cfg = SplitCfg<M>;
  constexpr int N = Cfg::N, SW = Cfg::SW, SWP = Cfg::SWP;
  const int line = strip * SW + lane;
  const bool act = lane < SW && line < N;
  float in[M], out[M];
  split4_inputs<M, ROLE>(lds_in + (act ? lane : 0), SW, in);
  split4_transform<M, ROLE>(in, out);
  if constexpr (FINAL) {
    float e = 0.f;
    dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int k = decltype(i)::value;
      e = fmaf(out[k], out[k], e);
    });
    if (!act) e = 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e += __shfl_down(e, off, 64);
    if (lane == 0) *part = e;
  } else {
    __syncthreads();  // every wave has consumed the staged strip: its LDS becomes the transpose slabs
    float* my = lds_tr + ROLE * (M * SWP);
    if (act) {
      dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int k = decltype(i)::value;
        my[k * SWP + lane] = out[k];
      });
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nl = (N - strip * SW) < SW ? (N - strip * SW) : SW;
#pragma unroll
    for (int k0 = 0; k0 < M; k0 += 64) {  // M may exceed the 64 lanes of a wave
      const int k = k0 + lane;
      if (k < M) {
        float* dst = t_b + (long long)(strip * SW) * N + ROLE * M + k;
#pragma unroll 8
        for (int j = 0; j < nl; ++j) dst[(long long)j * N] = my[k * SWP + j];
      }
    }
  }
}

// grid.x = nmaps_in_launch * STRIPS; block = 4 waves (wave = role).
// The strip In[b][0..N)[strip*SW .. +SW) is staged into LDS by direct-to-LDS loads
// (global_load_lds_dwordx4: no VGPRs, the whole 4*M*SW*4-byte strip in flight at once), then
// each role wave gathers its four mirrored rows per sample from LDS.
template <int M, bool FINAL>
__global__ __launch_bounds__(256, (split_waves_per_simd<M>())) void k_pass1d(
    const float* __restrict__ in, long long in_map_stride, float* __restrict__ t, float* __restrict__ partial) {
  using Cfg = SplitCfg<M>;
  constexpr int N = Cfg::N, SW = Cfg::SW;
  __shared__ __attribute__((aligned(16))) float lds[FINAL ? Cfg::IN_LDS : Cfg::LDS_NONFINAL];
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const long long b = blockIdx.x / Cfg::STRIPS;
  const int strip = blockIdx.x - (int)(b * Cfg::STRIPS);
  const float* in_b = in + b * in_map_stride;

  constexpr int NQUADS = N * SW / 4;
  constexpr int ITERS = (NQUADS + 255) / 256;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int qbase = it * 256 + role * 64;  // wave-uniform
    const int q = qbase + lane;
    const int e = 4 * q;
    const int row = e / SW, col = e - row * SW;
    if (q < NQUADS && strip * SW + col < N) {
      const float* g = in_b + (long long)row * N + strip * SW + col;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(lds + 4 * qbase), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float* t_b = FINAL ? nullptr : t + b * (long long)N * N;
  float* part = partial + (long long)blockIdx.x * 4 + role;
  switch (role) {
    case 0: split4_wave<M, 0, FINAL>(lds, t_b, lds, strip, lane, part); break;
    case 1: split4_wave<M, 1, FINAL>(lds, t_b, lds, strip, lane, part); break;
    case 2: split4_wave<M, 2, FINAL>(lds, t_b, lds, strip, lane, part); break;
    default: split4_wave<M, 3, FINAL>(lds, t_b, lds, strip, lane, part); break;
  }
}

// out[b] = scale * sum of the map's 4*STRIPS partials, fixed order
__global__ void k_split_reduce(const float* __restrict__ partial, int per_map, long long nmaps,
                               float scale, float* __restrict__ out) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nmaps) return;
  float s = 0.f;
  for (int i = 0; i < per_map; ++i) s += partial[b * per_map + i];
  out[b] = s * scale;
}

// ---------------------------------------------------------------------------------------
// direct family: basis tables + separable transform
// ---------------------------------------------------------------------------------------
// Bt[r*n + k] = s_k cos(pi (2r+1) k / (2n)), s_0 = sqrt(1/n), s_k = sqrt(2/n)
__global__ void k_basis(float* __restrict__ Bt, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const int r = idx / n, k = idx - r * n;
  const long long num = ((long long)(2 * r + 1) * k) % (4LL * n);
  const double cv = cospi(double(num) / double(2 * n));
  const double s = (k == 0) ? sqrt(1.0 / double(n)) : sqrt(2.0 / double(n));
  Bt[idx] = float(cv * s);
}

constexpr int kDirectThreads = 256;
constexpr int kKB = 8;  // output rows per basis block

template <bool STORE_COEFF>
__global__ __launch_bounds__(kDirectThreads) void k_energy_direct(
    MapGeom g, int pad, const float* __restrict__ CHt, const float* __restrict__ CWt,
    float* __restrict__ T, float* __restrict__ out) {
  const int HP = g.H + pad, WP = g.W + pad;
  __shared__ __attribute__((aligned(16))) float Bs[DCTS_MAX_EDGE][kKB];
  __shared__ float red[kDirectThreads / 64];
  const int tid = threadIdx.x;
  float* Tm = T + (size_t)blockIdx.x * HP * WP;

  for (long long m = blockIdx.x; m < g.nmaps; m += gridDim.x) {
    const float* xm = map_base(g, m);
    // ---- phase 1: Tm[k][c] = sum_r CH[k][r] x'[r][c] --------------------------------
    for (int k0 = 0; k0 < HP; k0 += kKB) {
      __syncthreads();
      for (int i = tid; i < HP * kKB; i += kDirectThreads) {
        const int r = i / kKB, kk = i - r * kKB;
        Bs[r][kk] = (k0 + kk < HP) ? CHt[r * HP + k0 + kk] : 0.f;
      }
      __syncthreads();
      for (int c = tid; c < WP; c += kDirectThreads) {
        float acc[kKB];
#pragma unroll
        for (int kk = 0; kk < kKB; ++kk) acc[kk] = 0.f;
        if (c >= pad) {
          const float* col = xm + (c - pad);
          for (int r = pad; r < HP; ++r) {
            const float xv = col[(long long)(r - pad) * g.strideH];
            const float4 b0 = *reinterpret_cast<const float4*>(&Bs[r][0]);
            const float4 b1 = *reinterpret_cast<const float4*>(&Bs[r][4]);
            acc[0] = fmaf(xv, b0.x, acc[0]);
            acc[1] = fmaf(xv, b0.y, acc[1]);
            acc[2] = fmaf(xv, b0.z, acc[2]);
            acc[3] = fmaf(xv, b0.w, acc[3]);
            acc[4] = fmaf(xv, b1.x, acc[4]);
            acc[5] = fmaf(xv, b1.y, acc[5]);
            acc[6] = fmaf(xv, b1.z, acc[6]);
            acc[7] = fmaf(xv, b1.w, acc[7]);
          }
        }
#pragma unroll
        for (int kk = 0; kk < kKB; ++kk)
          if (k0 + kk < HP) Tm[(k0 + kk) * WP + c] = acc[kk];
      }
    }
    // ---- phase 2: Y[k][l] = sum_c Tm[k][c] CW[l][c]; energy += Y^2 --------------------
    float e = 0.f;
    for (int k0 = 0; k0 < HP; k0 += kKB) {
      __syncthreads();  // also orders phase-1 global stores before these loads (same CU)
      for (int i = tid; i < WP * kKB; i += kDirectThreads) {
        const int cc = i / kKB, kk = i - cc * kKB;
        Bs[cc][kk] = (k0 + kk < HP) ? Tm[(k0 + kk) * WP + cc] : 0.f;
      }
      __syncthreads();
      for (int l = tid; l < WP; l += kDirectThreads) {
        float acc[kKB];
#pragma unroll
        for (int kk = 0; kk < kKB; ++kk) acc[kk] = 0.f;
        for (int cc = 0; cc < WP; ++cc) {
          const float wv = CWt[cc * WP + l];
          const float4 b0 = *reinterpret_cast<const float4*>(&Bs[cc][0]);
          const float4 b1 = *reinterpret_cast<const float4*>(&Bs[cc][4]);
          acc[0] = fmaf(wv, b0.x, acc[0]);
          acc[1] = fmaf(wv, b0.y, acc[1]);
          acc[2] = fmaf(wv, b0.z, acc[2]);
          acc[3] = fmaf(wv, b0.w, acc[3]);
          acc[4] = fmaf(wv, b1.x, acc[4]);
          acc[5] = fmaf(wv, b1.y, acc[5]);
          acc[6] = fmaf(wv, b1.z, acc[6]);
          acc[7] = fmaf(wv, b1.w, acc[7]);
        }
#pragma unroll
        for (int kk = 0; kk < kKB; ++kk) {
          if (k0 + kk < HP) {
            if constexpr (STORE_COEFF)
              out[(m * HP + k0 + kk) * WP + l] = acc[kk];
            else
              e = fmaf(acc[kk], acc[kk], e);
          }
        }
      }
    }
    if constexpr (!STORE_COEFF) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) e += __shfl_down(e, off, 64);
      __syncthreads();
      if ((tid & 63) == 0) red[tid >> 6] = e;
      __syncthreads();
      if (tid == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kDirectThreads / 64; ++i) s += red[i];
        out[m] = s;
      }
    }
  }
}

// Batch sum over n of E[n][j] for a 32-channel strip per block: 8 n-slices run in parallel
// (slice s takes n = s, s+8, ...), partials are combined in slice order -> a fixed,
// launch-independent summation order (bit-reproducible, no atomics).
constexpr int kSumCh = 32, kSumSl = 16;
__device__ __forceinline__ float strip_batch_sum(const float* __restrict__ e, long long N,
                                                 long long C, long long j, int slice,
                                                 float (*part)[kSumCh]) {
  float s = 0.f;
  if (j < C) {
    long long n = slice;
#pragma unroll 1
    for (; n + 3 * kSumSl < N; n += 4 * kSumSl) {
      const float a0 = e[n * C + j], a1 = e[(n + kSumSl) * C + j];
      const float a2 = e[(n + 2 * kSumSl) * C + j], a3 = e[(n + 3 * kSumSl) * C + j];
      s += a0;
      s += a1;
      s += a2;
      s += a3;
    }
    for (; n < N; n += kSumSl) s += e[n * C + j];
  }
  part[slice][threadIdx.x % kSumCh] = s;
  __syncthreads();
  float t = 0.f;
  if (slice == 0) {
#pragma unroll
    for (int i = 0; i < kSumSl; ++i) t += part[i][threadIdx.x % kSumCh];
  }
  return t;  // valid in slice 0
}

// out_c[j] = sum_n e[n*C + j]
__global__ __launch_bounds__(kSumCh * kSumSl) void k_batch_sum(const float* __restrict__ e, long long N,
                                                               long long C, float* __restrict__ out_c) {
  __shared__ float part[kSumSl][kSumCh];
  const int slice = threadIdx.x / kSumCh;
  const long long j = (long long)blockIdx.x * kSumCh + threadIdx.x % kSumCh;
  const float t = strip_batch_sum(e, N, C, j, slice, part);
  if (slice == 0 && j < C) out_c[j] = t;
}

// fr[j] <- (fr[j] * total + sum_n e[n*C + j]) / (total + N): the running-mean update of
// utils/common.py:274-277 fused with the batch sum of :273 (same three fp32 roundings)
__global__ __launch_bounds__(kSumCh * kSumSl) void k_running_mean(const float* __restrict__ e, long long N,
                                                                  long long C, float* __restrict__ fr,
                                                                  float total) {
  __shared__ float part[kSumSl][kSumCh];
  const int slice = threadIdx.x / kSumCh;
  const long long j = (long long)blockIdx.x * kSumCh + threadIdx.x % kSumCh;
  const float t = strip_batch_sum(e, N, C, j, slice, part);
  if (slice == 0 && j < C) {
    const float acc = __fadd_rn(__fmul_rn(fr[j], total), t);
    fr[j] = __fdiv_rn(acc, __fadd_rn(total, float(N)));
  }
}

// the same update for up to kMultiMax hook points in one launch (descriptors by value in the
// kernel arguments): blockIdx.y = hook point, blockIdx.x = 32-channel strip
constexpr int kMultiMax = 64;
struct UpdateBatch {
  dcts_update_desc d[kMultiMax];
};
__global__ __launch_bounds__(kSumCh * kSumSl) void k_running_mean_multi(UpdateBatch b) {
  __shared__ float part[kSumSl][kSumCh];
  const dcts_update_desc d = b.d[blockIdx.y];
  if ((long long)blockIdx.x * kSumCh >= d.C_count) return;  // whole block leaves together
  const int slice = threadIdx.x / kSumCh;
  const long long j = (long long)blockIdx.x * kSumCh + threadIdx.x % kSumCh;
  const float t = strip_batch_sum(d.energy_nc, d.N, d.C_count, j, slice, part);
  if (slice == 0 && j < d.C_count) {
    const float acc = __fadd_rn(__fmul_rn(d.feature_result[j], d.total_before), t);
    d.feature_result[j] = __fdiv_rn(acc, __fadd_rn(d.total_before, float(d.N)));
  }
}

// PMC calibration aid: streams n floats with the codelet kernels' access width (one dword per
// lane, consecutive lanes consecutive addresses) so FETCH_SIZE can be compared with a known
// byte count in this exact pattern (MI355X_MICROARCH.md, HBM section: widths other than
// 16 B/lane are uncalibrated).
__global__ __launch_bounds__(256) void k_calib_read(const float* __restrict__ x, long long n,
                                                    float* __restrict__ sink) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    s += x[i];
  if (s == 123456.789f) sink[0] = s;  // keeps the loads alive without a store in practice
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
constexpr int kDirectGridCap = 512;
constexpr int kNumCU = 256;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DirectWs {
  size_t off_ch, off_cw, off_t, total;
  int grid;
};
DirectWs direct_ws(long long nmaps, int HP, int WP) {
  DirectWs w;
  w.grid = (int)(nmaps < kDirectGridCap ? (nmaps > 0 ? nmaps : 1) : kDirectGridCap);
  w.off_ch = 0;
  w.off_cw = align_up(w.off_ch + (size_t)HP * HP * 4, 256);
  w.off_t = align_up(w.off_cw + (size_t)WP * WP * 4, 256);
  w.total = align_up(w.off_t + (size_t)w.grid * HP * WP * 4, 256);
  return w;
}

template <int HP, int WP, int PAD, bool STORE>
int launch_codelet(const MapGeom& g, float* out, hipStream_t st) {
  using Cfg = CodeletCfg<HP, WP>;
  const long long ngroups = (g.nmaps + Cfg::G - 1) / Cfg::G;
  long long blocks = (ngroups + Cfg::WAVES - 1) / Cfg::WAVES;
  const long long cap = (long long)kNumCU * 32 / Cfg::WAVES;  // one full residency of waves
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_energy_codelet<HP, WP, PAD, STORE>), dim3((unsigned)blocks),
                     dim3(64 * Cfg::WAVES), 0, st, g, out);
  return (int)hipGetLastError();
}

template <int N>
int launch_codelet_dma(const MapGeom& g, float* out, hipStream_t st) {
  using Cfg = CodeletCfg<N, N>;
  if constexpr ((Cfg::G * N * N) % 4 != 0) {
    return DCTS_E_UNSUPPORTED;
  } else {
    const long long ngroups = (g.nmaps + Cfg::G - 1) / Cfg::G;
    long long blocks = (ngroups + Cfg::WAVES - 1) / Cfg::WAVES;
    // persistent grid = exactly one residency: every wave then loops over many groups and the
    // prefetch of group i+1 overlaps the arithmetic of group i
    static const int per_cu = [] {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_energy_codelet_dma<N>, 64 * Cfg::WAVES, 0) != hipSuccess ||
          n < 1)
        n = 1;
      return n;
    }();
    const long long cap = (long long)kNumCU * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_energy_codelet_dma<N>), dim3((unsigned)blocks), dim3(64 * Cfg::WAVES), 0, st, g, out);
    return (int)hipGetLastError();
  }
}

// dense, 16-byte aligned, even-edge square tiles take the prefetching kernel
bool dma_ok(int HP, int WP, int pad, const MapGeom& g) {
  if (pad || HP != WP || (HP % 2) != 0) return false;
  if (!g.contiguous || g.strideC != (long long)HP * WP) return false;
  return (reinterpret_cast<uintptr_t>(g.x + (long long)g.c_begin * g.strideC) & 15) == 0;
}

int dispatch_codelet_dma(int N, const MapGeom& g, float* out, hipStream_t st) {
#define DCTS_CASE(N_) \
  case N_:            \
    return launch_codelet_dma<N_>(g, out, st);
  switch (N) {
    DCTS_CODELET_SIZES(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}

template <bool STORE>
int dispatch_codelet(int HP, int WP, int pad, const MapGeom& g, float* out, hipStream_t st) {
  if (HP != WP) return DCTS_E_UNSUPPORTED;
#define DCTS_CASE(N)                                                          \
  case N:                                                                     \
    if (pad) {                                                                \
      if constexpr ((N % 2) == 0 && N >= 2)                                   \
        return launch_codelet<N, N, 1, STORE>(g, out, st);                    \
      else                                                                    \
        return DCTS_E_UNSUPPORTED;                                            \
    }                                                                         \
    return launch_codelet<N, N, 0, STORE>(g, out, st);
  switch (HP) {
    DCTS_CODELET_SIZES(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}

// tile edges N = 4*M served by the split-4 family
#define DCTS_SPLIT_M(X) X(18) X(20) X(28) X(32) X(36) X(40) X(56) X(64) X(72) X(80)

bool has_split(long long HP, long long WP) {
  if (HP != WP) return false;
#define DCTS_CASE(M_) \
  if (HP == 4 * M_) return true;
  DCTS_SPLIT_M(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

// intermediate tile buffer per launch pair; DCTS_SPLIT_CHUNK_MB overrides (tuning knob)
size_t split_chunk_bytes() {
  static const size_t v = [] {
    const char* e = getenv("DCTS_SPLIT_CHUNK_MB");
    long mb = e ? atol(e) : 0;
    if (mb < 1 || mb > 4096) mb = 256;
    return (size_t)mb << 20;
  }();
  return v;
}

struct SplitWs {
  long long chunk_maps;
  size_t off_t, off_part, total;
};
SplitWs split_ws(long long nmaps, int N) {
  SplitWs w;
  const size_t map_bytes = (size_t)N * N * 4;
  long long chunk = (long long)(split_chunk_bytes() / map_bytes);
  if (chunk < 1) chunk = 1;
  if (chunk > nmaps) chunk = nmaps;
  w.chunk_maps = chunk;
  const int strips = (N + 63) / 64;
  w.off_t = 0;
  w.off_part = align_up((size_t)chunk * map_bytes, 256);
  w.total = align_up(w.off_part + (size_t)chunk * strips * 4 * 4, 256);
  return w;
}

template <int M>
int launch_split(const MapGeom& g, float* out, void* workspace, hipStream_t st) {
  using Cfg = SplitCfg<M>;
  constexpr int N = Cfg::N;
  const SplitWs ws = split_ws(g.nmaps, N);
  char* wsp = reinterpret_cast<char*>(workspace);
  float* T = reinterpret_cast<float*>(wsp + ws.off_t);
  float* part = reinterpret_cast<float*>(wsp + ws.off_part);
  const float* x0 = g.x + (long long)g.c_begin * g.strideC;
  const float scale = float(4.0 / (double(N) * double(N)));
  for (long long m0 = 0; m0 < g.nmaps; m0 += ws.chunk_maps) {
    const long long nb = (g.nmaps - m0) < ws.chunk_maps ? (g.nmaps - m0) : ws.chunk_maps;
    const unsigned grid = (unsigned)(nb * Cfg::STRIPS);
    hipLaunchKernelGGL((k_pass1d<M, false>), dim3(grid), dim3(256), 0, st, x0 + m0 * g.strideC, g.strideC, T,
                       part);
    hipLaunchKernelGGL((k_pass1d<M, true>), dim3(grid), dim3(256), 0, st, T, (long long)N * N, (float*)nullptr,
                       part);
    hipLaunchKernelGGL(k_split_reduce, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, part,
                       Cfg::STRIPS * 4, nb, scale, out + m0);
  }
  return (int)hipGetLastError();
}

int dispatch_split(int N, const MapGeom& g, float* out, void* workspace, hipStream_t st) {
#define DCTS_CASE(M_) \
  case 4 * M_:        \
    return launch_split<M_>(g, out, workspace, st);
  switch (N) {
    DCTS_SPLIT_M(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}

bool has_codelet(long long HP, long long WP) {
  if (HP != WP) return false;
#define DCTS_CASE(N) \
  if (HP == N) return true;
  DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

template <bool STORE>
int run(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W, int64_t strideN,
        int64_t strideC, int64_t strideH, int64_t strideW, int32_t c_begin, int32_t c_count,
        int32_t pad_front_if_odd, float* out, void* workspace, size_t workspace_bytes,
        void* stream, int32_t algo) {
  if (!x || !out) return DCTS_E_NULL;
  if (N <= 0 || C_total <= 0 || H <= 0 || W <= 0) return DCTS_E_SHAPE;
  if (c_count <= 0 || c_begin < 0 || (int64_t)c_begin + c_count > C_total) return DCTS_E_CHANNELS;
  if (strideW != 1 || strideH < W) return DCTS_E_STRIDE;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(out) & 3)) return DCTS_E_ALIGN;
  const int pad = (pad_front_if_odd && (H % 2 != 0)) ? 1 : 0;
  const int64_t HP = H + pad, WP = W + pad;
  if (HP > DCTS_MAX_EDGE || WP > DCTS_MAX_EDGE) return DCTS_E_SHAPE;
  if (N * (int64_t)c_count >= (1LL << 40)) return DCTS_E_SHAPE;

  MapGeom g;
  g.x = x;
  g.nmaps = N * (int64_t)c_count;
  g.strideN = strideN;
  g.strideC = strideC;
  g.strideH = strideH;
  g.c_count = c_count;
  g.c_begin = c_begin;
  g.H = (int)H;
  g.W = (int)W;
  g.contiguous = (N == 1 || strideN == (int64_t)c_count * strideC) ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  const bool codelet_ok = has_codelet(HP, WP) && strideH == W;
  if ((algo == DCTS_ALGO_CODELET || algo == DCTS_ALGO_PREFETCH) && !codelet_ok) return DCTS_E_UNSUPPORTED;
  if (algo != DCTS_ALGO_AUTO && algo != DCTS_ALGO_DIRECT && algo != DCTS_ALGO_CODELET &&
      algo != DCTS_ALGO_SPLIT && algo != DCTS_ALGO_PREFETCH)
    return DCTS_E_UNSUPPORTED;
  if (codelet_ok && algo != DCTS_ALGO_DIRECT) {
    if constexpr (!STORE) {
      // the prefetching variant is opt-in: on MI355X it measured equal to the register-load
      // kernel in steady state (both at the practical HBM rate) and ~2 % slower on the bench
      if (algo == DCTS_ALGO_PREFETCH) {
        if (!dma_ok((int)HP, (int)WP, pad, g)) return DCTS_E_UNSUPPORTED;
        return dispatch_codelet_dma((int)HP, g, out, st);
      }
    } else if (algo == DCTS_ALGO_PREFETCH) {
      return DCTS_E_UNSUPPORTED;
    }
    return dispatch_codelet<STORE>((int)HP, (int)WP, pad, g, out, st);
  }
  if constexpr (!STORE) {
    const bool split_ok = has_split(HP, WP) && pad == 0 && strideH == W && g.contiguous &&
                          strideC == H * W;
    if (algo == DCTS_ALGO_SPLIT && !split_ok) return DCTS_E_UNSUPPORTED;
    if (split_ok && algo != DCTS_ALGO_DIRECT) {
      const SplitWs sws = split_ws(g.nmaps, (int)HP);
      if (!workspace || workspace_bytes < sws.total) return DCTS_E_WORKSPACE;
      return dispatch_split((int)HP, g, out, workspace, st);
    }
  } else {
    if (algo == DCTS_ALGO_SPLIT) return DCTS_E_UNSUPPORTED;
  }

  const DirectWs ws = direct_ws(g.nmaps, (int)HP, (int)WP);
  if (!workspace) return ws.total ? DCTS_E_WORKSPACE : DCTS_E_NULL;
  if (workspace_bytes < ws.total) return DCTS_E_WORKSPACE;
  char* wsp = reinterpret_cast<char*>(workspace);
  float* CHt = reinterpret_cast<float*>(wsp + ws.off_ch);
  float* CWt = reinterpret_cast<float*>(wsp + ws.off_cw);
  float* T = reinterpret_cast<float*>(wsp + ws.off_t);
  hipLaunchKernelGGL(k_basis, dim3((unsigned)((HP * HP + 255) / 256)), dim3(256), 0, st, CHt, (int)HP);
  hipLaunchKernelGGL(k_basis, dim3((unsigned)((WP * WP + 255) / 256)), dim3(256), 0, st, CWt, (int)WP);
  hipLaunchKernelGGL((k_energy_direct<STORE>), dim3((unsigned)ws.grid), dim3(kDirectThreads), 0, st,
                     g, pad, CHt, CWt, T, out);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int dcts_version(void) { return DCTS_ABI_VERSION; }

const char* dcts_strerror(int code) {
  switch (code) {
    case DCTS_OK: return "ok";
    case DCTS_E_NULL: return "required pointer is NULL";
    case DCTS_E_SHAPE: return "bad shape (N, C, H, W must be > 0 and tile edges <= 512)";
    case DCTS_E_CHANNELS: return "channel slice outside [0, C_total)";
    case DCTS_E_STRIDE: return "rows must be dense: strideW == 1 and strideH >= W";
    case DCTS_E_WORKSPACE: return "workspace missing or smaller than dcts_workspace_bytes()";
    case DCTS_E_UNSUPPORTED: return "no kernel of the requested family for this shape";
    case DCTS_E_ALIGN: return "pointer not 4-byte aligned";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown dctscore error";
}

size_t dcts_workspace_bytes(int64_t N, int64_t C_count, int64_t H, int64_t W) {
  if (N <= 0 || C_count <= 0 || H <= 0 || W <= 0) return 0;
  // worst case: odd front pad taken, direct kernel used
  const int64_t HP = H + 1, WP = W + 1;
  size_t need = direct_ws(N * C_count, (int)HP, (int)WP).total;
  if (has_split(H, W)) {
    const size_t s = split_ws(N * C_count, (int)H).total;
    if (s > need) need = s;
  }
  return need;
}

int dcts_has_codelet(int64_t H, int64_t W) { return has_codelet(H, W) ? 1 : 0; }

int dcts_energy_f32_ex(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                       int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                       int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                       float* out_nc, void* workspace, size_t workspace_bytes, void* stream,
                       int32_t algo) {
  return run<false>(x, N, C_total, H, W, strideN, strideC, strideH, strideW, c_begin, c_count,
                    pad_front_if_odd, out_nc, workspace, workspace_bytes, stream, algo);
}

int dcts_energy_f32(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                    int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                    int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd, float* out_nc,
                    void* workspace, size_t workspace_bytes, void* stream) {
  return run<false>(x, N, C_total, H, W, strideN, strideC, strideH, strideW, c_begin, c_count,
                    pad_front_if_odd, out_nc, workspace, workspace_bytes, stream, DCTS_ALGO_AUTO);
}

int dcts_dct2d_f32_ex(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                      int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                      int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                      float* out_coeff, void* workspace, size_t workspace_bytes, void* stream,
                      int32_t algo) {
  return run<true>(x, N, C_total, H, W, strideN, strideC, strideH, strideW, c_begin, c_count,
                   pad_front_if_odd, out_coeff, workspace, workspace_bytes, stream, algo);
}

int dcts_dct2d_f32(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                   int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                   int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd, float* out_coeff,
                   void* workspace, size_t workspace_bytes, void* stream) {
  return run<true>(x, N, C_total, H, W, strideN, strideC, strideH, strideW, c_begin, c_count,
                   pad_front_if_odd, out_coeff, workspace, workspace_bytes, stream, DCTS_ALGO_AUTO);
}

int dcts_batch_sum_f32(const float* energy_nc, int64_t N, int64_t C_count, float* out_c,
                       void* stream) {
  if (!energy_nc || !out_c) return DCTS_E_NULL;
  if (N <= 0 || C_count <= 0) return DCTS_E_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_batch_sum, dim3((unsigned)((C_count + kSumCh - 1) / kSumCh)), dim3(kSumCh * kSumSl), 0, st,
                     energy_nc, (long long)N, (long long)C_count, out_c);
  return (int)hipGetLastError();
}

int dcts_running_mean_update_f32(const float* energy_nc, int64_t N, int64_t C_count,
                                 float* feature_result, float total_before, void* stream) {
  if (!energy_nc || !feature_result) return DCTS_E_NULL;
  if (N <= 0 || C_count <= 0) return DCTS_E_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_running_mean, dim3((unsigned)((C_count + kSumCh - 1) / kSumCh)), dim3(kSumCh * kSumSl), 0, st,
                     energy_nc, (long long)N, (long long)C_count, feature_result, total_before);
  return (int)hipGetLastError();
}

int dcts_running_mean_update_multi_f32(const dcts_update_desc* descs, int32_t count, void* stream) {
  if (!descs) return DCTS_E_NULL;
  if (count <= 0) return DCTS_E_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int32_t i0 = 0; i0 < count; i0 += kMultiMax) {
    const int n = (count - i0) < kMultiMax ? (count - i0) : kMultiMax;
    UpdateBatch b;
    int64_t cmax = 0;
    for (int i = 0; i < n; ++i) {
      b.d[i] = descs[i0 + i];
      if (!b.d[i].energy_nc || !b.d[i].feature_result) return DCTS_E_NULL;
      if (b.d[i].N <= 0 || b.d[i].C_count <= 0) return DCTS_E_SHAPE;
      if (b.d[i].C_count > cmax) cmax = b.d[i].C_count;
    }
    for (int i = n; i < kMultiMax; ++i) b.d[i] = b.d[0];
    hipLaunchKernelGGL(k_running_mean_multi, dim3((unsigned)((cmax + kSumCh - 1) / kSumCh), (unsigned)n),
                       dim3(kSumCh * kSumSl), 0, st, b);
  }
  return (int)hipGetLastError();
}

int dcts_debug_stream_read_f32(const float* x, int64_t n, float* sink, void* stream) {
  if (!x || !sink) return DCTS_E_NULL;
  if (n <= 0) return DCTS_E_SHAPE;
  hipLaunchKernelGGL(k_calib_read, dim3(256 * 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     (long long)n, sink);
  return (int)hipGetLastError();
}

}  // extern "C"
