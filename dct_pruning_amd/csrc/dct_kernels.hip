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

#include <mutex>

#include "../../include/dctscore.h"
#include "codelet_sizes.h"
#include "dct_codelets.hpp"
#include "split_roles.hpp"
#include "rect.h"

// The file is compiled seven times in parallel (Makefile: -DDCTS_TU=1..7), each translation unit
// instantiating one kernel family; DCTS_TU=0 (default) builds everything in one unit. Only the
// dispatchers that instantiate kernels cross units: they are declared here with the geometry
// structs passed as opaque pointers (the structs live in the anonymous namespace of every unit).
#ifndef DCTS_TU
#define DCTS_TU 0
#endif
#ifndef DCTS_ADDTID
// ds_write_addtid_b32 for the codelet kernel's transposing stores where a wave holds one map (edges 36 ... 64): same box,
// 200 MB launches, % of the HBM peak: 56: 60.7 -> 61.5-62.3, 48: 60.8 -> 63.0, 36: 52.9 -> 54.0, 64: 52.9 -> 54.2; the
// 4.3 GB in-step launch is unchanged within noise (the kernel is VALU-bound there). Bit-identical results.
#define DCTS_ADDTID 1
#endif
#ifndef DCTS_F2_EXP
#define DCTS_F2_EXP 0  // timing experiments on k_split_fused2 (wrong results): 1 no staging loads, 2 no pass-1 butterflies, 3 no pass-2 butterflies, 4 no codelet arithmetic
#endif
#ifndef DCTS_FUSED2_AUTO
#define DCTS_FUSED2_AUTO 1  // AUTO uses the two-roles-per-wave fused kernel where it exists (288: 31 % vs 18 %, 320: 31 % vs 17 % of the HBM peak)
#endif
#define DCTS_PART(n) (DCTS_TU == 0 || DCTS_TU == (n))  // 1 codelet+lane, 2 two-launch split, 3 fused, 4 pipelined, 5 rest + C ABI, 6 fused with two roles per wave, 7 two-launch split: the 8 * M edges of round 3
namespace dctsi {
int dispatch_codelet(int store, int HP, int WP, int pad, const void* geom, float* out, hipStream_t st);
int dispatch_codelet_dma(int N, const void* geom, float* out, hipStream_t st);
int dispatch_codelet_multi(int HP, int pad, const void* multi_geom, hipStream_t st);
int dispatch_lane(int n, const void* multi_geom, hipStream_t st);
int dispatch_codelet_mixed(const void* mixed_geom, hipStream_t st);
int dispatch_split(int N, const void* geom, float* out, void* workspace, hipStream_t st);
int dispatch_split_more(int N, const void* geom, float* out, void* workspace, hipStream_t st);  // the 8 * M entries added in round 3 (DCTS_TU=7)
int dispatch_fused(int N, const void* tile_batch, hipStream_t st);
int dispatch_fused2(int N, const void* tile_batch, hipStream_t st);
int dispatch_pipe(int N, const void* tile_batch, hipStream_t st);
int dispatch_tile2d(int N, const void* tile_batch, hipStream_t st);  // tile2d.hip
// coefficient output through the large-tile kernels (debug / parity): leaf outputs into `scratch`
// (scratch_maps tiles), then k_assemble
int dispatch_fused_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps, hipStream_t st);
int dispatch_fused2_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps, hipStream_t st);
int dispatch_tile2d_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps, hipStream_t st);
// tile2g.hip: mid-size edges as a 2-D radix split with several maps per round
int has_tile2g(int N);
int has_tile2g_pad(int N);
int dispatch_tile2g(int N, const void* tile_batch, hipStream_t st);
int dispatch_tile2g_pad(int N, const void* tile_batch, hipStream_t st);  // tiles with the odd front pad (N = H + 1)
int dispatch_tile2g_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps, hipStream_t st);
}  // namespace dctsi

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
  // Waves launched per CU at most (the grid-stride loop takes the rest). NOT one residency (12 waves per CU
  // at 56 x 56): a grid several times the residency, whose workgroups the dispatcher hands out as CUs free up,
  // is faster than persistent waves in lock step - sweep of this cap on the bench's own launches, waves per
  // CU -> % of the HBM peak: 56 x 56 (344 k maps) 32: 67.2, 128...512: 69.6, 2048: 66.3; 28 x 28 (819 k) 32:
  // 68.4, 256: 74.7, 512: 76.4, 2048: 72.5; 14 x 14 (2.4 M) 32: 69.4, 512: 74.7, 2048: 74.9; 200 MB launches
  // of 8 / 14 / 28 / 32: 62 -> 71, 62 -> 71, 66 -> 72.5, 68 -> 73.5; whole ResNet-50 step 3259 -> 3561 Mmaps/s.
  // (4 x 4 and 2 x 2 groups are 1 KB and 512 B: there the wider grid costs more in wave launches than it
  // gains - 70 -> 61 % and 50 -> 46 % - and the cap stays at 32.)
  static constexpr int GRID_WAVES_PER_CU = (HP * WP >= 48 * 48) ? 256 : ((HP * WP >= 8 * 8) ? 512 : 32);
};

// four lane-consecutive LDS stores at byte offsets O0..O3 from `base` (an LDS byte address below 64 KiB: M0[15:0])
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void lds_write_addtid4(unsigned base, float a, float b, float c, float d) {
  static_assert(O3 < 65536 && O0 >= 0, "16-bit offset field");
  asm volatile(
      "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
      "ds_write_addtid_b32 %1 offset:%5\n\tds_write_addtid_b32 %2 offset:%6\n\t"
      "ds_write_addtid_b32 %3 offset:%7\n\tds_write_addtid_b32 %4 offset:%8"
      :
      : "s"(base), "v"(a), "v"(b), "v"(c), "v"(d), "n"(O0), "n"(O1), "n"(O2), "n"(O3)
      : "memory", "m0");
}

// one group of G maps: both passes, the LDS transpose and the reduction (see the header comment)
template <int HP, int WP, int PAD, bool STORE_COEFF>
__device__ __forceinline__ void codelet_group(const MapGeom& g, float* __restrict__ out, long long grp,
                                              float* my, int g1, int c, int g2, int k, bool act1, bool act2) {
  using Cfg = CodeletCfg<HP, WP>;
  constexpr int G = Cfg::G, S = Cfg::S, MAP_LDS = Cfg::MAP_LDS;
  constexpr int W = WP - PAD;  // data row length == row stride (dense rows)
  // ---- pass 1: column DCT-II of length HP, lane = column -------------------------
  const long long m1 = grp * G + g1;
  float xr[HP];
  if constexpr (PAD == 0) {
    // No branch and no zero fill: lanes without a map (beyond G*WP, or past the last map of a ragged
    // group) load some valid map instead and their results are never stored. The kernel is
    // VALU-issue-bound; the HP v_mov 0 per iteration of the zero fill were 4-6 % of its instructions.
    const bool has = act1 && m1 < g.nmaps;
    const float* p = map_base(g, has ? m1 : g.nmaps - 1) + (has ? c : 0);
    dcts::static_for<HP>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int r = decltype(i)::value;
      xr[r] = p[r * W];
    });
  } else {
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
  }
  float y[HP];
  dcts::Dct2<HP>::run(xr, y);
  y[0] *= dcts::kInvSqrt2;
#if DCTS_ADDTID
  if constexpr (G == 1 && HP % 4 == 0) {
    // one map per wave: lane = column, so row kk of the transposed slab is lane-consecutive words - ds_write_addtid_b32
    // (address = M0 + offset + 4 * lane: no address VGPR, 2 cycles per wave instruction instead of 4)
    if (act1) {
      const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_ptr)my);  // the wave's slab: uniform
      dcts::static_for<HP / 4>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int k0 = 4 * decltype(i)::value;
        lds_write_addtid4<k0 * S * 4, (k0 + 1) * S * 4, (k0 + 2) * S * 4, (k0 + 3) * S * 4>(base, y[k0], y[k0 + 1], y[k0 + 2], y[k0 + 3]);
      });
    }
  } else
#endif
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

template <int HP, int WP, int PAD, bool STORE_COEFF>
__global__ __launch_bounds__((64 * CodeletCfg<HP, WP>::WAVES)) void k_energy_codelet(
    MapGeom g, float* __restrict__ out) {
  using Cfg = CodeletCfg<HP, WP>;
  constexpr int G = Cfg::G, WAVES = Cfg::WAVES;
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

  for (long long grp = wave_gid; grp < ngroups; grp += nwaves)
    codelet_group<HP, WP, PAD, STORE_COEFF>(g, out, grp, my, g1, c, g2, k, act1, act2);
}

// Several hooked tensors of the same tile shape in ONE launch (single-sweep harness, bench): the
// groups of all tensors form one index space; a wave walks it with a grid stride and tracks which
// tensor its current group belongs to. CIFAR-sized layers are 5-20 us kernels when launched one
// by one - the launch ramp and tail cost more than the work.
constexpr int kMultiItems = 32;
struct MultiItem {
  MapGeom g;
  float* out;
  long long group_begin;  // first global group index of this tensor
};
struct MultiGeom {
  MultiItem it[kMultiItems];
  long long total_groups;
  int count;
};


template <int HP, int WP, int PAD>
__global__ __launch_bounds__((64 * CodeletCfg<HP, WP>::WAVES)) void k_energy_codelet_multi(MultiGeom mg) {
  using Cfg = CodeletCfg<HP, WP>;
  constexpr int G = Cfg::G, WAVES = Cfg::WAVES;
  __shared__ float slab[WAVES][Cfg::WAVE_LDS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* my = slab[wave];
  const int g1 = lane / WP, c = lane - g1 * WP;
  const int g2 = lane / HP, k = lane - g2 * HP;
  const bool act1 = g1 < G, act2 = g2 < G;

  const long long wave_gid = (long long)blockIdx.x * WAVES + wave;
  const long long nwaves = (long long)gridDim.x * WAVES;
  int t = 0;
  for (long long grp = wave_gid; grp < mg.total_groups; grp += nwaves) {
    while (t + 1 < mg.count && grp >= mg.it[t + 1].group_begin) ++t;  // wave-uniform, monotone
    t = __builtin_amdgcn_readfirstlane(t);
    const MultiItem& item = mg.it[t];
    codelet_group<HP, WP, PAD, false>(item.g, item.out, grp - item.group_begin, my, g1, c, g2, k, act1, act2);
  }
}

// Tensors of DIFFERENT small tile shapes in one launch (edges 2, 4, 8, 16, 32: every hooked tensor of
// the CIFAR nets). VGG-16-bn at batch 256 is 187 MB of activations in five tile shapes: five launches
// plus the running-mean update were 58 us, launch ramps and tails costing as much as the work. The
// groups of all tensors form one index space (a group = floor(64 / edge) maps of ITS tensor's shape);
// a wave switches on the shape of the tensor its group belongs to and runs that shape's codelet
// group: the same code as k_energy_codelet, results bit for bit those of one call per tensor.
constexpr int kMixedItems = 48;
struct MixedGeom {
  MultiItem it[kMixedItems];
  long long total_groups;
  int count;
};
#define DCTS_MIXED_SIZES(X) X(2) X(4) X(8) X(16) X(32)
constexpr bool mixed_has(int e) {
#define DCTS_CASE(N) \
  if (e == N) return true;
  DCTS_MIXED_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return false;
}
constexpr int mixed_slab_floats() {
  int m = 0;
#define DCTS_CASE(N) \
  if (CodeletCfg<N, N>::WAVE_LDS > m) m = CodeletCfg<N, N>::WAVE_LDS;
  DCTS_MIXED_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return m;
}
constexpr int kMixedWaves = 4;

template <int E>
__device__ __forceinline__ void mixed_group(const MultiItem& item, long long grp, float* my, int lane) {
  using Cfg = CodeletCfg<E, E>;
  const int g1 = lane / E, c = lane - g1 * E;  // square tile: pass-1 and pass-2 roles coincide
  const bool act = g1 < Cfg::G;
  codelet_group<E, E, 0, false>(item.g, item.out, grp, my, g1, c, g1, c, act, act);
}

__global__ __launch_bounds__((64 * kMixedWaves)) void k_energy_codelet_mixed(MixedGeom mg) {
  __shared__ float slab[kMixedWaves][mixed_slab_floats()];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* my = slab[wave];
  const long long wave_gid = (long long)blockIdx.x * kMixedWaves + wave;
  const long long nwaves = (long long)gridDim.x * kMixedWaves;
  int t = 0;
  for (long long grp = wave_gid; grp < mg.total_groups; grp += nwaves) {
    while (t + 1 < mg.count && grp >= mg.it[t + 1].group_begin) ++t;  // wave-uniform, monotone
    t = __builtin_amdgcn_readfirstlane(t);
    const MultiItem& item = mg.it[t];
    const long long local = grp - item.group_begin;
    switch (__builtin_amdgcn_readfirstlane(item.g.H)) {
#define DCTS_CASE(N)                          \
  case N:                                     \
    mixed_group<N>(item, local, my, lane);    \
    break;
      DCTS_MIXED_SIZES(DCTS_CASE)
#undef DCTS_CASE
      default:
        break;
    }
  }
}

// ---------------------------------------------------------------------------------------
// lane-per-map kernels for tiny odd tiles (7x7: the last stage of ResNet-50; 9x9: U2-Net-p)
// ---------------------------------------------------------------------------------------
// The codelet kernel above gives a 7x7 map to 7 lanes: 243 VALU instructions per 9 maps, of which
// 70 are arithmetic (the rest: addressing, the transpose, a segmented reduction over 7 lanes), and
// every load instruction touches nine 28-byte segments: VALU-issue-bound at 45-50 % of the HBM peak.
// Here a lane owns a whole map: a wave streams 64 consecutive maps (64*N*N floats, contiguous in
// memory) into its private LDS slab with direct-to-LDS loads, every lane reads its N*N values
// (stride N*N floats between lanes: odd, conflict-free), and both DCT passes run in registers with
// no transpose and no cross-lane reduction: ~9 instructions per map instead of 27. The slab is free
// as soon as the lanes have read it, so the next group's loads are in flight during the arithmetic.
// Channel-sliced (non-dense) tensors take per-lane loads into the same arithmetic: same results.
template <int N>
struct LaneCfg {
  static constexpr int NN = N * N;
  static constexpr int G = 64;                         // maps per wave per iteration
  static constexpr int SLAB = (G * NN + 3) / 4 * 4;    // floats
  static constexpr int ITERS = (G * NN / 4 + 63) / 64;  // direct-to-LDS instructions per group
  static constexpr int WAVES = 2;
  static_assert(NN % 2 == 1, "lane stride must be odd (bank conflicts) - even tiles use the codelet kernel");
};

struct LaneGroup {  // wave-uniform description of one group of <= 64 maps
  const float* src;  // dense: first float of the group
  float* out;        // &out[m0]
  long long m0;
  int nm;            // maps in the group
  int dense;
  int item;
};

template <int N>
__device__ __forceinline__ void lane_stage(const LaneGroup& gr, lds_ptr my, int lane) {
  using Cfg = LaneCfg<N>;
  if (!gr.dense) return;
  const int nfl = gr.nm * Cfg::NN, nq = nfl >> 2, rem = nfl & 3;
  // wave-uniform operands, made so explicitly (they derive from the wave index)
  const unsigned long long sa = reinterpret_cast<unsigned long long>(gr.src);
  const unsigned long long src = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)sa);
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)my);
#pragma unroll
  for (int it = 0; it < Cfg::ITERS; ++it) {
    const int q = it * 64 + lane;
    if (q < nq) {
      const unsigned dst = base + it * 1024;
      const unsigned off = (unsigned)q * 16u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :
                   : "s"(dst), "v"(off), "s"(src)
                   : "memory", "m0");
    }
  }
  if (lane < rem) my[4 * nq + lane] = gr.src[4 * nq + lane];  // last 1-3 floats of a ragged tail
}

template <int N>
__device__ __forceinline__ void lane_compute(const MapGeom& g, const LaneGroup& gr, lds_ptr my, int lane,
                                             float (&v)[N * N]) {
  constexpr int NN = N * N;
  const bool act = lane < gr.nm;
  if (gr.dense) {
    lds_cptr p = my + (act ? lane : 0) * NN;
    dcts::static_for<NN>([&](auto i) DCTS_LAMBDA_INLINE { v[decltype(i)::value] = p[decltype(i)::value]; });
  } else {
    const float* p = map_base(g, gr.m0 + (act ? lane : 0));
    dcts::static_for<NN>([&](auto i) DCTS_LAMBDA_INLINE { v[decltype(i)::value] = p[decltype(i)::value]; });
  }
}

template <int N>
__device__ __forceinline__ float lane_energy(float (&v)[N * N]) {
  dcts::static_for<N>([&](auto ic) DCTS_LAMBDA_INLINE {  // columns, in place
    constexpr int c = decltype(ic)::value;
    float in[N], o[N];
    dcts::static_for<N>([&](auto ir) DCTS_LAMBDA_INLINE { in[decltype(ir)::value] = v[decltype(ir)::value * N + c]; });
    dcts::Dct2<N>::run(in, o);
    o[0] *= dcts::kInvSqrt2;
    dcts::static_for<N>([&](auto ir) DCTS_LAMBDA_INLINE { v[decltype(ir)::value * N + c] = o[decltype(ir)::value]; });
  });
  float e = 0.f;
  dcts::static_for<N>([&](auto ir) DCTS_LAMBDA_INLINE {  // rows
    constexpr int r = decltype(ir)::value;
    float in[N], o[N];
    dcts::static_for<N>([&](auto ic) DCTS_LAMBDA_INLINE { in[decltype(ic)::value] = v[r * N + decltype(ic)::value]; });
    dcts::Dct2<N>::run(in, o);
    o[0] *= dcts::kInvSqrt2;
    dcts::static_for<N>([&](auto ic) DCTS_LAMBDA_INLINE { e = fmaf(o[decltype(ic)::value], o[decltype(ic)::value], e); });
  });
  constexpr float sc = float(4.0 / (double(N) * double(N)));
  return e * sc;
}

__device__ __forceinline__ LaneGroup lane_group_of(const MapGeom& g, float* out, long long grp, int nn, int item) {
  LaneGroup gr;
  gr.m0 = grp * 64;
  const long long left = g.nmaps - gr.m0;
  gr.nm = (int)(left < 64 ? left : 64);
  gr.src = g.x + (long long)g.c_begin * g.strideC + gr.m0 * nn;
  gr.dense = (g.contiguous && g.strideC == nn && ((reinterpret_cast<unsigned long long>(gr.src) & 15) == 0)) ? 1 : 0;
  gr.out = out + gr.m0;
  gr.item = item;
  return gr;
}

// one kernel for the single-tensor and the multi-tensor entry points (count == 1 for the former)
template <int N>
__global__ __launch_bounds__((64 * LaneCfg<N>::WAVES)) void k_energy_lane_multi(MultiGeom mg) {
  using Cfg = LaneCfg<N>;
  __shared__ __attribute__((aligned(16))) float slab[Cfg::WAVES][Cfg::SLAB];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const lds_ptr my = (lds_ptr)slab[wave];
  const long long wave_gid = (long long)blockIdx.x * Cfg::WAVES + wave;
  const long long nwaves = (long long)gridDim.x * Cfg::WAVES;
  int t = 0;
  auto locate = [&](long long grp) DCTS_LAMBDA_INLINE {
    while (t + 1 < mg.count && grp >= mg.it[t + 1].group_begin) ++t;  // wave-uniform, monotone
    t = __builtin_amdgcn_readfirstlane(t);
    return lane_group_of(mg.it[t].g, mg.it[t].out, grp - mg.it[t].group_begin, Cfg::NN, t);
  };
  if (wave_gid >= mg.total_groups) return;
  LaneGroup cur = locate(wave_gid);
  lane_stage<N>(cur, my, lane);
  for (long long grp = wave_gid; grp < mg.total_groups; grp += nwaves) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group has landed in the slab
    float v[Cfg::NN];
    lane_compute<N>(mg.it[cur.item].g, cur, my, lane, v);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and is in registers: the slab is free
    const LaneGroup done = cur;
    if (grp + nwaves < mg.total_groups) {
      cur = locate(grp + nwaves);
      lane_stage<N>(cur, my, lane);
    }
    const float e = lane_energy<N>(v);
    if (lane < done.nm) done.out[lane] = e;
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
// split family: tiles whose edge N = 2^L * M is too long for one lane's registers
// ---------------------------------------------------------------------------------------
// The top L radix-2 levels of the codelet recursion (dct_codelets.hpp) are unrolled across
// 2^L "role" waves instead of inside a lane; each role runs an M-point codelet on a length-M
// input it gathers from 2^L mirrored samples. The role tree, with y the input of a node:
//   DCT-II node (length n):  child 0 = DCT-II(n/2) of y[j] + y[n-1-j]
//                            child 1 = DCT-IV(n/2) of y[j] - y[n-1-j]
//   DCT-IV node (length n):  child 0 = DCT-II(n/2) of  y[j] cos(b_j) + y[n-1-j] sin(b_j)
//                            child 1 = DCT-II(n/2) of (-1)^j (y[n-1-j] cos(b_j) - y[j] sin(b_j)),
//                            b_j = (2j+1) pi / (4n); outputs A (child 0), B (child 1)
// A DCT-II node's outputs are its children's, interleaved (exact). A DCT-IV node's outputs are
// X[0] = A[0], X[n-1] = -B[0], X[2j] = A[j] + B[n/2-j], X[2j-1] = A[j] - B[n/2-j]: that last
// add/sub layer is a rotation of each pair scaled by sqrt(2); the energy kernels fuse it into the
// reduction, (a+b)^2 + (a-b)^2 = 2a^2 + 2b^2, i.e. A[j], B[j] (j > 0) are carried with weight
// sqrt(2) (SplitNode::wt). Every other butterfly, rotation and twiddle is computed.
//
// k_pass1d transforms the row axis of In[b][n][line] (lines contiguous): one workgroup per
// (<= 64-line strip), one wave per role; the strip is staged once with direct-to-LDS loads and
// every role wave gathers its mirrored rows from LDS. Non-final pass: the result goes to
// T[b][line][role*M + k] through a per-wave LDS transpose (coalesced stores), so the second launch
// of the same kernel transforms the other axis. Final pass: squares are reduced per wave into
// partial sums which k_split_reduce adds in fixed order.

template <int M, int L>
struct SplitCfg {
  static constexpr int N = M << L;
  static constexpr int ROLES = 1 << L;
  static constexpr int STRIPS = (N + 63) / 64;
  // lines per strip: 64 when rows are whole 128-byte lines (N % 32 == 0), so every staged row
  // segment is two aligned cache lines (56-column strips of a 224-wide tile straddled lines:
  // PMC showed 1.43x read over-fetch); otherwise balanced strips, multiple of 4
  static constexpr int SW = (N % 32 == 0) ? 64 : (((N + STRIPS - 1) / STRIPS) + 3) / 4 * 4;
  static constexpr int SWP = SW | 1;                                    // odd LDS stride for the transpose
  static constexpr int IN_LDS = N * SW;                                 // floats: the staged input strip
  static constexpr int TR_LDS = ROLES * M * SWP;                        // floats: per-wave transpose slabs
  static constexpr int LDS_NONFINAL = IN_LDS > TR_LDS ? IN_LDS : TR_LDS;
};


// register budget (waves/SIMD): the role waves only hold an M-point codelet
template <int M>
constexpr int split_waves_per_simd() { return M <= 32 ? 4 : (M <= 48 ? 3 : 2); }

// LDS pointers stay in address space 3 end to end: a generic pointer handed through these helpers
// needs a flat->local cast (with a null check) at every use, which ROCm 7.2's gfx950 backend
// mis-selects inside the fused kernel ("V_CMP_NE_U32 0, $src_shared_base": illegal instruction)

// role butterflies, in place: `base` is an LDS image [N rows][rs floats], lanes = columns
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// `hook` runs once per sample iteration: the fused kernel uses it to trickle out the direct-to-LDS
// loads of the next strip between butterflies instead of issuing them in one burst (a burst of
// 8 x 8 KiB per CU back-pressures the issue: stamps showed 470 cycles per load instruction)
template <int M, int L, class Hook = NoHook>
__device__ __forceinline__ void split_butterflies_pk(lds_ptr base, int rs, bool lane_ok, int lane, int wave,
                                                     Hook hook = Hook{}) {
  // Two samples p, p+1 per iteration as the halves of packed-f32 registers (v_pk_add/mul/fma_f32:
  // two results per issue slot): the network is the same for every p, only the rotation constants
  // differ. This phase is VALU-issue-bound, and the pairs halve its instruction count.
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int S = 1 << L;
  constexpr int NPAIR = (M + 1) / 2;
  constexpr RolePlan<L> plan{};
  const RotTable<M, L>& tab = kRotTable<M, L>;
  lds_ptr colp = base + (lane_ok ? lane : 0);
  for (int j = wave; j < NPAIR; j += S) {
    const int p = 2 * j;                   // even: (-1)^p = +1, (-1)^(p+1) = -1
    const bool two = (M % 2 == 0) || (p + 1 < M);
    const int p1 = two ? p + 1 : p;
    f2 y[S];
    dcts::static_for<S>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(i)::value;
      const int row0 = (s % 2 == 0) ? s * M + p : s * M + M - 1 - p;
      const int row1 = (s % 2 == 0) ? s * M + p1 : s * M + M - 1 - p1;
      y[s] = f2{colp[row0 * rs], colp[row1 * rs]};
    });
    dcts::static_for<plan.NOPS>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int o = decltype(i)::value;
      constexpr int a = plan.op_a[o], bb = plan.op_b[o], r = plan.op_rot[o];
      const f2 ya = y[a], yb = y[bb];
      if constexpr (r < 0) {
        y[a] = ya + yb;
        y[bb] = ya - yb;
      } else {
        constexpr float k0 = RotTable<M, L>::sign0(r);
        const f2 c = f2{tab.c[r][p], tab.c[r][p1]}, sn = f2{tab.s[r][p], tab.s[r][p1]};
        const f2 cs = f2{k0 * c.x, -k0 * c.y}, ss = f2{k0 * sn.x, -k0 * sn.y};  // sign of the second output folded in
        y[a] = ya * c + yb * sn;
        y[bb] = yb * cs - ya * ss;
      }
    });
    if (lane_ok) {
      dcts::static_for<S>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int s = decltype(i)::value;
        const int row0 = (s % 2 == 0) ? s * M + p : s * M + M - 1 - p;
        const int row1 = (s % 2 == 0) ? s * M + p1 : s * M + M - 1 - p1;
        colp[row0 * rs] = y[s].x;
        if (two) colp[row1 * rs] = y[s].y;
      });
    }
    hook();
  }
}

template <int M, int L, class Hook = NoHook, int NW = (1 << L)>
__device__ __forceinline__ void split_butterflies_1(lds_ptr base, int rs, bool lane_ok, int lane, int wave,
                                                    Hook hook = Hook{}) {
  constexpr int S = 1 << L;  // samples of one item; NW waves share the M items (NW < S: two roles per wave)
  constexpr RolePlan<L> plan{};
  const RotTable<M, L>& tab = kRotTable<M, L>;
  lds_ptr colp = base + (lane_ok ? lane : 0);
  for (int p = wave; p < M; p += NW) {
    const float sp = (p & 1) ? -1.f : 1.f;  // (-1)^p
    float y[S];
    dcts::static_for<S>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(i)::value;
      const int row = (s % 2 == 0) ? s * M + p : s * M + M - 1 - p;
      y[s] = colp[row * rs];
    });
    dcts::static_for<plan.NOPS>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int o = decltype(i)::value;
      constexpr int a = plan.op_a[o], bb = plan.op_b[o], r = plan.op_rot[o];
      const float ya = y[a], yb = y[bb];
      if constexpr (r < 0) {
        y[a] = ya + yb;
        y[bb] = ya - yb;
      } else {
        const float c = tab.c[r][p], sn = tab.s[r][p];
        constexpr float k0 = RotTable<M, L>::sign0(r);
        y[a] = ya * c + yb * sn;
        y[bb] = (k0 * sp) * (yb * c - ya * sn);
      }
    });
    if (lane_ok) {
      dcts::static_for<S>([&](auto i) DCTS_LAMBDA_INLINE {
        constexpr int s = decltype(i)::value;
        const int row = (s % 2 == 0) ? s * M + p : s * M + M - 1 - p;
        colp[row * rs] = y[s];
      });
    }
    hook();
  }
}

// The packed form halves the instruction count but also the number of busy waves, and doubles the
// 2^L live samples. Measured: +3..5 % in the two-launch pass kernel (288, 320), -2..8 % in the
// eight-wave fused kernels (too few waves left to hide LDS latency), spills in the sixteen-wave
// ones. So only k_pass1d asks for it.
template <int M, int L, class Hook = NoHook, bool PACK = false, int NW = (1 << L)>
__device__ __forceinline__ void split_butterflies(lds_ptr base, int rs, bool lane_ok, int lane, int wave,
                                                  Hook hook = Hook{}) {
  if constexpr (PACK)
    split_butterflies_pk<M, L>(base, rs, lane_ok, lane, wave, hook);
  else
    split_butterflies_1<M, L, Hook, NW>(base, rs, lane_ok, lane, wave, hook);
}

// role r's M-point transform of one column of the butterflied image: gathers the role's input
// segment, runs the codelet, applies the role's amplitude weights
template <int M, int L, int ROLE>
__device__ __forceinline__ void split_role_transform(lds_cptr col, int rs, float (&out)[M]) {
  using Leaf = typename RoleLeaf<(M << L), L, ROLE>::type;
  constexpr RolePlan<L> plan{};
  constexpr int SLOT = plan.slot_of_role[ROLE];
  constexpr bool ASC = plan.asc_of_role[ROLE] != 0;
  static_assert(Leaf::len == M, "role tree depth");
  static_assert((plan.is4_of_role[ROLE] != 0) == Leaf::is4, "role plan and role tree disagree");
  float in[M];
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int q = decltype(i)::value;   // index of the sample in the role's input
    constexpr int p = ASC ? q : M - 1 - q;  // the (p, line) item that produced it
    constexpr int row = (SLOT % 2 == 0) ? SLOT * M + p : SLOT * M + M - 1 - p;
    in[q] = col[row * rs];
  });
#if DCTS_F2_EXP == 4
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { out[decltype(i)::value] = in[decltype(i)::value] * 1.5f; });
#else
  if constexpr (Leaf::is4)
    dcts::Dct4<M>::run(in, out);
  else
    dcts::Dct2<M>::run(in, out);
#endif
  constexpr float w0 = float(Leaf::wt(true)), w1 = float(Leaf::wt(false));
  if constexpr (w0 != 1.0f) out[0] *= w0;
  if constexpr (w1 != 1.0f)
    dcts::static_for<M - 1>([&](auto i) DCTS_LAMBDA_INLINE { out[decltype(i)::value + 1] *= w1; });
}

template <int M, int L, int ROLE, bool FINAL>
__device__ __forceinline__ void split_wave(lds_cptr lds_in, float* __restrict__ t_b, lds_ptr lds_tr,
                                           int strip, int lane, float* part) {
  using Cfg = SplitCfg<M, L>;
  constexpr int N = Cfg::N, SW = Cfg::SW, SWP = Cfg::SWP;
  const int line = strip * SW + lane;
  const bool act = lane < SW && line < N;
  float out[M];
  split_role_transform<M, L, ROLE>(lds_in + (act ? lane : 0), SW, out);
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
    lds_ptr my = lds_tr + ROLE * (M * SWP);
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
        lds_cptr src = my + k * SWP;
#pragma unroll 4
        for (int j = 0; j < nl; ++j) {
          *dst = src[j];
          dst += N;
        }
      }
    }
  }
}

template <int M, int L, bool FINAL, int... R>
__device__ __forceinline__ void split_dispatch(int role, lds_cptr lds_in, float* t_b, lds_ptr lds_tr,
                                               int strip, int lane, float* part,
                                               std::integer_sequence<int, R...>) {
  // exactly one branch is taken per wave (role is wave-uniform); every branch reaches the
  // barrier inside split_wave, so the workgroup stays in step
  ((role == R ? split_wave<M, L, R, FINAL>(lds_in, t_b, lds_tr, strip, lane, part) : (void)0), ...);
}

// grid.x = nmaps_in_launch * STRIPS; block = 2^L waves.
//  1. the strip In[b][0..N)[strip*SW .. +SW) is staged into LDS by direct-to-LDS loads
//     (global_load_lds_dwordx4: no VGPRs, the whole N*SW*4-byte strip in flight at once);
//  2. role butterflies, in place in LDS: wave w takes the samples p = w, w + 2^L, ... of every
//     line (lane = line): 2^L reads, L*2^(L-1) butterflies/rotations, 2^L writes per (p, line);
//  3. wave r = role r: reads its M inputs (one contiguous segment of rows), M-point codelet,
//     then the transposed store or the energy reduction.
template <int M, int L, bool FINAL>
__global__ __launch_bounds__((64 << L), (split_waves_per_simd<M>())) void k_pass1d(
    const float* __restrict__ in, long long in_map_stride, float* __restrict__ t, float* __restrict__ partial) {
  using Cfg = SplitCfg<M, L>;
  constexpr int N = Cfg::N, SW = Cfg::SW, THREADS = 64 << L, S = Cfg::ROLES;
  __shared__ __attribute__((aligned(16))) float lds[FINAL ? Cfg::IN_LDS : Cfg::LDS_NONFINAL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long b = blockIdx.x / Cfg::STRIPS;
  const int strip = blockIdx.x - (int)(b * Cfg::STRIPS);
  const float* in_b = in + b * in_map_stride;

  constexpr int NQUADS = N * SW / 4;
  constexpr int ITERS = (NQUADS + THREADS - 1) / THREADS;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int qbase = it * THREADS + wave * 64;  // wave-uniform
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

  const lds_ptr lds3 = (lds_ptr)lds;
  split_butterflies<M, L, NoHook, true>(lds3, SW, lane < SW, lane, wave);
  __syncthreads();

  float* t_b = FINAL ? nullptr : t + b * (long long)N * N;
  float* part = partial + (long long)blockIdx.x * Cfg::ROLES + wave;
  split_dispatch<M, L, FINAL>(wave, lds3, t_b, lds3, strip, lane, part,
                              std::make_integer_sequence<int, Cfg::ROLES>{});
}


// ---------------------------------------------------------------------------------------
// fused split kernel: one launch, HBM traffic = the input, for tiles the register file can park
// ---------------------------------------------------------------------------------------
// One persistent workgroup (2^L role waves) per CU walks over maps. Pass 1 as in k_pass1d, strip
// by strip (64 columns, double-buffered direct-to-LDS staging: strip s+1 streams in while strip s
// is transformed), but the role outputs are not written out: wave q keeps T[line][q*M + k] for
// all its lines in VGPRs (STRIPS*M registers per lane: the whole N x N intermediate tile lives in
// the register file). Pass 2 runs in rounds of 56-64 coefficient columns (KPR from every role):
// the waves dump those parked rows into LDS as an image [line][column], the role butterflies
// run in place along the lines, every wave runs one W-role codelet with lane = column, and the
// squares are accumulated.
template <int M, int L>
struct FusedCfg {
  static constexpr int N = M << L;
  static constexpr int S = 1 << L;
  static constexpr int SW = 64;
  static constexpr int STRIPS = (N + SW - 1) / SW;
  // whole roles per unbalanced round: a power of two, or as many as fit the 64 lanes where that
  // saves a round on the eight-wave kernels (144 = 18 x 8: rounds of 3+3+2 roles instead of four
  // rounds of 2, 26.7 -> 30.9 % of peak; no gain measured at 160 = 10 x 16)
  static constexpr int RPR_P2 = (64 / M) >= 4 ? 4 : ((64 / M) >= 2 ? 2 : 1);
  static constexpr int RPR_FIT = (64 / M) > (1 << L) ? (1 << L) : ((64 / M) >= 1 ? 64 / M : 1);
  static constexpr int RPR = (L <= 3 && (S + RPR_FIT - 1) / RPR_FIT < S / RPR_P2) ? RPR_FIT : RPR_P2;
  // which parked rows go into a pass-2 round:
  //  BALANCED: KPR = 64/S coefficients of EVERY role (all waves dump, equal work; M is padded up to
  //            ROUNDS*KPR with zero columns) - used where the padding wastes <= 1/6 of the columns;
  //  otherwise RPR whole roles per round (only their waves dump).
  static constexpr int KPR_B = 64 / S;
  static constexpr int ROUNDS_B = (M + KPR_B - 1) / KPR_B;
  static constexpr bool BALANCED = (S <= 64) && (6 * (ROUNDS_B * KPR_B - M) <= ROUNDS_B * KPR_B) &&
                                   !(M == 14 && L == 4) && M != 28;  // those two spill when every wave keeps its parked set live
  static constexpr int KPR = KPR_B;
  static constexpr int COLS = BALANCED ? S * KPR_B : RPR * M;  // pass-2 columns (lanes) per round
  static constexpr int ROUNDS = BALANCED ? ROUNDS_B : (S + RPR - 1) / RPR;  // the last round may hold fewer roles
  static constexpr int RW = COLS | 1;                          // pass-2 image row stride (odd: conflict-free dump)
  static constexpr int BUF = (N * SW > N * RW ? N * SW : N * RW);  // floats per LDS buffer
  static_assert(N % 4 == 0, "shape");
};

// one direct-to-LDS instruction (64 lanes x 16 B) of a strip's staging: piece `it` of PIECES.
// lane q = it*THREADS + wave*64 + lane covers row q/16, columns 4*(q%16).. of the 64-wide strip
template <int M, int L, int NW = (1 << L)>
struct FusedStage {
  static constexpr int N = M << L, SW = 64, THREADS = 64 * NW;
  static constexpr int NQUADS = N * SW / 4;
  static constexpr int PIECES = (NQUADS + THREADS - 1) / THREADS;
  static_assert(SW == 64, "piece addressing assumes 16 quads per row");
  // One direct-to-LDS load, issued as a raw instruction. The compiler tracks direct-to-LDS loads it knows about and
  // puts s_waitcnt vmcnt(0) in front of the next LDS access that may alias the destination; its
  // alias information does not survive this kernel's pointer arithmetic, so EVERY following
  // ds_read/ds_write waited for the prefetch to land (one memory round trip per instalment). The
  // pipelined kernel orders these loads itself: s_waitcnt vmcnt(0) + barrier before the strip is read.
  static __device__ __forceinline__ void piece_raw(const float* __restrict__ in_b, int strip, lds_ptr buf, int lane,
                                                   int wave, int it) {
    const int qbase = it * THREADS + wave * 64;  // wave-uniform
    const int q = qbase + lane;
    const int row = q >> 4, col = (q & 15) << 2;
    if (q < NQUADS && strip * SW + col < N) {
      const float* base = in_b + strip * SW;                // wave-uniform (tile_in)
      const unsigned off = (unsigned)(row * N + col) * 4u;  // bytes
      const unsigned dst = (unsigned)(unsigned long long)(buf + 4 * qbase);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :
                   : "s"(dst), "v"(off), "s"(base)
                   : "memory", "m0");
    }
  }
};

// Diagnostic build only (-DDCTS_FUSED_STAMPS, tools/stamp_fused.sh): s_memtime stamps at the phase
// boundaries of the fused kernel, summed per wave into g_fused_stamps (never touches an output).
#ifdef DCTS_FUSED_STAMPS
__device__ unsigned long long g_fused_stamps[16][16];
#define DCTS_STAMP(slot)                                                          \
  do {                                                                            \
    unsigned long long t_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
    __builtin_amdgcn_sched_barrier(0);                                            \
    acc_[slot] += t_ - last_;                                                     \
    last_ = t_;                                                                   \
  } while (0)
#else
#define DCTS_STAMP(slot) ((void)0)
#endif

// sum of a map's per-wave partials in fixed order (wave 0, lane 0) and the final scale
template <int M, int L, int ROLE, class Src>
__device__ __forceinline__ void fused_finish(lds_ptr partials, int slot, long long m, const Src& tb, int lane, int* hint = nullptr) {
  constexpr int S = 1 << L, N = M << L;
  if (ROLE == 0 && lane == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i) t += partials[slot * S + i];
    constexpr float sc = float(4.0 / (double(N) * double(N)));
    *tile_out(tb, m, hint) = t * sc;
  }
}

// STORE: debug / parity instantiation (dcts_dct2d_f32_ex with DCTS_ALGO_FUSED): the weighted leaf outputs
// of pass 2 also go to leaf_out[map][roleH * M + kH][roleW * M + kW]; k_assemble (split_roles.hpp) applies
// the DCT-IV add/sub layers the energy path folds into its weights. No energy is written.
// (defined with the two-roles kernel below)
template <int M, int L, int P, int STRIP>
__device__ __forceinline__ void f2_load_item(__amdgpu_buffer_rsrc_t rs, int voff, float (&y)[1 << L]);
template <int M, int L, int P>
__device__ __forceinline__ void f2_network_store(float (&y)[1 << L], lds_ptr image, int rs_lds, int lane, bool act);
// The one-role-per-wave fused kernel with pass 1 on samples loaded into registers and alternating pass-2 buffers (see
// DCTS_F2_REGLOAD below). Same box, staged -> register loads, % of the HBM peak: 96: 38.1 -> 43.2, 192: 33.0 -> 33.9, 256: 35.3 -> 35.8
// (762 maps) / 39.0 -> 41.0 (3000), 112: 38.2 -> 38.3; the shapes AUTO gives to other kernels: 128 45.3 -> 48.9, 144 31.9 -> 34.6,
// 224 29.4 -> 30.8, 160 32.9 -> 33.8, 72 33.4 -> 31.8. No scratch (one coefficient instantiation: 12 B).
#ifndef DCTS_F1_REGLOAD
#define DCTS_F1_REGLOAD 1
#endif

template <int M, int L, int ROLE, bool STORE = false>
__device__ __forceinline__ void fused_body(const TileBatch& tb, lds_ptr lds, lds_ptr partials, int lane,
                                           float* leaf_out = nullptr) {
  using Cfg = FusedCfg<M, L>;
  constexpr int N = Cfg::N, S = Cfg::S, SW = Cfg::SW, STRIPS = Cfg::STRIPS, COLS = Cfg::COLS, KPR = Cfg::KPR,
                RPR = Cfg::RPR, ROUNDS = Cfg::ROUNDS, RW = Cfg::RW, BUF = Cfg::BUF;
  int cur = 0, pslot = 0, pending_slot = 0;
  long long pending_m = -1;
  long long m = blockIdx.x;
#ifdef DCTS_FUSED_STAMPS
  unsigned long long acc_[16] = {}, last_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
  const long long nmaps = tb.total;
  int hint_in = 0, hint_next = 0, hint_out = 0;  // tensor of the current / next / finished map (tile_item)
  constexpr int ITEMS = ROLE < M ? (M - ROLE + S - 1) / S : 0;  // this wave's butterfly items p = ROLE, ROLE + S, ...
  static_assert(ITEMS <= ROUNDS, "one item of the next map per pass-2 round");
  float pre[DCTS_F1_REGLOAD && ITEMS > 0 ? ITEMS : 1][1 << L];
  auto map_rsrc = [&](const float* base, bool valid) DCTS_LAMBDA_INLINE {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, valid ? (unsigned)(N * N * 4) : 0u, 0x00020000);
  };
  auto lane_voff = [&](int strip) DCTS_LAMBDA_INLINE {
    const int ln = launder(lane);
    return (strip * SW + ln < N) ? ln * 4 : 0x7ffffff0;
  };
  if (m < nmaps) {
    const float* first = tile_in(tb, m);
    if constexpr (DCTS_F1_REGLOAD != 0) {
      const __amdgpu_buffer_rsrc_t rs = map_rsrc(first, true);
      const int vo = lane_voff(0);
      dcts::static_for<ITEMS>([&](auto ii) DCTS_LAMBDA_INLINE {
        constexpr int i = decltype(ii)::value;
        f2_load_item<M, L, ROLE + S * i, 0>(rs, vo, pre[i]);
      });
    } else {
#pragma unroll
      for (int it = 0; it < FusedStage<M, L>::PIECES; ++it) FusedStage<M, L>::piece_raw(first, 0, lds, lane, ROLE, it);
    }
  }
  for (; m < nmaps; m += gridDim.x) {
    const float* in_b = tile_in(tb, m, &hint_in);
    float parked[STRIPS][M];
#if DCTS_F1_REGLOAD
    const bool more_maps = m + gridDim.x < nmaps;
    const float* next_b = more_maps ? tile_in(tb, m + gridDim.x, &hint_next) : in_b;
    // ---- pass 1 on samples in registers: one barrier per strip, the two buffers alternate as role images ---------------
    dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(is)::value;
      const lds_ptr buf = lds + cur * BUF;
      const int ln = launder(lane);
      const bool act = s * SW + ln < N;
      DCTS_STAMP(2);
      {
        const __amdgpu_buffer_rsrc_t rs = map_rsrc(in_b, true);
        const int vo = (s + 1 < STRIPS) ? lane_voff(s + 1) : 0;
        dcts::static_for<ITEMS>([&](auto ii) DCTS_LAMBDA_INLINE {
          constexpr int i = decltype(ii)::value;
          f2_network_store<M, L, ROLE + S * i>(pre[i], buf, SW, ln, act);
          if constexpr (s + 1 < STRIPS) f2_load_item<M, L, ROLE + S * i, (s + 1 < STRIPS ? s + 1 : 0)>(rs, vo, pre[i]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      DCTS_STAMP(3);
      lds_barrier();
      DCTS_STAMP(4);
      if constexpr (s == 0) {
        if (pending_m >= 0) {
          if constexpr (!STORE) fused_finish<M, L, ROLE>(partials, pending_slot, pending_m, tb, lane, &hint_out);
          pending_m = -1;
        }
      }
      split_role_transform<M, L, ROLE>(buf + (act ? launder(lane) : 0), SW, parked[s]);
      DCTS_STAMP(5);
      cur ^= 1;
    });
#else
    // ---- pass 1: H axis, strip by strip -------------------------------------------------
    dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(is)::value;
      DCTS_STAMP(11);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this strip has landed
      DCTS_STAMP(0);
      lds_barrier();                                   // ... for everyone; the other buffer is free
      if constexpr (s == 0) {
        if (pending_m >= 0) {
          if constexpr (!STORE) fused_finish<M, L, ROLE>(partials, pending_slot, pending_m, tb, lane, &hint_out);
          pending_m = -1;
        }
      }
      DCTS_STAMP(1);
      const lds_ptr buf = lds + cur * BUF;
      const lds_ptr nxt = lds + (cur ^ 1) * BUF;
      // the next strip (or the next map's first one) streams into the other buffer while this one
      // is transformed; its load instructions are trickled out between the butterflies
      const bool more = (s + 1 < STRIPS) || (m + gridDim.x < nmaps);
      const float* nsrc = (s + 1 < STRIPS || !more) ? in_b : tile_in(tb, m + gridDim.x, &hint_next);
      constexpr int nstrip = (s + 1 < STRIPS) ? s + 1 : 0;
      int piece = more ? 0 : FusedStage<M, L>::PIECES;
      auto trickle = [&]() DCTS_LAMBDA_INLINE {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (piece < FusedStage<M, L>::PIECES) FusedStage<M, L>::piece_raw(nsrc, nstrip, nxt, lane, ROLE, piece++);
      };
      const bool act = s * SW + lane < N;  // lane < 64 == SW always
      DCTS_STAMP(2);
      split_butterflies<M, L>(buf, SW, act, lane, ROLE, trickle);
      while (piece < FusedStage<M, L>::PIECES) FusedStage<M, L>::piece_raw(nsrc, nstrip, nxt, lane, ROLE, piece++);
      DCTS_STAMP(3);
      lds_barrier();
      DCTS_STAMP(4);
      split_role_transform<M, L, ROLE>(buf + (act ? lane : 0), SW, parked[s]);
      DCTS_STAMP(5);
      cur ^= 1;
    });
#endif
    // ---- pass 2: W axis, RPR role groups of parked rows per round ---------------------------
    const lds_ptr blk0 = lds + (cur ^ 1) * BUF;  // the last strip's buffer; the other one is receiving (staged) / free (register loads)
    const lds_ptr blk1 = lds + cur * BUF;
    constexpr bool ALT = DCTS_F1_REGLOAD != 0;  // rounds alternate between the buffers: two barriers per round (see fused2_body)
    float e = 0.f;
    dcts::static_for<ROUNDS>([&](auto ir) DCTS_LAMBDA_INLINE {
      constexpr int r = decltype(ir)::value;
      const lds_ptr blk = (ALT && r % 2 == 1) ? blk1 : blk0;
      DCTS_STAMP(11);
      if constexpr (!ALT || r == 0) lds_barrier();  // previous readers of blk are done
      DCTS_STAMP(6);
      if constexpr (Cfg::BALANCED) {
        dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
          constexpr int s = decltype(is)::value;
          const int line = s * SW + lane;
          const int off = (line < N ? line : 0) * RW + ROLE * KPR;
          dcts::static_for<KPR>([&](auto ik) DCTS_LAMBDA_INLINE {
            constexpr int k = decltype(ik)::value;
            if constexpr (r * KPR + k < M) {
              if (line < N) blk[off + k] = parked[s][r * KPR + k];
            } else {
              if (line < N) blk[off + k] = 0.f;  // padding column: contributes exactly zero energy
            }
          });
        });
      } else if constexpr (ROLE / RPR == r) {
        dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
          constexpr int s = decltype(is)::value;
          const int line = s * SW + lane;
          const int off = (line < N ? line : 0) * RW + (ROLE % RPR) * M;
          dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
            constexpr int k = decltype(ik)::value;
            if (line < N) blk[off + k] = parked[s][k];
          });
        });
      }
      DCTS_STAMP(7);
#if DCTS_F1_REGLOAD
      if constexpr (r < ITEMS) {  // item r of the next map's first strip
        const __amdgpu_buffer_rsrc_t rs = map_rsrc(next_b, more_maps);
        f2_load_item<M, L, ROLE + S * (r < ITEMS ? r : 0), 0>(rs, lane_voff(0), pre[r < ITEMS ? r : 0]);
      }
#endif
      lds_barrier();
      DCTS_STAMP(8);
      // columns of this round: all of them, or fewer whole roles in the last unbalanced round
      constexpr int cols_r = Cfg::BALANCED ? COLS : ((S - r * RPR) < RPR ? (S - r * RPR) : RPR) * M;
      const bool act = lane < cols_r;
      split_butterflies<M, L>(blk, RW, act, lane, ROLE);
      DCTS_STAMP(9);
      lds_barrier();
      DCTS_STAMP(10);
      float o[M];
      split_role_transform<M, L, ROLE>(blk + (act ? lane : 0), RW, o);
      if constexpr (STORE) {
        // image column `lane` of round r holds the H-axis leaf output iH
        int iH;
        if constexpr (Cfg::BALANCED) {
          const int q = lane / KPR, kh = r * KPR + (lane - q * KPR);
          iH = kh < M ? q * M + kh : -1;  // padding columns
        } else {
          const int q = lane / M;
          iH = (r * RPR + q) * M + (lane - q * M);
        }
        if (act && iH >= 0) {
          float* dst = leaf_out + ((long long)m * N + iH) * N + ROLE * M;
          dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE { dst[decltype(ik)::value] = o[decltype(ik)::value]; });
        }
      }
      float er = 0.f;
      dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
        constexpr int k = decltype(ik)::value;
        er = fmaf(o[k], o[k], er);
      });
      if (act) e += er;
      DCTS_STAMP(12);
    });
    // ---- reduce: lanes -> wave -> workgroup, fixed order -------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e += __shfl_down(e, off, 64);
    // the workgroup-level sum is deferred past the next barrier the loop executes anyway (the first
    // one of the next map, or the one after the loop): partials are double-buffered by map parity
    if (lane == 0) partials[pslot * S + ROLE] = e;
    pending_m = m;
    pending_slot = pslot;
    pslot ^= 1;
    if constexpr (DCTS_F1_REGLOAD != 0 && ROUNDS % 2 == 0) cur ^= 1;  // the next map's first strip must not overwrite the last round's image
    DCTS_STAMP(13);
  }
  if (pending_m >= 0) {
    lds_barrier();
    if constexpr (!STORE) fused_finish<M, L, ROLE>(partials, pending_slot, pending_m, tb, lane, &hint_out);
  }
#ifdef DCTS_FUSED_STAMPS
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_fused_stamps[ROLE][i], acc_[i]);
#endif
}

template <int M, int L, bool STORE, int... R>
__device__ __forceinline__ void fused_dispatch(int role, const TileBatch& tb, lds_ptr lds, lds_ptr partials, int lane,
                                               float* leaf_out, std::integer_sequence<int, R...>) {
  ((role == R ? fused_body<M, L, R, STORE>(tb, lds, partials, lane, leaf_out) : (void)0), ...);
}

// waves per SIMD the register file allows: STRIPS*M parked values + the codelet's working set
template <int M, int L>
constexpr int fused_waves_per_simd() {
  const int need = FusedCfg<M, L>::STRIPS * M + 72;
  int w = 512 / ((need + 7) / 8 * 8);
  const int per_wg = (1 << L) / 4 > 0 ? (1 << L) / 4 : 1;
  if (w < per_wg) w = per_wg;
  if (w > 8) w = 8;
  return w;
}

template <int M, int L>
__global__ __launch_bounds__((64 << L), (fused_waves_per_simd<M, L>())) void k_split_fused(TileBatch tb) {
  using Cfg = FusedCfg<M, L>;
  __shared__ __attribute__((aligned(16))) float lds[2 * Cfg::BUF];
  __shared__ float partials[2 * Cfg::S];
  fused_dispatch<M, L, false>(threadIdx.x >> 6, tb, (lds_ptr)lds, (lds_ptr)partials, threadIdx.x & 63, nullptr,
                              std::make_integer_sequence<int, Cfg::S>{});
}
template <int M, int L>
__global__ __launch_bounds__((64 << L), (fused_waves_per_simd<M, L>())) void k_split_fused_coeff(TileBatch tb, float* leaf_out) {
  using Cfg = FusedCfg<M, L>;
  __shared__ __attribute__((aligned(16))) float lds[2 * Cfg::BUF];
  __shared__ float partials[2 * Cfg::S];
  fused_dispatch<M, L, true>(threadIdx.x >> 6, tb, (lds_ptr)lds, (lds_ptr)partials, threadIdx.x & 63, leaf_out,
                             std::make_integer_sequence<int, Cfg::S>{});
}

// ---------------------------------------------------------------------------------------
// fused kernel with two roles per wave: 288 = 18 x 16 roles on eight waves
// ---------------------------------------------------------------------------------------
// A 288 x 288 tile (81 K floats) fits the register file of a CU (128 K floats) but not next to the
// codelet working set at 16 waves x 128 VGPRs (5 strips x 18 parked + ~60). Eight waves own 256
// VGPRs each: wave w runs roles 2w and 2w+1 one after the other (2 x 5 x 18 = 180 parked values),
// the butterfly items are shared by the eight waves, the pass-2 dump is the balanced one (KPR
// coefficients of every role per round). Otherwise the fused kernel above: double-buffered
// direct-to-LDS staging, LDS-only barriers, deferred workgroup sum. One launch, HBM traffic = the
// input once, instead of the 3x of the two-launch path.
template <int M, int L>
struct Fused2Cfg {
  static constexpr int N = M << L, S = 1 << L, NW = S / 2, SW = 64;
  static constexpr int STRIPS = (N + SW - 1) / SW;
  // Two LDS buffers of max(strip, pass-2 image) floats. With all 64 columns per round the image
  // (N x 65) is the larger one; where two of those exceed the 160 KiB (320: 166 KB) a round takes
  // 48 columns (KPR = 3 per role, image N x 49) and the workgroup partials move into the slack behind
  // the image, which costs the deferred workgroup sum (one more barrier per map).
  static constexpr int LDS_FLOATS = 160 * 1024 / 4;
  static constexpr bool WIDE = 2 * N * 65 + 2 * NW <= LDS_FLOATS;
  static constexpr int KPR = WIDE ? 64 / S : 48 / S;
  static constexpr int COLS = S * KPR;
  static constexpr int ROUNDS = (M + KPR - 1) / KPR;
  static constexpr int RW = COLS + 1;
  static constexpr int BUF = N * RW > N * SW ? N * RW : N * SW;
  static constexpr bool DEFER = 2 * BUF + 2 * NW <= LDS_FLOATS;  // room for separate partials
  static_assert(S >= 2 && S <= 16 && N % 4 == 0 && KPR >= 1, "shape");
  static_assert(DEFER || N * RW + NW <= BUF, "partials must fit the slack behind the image");
  static_assert(2 * BUF <= LDS_FLOATS, "LDS");
};

// ---- pass 1 of the two-roles kernel with the samples loaded straight into registers (DCTS_F2_REGLOAD) ------------------
// The staged version brings a strip into LDS with direct-to-LDS loads, runs the role butterflies IN PLACE (16 reads, the
// network, 16 writes per item) and then the role codelets read their rows: two LDS writes and two reads per sample, and
// during pass 2 - no free buffer - nothing can stream in. Here an item's 16 samples x[a*M + p~][line] are buffer loads
// (lane = line: 256 contiguous bytes per wave instruction; p is a compile-time constant of the wave, so every row offset
// is an immediate and the rotation constants are literals), the network runs on them in registers and its outputs are
// written once into the role image: one LDS write and one read per sample, one workgroup barrier per strip (the two
// buffers alternate as images), no alignment requirement. The samples of the next strip are requested as soon as an
// item's registers are free (they fly during the remaining butterflies, the barrier and the codelets); those of the next
// map's first strip during pass 2, item by item as the dumps free registers.
// Same box, % of the HBM peak, staged -> register loads -> + pass-2 rounds alternating between the two buffers (two barriers
// per round instead of three): 288 x 288: 28.7 -> 30.4 -> 31.2 (2048 maps), 30.8 -> 33.0 -> 34.2 (4999), 27.3 -> 28.9 -> 29.5 (768);
// 320 x 320: 29.7 -> 31.0 -> 31.5 (2048). 219 / 248 VGPRs, no scratch. On by default; 0 restores the staged pass 1.
#ifndef DCTS_F2_REGLOAD
#define DCTS_F2_REGLOAD 1
#endif
constexpr int kF2Out = 0x7ffffff0;  // a lane offset beyond any map: the load returns 0 and makes no request
template <int M, int L, int P, int STRIP>
__device__ __forceinline__ void f2_load_item(__amdgpu_buffer_rsrc_t rs, int voff, float (&y)[1 << L]) {
  constexpr int S = 1 << L, N = M << L;
  dcts::static_for<S>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int s = decltype(i)::value;
    constexpr int row = (s % 2 == 0) ? s * M + P : s * M + M - 1 - P;
    y[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (row * N + STRIP * 64) * 4, 0));
  });
}
template <int M, int L, int P>
__device__ __forceinline__ void f2_network_store(float (&y)[1 << L], lds_ptr image, int rs_lds, int lane, bool act) {
  constexpr int S = 1 << L;
  constexpr RolePlan<L> plan{};
  constexpr RotTable<M, L> tab{};
  constexpr float sp = (P & 1) ? -1.f : 1.f;
  dcts::static_for<plan.NOPS>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int o = decltype(i)::value;
    constexpr int a = plan.op_a[o], bb = plan.op_b[o], r = plan.op_rot[o];
    const float ya = y[a], yb = y[bb];
    if constexpr (r < 0) {
      y[a] = ya + yb;
      y[bb] = ya - yb;
    } else {
      constexpr float c = tab.c[r][P], sn = tab.s[r][P];
      constexpr float k0 = RotTable<M, L>::sign0(r);
      y[a] = ya * c + yb * sn;
      y[bb] = (k0 * sp) * (yb * c - ya * sn);
    }
  });
  if (act) {
    lds_ptr colp = image + lane;
    dcts::static_for<S>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(i)::value;
      constexpr int row = (s % 2 == 0) ? s * M + P : s * M + M - 1 - P;
      colp[row * rs_lds] = y[s];
    });
  }
}

template <int M, int L, int W, bool STORE = false>
__device__ __forceinline__ void fused2_body(const TileBatch& tb, lds_ptr buf0, lds_ptr buf1, lds_ptr partials,
                                            int lane_in, float* leaf_out = nullptr) {
  using Cfg = Fused2Cfg<M, L>;
  using Stage = FusedStage<M, L, Cfg::NW>;
  constexpr int N = Cfg::N, NW = Cfg::NW, SW = Cfg::SW, STRIPS = Cfg::STRIPS, KPR = Cfg::KPR,
                ROUNDS = Cfg::ROUNDS, RW = Cfg::RW, COLS = Cfg::COLS;
  constexpr int R0 = 2 * W, R1 = 2 * W + 1;
  int cur = 0, pslot = 0, pending_slot = 0;
  long long pending_m = -1;
  long long m = blockIdx.x;
  const long long nmaps = tb.total;
#ifdef DCTS_FUSED_STAMPS
  unsigned long long acc_[16] = {}, last_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
  int hint_in = 0, hint_next = 0, hint_out = 0;  // tensor of the current / next / finished map (tile_item)
  auto finish = [&](lds_ptr part, int slot, long long mm) DCTS_LAMBDA_INLINE {
    if (W == 0 && lane_in == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) t += part[slot * NW + i];
      constexpr float sc = float(4.0 / (double(N) * double(N)));
      if constexpr (!STORE) *tile_out(tb, mm, &hint_out) = t * sc;
    }
  };
  // register-load pass 1: this wave's butterfly items are p = W, W + NW, ... (compile time); pre[i] holds item i's samples
  constexpr int ITEMS = (M - W + NW - 1) / NW;
  static_assert(ITEMS <= ROUNDS, "one item of the next map per pass-2 round");
  float pre[DCTS_F2_REGLOAD ? ITEMS : 1][1 << L];
  auto map_rsrc = [&](const float* base, bool valid) DCTS_LAMBDA_INLINE {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, valid ? (unsigned)(N * N * 4) : 0u, 0x00020000);
  };
  auto lane_voff = [&](int strip) DCTS_LAMBDA_INLINE {
    const int lane = launder(lane_in);
    return (strip * SW + lane < N) ? lane * 4 : kF2Out;
  };
  if (m < nmaps) {
    const float* first = tile_in(tb, m);
    if constexpr (DCTS_F2_REGLOAD) {
      const __amdgpu_buffer_rsrc_t rs = map_rsrc(first, true);
      const int vo = lane_voff(0);
      dcts::static_for<ITEMS>([&](auto ii) DCTS_LAMBDA_INLINE {
        constexpr int i = decltype(ii)::value;
        f2_load_item<M, L, W + NW * i, 0>(rs, vo, pre[i]);
      });
    } else {
#pragma unroll
      for (int it = 0; it < Stage::PIECES; ++it) Stage::piece_raw(first, 0, buf0, lane_in, W, it);
    }
  }
  for (; m < nmaps; m += gridDim.x) {
    const float* in_b = tile_in(tb, m, &hint_in);
    float parked[2][STRIPS][M];
#if DCTS_F2_REGLOAD
    const bool more_maps = m + gridDim.x < nmaps;
    const float* next_b = more_maps ? tile_in(tb, m + gridDim.x, &hint_next) : in_b;
    // ---- pass 1: H axis, strip by strip; butterflies on samples in registers, one barrier per strip ------------
    dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(is)::value;
      const lds_ptr buf = cur ? buf1 : buf0;
      int lane = launder(lane_in);
      const bool act = s * SW + lane < N;
      DCTS_STAMP(2);
      {
        const __amdgpu_buffer_rsrc_t rs = map_rsrc(in_b, true);
        const int vo = (s + 1 < STRIPS) ? lane_voff(s + 1) : 0;
        dcts::static_for<ITEMS>([&](auto ii) DCTS_LAMBDA_INLINE {
          constexpr int i = decltype(ii)::value;
          f2_network_store<M, L, W + NW * i>(pre[i], buf, SW, lane, act);
          // the registers of this item are free: request its samples of the next strip
          if constexpr (s + 1 < STRIPS) f2_load_item<M, L, W + NW * i, (s + 1 < STRIPS ? s + 1 : 0)>(rs, vo, pre[i]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      DCTS_STAMP(3);
      lds_barrier();  // the image of strip s is complete; everyone is past the codelets of strip s - 1 (the other buffer)
      DCTS_STAMP(4);
      if constexpr (s == 0) {
        if (pending_m >= 0) {
          finish(partials, pending_slot, pending_m);
          pending_m = -1;
        }
      }
      dcts::static_for<2>([&](auto ii) DCTS_LAMBDA_INLINE {
        constexpr int i = decltype(ii)::value;
        const int ln = launder(lane_in);
        float o[M];
        split_role_transform<M, L, 2 * W + i>(buf + (act ? ln : 0), SW, o);
        dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
          constexpr int k = decltype(ik)::value;
          asm volatile("" : "+v"(o[k]));  // pin the codelet here (LLVM would sink it to the dump)
          parked[i][s][k] = o[k];
        });
      });
      DCTS_STAMP(5);
      cur ^= 1;
    });
#else
    // ---- pass 1: H axis, strip by strip -------------------------------------------------
    dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
      constexpr int s = decltype(is)::value;
      DCTS_STAMP(11);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this strip has landed
      DCTS_STAMP(0);
      lds_barrier();                                     // ... for everyone; the other buffer is free
      DCTS_STAMP(1);
      if constexpr (s == 0) {
        if (pending_m >= 0) {
          finish(partials, pending_slot, pending_m);
          pending_m = -1;
        }
      }
      const lds_ptr buf = cur ? buf1 : buf0;
      const lds_ptr nxt = cur ? buf0 : buf1;
      const bool more = (s + 1 < STRIPS) || (m + gridDim.x < nmaps);
      const float* nsrc = (s + 1 < STRIPS || !more) ? in_b : tile_in(tb, m + gridDim.x, &hint_next);
      constexpr int nstrip = (s + 1 < STRIPS) ? s + 1 : 0;
#if DCTS_F2_EXP != 1
      if (more) {
#pragma unroll
        for (int it = 0; it < Stage::PIECES; it += 2) Stage::piece_raw(nsrc, nstrip, nxt, launder(lane_in), W, it);
      }
#endif
      int lane = launder(lane_in);
      const bool act = s * SW + lane < N;
      DCTS_STAMP(2);
#if DCTS_F2_EXP != 2
      split_butterflies<M, L, NoHook, false, NW>(buf, SW, act, lane, W);
#endif
      DCTS_STAMP(3);
#if DCTS_F2_EXP != 1
      if (more) {
#pragma unroll
        for (int it = 1; it < Stage::PIECES; it += 2) Stage::piece_raw(nsrc, nstrip, nxt, launder(lane_in), W, it);
      }
#endif
      DCTS_STAMP(2);
      lds_barrier();
      DCTS_STAMP(4);
      lane = launder(lane_in);
      dcts::static_for<2>([&](auto ii) DCTS_LAMBDA_INLINE {
        constexpr int i = decltype(ii)::value;
        const int ln = launder(lane_in);
        float o[M];
        split_role_transform<M, L, 2 * W + i>(buf + (act ? ln : 0), SW, o);
        dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
          constexpr int k = decltype(ik)::value;
          asm volatile("" : "+v"(o[k]));  // pin the codelet here (LLVM would sink it to the dump)
          parked[i][s][k] = o[k];
        });
      });
      DCTS_STAMP(5);
      cur ^= 1;
    });
#endif
    // ---- pass 2: W axis, KPR coefficients of every role per round ---------------------------
    const lds_ptr blk0 = cur ? buf0 : buf1;  // the last strip's buffer; the other one is receiving (staged) / free (register loads)
    const lds_ptr blk1 = cur ? buf1 : buf0;
    // With the samples loaded into registers nothing streams into the second buffer during pass 2: the rounds alternate
    // between the two, and a round's dump need not wait for the readers of the previous round (they use the other buffer;
    // the readers of the round before that are two barriers back): two barriers per round instead of three.
    constexpr bool ALT = DCTS_F2_REGLOAD != 0;
    float e = 0.f;
    dcts::static_for<ROUNDS>([&](auto ir) DCTS_LAMBDA_INLINE {
      constexpr int r = decltype(ir)::value;
      const lds_ptr blk = (ALT && r % 2 == 1) ? blk1 : blk0;
      DCTS_STAMP(11);
      if constexpr (!ALT || r == 0) lds_barrier();  // previous readers of blk are done
      DCTS_STAMP(6);
      int lane = launder(lane_in);
      dcts::static_for<2>([&](auto ii) DCTS_LAMBDA_INLINE {
        constexpr int i = decltype(ii)::value;
        dcts::static_for<STRIPS>([&](auto is) DCTS_LAMBDA_INLINE {
          constexpr int s = decltype(is)::value;
          const int line = s * SW + lane;
          const int off = (line < N ? line : 0) * RW + (2 * W + i) * KPR;
          dcts::static_for<KPR>([&](auto ic) DCTS_LAMBDA_INLINE {
            constexpr int c = decltype(ic)::value;
            if constexpr (r * KPR + c < M) {
              if (line < N) blk[off + c] = parked[i][s][r * KPR + c];
            } else {
              if (line < N) blk[off + c] = 0.f;  // padding column: contributes exactly zero energy
            }
          });
        });
      });
      DCTS_STAMP(7);
#if DCTS_F2_REGLOAD
      if constexpr (r < ITEMS) {  // round r's dump has freed registers: item r of the next map's first strip
        const __amdgpu_buffer_rsrc_t rs = map_rsrc(next_b, more_maps);
        f2_load_item<M, L, W + NW * (r < ITEMS ? r : 0), 0>(rs, lane_voff(0), pre[r < ITEMS ? r : 0]);
      }
#endif
      lds_barrier();
      DCTS_STAMP(8);
      lane = launder(lane_in);
      const bool colact = lane < COLS;
#if DCTS_F2_EXP != 3
      split_butterflies<M, L, NoHook, false, NW>(blk, RW, colact, lane, W);
#endif
      DCTS_STAMP(9);
      lds_barrier();
      DCTS_STAMP(10);
      dcts::static_for<2>([&](auto ii) DCTS_LAMBDA_INLINE {
        constexpr int i = decltype(ii)::value;
        const int ln = launder(lane_in);
        float o[M];
        split_role_transform<M, L, 2 * W + i>(blk + (ln < COLS ? ln : 0), RW, o);
        if constexpr (STORE) {  // see fused_body
          const int q = ln / KPR, kh = r * KPR + (ln - q * KPR);
          if (ln < COLS && kh < M) {
            float* dst = leaf_out + ((long long)m * N + q * M + kh) * N + (2 * W + i) * M;
            dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE { dst[decltype(ik)::value] = o[decltype(ik)::value]; });
          }
        }
        float er = 0.f;
        dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
          constexpr int k = decltype(ik)::value;
          er = fmaf(o[k], o[k], er);
        });
        asm volatile("" : "+v"(er));
        if (ln < COLS) e += er;
      });
      DCTS_STAMP(12);
    });
    e = wave_sum_dpp(e);
    if constexpr (Cfg::DEFER) {
      if (lane_in == 0) partials[pslot * NW + W] = e;
      pending_m = m;
      pending_slot = pslot;
      pslot ^= 1;
    } else {
      // no room for a partials array: it lives behind the image, and the sum is taken right away
      // (the next strip only streams into this buffer after the next top-of-strip barrier)
      lds_barrier();  // every wave has finished reading the image
      const lds_ptr blk = (ALT && (ROUNDS - 1) % 2 == 1) ? blk1 : blk0;  // the last round's buffer
      const lds_ptr part = blk + N * RW;
      if (lane_in == 0) part[W] = e;
      lds_barrier();
      finish(part, 0, m);
    }
    if constexpr (DCTS_F2_REGLOAD != 0 && ROUNDS % 2 == 0) cur ^= 1;  // the next map's first strip must not overwrite the last round's image
  }
  if (pending_m >= 0) {
    lds_barrier();
    finish(partials, pending_slot, pending_m);
  }
#ifdef DCTS_FUSED_STAMPS
  if (lane_in == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_fused_stamps[W][i], acc_[i]);
#endif
}

template <int M, int L, bool STORE, int... Wv>
__device__ __forceinline__ void fused2_dispatch(int wave, const TileBatch& tb, lds_ptr buf0, lds_ptr buf1,
                                                lds_ptr partials, int lane, float* leaf_out,
                                                std::integer_sequence<int, Wv...>) {
  ((wave == Wv ? fused2_body<M, L, Wv, STORE>(tb, buf0, buf1, partials, lane, leaf_out) : (void)0), ...);
}

template <int M, int L>
__global__ __launch_bounds__((64 * Fused2Cfg<M, L>::NW), 2) void k_split_fused2(TileBatch tb) {
  using Cfg = Fused2Cfg<M, L>;
  __shared__ __attribute__((aligned(16))) float buf0[Cfg::BUF];
  __shared__ __attribute__((aligned(16))) float buf1[Cfg::BUF];
  __shared__ float partials[Cfg::DEFER ? 2 * Cfg::NW : 1];
  fused2_dispatch<M, L, false>(threadIdx.x >> 6, tb, (lds_ptr)buf0, (lds_ptr)buf1, (lds_ptr)partials, threadIdx.x & 63,
                               nullptr, std::make_integer_sequence<int, Cfg::NW>{});
}
template <int M, int L>
__global__ __launch_bounds__((64 * Fused2Cfg<M, L>::NW), 2) void k_split_fused2_coeff(TileBatch tb, float* leaf_out) {
  using Cfg = Fused2Cfg<M, L>;
  __shared__ __attribute__((aligned(16))) float buf0[Cfg::BUF];
  __shared__ __attribute__((aligned(16))) float buf1[Cfg::BUF];
  __shared__ float partials[Cfg::DEFER ? 2 * Cfg::NW : 1];
  fused2_dispatch<M, L, true>(threadIdx.x >> 6, tb, (lds_ptr)buf0, (lds_ptr)buf1, (lds_ptr)partials, threadIdx.x & 63,
                              leaf_out, std::make_integer_sequence<int, Cfg::NW>{});
}

// ---------------------------------------------------------------------------------------
// pipelined fused kernel: pass 2 of map m interleaved with pass 1 of map m+1
// ---------------------------------------------------------------------------------------
// The fused kernel above streams a map in (pass 1, HBM-bound), then transforms the parked tile
// (pass 2, no HBM traffic at all): the two halves alternate and neither the memory system nor the
// VALUs are ever busy for more than half of the time. Here one step = pass-2 round r of the
// previous map followed by pass-1 strip r of the current one, so the direct-to-LDS loads of strip
// r+1 are in flight for a whole step (both halves) and the kernel becomes VALU-issue-bound.
// Registers: every round dumps KPR coefficients of every wave's parked rows (the balanced dump),
// which frees exactly the registers the next strip's M outputs need, so the parked set never
// exceeds one tile: slots P[T][T][KPR], T = strips = rounds. A map parked with layout 0 keeps
// T[strip s][coef k] in P[k/KPR][s][k%KPR], layout 1 in P[s][k/KPR][k%KPR]: round r of a
// layout-0 map frees P[r][*][*], which is where strip r of the next map (layout 1) goes, and vice
// versa; maps alternate layouts, the loop body is unrolled over the two parities.
// LDS: two buffers. Step k transforms the staged strip in B[k%2]; the pass-2 image of that step
// lives in B[(k+1)%2], which then receives strip k+1 while B[k%2] is transformed.
template <int M, int L>
struct PipeCfg {
  static constexpr int N = M << L, S = 1 << L, SW = 64;
  static constexpr int T = (N + SW - 1) / SW;
  static constexpr int KPR = 64 / S;
  static constexpr int RW = 65;  // pass-2 image row stride: 64 columns, odd
  static constexpr int BUF = N * RW;
  static_assert((M + KPR - 1) / KPR == T, "rounds == strips");
  static_assert(S <= 16 && N % 4 == 0, "shape");
  // Register relief: NP of the M outputs of strip s are parked in LDS instead (one dword per thread
  // and value, conflict-free), namely the last NP real coefficients of round s. Those are dumped
  // in step s of the next map, before strip s of that map overwrites them, so one copy is enough.
  static constexpr int LDS_FLOATS = 160 * 1024 / 4 - 2 * S - 64;
  static constexpr int last_real = M - (T - 1) * KPR;  // real coefficients in the last round
  static constexpr int NP_FIT = (LDS_FLOATS - 2 * BUF) / (T * 64 * S);
  static constexpr int NP = NP_FIT < 0 ? 0 : (NP_FIT > 2 ? 2 : NP_FIT) > last_real ? last_real : (NP_FIT > 2 ? 2 : NP_FIT);
  static constexpr int nreal(int r) { return r == T - 1 ? last_real : KPR; }
  // index (0..NP-1) of coefficient k of strip s in the LDS park, or -1 if it stays in a register
  static constexpr int park_index(int s, int k) {
    if (k / KPR != s) return -1;
    const int c = k % KPR, first = nreal(s) - NP;
    return (c >= first && c < nreal(s)) ? c - first : -1;
  }
};

template <int PAR, int SI, int K, int T, int KPR>
__device__ __forceinline__ float& pipe_slot(float (&P)[T][T][KPR]) {
  if constexpr (PAR == 0)
    return P[K / KPR][SI][K % KPR];
  else
    return P[SI][K / KPR][K % KPR];
}

template <int M, int L, int ROLE>
__device__ __forceinline__ void pipe_body(const PlainMaps& tb, lds_ptr buf0, lds_ptr buf1, lds_ptr parkbuf,
                                          lds_ptr partials, int lane_in) {
  using Cfg = PipeCfg<M, L>;
  using Stage = FusedStage<M, L>;
  constexpr int N = Cfg::N, S = Cfg::S, SW = Cfg::SW, T = Cfg::T, KPR = Cfg::KPR, RW = Cfg::RW, BUF = Cfg::BUF;
  float P[T][T][KPR];
  const lds_ptr park = parkbuf + (ROLE * 64 + lane_in);  // [T * NP][64 * S]
  long long m_cur = blockIdx.x, m_prev = -1, pending_m = -1;
  int pslot = 0, pending_slot = 0;
#ifdef DCTS_FUSED_STAMPS
  unsigned long long acc_[16] = {}, last_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
  const long long nmaps = tb.total;
  if (m_cur < nmaps) {
    const float* first = tile_in(tb, m_cur);
#pragma unroll
    for (int it = 0; it < Stage::PIECES; ++it) Stage::piece_raw(first, 0, buf0, lane_in, ROLE, it);
  }

  // The direct-to-LDS loads of a strip are issued in four instalments spread over one whole step
  // (pass-1 butterflies and transform of the previous strip, then the dump and the butterflies of
  // the following pass-2 round): a wave stalls on such an instruction while the CU's memory queue
  // is full, and with all of a strip's loads in one phase every wave sat out that stall at the
  // phase's barrier (stamps: 4.8k of a step's 13k cycles) while the queue idled in the other four.
  auto issue = [&](const float* src, int strip, lds_ptr buf, auto slot) DCTS_LAMBDA_INLINE {
#pragma unroll
    for (int it = decltype(slot)::value; it < Stage::PIECES; it += 4)
      Stage::piece_raw(src, strip, buf, launder(lane_in), ROLE, it);
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  using Q2 = std::integral_constant<int, 2>;
  using Q3 = std::integral_constant<int, 3>;
  auto iteration = [&](auto par, auto hp, auto hc) DCTS_LAMBDA_INLINE {
    constexpr int PAR = decltype(par)::value;  // layout of the previous map; the current one gets 1 - PAR
    constexpr bool have_prev = decltype(hp)::value, have_cur = decltype(hc)::value;
    const float* in_b = tile_in(tb, have_cur ? m_cur : 0);
    float e = 0.f;
    dcts::static_for<T>([&](auto ir) DCTS_LAMBDA_INLINE {
      constexpr int r = decltype(ir)::value;
      constexpr int k = PAR * T + r;
      // The two buffers are separate __shared__ objects, statically selected: that is what lets the
      // compiler see that LDS reads of one do not alias direct-to-LDS loads in flight to the other.
      // With one array and offsets it put s_waitcnt vmcnt(0) in front of the first LDS access after
      // every such load: a full memory round trip per instalment.
      const lds_ptr dat = (k % 2) ? buf1 : buf0;  // strip r of the current map (landing / landed)
      const lds_ptr img = (k % 2) ? buf0 : buf1;  // pass-2 image of this step, then strip r+1
      if constexpr (have_prev) {
        DCTS_STAMP(12);
        lds_barrier();  // the strip that lived in img has been consumed by everyone
        DCTS_STAMP(0);
        if constexpr (r == 0) {
          if (pending_m >= 0) {
            fused_finish<M, L, ROLE>(partials, pending_slot, pending_m, tb, lane_in);
            pending_m = -1;
          }
        }
        if constexpr (have_cur) issue(in_b, r, dat, Q2{});
        int lane = launder(lane_in);
        dcts::static_for<T>([&](auto is) DCTS_LAMBDA_INLINE {
          constexpr int s = decltype(is)::value;
          const int line = s * SW + lane;
          const int off = (line < N ? line : 0) * RW + ROLE * KPR;
          dcts::static_for<KPR>([&](auto ic) DCTS_LAMBDA_INLINE {
            constexpr int c = decltype(ic)::value;
            if constexpr (r * KPR + c < M) {
              constexpr int pi = Cfg::park_index(s, r * KPR + c);
              if constexpr (pi >= 0) {
                const float v = park[(s * Cfg::NP + pi) * (64 * S)];
                if (line < N) img[off + c] = v;
              } else {
                if (line < N) img[off + c] = pipe_slot<PAR, s, r * KPR + c>(P);
              }
            } else {
              if (line < N) img[off + c] = 0.f;  // padding column: contributes exactly zero energy
            }
          });
        });
        DCTS_STAMP(1);
        lds_barrier();
        DCTS_STAMP(2);
        if constexpr (have_cur) issue(in_b, r, dat, Q3{});
        lane = launder(lane_in);
        split_butterflies<M, L>(img, RW, true, lane, ROLE);
        DCTS_STAMP(3);
        lds_barrier();
        DCTS_STAMP(4);
        lane = launder(lane_in);
        float o[M];
        split_role_transform<M, L, ROLE>(img + lane, RW, o);
        float er = 0.f;
        dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
          constexpr int kk = decltype(ik)::value;
          er = fmaf(o[kk], o[kk], er);
        });
        asm volatile("" : "+v"(er));
        e += er;
        DCTS_STAMP(5);
      }
      if constexpr (have_cur) {
        DCTS_STAMP(12);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of this strip have landed
        DCTS_STAMP(6);
        lds_barrier();                                   // ... everyone's; img has been consumed
        DCTS_STAMP(7);
        const bool more = (r + 1 < T) || (m_cur + gridDim.x < nmaps);
        const float* nsrc = (r + 1 < T || !more) ? in_b : tile_in(tb, m_cur + gridDim.x);
        constexpr int nstrip = (r + 1 < T) ? r + 1 : 0;
        // the step that transforms the next strip starts with a pass-2 round (which issues the other
        // two instalments) unless this is the first map of the workgroup
        constexpr bool next_has_p2 = have_prev || (r + 1 == T);
        if (more) {
          issue(nsrc, nstrip, img, Q0{});
          if constexpr (!next_has_p2) issue(nsrc, nstrip, img, Q2{});
        }
        int lane = launder(lane_in);
        const bool act = r * SW + lane < N;
        split_butterflies<M, L>(dat, SW, act, lane, ROLE);
        DCTS_STAMP(8);
        lds_barrier();
        DCTS_STAMP(9);
        if (more) {
          issue(nsrc, nstrip, img, Q1{});
          if constexpr (!next_has_p2) issue(nsrc, nstrip, img, Q3{});
        }
        lane = launder(lane_in);
        float o[M];
        split_role_transform<M, L, ROLE>(dat + (act ? lane : 0), SW, o);
        dcts::static_for<M>([&](auto ik) DCTS_LAMBDA_INLINE {
          constexpr int kk = decltype(ik)::value;
          // pin the codelet here: LLVM otherwise sinks its arithmetic down to the dump one map later
          // (the first use of the outputs) and keeps the inputs and half-finished temporaries alive
          constexpr int pi = Cfg::park_index(r, kk);
          if constexpr (pi >= 0) {
            park[(r * Cfg::NP + pi) * (64 * S)] = o[kk];
          } else {
            asm volatile("" : "+v"(o[kk]));
            pipe_slot<1 - PAR, r, kk>(P) = o[kk];
          }
        });
        DCTS_STAMP(10);
      }
    });
    if constexpr (have_prev) {
      e = wave_sum_dpp(e);
      if (lane_in == 0) partials[pslot * S + ROLE] = e;
      pending_m = m_prev;
      pending_slot = pslot;
      pslot ^= 1;
      DCTS_STAMP(11);
    }
    m_prev = have_cur ? m_cur : -1;
    m_cur += gridDim.x;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // every workgroup owns at least one map (grid <= nmaps): prologue, steady pairs, epilogue
  iteration(I0{}, std::false_type{}, std::true_type{});
  for (;;) {
    if (m_cur >= nmaps) {
      iteration(I1{}, std::true_type{}, std::false_type{});
      break;
    }
    iteration(I1{}, std::true_type{}, std::true_type{});
    if (m_cur >= nmaps) {
      iteration(I0{}, std::true_type{}, std::false_type{});
      break;
    }
    iteration(I0{}, std::true_type{}, std::true_type{});
  }
  if (pending_m >= 0) {
    lds_barrier();
    fused_finish<M, L, ROLE>(partials, pending_slot, pending_m, tb, lane_in);
  }
#ifdef DCTS_FUSED_STAMPS
  if (lane_in == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_fused_stamps[ROLE][i], acc_[i]);
#endif
}

template <int M, int L, int... R>
__device__ __forceinline__ void pipe_dispatch(int role, const PlainMaps& tb, lds_ptr buf0, lds_ptr buf1, lds_ptr park,
                                              lds_ptr partials, int lane, std::integer_sequence<int, R...>) {
  ((role == R ? pipe_body<M, L, R>(tb, buf0, buf1, park, partials, lane) : (void)0), ...);
}

template <int M, int L>
__global__ __launch_bounds__((64 << L), (fused_waves_per_simd<M, L>())) void k_split_pipe(const float* __restrict__ x, long long map_stride,
                                                             long long nmaps, float* __restrict__ out) {
  const PlainMaps tb{x, out, map_stride, nmaps};
  using Cfg = PipeCfg<M, L>;
  __shared__ __attribute__((aligned(16))) float buf0[Cfg::BUF];
  __shared__ __attribute__((aligned(16))) float buf1[Cfg::BUF];
  __shared__ float park[Cfg::T * Cfg::NP * 64 * Cfg::S > 0 ? Cfg::T * Cfg::NP * 64 * Cfg::S : 1];
  __shared__ float partials[2 * Cfg::S];
  pipe_dispatch<M, L>(threadIdx.x >> 6, tb, (lds_ptr)buf0, (lds_ptr)buf1, (lds_ptr)park, (lds_ptr)partials,
                      threadIdx.x & 63, std::make_integer_sequence<int, Cfg::S>{});
}

// out[b] = scale * sum of the map's ROLES*STRIPS partials, fixed order
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

// Batch sum over n of E[n][j] for a 32-channel strip per block: kSumSl = 16 n-slices run in parallel
// (slice s takes n = s, s+16, ...), partials are combined in slice order -> a fixed,
// launch-independent summation order (bit-reproducible, no atomics).
constexpr int kSumCh = 32, kSumSl = 16;
__device__ __forceinline__ float strip_batch_sum(const float* __restrict__ e, long long N,
                                                 long long C, long long j, int slice,
                                                 float (*part)[kSumCh]) {
  float s = 0.f;
  if (j < C) {
    long long n = slice;
    // sixteen loads in flight per lane and round trip (a batch of 256 samples is ONE round trip: the
    // kernel is pure latency, 3.8 us with four loads per trip); the additions keep their order
#pragma unroll 1
    for (; n + 15 * kSumSl < N; n += 16 * kSumSl) {
      float a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = e[(n + i * kSumSl) * C + j];
#pragma unroll
      for (int i = 0; i < 16; ++i) s += a[i];
    }
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

// Score variant in the coefficient domain: out[m] = sum_{u,v} weights[u,v] * coeff[m][u][v]^2.
// One wave per map over dense [HW] coefficient tiles; lanes stride the tile, fixed-order wave sum.
__global__ __launch_bounds__(256) void k_weighted_energy(const float* __restrict__ coeff, const float* __restrict__ weights,
                                                         long long nmaps, int hw, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long m = wave; m < nmaps; m += nwaves) {
    const float* c = coeff + m * hw;
    float e = 0.f;
    for (int i = lane; i < hw; i += 64) {
      const float v = c[i];
      e = fmaf(weights[i] * v, v, e);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e += __shfl_down(e, off, 64);
    if (lane == 0) out[m] = e;
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
// compute units of the current device, queried once (256 on MI355X; the persistent grids are sized by it)
inline int num_cus() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 1)
      cu = 256;
    return cu;
  }();
  return n;
}

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
  const long long cap = (long long)num_cus() * Cfg::GRID_WAVES_PER_CU / Cfg::WAVES;
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
    const long long cap = (long long)num_cus() * per_cu;
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

}  // namespace
#if DCTS_PART(1)
namespace dctsi {
int dispatch_codelet_dma(int N, const void* geom, float* out, hipStream_t st) {
  const MapGeom& g = *static_cast<const MapGeom*>(geom);

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
}  // namespace dctsi
#endif
namespace {
inline int dispatch_codelet_dma(int N, const MapGeom& g, float* out, hipStream_t st) {
  return dctsi::dispatch_codelet_dma(N, &g, out, st);
}

template <bool STORE>
int dispatch_codelet_impl(int HP, int WP, int pad, const MapGeom& g, float* out, hipStream_t st) {
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
}  // namespace
#if DCTS_PART(1)
namespace dctsi {
int dispatch_codelet(int store, int HP, int WP, int pad, const void* geom, float* out, hipStream_t st) {
  const MapGeom& g = *static_cast<const MapGeom*>(geom);
  return store ? dispatch_codelet_impl<true>(HP, WP, pad, g, out, st) : dispatch_codelet_impl<false>(HP, WP, pad, g, out, st);
}
}  // namespace dctsi
#endif
namespace {
template <bool STORE>
inline int dispatch_codelet(int HP, int WP, int pad, const MapGeom& g, float* out, hipStream_t st) {
  return dctsi::dispatch_codelet(STORE ? 1 : 0, HP, WP, pad, &g, out, st);
}

// tile edges served by the split family: X(N, M, L) with N = M << L. L = 3 (eight M-point roles)
// where the four-role codelets would be too register-hungry for more than 1-2 waves per SIMD.
#ifndef DCTS_SPLIT_TABLE
// Round 3: the other multiples of 4 up to 256 (N = 4 * M) and of 8 up to 512 (N = 8 * M) with M <= 64 even or M <= 32 - the
// codelet template factorises any M; an odd M beyond 32 would be a direct M x M sum per leaf: minutes of build time each and
// compute-bound. These edges (an --input_size such as 272 / 304 / 352 / 384 / 448 / 512 and their halves) have no single-launch
// kernel: they take the two-launch path (3 x the algorithmic traffic: <= 0.2 of the HBM peak) instead of the cosine-matrix kernel
// (< 0.01). The 8 * M entries are a translation unit of their own (DCTS_TU=7).
#define DCTS_SPLIT_TABLE_MORE_A(X) \
  X(68, 17, 2) X(76, 19, 2) X(84, 21, 2) X(88, 22, 2) X(92, 23, 2) X(100, 25, 2) X(104, 26, 2) \
  X(108, 27, 2) X(116, 29, 2) X(120, 30, 2) X(124, 31, 2) X(136, 34, 2) X(152, 38, 2) X(168, 42, 2) \
  X(176, 44, 2) X(184, 46, 2) X(200, 50, 2) X(208, 52, 2) X(216, 54, 2) X(232, 58, 2) X(240, 60, 2) \
  X(248, 62, 2)
#define DCTS_SPLIT_TABLE_MORE_B(X) \
  X(272, 34, 3) X(304, 38, 3) X(336, 42, 3) X(352, 44, 3) X(368, 46, 3) X(384, 48, 3) X(400, 50, 3) \
  X(416, 52, 3) X(432, 54, 3) X(448, 56, 3) X(464, 58, 3) X(480, 60, 3) X(496, 62, 3) X(512, 64, 3)
#define DCTS_SPLIT_TABLE_MORE(X) DCTS_SPLIT_TABLE_MORE_A(X) DCTS_SPLIT_TABLE_MORE_B(X)
#define DCTS_SPLIT_TABLE_BASE(X)                                                         \
  X(72, 18, 2) X(80, 20, 2) X(96, 24, 2) X(112, 28, 2) X(128, 32, 2) X(144, 36, 2) X(160, 40, 2)      \
  X(192, 24, 3) X(224, 28, 3) X(256, 32, 3) X(288, 36, 3) X(320, 40, 3)
#define DCTS_SPLIT_TABLE(X) DCTS_SPLIT_TABLE_BASE(X) DCTS_SPLIT_TABLE_MORE(X)
#endif

bool has_split(long long HP, long long WP) {
  if (HP != WP) return false;
#define DCTS_CASE(N_, M_, L_) \
  if (HP == N_) return true;
  DCTS_SPLIT_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

int split_partials_per_map(int N) {
#define DCTS_CASE(N_, M_, L_) \
  if (N == N_) return SplitCfg<M_, L_>::STRIPS * SplitCfg<M_, L_>::ROLES;
  DCTS_SPLIT_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return 0;
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
  w.off_t = 0;
  w.off_part = align_up((size_t)chunk * map_bytes, 256);
  w.total = align_up(w.off_part + (size_t)chunk * split_partials_per_map(N) * 4, 256);
  return w;
}

template <int M, int L>
int launch_split(const MapGeom& g, float* out, void* workspace, hipStream_t st) {
  using Cfg = SplitCfg<M, L>;
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
    hipLaunchKernelGGL((k_pass1d<M, L, false>), dim3(grid), dim3(64 << L), 0, st, x0 + m0 * g.strideC,
                       g.strideC, T, part);
    hipLaunchKernelGGL((k_pass1d<M, L, true>), dim3(grid), dim3(64 << L), 0, st, T, (long long)N * N,
                       (float*)nullptr, part);
    hipLaunchKernelGGL(k_split_reduce, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, part,
                       Cfg::STRIPS * Cfg::ROLES, nb, scale, out + m0);
  }
  return (int)hipGetLastError();
}

// tiles whose intermediate fits the register file of one CU: single fused launch X(N, M, L).
// (M, L) per edge is the fastest measured factorisation (e.g. 224: 28x8 roles 19 %, 14x16 roles 26 %;
// 128: 16x8 38 %, 32x4 35 %, 8x16 22 %; 288 = 18x16 spills at 128 VGPRs and loses to two launches)
#ifndef DCTS_FUSED_TABLE
#define DCTS_FUSED_TABLE(X) \
  X(72, 9, 3) X(80, 10, 3) X(96, 12, 3) X(112, 14, 3) X(128, 16, 3) X(144, 18, 3) X(160, 10, 4) X(192, 12, 4) X(224, 14, 4) \
  X(256, 16, 4)
#endif

bool has_fused(long long N) {
#define DCTS_CASE(N_, M_, L_) \
  if (N == N_) return true;
  DCTS_FUSED_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

template <int M, int L>
int launch_fused(const TileBatch& tb, hipStream_t st) {
  // persistent grid: exactly the workgroups one residency holds (LDS- or register-limited)
  static const int per_cu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_split_fused<M, L>, 64 << L, 0) != hipSuccess || n < 1)
      n = 1;
    return n;
  }();
  const long long cap = (long long)num_cus() * per_cu;
  const long long grid = tb.total < cap ? tb.total : cap;
  hipLaunchKernelGGL((k_split_fused<M, L>), dim3((unsigned)grid), dim3(64 << L), 0, st, tb);
  return (int)hipGetLastError();
}

template <class Kernel, class Assemble>
int run_coeff_chunks(Kernel kernel, Assemble assemble, int N, int threads, const float* x, long long nmaps, float* out,
                     float* scratch, long long scratch_maps, hipStream_t st) {
  if (!scratch || scratch_maps < 1) return DCTS_E_WORKSPACE;
  for (long long m0 = 0; m0 < nmaps; m0 += scratch_maps) {
    const long long nb = (nmaps - m0) < scratch_maps ? (nmaps - m0) : scratch_maps;
    TileBatch tb;
    for (int i = 0; i < kTileItems; ++i) {
      tb.x[i] = x + m0 * (long long)N * N;
      tb.out[i] = nullptr;  // the coefficient instantiations write no energies
      tb.begin[i] = 0;
    }
    tb.begin[1] = tb.begin[kTileItems] = nb;
    tb.map_elems = (long long)N * N;
    tb.total = nb;
    tb.count = 1;
    const long long grid = nb < num_cus() ? nb : num_cus();
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), 0, st, tb, scratch);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    rc = assemble(scratch, nb, out + m0 * (long long)N * N, st);
    if (rc) return rc;
  }
  return DCTS_OK;
}

}  // namespace
#if DCTS_PART(3)
namespace dctsi {
int dispatch_fused_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps,
                         hipStream_t st) {
#define DCTS_CASE(N_, M_, L_)                                                                                        \
  case N_:                                                                                                           \
    return run_coeff_chunks(k_split_fused_coeff<M_, L_>, launch_assemble<M_, L_, true>, N_, 64 << L_, x, nmaps, out, \
                            scratch, scratch_maps, st);
  switch (N) {
    DCTS_FUSED_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
int dispatch_fused(int N, const void* tile_batch, hipStream_t st) {
  const TileBatch& tb = *static_cast<const TileBatch*>(tile_batch);
#define DCTS_CASE(N_, M_, L_) \
  case N_:                    \
    return launch_fused<M_, L_>(tb, st);
  switch (N) {
    DCTS_FUSED_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
}  // namespace dctsi
#endif
namespace {
inline int dispatch_fused(int N, const TileBatch& tb, hipStream_t st) { return dctsi::dispatch_fused(N, &tb, st); }

// two roles per wave X(N, M, L): tiles the 16-wave kernels cannot park (320 runs 48-column rounds
// so that its two LDS buffers fit the 160 KiB exactly, see Fused2Cfg)
#ifndef DCTS_FUSED2_TABLE
#define DCTS_FUSED2_TABLE(X) X(288, 18, 4) X(320, 20, 4)
#endif

bool has_fused2(long long N) {
#define DCTS_CASE(N_, M_, L_) \
  if (N == N_) return true;
  DCTS_FUSED2_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

template <int M, int L>
int launch_fused2(const TileBatch& tb, hipStream_t st) {
  const long long cap = num_cus();  // LDS: one workgroup per CU
  const long long grid = tb.total < cap ? tb.total : cap;
  hipLaunchKernelGGL((k_split_fused2<M, L>), dim3((unsigned)grid), dim3(64 * Fused2Cfg<M, L>::NW), 0, st, tb);
  return (int)hipGetLastError();
}
}  // namespace
#if DCTS_PART(6)
namespace dctsi {
int dispatch_fused2_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps,
                          hipStream_t st) {
#define DCTS_CASE(N_, M_, L_)                                                                                      \
  case N_:                                                                                                         \
    return run_coeff_chunks(k_split_fused2_coeff<M_, L_>, launch_assemble<M_, L_, true>, N_,                       \
                            64 * Fused2Cfg<M_, L_>::NW, x, nmaps, out, scratch, scratch_maps, st);
  switch (N) {
    DCTS_FUSED2_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
int dispatch_fused2(int N, const void* tile_batch, hipStream_t st) {
  const TileBatch& tb = *static_cast<const TileBatch*>(tile_batch);
#define DCTS_CASE(N_, M_, L_) \
  case N_:                    \
    return launch_fused2<M_, L_>(tb, st);
  switch (N) {
    DCTS_FUSED2_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
}  // namespace dctsi
#endif
namespace {

// pipelined variant X(N, M, L)
#ifndef DCTS_PIPE_TABLE
// (measured against the fused kernel, % of 8 TB/s: 128: 49.2 vs 46.9, 224: 36.8 vs 33.1; it loses
// where the balanced dump pads much (72, 80, 144, 160) or spills (256), and ties at 112)
#define DCTS_PIPE_TABLE(X) X(128, 16, 3) X(224, 14, 4)
#endif

bool has_pipe(long long N) {
#define DCTS_CASE(N_, M_, L_) \
  if (N == N_) return true;
  DCTS_PIPE_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

template <int M, int L>
int launch_pipe(const TileBatch& tb, hipStream_t st) {  // one tensor per launch (PlainMaps)
  static const int per_cu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_split_pipe<M, L>, 64 << L, 0) != hipSuccess || n < 1)
      n = 1;
    return n;
  }();
  const long long cap = (long long)num_cus() * per_cu;
  int rc = 0;
  for (int i = 0; i < tb.count && !rc; ++i) {
    const long long nm = tb.begin[i + 1] - tb.begin[i];
    const long long grid = nm < cap ? nm : cap;
    hipLaunchKernelGGL((k_split_pipe<M, L>), dim3((unsigned)grid), dim3(64 << L), 0, st, tb.x[i], tb.map_elems, nm,
                       tb.out[i]);
    rc = (int)hipGetLastError();
  }
  return rc;
}

}  // namespace
#if DCTS_PART(4)
namespace dctsi {
int dispatch_pipe(int N, const void* tile_batch, hipStream_t st) {
  const TileBatch& tb = *static_cast<const TileBatch*>(tile_batch);
#define DCTS_CASE(N_, M_, L_) \
  case N_:                    \
    return launch_pipe<M_, L_>(tb, st);
  switch (N) {
    DCTS_PIPE_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
}  // namespace dctsi
#endif
namespace {
inline int dispatch_pipe(int N, const TileBatch& tb, hipStream_t st) { return dctsi::dispatch_pipe(N, &tb, st); }

}  // namespace
#if DCTS_PART(2)
namespace dctsi {
int dispatch_split(int N, const void* geom, float* out, void* workspace, hipStream_t st) {
  const MapGeom& g = *static_cast<const MapGeom*>(geom);

#define DCTS_CASE(N_, M_, L_) \
  case N_:                    \
    return launch_split<M_, L_>(g, out, workspace, st);
  switch (N) {
#ifdef DCTS_SPLIT_TABLE_MORE_A
    DCTS_SPLIT_TABLE_BASE(DCTS_CASE)
    DCTS_SPLIT_TABLE_MORE_A(DCTS_CASE)
    default:
      return dispatch_split_more(N, geom, out, workspace, st);
#else
    DCTS_SPLIT_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
#endif
  }
#undef DCTS_CASE
}
}  // namespace dctsi
#endif
#if DCTS_PART(7) && defined(DCTS_SPLIT_TABLE_MORE_B)
namespace dctsi {
int dispatch_split_more(int N, const void* geom, float* out, void* workspace, hipStream_t st) {
  const MapGeom& g = *static_cast<const MapGeom*>(geom);
#define DCTS_CASE(N_, M_, L_) \
  case N_:                    \
    return launch_split<M_, L_>(g, out, workspace, st);
  switch (N) {
    DCTS_SPLIT_TABLE_MORE_B(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
}  // namespace dctsi
#endif
namespace {
inline int dispatch_split(int N, const MapGeom& g, float* out, void* workspace, hipStream_t st) {
  return dctsi::dispatch_split(N, &g, out, workspace, st);
}

template <int HP, int WP, int PAD>
int launch_codelet_multi(const MultiGeom& mg, hipStream_t st) {
  using Cfg = CodeletCfg<HP, WP>;
  long long blocks = (mg.total_groups + Cfg::WAVES - 1) / Cfg::WAVES;
  const long long cap = (long long)num_cus() * Cfg::GRID_WAVES_PER_CU / Cfg::WAVES;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_energy_codelet_multi<HP, WP, PAD>), dim3((unsigned)blocks), dim3(64 * Cfg::WAVES), 0, st,
                     mg);
  return (int)hipGetLastError();
}

constexpr bool has_lane_kernel(int n) { return n == 7 || n == 9; }

template <int N>
int launch_lane(const MultiGeom& mg, hipStream_t st) {
  using Cfg = LaneCfg<N>;
  static const int per_cu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_energy_lane_multi<N>, 64 * Cfg::WAVES, 0) != hipSuccess || n < 1)
      n = 1;
    return n;
  }();
  long long blocks = (mg.total_groups + Cfg::WAVES - 1) / Cfg::WAVES;
  const long long cap = (long long)num_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_energy_lane_multi<N>), dim3((unsigned)blocks), dim3(64 * Cfg::WAVES), 0, st, mg);
  return (int)hipGetLastError();
}

}  // namespace
#if DCTS_PART(1)
namespace dctsi {
int dispatch_codelet_mixed(const void* mixed_geom, hipStream_t st) {
  const MixedGeom& mg = *static_cast<const MixedGeom*>(mixed_geom);
  static const int per_cu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_energy_codelet_mixed, 64 * kMixedWaves, 0) != hipSuccess || n < 1)
      n = 1;
    return n;
  }();
  long long blocks = (mg.total_groups + kMixedWaves - 1) / kMixedWaves;
  const long long cap = (long long)num_cus() * per_cu;  // one residency of persistent waves
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_energy_codelet_mixed, dim3((unsigned)blocks), dim3(64 * kMixedWaves), 0, st, mg);
  return (int)hipGetLastError();
}
int dispatch_lane(int n, const void* multi_geom, hipStream_t st) {
  const MultiGeom& mg = *static_cast<const MultiGeom*>(multi_geom);

  switch (n) {
    case 7:
      return launch_lane<7>(mg, st);
    case 9:
      return launch_lane<9>(mg, st);
    default:
      return DCTS_E_UNSUPPORTED;
  }
}
}  // namespace dctsi
#endif
namespace {
inline int dispatch_lane(int n, const MultiGeom& mg, hipStream_t st) {
  return dctsi::dispatch_lane(n, &mg, st);
}

int codelet_group_size(int HP) {
  if (has_lane_kernel(HP)) return 64;
#define DCTS_CASE(N) \
  if (HP == N) return CodeletCfg<N, N>::G;
  DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return 0;
}

}  // namespace
#if DCTS_PART(1)
namespace dctsi {
int dispatch_codelet_multi(int HP, int pad, const void* multi_geom, hipStream_t st) {
  const MultiGeom& mg = *static_cast<const MultiGeom*>(multi_geom);

  if (has_lane_kernel(HP) && pad == 0) return dctsi::dispatch_lane(HP, &mg, st);
#define DCTS_CASE(N)                                        \
  case N:                                                   \
    if (pad) {                                              \
      if constexpr ((N % 2) == 0 && N >= 2)                 \
        return launch_codelet_multi<N, N, 1>(mg, st);       \
      else                                                  \
        return DCTS_E_UNSUPPORTED;                          \
    }                                                       \
    return launch_codelet_multi<N, N, 0>(mg, st);
  switch (HP) {
    DCTS_CODELET_SIZES(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
}  // namespace dctsi
#endif
namespace {
inline int dispatch_codelet_multi(int HP, int pad, const MultiGeom& mg, hipStream_t st) {
  return dctsi::dispatch_codelet_multi(HP, pad, &mg, st);
}

bool has_codelet(long long HP, long long WP) {
  if (HP != WP) return false;
#define DCTS_CASE(N) \
  if (HP == N) return true;
  DCTS_CODELET_SIZES(DCTS_CASE)
#undef DCTS_CASE
  return false;
}

// which single-launch large-tile kernel serves an edge: 0 none, 1 fused, 2 fused with two roles per
// wave, 3 pipelined (AUTO order: pipelined, two-roles, fused)
bool has_tile2d(long long N) { return N == 224 || dctsi::has_tile2g((int)N) != 0; }

// Does AUTO take the several-maps-per-round 2-D split (tile2g.hip, family 5) for `nmaps` maps of edge HP? Same box,
// % of the HBM peak, fused / pipelined kernel -> tile2g: 72: 30.9 -> 44.1 (9645 maps), 29.7 -> 45.1 (32768); 80: 33.2 ->
// 40.3, 32.9 -> 37.6; 144: 30.1 -> 33.7 (2411), 32.0 -> 42.4 (4992), 31.8 -> 41.0 (8192); 160: 31.2 -> 34.1 (1953), 33.5 ->
// 37.3 (4096); 128: 45.1 -> 43.3 (3051) but 44.1 -> 50.6 (8192); 112: 38.8 -> 34.6, 39.8 -> 38.2
// (profiles/r03_tile2g_vs_fused_same_box.txt). The choice must not depend on the map count: dcts_energy_multi_f32
// promises the bits of one call per tensor, whatever the tensors' sizes. So 72, 80, 144, 160 take it, 112 and 128
// keep the fused / pipelined kernels (DCTS_ALGO_TILE2D still selects it for them).
bool tile2g_auto(int HP, long long /*nmaps*/) {
  return dctsi::has_tile2g(HP) && HP != 96 && HP != 112 && HP != 128;
}

int tile_family(int HP, int algo, long long nmaps) {
  if (algo == DCTS_ALGO_TILE2D) return HP == 224 ? 4 : (dctsi::has_tile2g(HP) ? 5 : 0);
  if (algo == DCTS_ALGO_AUTO && tile2g_auto(HP, nmaps)) return 5;
  // AUTO order: 2-D split (tile2d.hip), pipelined, two-roles, fused. 224: 2-D split 33-42 % of the HBM
  // peak against 31-37 % pipelined, same box, 996...16384 maps
  if (algo == DCTS_ALGO_AUTO && HP == 224) return 4;
  if (algo == DCTS_ALGO_PIPE) return has_pipe(HP) && has_fused(HP) ? 3 : 0;
  if (algo == DCTS_ALGO_AUTO && has_pipe(HP) && has_fused(HP)) return 3;
  if (has_fused2(HP) && (algo == DCTS_ALGO_FUSED || DCTS_FUSED2_AUTO)) return 2;
  if (has_fused(HP)) return 1;
  return 0;
}
int dispatch_tile_family(int fam, int HP, const TileBatch& tb, hipStream_t st) {
  switch (fam) {
    case 6:
      return dctsi::dispatch_tile2g_pad(HP, &tb, st);
    case 5:
      return dctsi::dispatch_tile2g(HP, &tb, st);
    case 4:
      return dctsi::dispatch_tile2d(HP, &tb, st);
    case 3:
      return dispatch_pipe(HP, tb, st);
    case 2:
      return dctsi::dispatch_fused2(HP, &tb, st);
    case 1:
      return dispatch_fused(HP, tb, st);
    default:
      return DCTS_E_UNSUPPORTED;
  }
}

// The direct kernel's basis tables live at the head of the caller's workspace. They are built once per
// (workspace, stream, H', W') and reused by later calls: the library remembers - on the host, nothing is read
// back - which BYTE RANGE of which workspace holds tables, and forgets an entry whenever any of its own paths
// is about to write bytes that overlap that range (another shape's tables, the direct kernel's T tiles, the
// split path's intermediate, the coefficient path's leaf outputs, the weighted path's coefficient chunk) or the
// caller says so (dcts_workspace_invalidate[_range]). Same stream only: that is what orders the build before
// the reuse. Calls that receive an INTERIOR pointer of a caller's workspace (the weighted path's inner calls)
// never cache: an interior offset depends on the tile shape, and the caller cannot name it to invalidate it.
struct BasisSlot {
  uintptr_t lo, hi;  // bytes [lo, hi) hold the two tables
  void* ws;          // the workspace pointer the call was made with
  void* stream;
  int HP, WP;
};
constexpr int kBasisSlots = 16;
BasisSlot g_basis[kBasisSlots] = {};
int g_basis_next = 0;
std::mutex g_basis_mu;

bool basis_cached(void* ws, void* stream, int HP, int WP) {
  std::lock_guard<std::mutex> lk(g_basis_mu);
  for (const BasisSlot& b : g_basis)
    if (b.hi && b.ws == ws && b.stream == stream && b.HP == HP && b.WP == WP) return true;
  return false;
}
// forget every entry whose tables overlap [p, p + bytes)
void basis_forget_range(const void* p, size_t bytes) {
  if (!p || !bytes) return;
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
  std::lock_guard<std::mutex> lk(g_basis_mu);
  for (BasisSlot& b : g_basis)
    if (b.hi && b.lo < hi && lo < b.hi) b = BasisSlot{};
}
// the caller names a workspace by its base pointer only: forget what was cached under that pointer and
// whatever tables contain that address
void basis_forget(void* ws) {
  if (!ws) return;
  const uintptr_t a = reinterpret_cast<uintptr_t>(ws);
  std::lock_guard<std::mutex> lk(g_basis_mu);
  for (BasisSlot& b : g_basis)
    if (b.hi && (b.ws == ws || (b.lo <= a && a < b.hi))) b = BasisSlot{};
}
void basis_remember(void* ws, const void* tables, size_t table_bytes, void* stream, int HP, int WP) {
  const uintptr_t lo = reinterpret_cast<uintptr_t>(tables), hi = lo + table_bytes;
  std::lock_guard<std::mutex> lk(g_basis_mu);
  for (BasisSlot& b : g_basis)
    if (b.hi && (b.ws == ws || (b.lo < hi && lo < b.hi))) b = BasisSlot{};  // one shape per workspace, no overlapping tables
  g_basis[g_basis_next] = BasisSlot{lo, hi, ws, stream, HP, WP};
  g_basis_next = (g_basis_next + 1) % kBasisSlots;
}

template <bool STORE>
int run(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W, int64_t strideN,
        int64_t strideC, int64_t strideH, int64_t strideW, int32_t c_begin, int32_t c_count,
        int32_t pad_front_if_odd, float* out, void* workspace, size_t workspace_bytes,
        void* stream, int32_t algo, bool cache_basis = true) {
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
      algo != DCTS_ALGO_SPLIT && algo != DCTS_ALGO_PREFETCH && algo != DCTS_ALGO_FUSED && algo != DCTS_ALGO_PIPE && algo != DCTS_ALGO_LANE && algo != DCTS_ALGO_TILE2D && algo != DCTS_ALGO_RECT)
    return DCTS_E_UNSUPPORTED;
  if (algo == DCTS_ALGO_LANE && !(codelet_ok && !STORE && pad == 0 && has_lane_kernel((int)HP))) return DCTS_E_UNSUPPORTED;
  // both edges have a 1-D codelet, but the maps are not square or their rows not dense: the run-time pair of codelets
  // (rect.hip). Square dense maps keep their own kernels unless ALGO_RECT asks (tests compare the two).
  const bool rect_ok = HP <= 64 && WP <= 64 && dctsi::has_rect((int)HP, (int)WP) != 0;
  if (algo == DCTS_ALGO_RECT && !rect_ok) return DCTS_E_UNSUPPORTED;
  if (rect_ok && (algo == DCTS_ALGO_RECT || (algo == DCTS_ALGO_AUTO && !codelet_ok))) {
    dctsi::RectGeom r{};
    r.x = x;
    r.nmaps = g.nmaps;
    r.strideN = strideN;
    r.strideC = strideC;
    r.strideH = strideH;
    r.c_count = c_count;
    r.c_begin = c_begin;
    r.H = (int)H;
    r.W = (int)W;
    r.HP = (int)HP;
    r.WP = (int)WP;
    r.pad = pad;
    r.contiguous = g.contiguous;
    return dctsi::dispatch_rect(r, out, STORE ? 1 : 0, st);
  }
  if (codelet_ok && algo != DCTS_ALGO_DIRECT) {
    if constexpr (!STORE) {
      if ((algo == DCTS_ALGO_AUTO || algo == DCTS_ALGO_LANE) && pad == 0 && has_lane_kernel((int)HP)) {
        MultiGeom mg;
        for (int i = 0; i < kMultiItems; ++i) {
          mg.it[i].g = g;
          mg.it[i].out = out;
          mg.it[i].group_begin = 0;
        }
        mg.total_groups = (g.nmaps + 63) / 64;
        mg.count = 1;
        return dispatch_lane((int)HP, mg, st);
      }
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
    // every split kernel stages with 16-byte direct-to-LDS loads: a base that is only 4-byte aligned takes the
    // direct kernel - except where tile2g.hip has the shape: it gathers single dwords and needs neither the alignment
    // nor (for 71 / 79 / 143 / 159: the cv2 path on odd maps) an unpadded tile
    const bool aligned16 = (reinterpret_cast<uintptr_t>(x + (long long)c_begin * strideC) & 15) == 0;
    const bool dense_maps = H == W && strideH == W && g.contiguous && strideC == H * W;
    const bool split_ok = has_split(HP, WP) && pad == 0 && dense_maps && aligned16;
    if (algo == DCTS_ALGO_SPLIT && !split_ok) return DCTS_E_UNSUPPORTED;
    if (algo == DCTS_ALGO_AUTO || algo == DCTS_ALGO_FUSED || algo == DCTS_ALGO_PIPE || algo == DCTS_ALGO_TILE2D) {
      int fam = (split_ok && aligned16) ? tile_family((int)HP, algo, g.nmaps) : 0;
      if (!fam && dense_maps && (algo == DCTS_ALGO_AUTO || algo == DCTS_ALGO_TILE2D || algo == DCTS_ALGO_FUSED)) {
        const bool t2 = algo != DCTS_ALGO_FUSED, fu = algo != DCTS_ALGO_TILE2D;  // an explicit family request is kept
        if (t2 && pad == 0 && !aligned16 && dctsi::has_tile2g((int)HP)) fam = 5;
        if (!fam && fu && pad == 0 && !aligned16 && has_fused2(HP) && DCTS_F2_REGLOAD) fam = 2;  // the two-roles kernel loads dwords into registers
        if (!fam && fu && pad == 0 && !aligned16 && has_fused(HP) && DCTS_F1_REGLOAD) fam = 1;   // so does the fused kernel
        if (t2 && pad == 1 && dctsi::has_tile2g_pad((int)HP)) fam = 6;
      }
      if (fam) {
        TileBatch tb;
        for (int i = 0; i < kTileItems; ++i) {
          tb.x[i] = x + (long long)c_begin * strideC;
          tb.out[i] = out;
          tb.begin[i] = 0;
        }
        tb.begin[1] = tb.begin[kTileItems] = g.nmaps;
        tb.map_elems = strideC;
        tb.total = g.nmaps;
        tb.count = 1;
        return dispatch_tile_family(fam, (int)HP, tb, st);
      }
      if (algo != DCTS_ALGO_AUTO) return DCTS_E_UNSUPPORTED;
    }
    if (split_ok && algo != DCTS_ALGO_DIRECT) {
      const SplitWs sws = split_ws(g.nmaps, (int)HP);
      if (!workspace || workspace_bytes < sws.total) return DCTS_E_WORKSPACE;
      if (reinterpret_cast<uintptr_t>(workspace) & 15) return DCTS_E_ALIGN;  // pass 2 stages the intermediate the same way
      basis_forget_range(workspace, sws.total);
      return dispatch_split((int)HP, g, out, workspace, st);
    }
  } else {
    if (algo == DCTS_ALGO_SPLIT || algo == DCTS_ALGO_PIPE) return DCTS_E_UNSUPPORTED;
    if (algo == DCTS_ALGO_FUSED || algo == DCTS_ALGO_TILE2D) {
      // coefficients through the large-tile kernels themselves (leaf outputs + k_assemble): what the
      // parity tests use to check that those kernels compute the DCT and not merely its energy
      const bool dense = has_split(HP, WP) && pad == 0 && strideH == W && g.contiguous && strideC == H * W &&
                         (reinterpret_cast<uintptr_t>(x + (long long)c_begin * strideC) & 15) == 0;
      if (!dense) return DCTS_E_UNSUPPORTED;
      const long long tile_bytes = (long long)HP * WP * 4;
      const long long ws_maps = workspace ? (long long)(workspace_bytes / (size_t)tile_bytes) : 0;
      if (ws_maps < 1 || (reinterpret_cast<uintptr_t>(workspace) & 15)) return DCTS_E_WORKSPACE;
      const float* x0 = x + (long long)c_begin * strideC;
      float* scratch = reinterpret_cast<float*>(workspace);
      basis_forget_range(workspace, workspace_bytes);
      if (algo == DCTS_ALGO_TILE2D)
        return HP == 224 ? dctsi::dispatch_tile2d_coeff((int)HP, x0, g.nmaps, out, scratch, ws_maps, st)
                         : dctsi::dispatch_tile2g_coeff((int)HP, x0, g.nmaps, out, scratch, ws_maps, st);
      if (has_fused2(HP)) return dctsi::dispatch_fused2_coeff((int)HP, x0, g.nmaps, out, scratch, ws_maps, st);
      return dctsi::dispatch_fused_coeff((int)HP, x0, g.nmaps, out, scratch, ws_maps, st);
    }
  }

  const DirectWs ws = direct_ws(g.nmaps, (int)HP, (int)WP);
  if (!workspace) return ws.total ? DCTS_E_WORKSPACE : DCTS_E_NULL;
  if (workspace_bytes < ws.total) return DCTS_E_WORKSPACE;
  char* wsp = reinterpret_cast<char*>(workspace);
  float* CHt = reinterpret_cast<float*>(wsp + ws.off_ch);
  float* CWt = reinterpret_cast<float*>(wsp + ws.off_cw);
  float* T = reinterpret_cast<float*>(wsp + ws.off_t);
  if (!cache_basis || !basis_cached(workspace, stream, (int)HP, (int)WP)) {
    basis_forget_range(workspace, ws.total);  // whatever tables lay in the bytes this call uses are gone
    hipLaunchKernelGGL(k_basis, dim3((unsigned)((HP * HP + 255) / 256)), dim3(256), 0, st, CHt, (int)HP);
    hipLaunchKernelGGL(k_basis, dim3((unsigned)((WP * WP + 255) / 256)), dim3(256), 0, st, CWt, (int)WP);
    if (cache_basis && hipGetLastError() == hipSuccess)
      basis_remember(workspace, wsp + ws.off_ch, ws.off_t - ws.off_ch, stream, (int)HP, (int)WP);
  } else {
    basis_forget_range(wsp + ws.off_t, ws.total - ws.off_t);  // the T tiles may cover another entry's tables
  }
  hipLaunchKernelGGL((k_energy_direct<STORE>), dim3((unsigned)ws.grid), dim3(kDirectThreads), 0, st,
                     g, pad, CHt, CWt, T, out);
  return (int)hipGetLastError();
}

}  // namespace

#if DCTS_PART(5)
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
    case DCTS_E_ALIGN: return "pointer not 4-byte aligned (tensors) / 16-byte aligned (workspace)";
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

void dcts_workspace_invalidate(void* workspace) { basis_forget(workspace); }

void dcts_workspace_invalidate_range(void* workspace, size_t bytes) {
  basis_forget(workspace);
  basis_forget_range(workspace, bytes);
}

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

size_t dcts_weighted_workspace_bytes(int64_t N, int64_t C_count, int64_t H, int64_t W) {
  if (N <= 0 || C_count <= 0 || H <= 0 || W <= 0) return 0;
  // coefficients of a chunk of maps + what the coefficient path itself needs for that chunk
  const int64_t HP = H + 1, WP = W + 1;
  const long long tile = (long long)HP * WP * 4;
  long long chunk = (256LL << 20) / tile;  // 256 MiB of coefficients per chunk at most
  if (chunk < 1) chunk = 1;
  if (chunk > N * C_count) chunk = N * C_count;
  return align_up((size_t)(chunk * tile), 256) + align_up((size_t)(chunk * tile), 256) + dcts_workspace_bytes(N, C_count, H, W);
}

int dcts_weighted_energy_f32(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W, int64_t strideN,
                             int64_t strideC, int64_t strideH, int64_t strideW, int32_t c_begin, int32_t c_count,
                             int32_t pad_front_if_odd, const float* weights, float* out_nc, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!x || !out_nc || !weights) return DCTS_E_NULL;
  if (N <= 0 || C_total <= 0 || H <= 0 || W <= 0) return DCTS_E_SHAPE;
  if (c_count <= 0 || c_begin < 0 || (int64_t)c_begin + c_count > C_total) return DCTS_E_CHANNELS;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return workspace ? DCTS_E_ALIGN : DCTS_E_WORKSPACE;
  const int pad = (pad_front_if_odd && (H % 2 != 0)) ? 1 : 0;
  const int64_t HP = H + pad, WP = W + pad;
  const long long tile = (long long)HP * WP * 4;
  // workspace = [coefficients of a chunk][scratch the coefficient path may use]
  const size_t inner_min = dcts_workspace_bytes(1, 1, H, W);
  if (workspace_bytes < (size_t)(2 * tile) + inner_min) return DCTS_E_WORKSPACE;
  long long chunk = (long long)((workspace_bytes - inner_min) / (size_t)(2 * tile));
  if (chunk < 1) return DCTS_E_WORKSPACE;
  const size_t off_inner = align_up((size_t)(chunk * tile), 256);
  if (off_inner + (size_t)(chunk * tile) > workspace_bytes) --chunk;
  if (chunk < 1) return DCTS_E_WORKSPACE;
  char* wsp = reinterpret_cast<char*>(workspace);
  float* coeff = reinterpret_cast<float*>(wsp);
  void* inner = wsp + align_up((size_t)(chunk * tile), 256);
  const size_t inner_bytes = workspace_bytes - align_up((size_t)(chunk * tile), 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // this call writes coefficients and scratch all over the workspace: no table cached in it survives, and the
  // inner calls (interior pointer, offset depends on the tile shape) do not cache theirs
  basis_forget(workspace);
  basis_forget_range(workspace, workspace_bytes);
  // the large-tile kernels where the tensor suits them, else whatever AUTO picks (codelet / direct)
  // (every inner call covers channels of ONE sample, so the batch stride does not matter)
  const bool dense = pad == 0 && H == W && strideH == W && strideW == 1 && strideC == H * W && (strideN * 4) % 16 == 0 &&
                     (reinterpret_cast<uintptr_t>(x + (long long)c_begin * strideC) & 15) == 0;
  const int algo = (dense && has_tile2d(HP)) ? DCTS_ALGO_TILE2D : (dense && (has_fused(HP) || has_fused2(HP))) ? DCTS_ALGO_FUSED : DCTS_ALGO_AUTO;
  // maps are taken sample by sample in runs of channels so that a chunk is one strided view of x
  const long long per_sample = c_count;
  for (int64_t n = 0; n < N; ++n) {
    for (long long c0 = 0; c0 < per_sample; c0 += chunk) {
      const long long nc = (per_sample - c0) < chunk ? (per_sample - c0) : chunk;
      int rc = run<true>(x + n * strideN, 1, C_total, H, W, strideN, strideC, strideH, strideW, (int32_t)(c_begin + c0), (int32_t)nc,
                         pad_front_if_odd, coeff, inner, inner_bytes, stream, algo, /*cache_basis=*/false);
      if (rc) return rc;
      long long blocks = (nc * 64 + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(k_weighted_energy, dim3((unsigned)blocks), dim3(256), 0, st, coeff, weights, nc, (int)(HP * WP),
                         out_nc + n * c_count + c0);
      rc = (int)hipGetLastError();
      if (rc) return rc;
    }
  }
  return DCTS_OK;
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

int dcts_energy_multi_f32(const dcts_tensor_item* items, int32_t count, int64_t H, int64_t W,
                          int32_t pad_front_if_odd, void* workspace, size_t workspace_bytes, void* stream) {
  if (!items) return DCTS_E_NULL;
  if (count <= 0 || H <= 0 || W <= 0) return DCTS_E_SHAPE;
  const int pad = (pad_front_if_odd && (H % 2 != 0)) ? 1 : 0;
  const int64_t HP = H + pad, WP = W + pad;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int32_t i = 0; i < count; ++i) {
    const dcts_tensor_item& t = items[i];
    if (!t.x || !t.out_nc) return DCTS_E_NULL;
    if (t.N <= 0 || t.C_total <= 0) return DCTS_E_SHAPE;
    if (t.c_count <= 0 || t.c_begin < 0 || (int64_t)t.c_begin + t.c_count > t.C_total) return DCTS_E_CHANNELS;
    if ((reinterpret_cast<uintptr_t>(t.x) & 3) || (reinterpret_cast<uintptr_t>(t.out_nc) & 3)) return DCTS_E_ALIGN;
  }
  if (has_codelet(HP, WP)) {
    const int G = codelet_group_size((int)HP);
    for (int32_t i0 = 0; i0 < count; i0 += kMultiItems) {
      const int n = (count - i0) < kMultiItems ? (count - i0) : kMultiItems;
      MultiGeom mg;
      long long groups = 0;
      for (int i = 0; i < n; ++i) {
        const dcts_tensor_item& t = items[i0 + i];
        MapGeom& g = mg.it[i].g;
        g.x = t.x;
        g.nmaps = t.N * (int64_t)t.c_count;
        g.strideN = t.strideN;
        g.strideC = t.strideC;
        g.strideH = W;
        g.c_count = t.c_count;
        g.c_begin = t.c_begin;
        g.H = (int)H;
        g.W = (int)W;
        g.contiguous = (t.N == 1 || t.strideN == (int64_t)t.c_count * t.strideC) ? 1 : 0;
        mg.it[i].out = t.out_nc;
        mg.it[i].group_begin = groups;
        groups += (g.nmaps + G - 1) / G;
      }
      for (int i = n; i < kMultiItems; ++i) mg.it[i] = mg.it[0];
      mg.total_groups = groups;
      mg.count = n;
      const int rc = dispatch_codelet_multi((int)HP, pad, mg, st);
      if (rc) return rc;
    }
    return DCTS_OK;
  }
  // large tiles with a single-launch kernel: the dense tensors go into ONE launch per 32 of them (their
  // maps form one index space: a CU that would get a fraction of a map from one small tensor now
  // draws from all of them); results are those of one call per tensor, bit for bit
  const int fam = (pad == 0 && H == W && has_split(HP, WP)) ? tile_family((int)HP, DCTS_ALGO_AUTO, 0)
                  : ((pad == 1 && H == W && dctsi::has_tile2g_pad((int)HP)) ? 6 : 0);
  TileBatch tb;
  int nb = 0;
  auto flush = [&]() -> int {
    if (!nb) return DCTS_OK;
    for (int i = nb; i < kTileItems; ++i) {
      tb.x[i] = tb.x[0];
      tb.out[i] = tb.out[0];
      tb.begin[i + 1] = tb.begin[nb];
    }
    tb.map_elems = H * W;
    tb.total = tb.begin[nb];
    tb.count = nb;
    nb = 0;
    return dispatch_tile_family(fam, (int)HP, tb, st);
  };
  for (int32_t i = 0; i < count; ++i) {
    const dcts_tensor_item& t = items[i];
    const float* x0 = t.x + (int64_t)t.c_begin * t.strideC;
    // (the dword-loading kernels - tile2g, families 5 and 6, and the two-roles kernel, family 2 - take any 4-byte-aligned base, the others need 16)
    const bool dense = fam && t.strideC == H * W && (t.N == 1 || t.strideN == (int64_t)t.c_count * t.strideC) &&
                       ((reinterpret_cast<uintptr_t>(x0) & 15) == 0 || fam >= 5 || (fam == 2 && DCTS_F2_REGLOAD) || (fam == 1 && DCTS_F1_REGLOAD));
    if (dense) {
      if (nb == 0) tb.begin[0] = 0;
      tb.x[nb] = x0;
      tb.out[nb] = t.out_nc;
      tb.begin[nb + 1] = tb.begin[nb] + t.N * (int64_t)t.c_count;
      if (++nb == kTileItems) {
        const int rc = flush();
        if (rc) return rc;
      }
      continue;
    }
    // everything else: one call per tensor (split / direct), same stream
    const int rc = run<false>(t.x, t.N, t.C_total, H, W, t.strideN, t.strideC, W, 1, t.c_begin, t.c_count,
                              pad_front_if_odd, t.out_nc, workspace, workspace_bytes, stream, DCTS_ALGO_AUTO);
    if (rc) return rc;
  }
  return flush();
}

int dcts_energy_mixed_f32(const dcts_shaped_item* items, int32_t count, void* workspace, size_t workspace_bytes,
                          void* stream) {
  if (!items) return DCTS_E_NULL;
  if (count <= 0) return DCTS_E_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int32_t i = 0; i < count; ++i) {
    const dcts_tensor_item& t = items[i].t;
    if (!t.x || !t.out_nc) return DCTS_E_NULL;
    if (t.N <= 0 || t.C_total <= 0 || items[i].H <= 0 || items[i].W <= 0) return DCTS_E_SHAPE;
    if (t.c_count <= 0 || t.c_begin < 0 || (int64_t)t.c_begin + t.c_count > t.C_total) return DCTS_E_CHANNELS;
    if ((reinterpret_cast<uintptr_t>(t.x) & 3) || (reinterpret_cast<uintptr_t>(t.out_nc) & 3)) return DCTS_E_ALIGN;
  }
  auto eligible = [&](const dcts_shaped_item& it) {
    return it.H == it.W && mixed_has((int)it.H) && !(it.pad_front_if_odd && (it.H % 2 != 0));
  };
  // 1. every small-tile tensor, whatever its shape, in one launch per kMixedItems of them
  MixedGeom mg;
  int n = 0;
  long long groups = 0;
  auto flush = [&]() -> int {
    if (!n) return DCTS_OK;
    for (int i = n; i < kMixedItems; ++i) mg.it[i] = mg.it[0];
    mg.total_groups = groups;
    mg.count = n;
    n = 0;
    groups = 0;
    return dctsi::dispatch_codelet_mixed(&mg, st);
  };
  for (int32_t i = 0; i < count; ++i) {
    if (!eligible(items[i])) continue;
    const dcts_tensor_item& t = items[i].t;
    MapGeom& g = mg.it[n].g;
    g.x = t.x;
    g.nmaps = t.N * (int64_t)t.c_count;
    g.strideN = t.strideN;
    g.strideC = t.strideC;
    g.strideH = items[i].W;
    g.c_count = t.c_count;
    g.c_begin = t.c_begin;
    g.H = (int)items[i].H;
    g.W = (int)items[i].W;
    g.contiguous = (t.N == 1 || t.strideN == (int64_t)t.c_count * t.strideC) ? 1 : 0;
    mg.it[n].out = t.out_nc;
    mg.it[n].group_begin = groups;
    const int G = 64 / (int)items[i].H;
    groups += (g.nmaps + G - 1) / G;
    if (++n == kMixedItems) {
      const int rc = flush();
      if (rc) return rc;
    }
  }
  int rc = flush();
  if (rc) return rc;
  // 2. the rest shape by shape (first occurrence order), through dcts_energy_multi_f32
  dcts_tensor_item buf[64];
  for (int32_t i = 0; i < count; ++i) {
    if (eligible(items[i])) continue;
    bool seen = false;
    for (int32_t k = 0; k < i && !seen; ++k)
      seen = !eligible(items[k]) && items[k].H == items[i].H && items[k].W == items[i].W &&
             (items[k].pad_front_if_odd != 0) == (items[i].pad_front_if_odd != 0);
    if (seen) continue;
    int m = 0;
    for (int32_t k = i; k < count; ++k) {
      if (eligible(items[k]) || items[k].H != items[i].H || items[k].W != items[i].W ||
          (items[k].pad_front_if_odd != 0) != (items[i].pad_front_if_odd != 0))
        continue;
      buf[m++] = items[k].t;
      if (m == 64) {
        rc = dcts_energy_multi_f32(buf, m, items[i].H, items[i].W, items[i].pad_front_if_odd, workspace, workspace_bytes, stream);
        if (rc) return rc;
        m = 0;
      }
    }
    if (m) {
      rc = dcts_energy_multi_f32(buf, m, items[i].H, items[i].W, items[i].pad_front_if_odd, workspace, workspace_bytes, stream);
      if (rc) return rc;
    }
  }
  return DCTS_OK;
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

#ifdef DCTS_FUSED_STAMPS
int dcts_debug_fused_stamps(unsigned long long* host_out /*[16][16]*/, int reset) {
  if (reset) {
    static unsigned long long zeros[16][16] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fused_stamps), 16 * 16 * sizeof(unsigned long long));
}
#endif

int dcts_debug_stream_read_f32(const float* x, int64_t n, float* sink, void* stream) {
  if (!x || !sink) return DCTS_E_NULL;
  if (n <= 0) return DCTS_E_SHAPE;
  hipLaunchKernelGGL(k_calib_read, dim3(256 * 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     (long long)n, sink);
  return (int)hipGetLastError();
}

}  // extern "C"
#endif  // DCTS_PART(5)
