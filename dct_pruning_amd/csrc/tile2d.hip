// tile2d.hip - large tiles (edge N = 8*M) as a 2-D radix-8 split: one in-register butterfly
// pre-pass over BOTH axes, then 64 independent M x M leaf blocks.
//
// Replaces, for these tile shapes, the per-map loop of the reference hooks
// (utils/common.py:262-277: dct.dct_2d(output[i,j,:,:], norm='ortho'), then sum(coeff^2)).
//
// Why another family. The split kernels of dct_kernels.hip transform one axis at a time: per axis
// the strip is staged in LDS, butterflied in place (read + write), read again by the M-point role
// codelets, and dumped once more for the other axis: 3 LDS writes + 4 reads per point and five
// workgroup barriers per 64-column step. tools/probes/valu_probe.hip shows what that costs on
// gfx950: a SIMD sustains one wave64 VALU instruction every 2 cycles once four waves feed it, while
// ds_write_b32 / ds_read_b32 cost 4 / 2 cycles per CU: at 224x224 those kernels spend ~20 k cycles
// per map in the LDS pipe and ~18 k waiting at barriers against ~13 k of VALU issue.
//
// Here the top three radix-2 levels of the DCT-II recursion (dct_codelets.hpp) are applied along H
// and W at once, in registers, straight from global memory: the 8 x 8 mirrored samples
// x[a*M + p~][b*M + q~] of an item (p, q) (p~ = p for even a, M-1-p for odd a; same for q~) go through
// the role network of split_roles.hpp along a (rotation constants by p), then along b (by q), and
// become one input sample of each of the 64 leaf problems Z[ra][rb] (row role ra, column role rb;
// a leaf is a 2-D transform, DCT-II or DCT-IV of length M per axis, RoleLeaf<>::is4). The transforms
// commute: (C_leaf o B_col) X (B_row^T o C_leaf^T) = C_leaf (B_col X B_row^T) C_leaf^T.
// A leaf block is then transformed by one wave exactly like a small tile in k_energy_codelet:
// lane = column, M-point codelet, transposed IN PLACE in the block's own LDS image (the wave that owns
// the block reads it by columns and writes 16-byte pieces of rows: T2Cfg), lane = row, second codelet,
// squares. LDS traffic: 2 writes + 2 reads per point; workgroup barriers: 4 per map.
//
// One workgroup of 16 waves per CU, persistent over maps. The 64 blocks of a map go through LDS in
// two sets of 32 (a 224 x 224 tile is 196 KiB, the LDS 160 KiB): an item's 32 outputs of the second
// set wait in the producer's registers while the first set is transformed. The loads of the next
// map are issued before the second set's transforms, so their latency hides there.
//
// As in the split family the last add/sub layer of every DCT-IV node above the leaves is folded into
// the reduction ((a+b)^2 + (a-b)^2 = 2a^2 + 2b^2, SplitNode::wt): energy-path-only shortcut;
// dcts_dct2d_f32_ex(DCTS_ALGO_TILE2D) stores the leaf outputs and applies that layer explicitly
// (k_t2_tail), which is how tests compare this kernel's coefficients with the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dctscore.h"
#include "split_roles.hpp"

namespace dctsi {
int dispatch_tile2d(int N, const void* tile_batch, hipStream_t st);
int dispatch_tile2d_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps,
                          hipStream_t st);
}  // namespace dctsi

namespace {

// Instruction-mix replay (tools/t2_dev.py on a -DDCTS_T2_EXP=3 build): the product kernel with every
// instruction in place but NO memory traffic - the loads of the next map read a zero-length buffer (the hardware
// returns 0 without a request) and the direct-to-LDS pieces are not issued. What it measures is the ceiling of the
// CU-side work of this design: cycles per map at 16 waves if HBM cost nothing (results are wrong by construction).
#ifndef DCTS_T2_EXP
#define DCTS_T2_EXP 0
#endif
constexpr int kT2L = 3, kT2S = 8, kT2Waves = 16;
constexpr RolePlan<kT2L> kT2Plan{};

// Which two blocks a consumer wave transforms together (lanes [0, M) and [M, 2M)), per set.
// Both blocks of a pass should run the same codelets (DCT-II or DCT-IV per axis): blocks are
// paired inside their class (type of ra, type of rb); the class sizes are odd (25/15/15/9 for
// L = 3), so the four left-over blocks form two passes whose second-axis codelets differ between
// the lane halves (both run, exec-masked). Passes are sorted by cost; set 0 takes the cheaper
// half in order, set 1 the dearer half in reverse: a wave's two passes add up evenly.
struct T2Sched {
  static constexpr int S = kT2S, NB = S * S, NP = NB / 2;
  int blk[2][NP] = {};   // [set][li] -> ra * S + rb     (li = 2 * wave + lane half)
  int set_of[NB] = {}, li_of[NB] = {};
  constexpr T2Sched() {
    constexpr RolePlan<kT2L> plan{};
    int cls[4][NB] = {}, ncls[4] = {};
    for (int ra = 0; ra < S; ++ra)
      for (int rb = 0; rb < S; ++rb) {
        const int c = plan.is4_of_role[ra] * 2 + plan.is4_of_role[rb];
        cls[c][ncls[c]++] = ra * S + rb;
      }
    int pairs[NP][2] = {}, np = 0, single[4] = {-1, -1, -1, -1};
    for (int c = 0; c < 4; ++c) {
      for (int i = 0; i + 1 < ncls[c]; i += 2) {
        pairs[np][0] = cls[c][i];
        pairs[np][1] = cls[c][i + 1];
        ++np;
      }
      if (ncls[c] % 2) single[c] = cls[c][ncls[c] - 1];
    }
    for (int c = 0; c < 4; c += 2)
      if (single[c] >= 0 && single[c + 1] >= 0) {
        pairs[np][0] = single[c];
        pairs[np][1] = single[c + 1];
        ++np;
      }
    for (int i = 0; i < NP; ++i) {
      const int set = i < NP / 2 ? 0 : 1;
      // wave (putting set 0's dearer passes on the older waves as well measured neutral)
      const int w = set == 0 ? i : NP - 1 - i;
      for (int g = 0; g < 2; ++g) {
        const int b = pairs[i][g];
        blk[set][2 * w + g] = b;
        set_of[b] = set;
        li_of[b] = 2 * w + g;
      }
    }
  }
};

constexpr T2Sched kT2Sch{};

// The item slots (a, b) whose outputs belong to `set`, ordered by column slot b then a: slot i of
// the set as a * S + b. A slot's register is free for the next map's sample once the set has been
// written to LDS, so the next map is loaded set by set, eight slots per hook point.
constexpr int t2_set_slot(int set, int i, int bmax) {
  int role_of_slot[kT2S] = {};
  for (int r = 0; r < kT2S; ++r) role_of_slot[kT2Plan.slot_of_role[r]] = r;
  int n = 0;
#ifdef DCTS_T2_AMAJOR
  for (int a = 0; a < kT2S; ++a)
    for (int b = 0; b < bmax; ++b)
#else
  for (int b = 0; b < bmax; ++b)
    for (int a = 0; a < kT2S; ++a)
#endif
      if (kT2Sch.set_of[role_of_slot[a] * kT2S + role_of_slot[b]] == set) {
        if (n == i) return a * kT2S + b;
        ++n;
      }
  return -1;
}
constexpr int t2_set_count(int set, int bmax) {
  int n = 0;
  while (t2_set_slot(set, n, bmax) >= 0) ++n;
  return n;
}
// load order of the slots with b < bmax: set 0's (free first), then set 1's
constexpr int t2_load_slot(int i, int bmax) {
  const int n0 = t2_set_count(0, bmax);
  return i < n0 ? t2_set_slot(0, i, bmax) : t2_set_slot(1, i - n0, bmax);
}

template <int M>
struct T2Cfg {
  static constexpr int L = kT2L, S = kT2S, N = M * S, NW = kT2Waves;
  static_assert(2 * M <= 64 && 2 * M > 32, "two blocks per wave");
  // A block is read by columns on BOTH axes (lane = column, consecutive words: conflict-free for any row
  // stride) and transposed by axis A's WRITES: lane q' puts its outputs into row q' of the block, 16
  // bytes at a time (ds_write_b128; rows of M = 28 floats start 28 banks apart: eight lanes cover the 32
  // banks). ds_write_b32 costs 3-4 cycles per wave (tools/probes/valu_probe.hip), as much as a
  // ds_read_b128, so this halves the LDS write time of the transforms. The second block / item row of a
  // wave starts at lane 32: half-waves never share banks, row and block strides need no padding.
  // (Before: in-place column writes, odd row stride 29, the second block at lane 28.)
  static constexpr int RS = M;                // block row stride
  static constexpr int BS = M * RS + 4;       // floats per block; 4 pad words take the unconditional stores of lanes without an item
  static constexpr int HW = 32;               // first lane of the second block / item row
  static_assert(M % 4 == 0 && (M / 2) % 2 == 0 && M <= 32, "16-byte rows, even halves");
  // position of axis A's output k in the transposed row: even k first, then odd k (the order the
  // staged codelets finish them in, so every 16-byte group is complete when its stage ends)
  static constexpr int pos_of_k(int k) { return (k % 2 == 0) ? k / 2 : M / 2 + k / 2; }
  static constexpr int k_of_pos(int j) { return j < M / 2 ? 2 * j : 2 * (j - M / 2) + 1; }
  static constexpr int ZSET = (S * S / 2) * BS;       // floats: one set of 32 blocks
  static constexpr int PI = 64 / M;                   // item rows per producer wave (2)
  static constexpr int PWAVES = (M + PI - 1) / PI;    // producer waves (14 for M = 28)
  static_assert(PWAVES <= NW, "producers");
  static constexpr int NROT = 3;                      // rotations of the L = 3 network
#ifndef DCTS_T2_DMACOLS
#define DCTS_T2_DMACOLS 2
#endif
#ifndef DCTS_T2_HOOKS
#define DCTS_T2_HOOKS 4, 4, 4, 12, 12, 12
#endif
  // The last DB column slots of the NEXT map do not land in registers but in LDS (direct-to-LDS
  // loads into the 56 KB the Z set leaves free: no VGPRs, nothing for the register allocator to
  // spill), issued while set 0 is transformed; phase A reads them from there.
  static constexpr int DB = DCTS_T2_DMACOLS;
  static constexpr int VB = S - DB;                    // column slots loaded into registers
  static constexpr int NV = S * VB;                    // ... = loads per lane and map
  static constexpr int RAWW = DB * M;                  // floats per row of the raw image in LDS
  static constexpr int RAW = N * RAWW;                 // floats
  static constexpr int RAW_PIECES = (RAW / 4 + 63) / 64;  // 1 KiB direct-to-LDS instructions per map
  static_assert((RAWW % 4) == 0 && ((VB * M) % 4) == 0, "16-byte pieces");
  // next-map loads at the six hook points of a map (three per pass: before axis A, between the axes,
  // after axis B; first the pass of set 0, then that of set 1); what is left goes out after the passes
  static constexpr int HOOKS[6] = {DCTS_T2_HOOKS};
  static constexpr int hook_begin(int h) {
    int n = 0;
    for (int i = 0; i < h; ++i) n += HOOKS[i];
    return n;
  }
  static_assert(hook_begin(6) <= NV, "loads per map");
  // (a slot must not be loaded before its set has gone to LDS: checked where the hooks are built)
};

// rotation constants (c, s, sigma*c, sigma*s), sigma = (-1)^j of the pair index: [rot][p][4]
template <int M>
struct T2RotTable {
  float v[T2Cfg<M>::NROT][M][4] = {};
  constexpr T2RotTable() {
    constexpr RotTable<M, kT2L> t{};
    for (int r = 0; r < T2Cfg<M>::NROT; ++r)
      for (int p = 0; p < M; ++p) {
        const float sg = RotTable<M, kT2L>::sign0(r) * ((p & 1) ? -1.f : 1.f);
        v[r][p][0] = t.c[r][p];
        v[r][p][1] = t.s[r][p];
        v[r][p][2] = sg * t.c[r][p];
        v[r][p][3] = sg * t.s[r][p];
      }
  }
};
template <int M>
__device__ const T2RotTable<M> kT2Rot{};

// per (set, li): leaf types and squared amplitude weights of the block
struct T2BlockParam {
  int tA, tB;                    // 1: DCT-IV along p' (A) / q' (B)
  int variant;                   // tA * 4 + tB(first block) * 2 + tB(second block) of the pass
  int block;                     // ra * S + rb
  float wA0, wA1, wB0, wB1;      // squared weights of output 0 / outputs > 0 per axis
};
template <int M>
struct T2ParamTable {
  T2BlockParam v[2][kT2S * kT2S / 2] = {};
};
template <int M, int R>
constexpr void t2_role_weights(float& w0, float& w1) {
  using Leaf = typename RoleLeaf<M * kT2S, kT2L, R>::type;
  const double a = Leaf::wt(true), b = Leaf::wt(false);
  w0 = float(a * a);
  w1 = float(b * b);
}
template <int M, int... R>
constexpr T2ParamTable<M> t2_make_params(std::integer_sequence<int, R...>) {
  constexpr RolePlan<kT2L> plan{};
  constexpr T2Sched sch{};
  float w0[kT2S] = {}, w1[kT2S] = {};
  (t2_role_weights<M, R>(w0[R], w1[R]), ...);
  T2ParamTable<M> t{};
  for (int s = 0; s < 2; ++s)
    for (int li = 0; li < kT2S * kT2S / 2; ++li) {
      const int ra = sch.blk[s][li] / kT2S, rb = sch.blk[s][li] % kT2S;
      t.v[s][li].block = sch.blk[s][li];
      t.v[s][li].tA = plan.is4_of_role[ra];
      t.v[s][li].tB = plan.is4_of_role[rb];
      t.v[s][li].wA0 = w0[ra];
      t.v[s][li].wA1 = w1[ra];
      t.v[s][li].wB0 = w0[rb];
      t.v[s][li].wB1 = w1[rb];
    }
  for (int s = 0; s < 2; ++s)
    for (int li = 0; li < kT2S * kT2S / 2; li += 2)
      t.v[s][li].variant = t.v[s][li + 1].variant = t.v[s][li].tA * 4 + t.v[s][li].tB * 2 + t.v[s][li + 1].tB;
  return t;
}
template <int M>
__device__ const T2ParamTable<M> kT2Params = t2_make_params<M>(std::make_integer_sequence<int, kT2S>{});

// Diagnostic build only (-DDCTS_T2_STAMPS, tools/t2_dev.py): s_memtime stamps at the phase
// boundaries, summed per wave into g_t2_stamps (never touches an output).
#ifdef DCTS_T2_STAMPS
__device__ unsigned long long g_t2_stamps[16][16];
#define T2_STAMP(slot)                                                            \
  do {                                                                            \
    unsigned long long t_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
    __builtin_amdgcn_sched_barrier(0);                                            \
    acc_[slot] += t_ - last_;                                                     \
    last_ = t_;                                                                   \
  } while (0)
#else
#define T2_STAMP(slot) ((void)0)
#endif

__device__ __forceinline__ void t2_pin(float& x) { asm volatile("" : "+v"(x)); }

// the L = 3 role network on 8 values held in registers: y[slot], constants by lane
template <int M>
__device__ __forceinline__ void t2_network(float (&y)[kT2S], const float (&rc)[T2Cfg<M>::NROT][4]) {
  constexpr RolePlan<kT2L> plan{};
  dcts::static_for<plan.NOPS>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int o = decltype(i)::value;
    constexpr int a = plan.op_a[o], b = plan.op_b[o], r = plan.op_rot[o];
    const float ya = y[a], yb = y[b];
    if constexpr (r < 0) {
      y[a] = ya + yb;
      y[b] = ya - yb;
    } else {
      y[a] = ya * rc[r][0] + yb * rc[r][1];
      y[b] = yb * rc[r][2] - ya * rc[r][3];
    }
  });
}

// Leaf transforms in stages, with scheduling barriers between the half-size sub-transforms and the
// outputs handed to `sink(k, value)` as soon as a stage has them (an LDS store on axis A, a square on
// axis B). Left to itself the machine scheduler interleaves the independent halves of a codelet for
// ILP until it runs into the 128-VGPR limit, and the register allocator then spills a handful of the
// caller's long-lived values (the parked outputs, the next map's samples) around every codelet. One
// wave issues a VALU instruction every ~8 cycles whatever the ILP (tools/probes/valu_probe.hip), so
// the narrow schedule costs nothing. Same arithmetic as dcts::Dct2 / dcts::Dct4, bit for bit.
template <int M, class Sink>
__device__ __forceinline__ void t2_dct2_staged(const float (&x)[M], Sink sink) {
  static_assert(M % 2 == 0, "even leaf");
  constexpr int H = M / 2;
  float u[H], v[H];
  dcts::static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int n = decltype(i)::value;
    u[n] = x[n] + x[M - 1 - n];
    v[n] = x[n] - x[M - 1 - n];
  });
  __builtin_amdgcn_sched_barrier(0);
  {
    float E[H];
    dcts::Dct2<H>::run(u, E);
    dcts::static_for<H>([&](auto i) DCTS_LAMBDA_INLINE { sink(std::integral_constant<int, 2 * decltype(i)::value>{}, E[decltype(i)::value]); });
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    float O[H];
    dcts::Dct4<H>::run(v, O);
    dcts::static_for<H>([&](auto i) DCTS_LAMBDA_INLINE { sink(std::integral_constant<int, 2 * decltype(i)::value + 1>{}, O[decltype(i)::value]); });
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int M, class Sink>
__device__ __forceinline__ void t2_dct4_staged(const float (&v)[M], Sink sink) {
  static_assert(M % 2 == 0, "even leaf");
  constexpr int H = M / 2;
  float a[H], b[H];
  dcts::static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int n = decltype(i)::value;
    constexpr float c = float(dcts::cospi_frac(2 * n + 1, 4 * M));
    constexpr float sn = float(dcts::sinpi_frac(2 * n + 1, 4 * M));
    constexpr float sg = (n % 2 == 0) ? 1.0f : -1.0f;
    a[n] = v[n] * c + v[M - 1 - n] * sn;
    b[n] = v[M - 1 - n] * (sg * c) - v[n] * (sg * sn);
  });
  __builtin_amdgcn_sched_barrier(0);
  float A[H], B[H];
  dcts::Dct2<H>::run(a, A);
  __builtin_amdgcn_sched_barrier(0);
  dcts::Dct2<H>::run(b, B);
  __builtin_amdgcn_sched_barrier(0);
  sink(std::integral_constant<int, 0>{}, A[0]);
  sink(std::integral_constant<int, M - 1>{}, -B[0]);
  dcts::static_for<H - 1>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int jj = decltype(i)::value + 1;
    sink(std::integral_constant<int, 2 * jj>{}, A[jj] + B[H - jj]);
    sink(std::integral_constant<int, 2 * jj - 1>{}, A[jj] - B[H - jj]);
  });
  __builtin_amdgcn_sched_barrier(0);
}

// The same staged codelets with the outputs handed over in two groups of M/2: even k after the first
// stage, odd k after the second (DCT-II), or both at the end (DCT-IV), as arrays indexed by k / 2:
// axis A stores a group as 16-byte pieces of the transposed row (T2Cfg::pos_of_k).
template <int M, class Sink>
__device__ __forceinline__ void t2_dct2_groups(const float (&x)[M], Sink sink) {
  constexpr int H = M / 2;
  float u[H], v[H];
  dcts::static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int n = decltype(i)::value;
    u[n] = x[n] + x[M - 1 - n];
    v[n] = x[n] - x[M - 1 - n];
  });
  __builtin_amdgcn_sched_barrier(0);
  {
    float E[H];
    dcts::Dct2<H>::run(u, E);
    sink(std::integral_constant<int, 0>{}, E);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    float O[H];
    dcts::Dct4<H>::run(v, O);
    sink(std::integral_constant<int, 1>{}, O);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int M, class Sink>
__device__ __forceinline__ void t2_dct4_groups(const float (&v)[M], Sink sink) {
  constexpr int H = M / 2;
  float a[H], b[H];
  dcts::static_for<H>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int n = decltype(i)::value;
    constexpr float c = float(dcts::cospi_frac(2 * n + 1, 4 * M));
    constexpr float sn = float(dcts::sinpi_frac(2 * n + 1, 4 * M));
    constexpr float sg = (n % 2 == 0) ? 1.0f : -1.0f;
    a[n] = v[n] * c + v[M - 1 - n] * sn;
    b[n] = v[M - 1 - n] * (sg * c) - v[n] * (sg * sn);
  });
  __builtin_amdgcn_sched_barrier(0);
  float A[H], B[H];
  dcts::Dct2<H>::run(a, A);
  __builtin_amdgcn_sched_barrier(0);
  dcts::Dct2<H>::run(b, B);
  __builtin_amdgcn_sched_barrier(0);
  {
    float Ev[H];  // k = 0, 2, 4, ...
    Ev[0] = A[0];
    dcts::static_for<H - 1>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int jj = decltype(i)::value + 1;
      Ev[jj] = A[jj] + B[H - jj];
    });
    sink(std::integral_constant<int, 0>{}, Ev);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    float Od[H];  // k = 1, 3, 5, ...
    dcts::static_for<H - 1>([&](auto i) DCTS_LAMBDA_INLINE {
      constexpr int jj = decltype(i)::value + 1;
      Od[jj - 1] = A[jj] - B[H - jj];
    });
    Od[H - 1] = -B[0];
    sink(std::integral_constant<int, 1>{}, Od);
  }
  __builtin_amdgcn_sched_barrier(0);
}

typedef float t2_v4f __attribute__((ext_vector_type(4)));
typedef float t2_v2f __attribute__((ext_vector_type(2)));
// H consecutive floats of an LDS row starting at word BASE (row 16-byte aligned): 16-byte stores
// where BASE + i is a multiple of 4, 8-byte stores for the odd pair at either end
template <int H, int BASE>
__device__ __forceinline__ void t2_store_group(lds_ptr row, const float (&g)[H]) {
  static_assert(H % 2 == 0 && BASE % 2 == 0, "pairs");
  constexpr int lead = (BASE % 4 == 0) ? 0 : 2;
  if constexpr (lead) *(__attribute__((address_space(3))) t2_v2f*)(row + BASE) = t2_v2f{g[0], g[1]};
  constexpr int nq = (H - lead) / 4;
  dcts::static_for<nq>([&](auto i) DCTS_LAMBDA_INLINE {
    constexpr int o = lead + 4 * decltype(i)::value;
    *(__attribute__((address_space(3))) t2_v4f*)(row + BASE + o) = t2_v4f{g[o], g[o + 1], g[o + 2], g[o + 3]};
  });
  constexpr int done = lead + 4 * nq;
  if constexpr (done < H) *(__attribute__((address_space(3))) t2_v2f*)(row + BASE + done) = t2_v2f{g[done], g[done + 1]};
}

// One pass = two leaf blocks (lane halves). Each axis is straight-line code per codelet type, picked
// by a wave-uniform branch AROUND the whole axis (stores / squares inside the arms): a branch between
// the DCT-II and the DCT-IV codelet with the 28 outputs merged behind it doubled the footprint
// (41 -> 80 VGPRs) and spilled next to the parked values, and a scratch reload waits vmcnt(0), i.e. for
// the prefetch of the next map. The next map's loads are issued BETWEEN the axes, outside every arm
// (inside, each arm would define the 64 sample registers and the merge again costs registers).
template <int M>
struct T2Pass {
  lds_ptr blk;
  lds_cptr pp;
  int g, j, li;
  bool act;
  __device__ __forceinline__ T2Pass(lds_ptr zbuf, lds_cptr params, int set, int wave, int lane) {
    g = lane >= T2Cfg<M>::HW ? 1 : 0;
    j = lane - g * T2Cfg<M>::HW;
    act = j < M;
    li = 2 * wave + g;
    pp = params + (set * (kT2S * kT2S / 2) + li) * 8;
    blk = zbuf + li * T2Cfg<M>::BS;
  }
};

// axis A: lane = column q', transform along p'; the lane's outputs go to ROW q' of the block (the
// transposition; every lane of the wave has read its column before the first store is issued - the
// codelets consume all inputs in their first stage and a wave's LDS operations complete in order), as
// 16-byte pieces in the order pos_of_k.
template <int M, int TA>
__device__ __forceinline__ void t2_axis_a(const T2Pass<M>& ps) {
  constexpr int RS = T2Cfg<M>::RS;
  float in[M];
  lds_cptr src = ps.blk + (ps.act ? ps.j : 0);
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { in[decltype(i)::value] = src[decltype(i)::value * RS]; });
  lds_ptr row = ps.blk + ps.j * RS;
  const bool act = ps.act;
  // exec-masked per group of M/2 outputs (the group's values are alive at that point anyway; a mask
  // around single stores kept all M outputs alive at once: 62 VGPRs instead of 44)
  auto put = [&](auto half, const float (&g)[M / 2]) DCTS_LAMBDA_INLINE {
    if (act) t2_store_group<M / 2, decltype(half)::value * (M / 2)>(row, g);
  };
  if constexpr (TA)
    t2_dct4_groups<M>(in, put);
  else
    t2_dct2_groups<M>(in, put);
}

// axis B: lane = row k1, transform along q'; returns the lane's weighted energy. A mixed pass (the
// lane halves differ in type) runs both codelets on every lane, one after the other (the row is read
// twice), and keeps the sums of the lane's own type: straight-line, same footprint as the others.
template <int M, int TB0, int TB1, bool STORE>
__device__ __forceinline__ float t2_axis_b(const T2Pass<M>& ps, int set, float* leaf_out) {
  constexpr int RS = T2Cfg<M>::RS;
  float s0 = 0.f, s1 = 0.f;
  auto run = [&](auto tb, bool mine) DCTS_LAMBDA_INLINE {
    constexpr int TB = decltype(tb)::value;
    float z[M];
    // lane = row k1 of the half-transformed block = column j of the transposed image (k1 = k_of_pos(j))
    lds_cptr src = ps.blk + (ps.act ? ps.j : 0);
    dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { z[decltype(i)::value] = src[decltype(i)::value * RS]; });
    float t0 = 0.f, t1 = 0.f;
    // debug / parity path: leaf outputs, unweighted, as [ra * M + k1][rb * M + k2] (k_assemble's layout)
    float* o = nullptr;
    if constexpr (STORE) {
      const int blk_id = __builtin_bit_cast(int, ps.pp[7]);
      o = leaf_out + ((long long)((blk_id / kT2S) * M + T2Cfg<M>::k_of_pos(ps.act ? ps.j : 0)) * (M * kT2S) + (blk_id % kT2S) * M);
    }
    auto sq = [&](auto k, float val) DCTS_LAMBDA_INLINE {
      if constexpr (decltype(k)::value == 0)
        t0 = val * val;
      else
        t1 = fmaf(val, val, t1);
      if constexpr (STORE) {
        if (ps.act && mine) o[decltype(k)::value] = val;
      }
    };
    if constexpr (TB)
      t2_dct4_staged<M>(z, sq);
    else
      t2_dct2_staged<M>(z, sq);
    s0 = mine ? t0 : s0;
    s1 = mine ? t1 : s1;
  };
  if constexpr (TB0 == TB1) {
    run(std::integral_constant<int, TB0>{}, true);
  } else {
    run(std::integral_constant<int, TB0>{}, ps.g == 0);
    asm volatile("" ::: "memory");
    run(std::integral_constant<int, TB1>{}, ps.g != 0);
  }
  const float e = (ps.j == 0 ? ps.pp[2] : ps.pp[3]) * (ps.pp[4] * s0 + ps.pp[5] * s1);
  return ps.act ? e : 0.f;
}

// variant id of a pass: TA * 4 + TB0 * 2 + TB1; which ones occur in a set is known at compile time
constexpr int t2_variant(int set, int w) {
  const int b0 = kT2Sch.blk[set][2 * w], b1 = kT2Sch.blk[set][2 * w + 1];
  const int ta = kT2Plan.is4_of_role[b0 / kT2S];  // == that of b1 (pairs share axis A's type)
  return ta * 4 + kT2Plan.is4_of_role[b0 % kT2S] * 2 + kT2Plan.is4_of_role[b1 % kT2S];
}
// does a pass of `set` have (variant & mask) == value?
constexpr bool t2_set_has(int set, int mask, int value) {
  for (int w = 0; w < kT2Waves; ++w)
    if ((t2_variant(set, w) & mask) == value) return true;
  return false;
}
static_assert([] {
  for (int s = 0; s < 2; ++s)
    for (int w = 0; w < kT2Waves; ++w)
      if (kT2Plan.is4_of_role[kT2Sch.blk[s][2 * w] / kT2S] != kT2Plan.is4_of_role[kT2Sch.blk[s][2 * w + 1] / kT2S]) return false;
  return true;
}(), "the two blocks of a pass share the type of axis A");

// exhaustive if / else chains over the types a set contains (the last one is the else arm)
template <int M, int SET>
__device__ __forceinline__ void t2_axis_a_chain(int vid, const T2Pass<M>& ps) {
  constexpr bool has0 = t2_set_has(SET, 4, 0), has1 = t2_set_has(SET, 4, 4);
  if constexpr (has0 && has1) {
    if (vid & 4)
      t2_axis_a<M, 1>(ps);
    else
      t2_axis_a<M, 0>(ps);
  } else if constexpr (has1) {
    t2_axis_a<M, 1>(ps);
  } else {
    t2_axis_a<M, 0>(ps);
  }
}
template <int M, int SET, bool STORE, int TB>
__device__ __forceinline__ float t2_axis_b_chain(int vid, const T2Pass<M>& ps, float* leaf_out) {
  constexpr bool here = t2_set_has(SET, 3, TB);
  constexpr bool later = [] {
    for (int u = TB + 1; u < 4; ++u)
      if (t2_set_has(SET, 3, u)) return true;
    return false;
  }();
  if constexpr (here && later) {
    if ((vid & 3) == TB) return t2_axis_b<M, (TB >> 1) & 1, TB & 1, STORE>(ps, SET, leaf_out);
    return t2_axis_b_chain<M, SET, STORE, TB + 1>(vid, ps, leaf_out);
  } else if constexpr (here) {
    return t2_axis_b<M, (TB >> 1) & 1, TB & 1, STORE>(ps, SET, leaf_out);
  } else {
    static_assert(TB < 3, "a set has at least one variant");
    return t2_axis_b_chain<M, SET, STORE, TB + 1>(vid, ps, leaf_out);
  }
}

// (The four waves of a SIMD - wave, wave + 4, ... - are arbitrated oldest first and finish a phase one after
// the other, the youngest last. Rotating s_setprio through them per phase half or per codelet stage evens
// their progress out but only moves time from the barriers into the phases: measured neutral, removed.)
// One pass with the caller's hook at three points: before axis A, between the axes, after axis B.
// The caller trickles the next map's loads out there: a burst of loads blocks the issuing wave until
// the CU's miss queue has room (stamps: 10 k cycles per map for 32 loads per wave in one go, i.e. the
// HBM rate); spread over the transforms they do not queue up.
template <int M, int SET, bool STORE, class Hook>
__device__ __forceinline__ float t2_consume(int vid, lds_ptr zbuf, lds_cptr params, int wave, int lane, float* leaf_out, Hook hook) {
  const T2Pass<M> ps(zbuf, params, SET, wave, lane);
  hook(std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  t2_axis_a_chain<M, SET>(vid, ps);
  __builtin_amdgcn_sched_barrier(0);
  hook(std::integral_constant<int, 1>{});
  __builtin_amdgcn_sched_barrier(0);
  // the wave's own LDS traffic is in order; only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float e = t2_axis_b_chain<M, SET, STORE, 0>(vid, ps, leaf_out);
  asm volatile("" : "+v"(e));
  __builtin_amdgcn_sched_barrier(0);
  hook(std::integral_constant<int, 2>{});
  __builtin_amdgcn_sched_barrier(0);
  return e;
}

// issue the loads of item (p, q) of a map for the column slots b in [B0, B1): buffer loads (one
// wave-uniform descriptor per map, four lane offsets, the slot's offset as the scalar/immediate part)
// instead of 64 per-lane 64-bit addresses; out-of-range reads return 0
// (lanes without an item - ok false - and the loads behind the last map of a workgroup - bytes 0 - read out of the
// descriptor's range: the hardware returns 0 and makes no request. Round 2 let them re-read item 0 / the last map:
// the PMC passes showed 1.18 x the algorithmic bytes for this kernel, profiles/r03_pmc_traffic_large_*.json)
constexpr int kT2Out = 0x7ffffff0;
template <int M, int B0, int B1>
__device__ __forceinline__ void t2_load(const float* __restrict__ in_b, int p, int q, bool ok, float (&v)[kT2S][kT2S]) {
  constexpr int N = T2Cfg<M>::N;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, DCTS_T2_EXP == 3 ? 0 : N * N * 4, 0x00020000);
  const int pe = p, po = M - 1 - p, qe = q, qo = M - 1 - q;
  const int o_ee = ok ? (pe * N + qe) * 4 : kT2Out, o_eo = ok ? (pe * N + qo) * 4 : kT2Out, o_oe = ok ? (po * N + qe) * 4 : kT2Out,
            o_oo = ok ? (po * N + qo) * 4 : kT2Out;
  dcts::static_for<kT2S>([&](auto ia) DCTS_LAMBDA_INLINE {
    constexpr int a = decltype(ia)::value;
    dcts::static_for<B1 - B0>([&](auto ib) DCTS_LAMBDA_INLINE {
      constexpr int b = B0 + decltype(ib)::value;
      const int voff = (a % 2 == 0) ? ((b % 2 == 0) ? o_ee : o_eo) : ((b % 2 == 0) ? o_oe : o_oo);
      v[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (a * M * N + b * M) * 4, 0));
    });
  });
}

// the same for entries [I0, I1) of the load order (t2_load_slot)
template <int M, int I0, int I1>
__device__ __forceinline__ void t2_load_seq(const float* __restrict__ in_b, unsigned bytes, int p, int q, bool ok, float (&v)[kT2S][kT2S]) {
  constexpr int N = T2Cfg<M>::N;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, DCTS_T2_EXP == 3 ? 0u : bytes, 0x00020000);
  const int pe = p, po = M - 1 - p, qe = q, qo = M - 1 - q;
  const int o_ee = ok ? (pe * N + qe) * 4 : kT2Out, o_eo = ok ? (pe * N + qo) * 4 : kT2Out, o_oe = ok ? (po * N + qe) * 4 : kT2Out,
            o_oo = ok ? (po * N + qo) * 4 : kT2Out;
  dcts::static_for<(I1 > I0 ? I1 - I0 : 0)>([&](auto ii) DCTS_LAMBDA_INLINE {
    constexpr int sl = t2_load_slot(I0 + decltype(ii)::value, T2Cfg<M>::VB);
    static_assert(sl >= 0, "slot");
    constexpr int a = sl / kT2S, b = sl % kT2S;
    const int voff = (a % 2 == 0) ? ((b % 2 == 0) ? o_ee : o_eo) : ((b % 2 == 0) ? o_oe : o_oo);
    v[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (a * M * N + b * M) * 4, 0));
  });
}

// piece `i` (64 lanes x 16 B) of the raw image: rows 0..N-1, columns [VB*M, N) of the map, dense
// [row][RAWW] in LDS. Raw instruction (ordering is the kernel's own s_waitcnt vmcnt(0) + barrier;
// see FusedStage::piece_raw in dct_kernels.hip for why not the builtin).
template <int M>
__device__ __forceinline__ void t2_dma_piece(const float* __restrict__ in_b, lds_ptr raw, int i, int lane) {
  using Cfg = T2Cfg<M>;
  constexpr int QPR = Cfg::RAWW / 4;  // quads per row
  const int f = i * 64 + lane;
  if (f < Cfg::RAW / 4) {
    const int row = f / QPR, cq = f - row * QPR;
    const unsigned off = (unsigned)(row * Cfg::N + Cfg::VB * M + 4 * cq) * 4u;  // bytes
    const unsigned dst = (unsigned)(unsigned long long)(raw + i * 256);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(dst), "v"(off), "s"(in_b)
                 : "memory", "m0");
  }
}

template <int M, class Src, bool STORE>
__device__ __forceinline__ void t2_body(const Src& tb, lds_ptr zbuf, lds_ptr raw, lds_ptr rot, lds_ptr params,
                                        lds_ptr partials, float* leaf_out) {
  using Cfg = T2Cfg<M>;
  constexpr int S = kT2S, RS = Cfg::RS, BS = Cfg::BS, NROT = Cfg::NROT;
  const int lane_in = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DCTS_T2_STAMPS
  unsigned long long acc_[16] = {}, last_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
  // tables -> LDS (read per map instead of living in registers)
  for (int i = threadIdx.x; i < NROT * M * 4; i += blockDim.x) rot[i] = (&kT2Rot<M>.v[0][0][0])[i];
  for (int i = threadIdx.x; i < 2 * (S * S / 2); i += blockDim.x) {
    const T2BlockParam& bp = (&kT2Params<M>.v[0][0])[i];
    params[i * 8 + 0] = __builtin_bit_cast(float, bp.tA);
    params[i * 8 + 1] = __builtin_bit_cast(float, bp.tB);
    params[i * 8 + 2] = bp.wA0;
    params[i * 8 + 3] = bp.wA1;
    params[i * 8 + 4] = bp.wB0;
    params[i * 8 + 5] = bp.wB1;
    params[i * 8 + 6] = __builtin_bit_cast(float, bp.variant);
    params[i * 8 + 7] = __builtin_bit_cast(float, bp.block);
  }
  const bool producer = wave < Cfg::PWAVES;
  const long long nmaps = tb.total;
  long long m = blockIdx.x;
  long long pending_m = -1;
  int pslot = 0, pending_slot = 0;
  float v[S][S];
  auto item_pq = [&](int& p, int& q, bool& ok) DCTS_LAMBDA_INLINE {
    const int lane = launder(lane_in);
    const int pi = lane >= Cfg::HW ? 1 : 0;
    q = lane - pi * Cfg::HW;
    p = Cfg::PI * wave + pi;
    ok = q < M && pi < Cfg::PI && p < M;
    if (!ok) {
      p = 0;
      q = 0;
    }
  };
  {
    // every wave loads and butterflies (the two waves without items redo item 0 and store nothing):
    // a conditional load would merge with the OLD values behind it and keep all 64 of them alive
    // across the transforms - that merge, not the prefetch, was what spilled
    int p, q;
    bool ok;
    item_pq(p, q, ok);
    const float* first = tile_in(tb, m);  // grid <= nmaps: every workgroup owns a map
    t2_load<M, 0, Cfg::VB>(first, p, q, ok, v);
    if constexpr (Cfg::DB > 0) {
      for (int i = wave; i < Cfg::RAW_PIECES; i += kT2Waves) t2_dma_piece<M>(first, raw, i, lane_in);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed (the barrier below publishes them)
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  int hint_next = 0, hint_out = 0;  // tensor of the next / finished map (tile_item resumes its scan there)
  auto finish = [&]() DCTS_LAMBDA_INLINE {
    if (pending_m >= 0) {
      if (wave == 0 && lane_in == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < kT2Waves; ++i) t += partials[pending_slot * kT2Waves + i];
        constexpr float sc = float(4.0 / (double(Cfg::N) * double(Cfg::N)));
        if constexpr (!STORE) *tile_out(tb, pending_m, &hint_out) = t * sc;  // the coefficient path has no energy output
      }
      pending_m = -1;
    }
  };
  lds_barrier();  // tables are in LDS
  T2_STAMP(15);
  for (; m < nmaps; m += gridDim.x) {
    // ---- A: butterflies of this wave's items, both axes, in registers ---------------------------
    {
      int p, q;
      bool ok;
      item_pq(p, q, ok);
      float rp[NROT][4];
      dcts::static_for<NROT>([&](auto ir) DCTS_LAMBDA_INLINE {
        constexpr int r = decltype(ir)::value;
        dcts::static_for<4>([&](auto ic) DCTS_LAMBDA_INLINE { rp[r][decltype(ic)::value] = rot[(r * M + p) * 4 + decltype(ic)::value]; });
      });
      // along a (the H axis) for every b: constants by p
      dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE {
        constexpr int b = decltype(ib)::value;
        float y[S];
        if constexpr (b < Cfg::VB) {
          dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE { y[decltype(ia)::value] = v[decltype(ia)::value][b]; });
        } else {
          // this column slot of the map landed in LDS (raw image [row][RAWW])
          const int qq = (b % 2 == 0) ? q : M - 1 - q;
          dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE {
            constexpr int a = decltype(ia)::value;
            const int row = a * M + ((a % 2 == 0) ? p : M - 1 - p);
            y[a] = raw[row * Cfg::RAWW + (b - Cfg::VB) * M + qq];
          });
        }
        t2_network<M>(y, rp);
        dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE { v[decltype(ia)::value][b] = y[decltype(ia)::value]; });
        // one network at a time: interleaved for ILP the eight of them run the registers out, and what
        // the allocator then spills are the long-lived samples / parked outputs
        __builtin_amdgcn_sched_barrier(0);
      });
      float rq[NROT][4];
      dcts::static_for<NROT>([&](auto ir) DCTS_LAMBDA_INLINE {
        constexpr int r = decltype(ir)::value;
        dcts::static_for<4>([&](auto ic) DCTS_LAMBDA_INLINE { rq[r][decltype(ic)::value] = rot[(r * M + q) * 4 + decltype(ic)::value]; });
      });
      // along b (the W axis) for every a: constants by q
      dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE {
        constexpr int a = decltype(ia)::value;
        t2_network<M>(v[a], rq);
        // the outputs exist HERE: LLVM otherwise sinks the networks down to the LDS stores of the set that
        // uses them (behind the barrier, interleaved with the stores' address arithmetic: 15 spilled VGPRs,
        // and every scratch reload waits vmcnt(0), i.e. for the next map's loads as well)
        dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE { t2_pin(v[a][decltype(ib)::value]); });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    T2_STAMP(0);
    lds_barrier();  // #4 of the previous map: every consumer is done with set 1 in zbuf
    T2_STAMP(1);
    finish();
    auto write_set = [&](auto iset) DCTS_LAMBDA_INLINE {
      constexpr int SET = decltype(iset)::value;
      int p, q;
      bool ok;
      item_pq(p, q, ok);
      {
        // lanes without an item store unconditionally (a branch around the stores costs registers) into
        // the four pad words behind the block, which nobody reads
        const int dump = M * RS + (launder(lane_in) & 3);
        const int o_aa = ok ? p * RS + q : dump, o_ad = ok ? p * RS + (M - 1 - q) : dump;
        const int o_da = ok ? (M - 1 - p) * RS + q : dump, o_dd = ok ? (M - 1 - p) * RS + (M - 1 - q) : dump;
        dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE {
          constexpr int ra = decltype(ia)::value;
          dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE {
            constexpr int rb = decltype(ib)::value;
            constexpr int bid = ra * S + rb;
            if constexpr (kT2Sch.set_of[bid] == SET) {
              constexpr int a = kT2Plan.slot_of_role[ra], b = kT2Plan.slot_of_role[rb];
              constexpr bool asc_a = kT2Plan.asc_of_role[ra] != 0, asc_b = kT2Plan.asc_of_role[rb] != 0;
              const int off = asc_a ? (asc_b ? o_aa : o_ad) : (asc_b ? o_da : o_dd);
              zbuf[kT2Sch.li_of[bid] * BS + off] = v[a][b];
            }
          });
        });
      }
    };
    if (producer) write_set(std::integral_constant<int, 0>{});
    T2_STAMP(2);
    lds_barrier();  // #1
    T2_STAMP(3);
    float* lo = leaf_out ? leaf_out + m * (long long)Cfg::N * Cfg::N : nullptr;
    const int vid0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, params[(0 * (S * S / 2) + 2 * wave) * 8 + 6]));
    const int vid1 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, params[(1 * (S * S / 2) + 2 * wave) * 8 + 6]));
    // The next map's 64 loads per lane are trickled out during the transforms of BOTH sets, eight
    // per hook point, straight into the registers of this map's outputs that have just gone to
    // LDS: the slots of set 0 while set 0 is transformed, those of set 1 during set 1 (in column
    // order: phase A runs its H-axis networks column by column, which covers what is left of the
    // last loads' latency). 32 parked + 32 landing + 44 for a pass fit the 128 VGPRs of a 16-wave
    // workgroup; so do 64 landing + 44.
    const bool more = m + gridDim.x < nmaps;
    const float* nsrc = tile_in(tb, more ? m + gridDim.x : m, &hint_next);
    const unsigned nbytes = more ? (unsigned)(Cfg::N * Cfg::N * 4) : 0u;  // behind the last map: every load reads out of range
    static_assert(Cfg::hook_begin(3) <= t2_set_count(0, Cfg::VB), "a slot is loaded after its set has gone to LDS");
    auto trickle = [&](auto set) DCTS_LAMBDA_INLINE {
      return [&](auto k) DCTS_LAMBDA_INLINE {
        constexpr int h = 3 * decltype(set)::value + decltype(k)::value;
        constexpr int i0 = Cfg::hook_begin(h), n = Cfg::HOOKS[h];
        if constexpr (decltype(set)::value == 0 && Cfg::DB > 0) {
          // the raw image is free since barrier #1 (phase A has read it): the next map's pieces, one
          // per wave and hook point (49 for 224x224; the 16 waves issue 16 at a time)
          constexpr int kk = decltype(k)::value;
          if (DCTS_T2_EXP != 3 && more)
            for (int i = wave + kT2Waves * kk; i < Cfg::RAW_PIECES; i += 3 * kT2Waves) t2_dma_piece<M>(nsrc, raw, i, launder(lane_in));
        }
        if constexpr (n > 0) {
          int p, q;
          bool ok;
          item_pq(p, q, ok);
          t2_load_seq<M, i0, i0 + n>(nsrc, nbytes, p, q, ok, v);
        }
      };
    };
    float e = t2_consume<M, 0, STORE>(vid0, zbuf, params, wave, launder(lane_in), lo, trickle(std::integral_constant<int, 0>{}));
    asm volatile("" : "+v"(e));
    T2_STAMP(4);
    lds_barrier();  // #2
    T2_STAMP(5);
    if (producer) write_set(std::integral_constant<int, 1>{});
    T2_STAMP(6);
    if constexpr (Cfg::DB > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of the next map's raw image have landed
    lds_barrier();  // #3
    T2_STAMP(7);
    T2_STAMP(8);
    e += t2_consume<M, 1, STORE>(vid1, zbuf, params, wave, launder(lane_in), lo, trickle(std::integral_constant<int, 1>{}));
    asm volatile("" : "+v"(e));
    T2_STAMP(9);
    if constexpr (Cfg::hook_begin(6) < Cfg::NV) {
      asm volatile("" ::: "memory");
      int p, q;
      bool ok;
      item_pq(p, q, ok);
      t2_load_seq<M, Cfg::hook_begin(6), Cfg::NV>(nsrc, nbytes, p, q, ok, v);
      __builtin_amdgcn_sched_barrier(0);
    }
    e = wave_sum_dpp(e);
    if (lane_in == 0) partials[pslot * kT2Waves + wave] = e;
    pending_m = m;
    pending_slot = pslot;
    pslot ^= 1;
    T2_STAMP(10);
  }
  lds_barrier();
  finish();
#ifdef DCTS_T2_STAMPS
  if (lane_in == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_t2_stamps[wave][i], acc_[i]);
#endif
}

template <int M, bool STORE>
__global__ __launch_bounds__(64 * kT2Waves) void k_tile2d(TileBatch tb, float* leaf_out) {
  using Cfg = T2Cfg<M>;
  __shared__ __attribute__((aligned(16))) float zbuf[Cfg::ZSET];
  __shared__ __attribute__((aligned(16))) float raw[Cfg::RAW > 0 ? Cfg::RAW_PIECES * 256 : 4];
  __shared__ __attribute__((aligned(16))) float rot[Cfg::NROT * M * 4];
  __shared__ __attribute__((aligned(16))) float params[2 * (kT2S * kT2S / 2) * 8];
  __shared__ float partials[2 * kT2Waves];
  t2_body<M, TileBatch, STORE>(tb, (lds_ptr)zbuf, (lds_ptr)raw, (lds_ptr)rot, (lds_ptr)params, (lds_ptr)partials, leaf_out);
}

#define DCTS_TILE2D_TABLE(X) X(224, 28)

int t2_num_cus() {
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      n = 256;
    return n;
  }();
  return ncu;
}

template <int M>
int launch_tile2d(const TileBatch& tb, hipStream_t st) {
  const int ncu = t2_num_cus();
  const long long grid = tb.total < ncu ? tb.total : ncu;  // LDS: one workgroup per CU
  hipLaunchKernelGGL((k_tile2d<M, false>), dim3((unsigned)grid), dim3(64 * kT2Waves), 0, st, tb, (float*)nullptr);
  return (int)hipGetLastError();
}

}  // namespace

#ifdef DCTS_T2_DEV
// development entry points (tools/t2_dev.py builds this file alone: seconds instead of minutes)
extern "C" int t2_dev_run(const float* x, long long nmaps, int edge, float* out, void* stream) {
  TileBatch tb;
  for (int i = 0; i < kTileItems; ++i) {
    tb.x[i] = x;
    tb.out[i] = out;
    tb.begin[i] = 0;
  }
  tb.begin[1] = tb.begin[kTileItems] = nmaps;
  tb.map_elems = (long long)edge * edge;
  tb.total = nmaps;
  tb.count = 1;
  return dctsi::dispatch_tile2d(edge, &tb, reinterpret_cast<hipStream_t>(stream));
}
#ifdef DCTS_T2_STAMPS
extern "C" int t2_dev_stamps(unsigned long long* host_out /*[16][16]*/, int reset) {
  if (reset) {
    static unsigned long long zeros[16][16] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_t2_stamps), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_t2_stamps), 16 * 16 * sizeof(unsigned long long));
}
#endif
#endif

namespace dctsi {
int dispatch_tile2d(int N, const void* tile_batch, hipStream_t st) {
  const TileBatch& tb = *static_cast<const TileBatch*>(tile_batch);
#define DCTS_CASE(N_, M_) \
  case N_:                \
    return launch_tile2d<M_>(tb, st);
  switch (N) {
    DCTS_TILE2D_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
// coefficient output (debug / parity): the same kernel with the leaf outputs stored, then the DCT-IV
// add/sub layers above the leaves, the orthonormal scale and the role -> frequency index map
// (k_assemble). `scratch` holds scratch_maps tiles; maps go through it in chunks.
int dispatch_tile2d_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps,
                          hipStream_t st) {
  if (N != 224) return DCTS_E_UNSUPPORTED;
  if (!scratch || scratch_maps < 1) return DCTS_E_WORKSPACE;
  constexpr int M = 28;
  for (long long m0 = 0; m0 < nmaps; m0 += scratch_maps) {
    const long long nb = (nmaps - m0) < scratch_maps ? (nmaps - m0) : scratch_maps;
    TileBatch tb;
    for (int i = 0; i < kTileItems; ++i) {
      tb.x[i] = x + m0 * (long long)N * N;
      tb.out[i] = nullptr;  // the STORE instantiation writes no energies
      tb.begin[i] = 0;
    }
    tb.begin[1] = tb.begin[kTileItems] = nb;
    tb.map_elems = (long long)N * N;
    tb.total = nb;
    tb.count = 1;
    const long long grid = nb < t2_num_cus() ? nb : t2_num_cus();  // as the energy launch: one persistent workgroup per CU
    hipLaunchKernelGGL((k_tile2d<M, true>), dim3((unsigned)grid), dim3(64 * kT2Waves), 0, st, tb, scratch);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    rc = launch_assemble<M, kT2L, false>(scratch, nb, out + m0 * (long long)N * N, st);
    if (rc) return rc;
  }
  return DCTS_OK;
}
}  // namespace dctsi
