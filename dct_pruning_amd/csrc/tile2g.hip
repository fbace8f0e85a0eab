// tile2g.hip - mid-size tiles (edge N = 2^L * M: 72 ... 160) as a 2-D radix-2^L split with G MAPS PER ROUND.
//
// Replaces, for these tile shapes, the per-map loop of the reference hooks
// (utils/common.py:262-277: dct.dct_2d(output[i,j,:,:], norm='ortho'), then sum(coeff^2)).
//
// tile2d.hip does this for 224 x 224 (one map per round: 784 items of 8 x 8 mirrored samples fill 14 of the
// 16 waves, two 28 x 28 leaf blocks per wave pass). The same structure at 144 x 144 has 324 items - five
// waves busy in the network phase - and 18 x 18 leaf blocks that leave a wave pass a third full, which is why
// the strip / park kernels of dct_kernels.hip (k_split_fused) kept these shapes at 0.31-0.42 of the HBM peak:
// three LDS writes + four reads per point and five workgroup barriers per 64-column strip.
// Here a ROUND is G maps (G = floor(64 / M): 3 at 144, 4 at 128 and 112, 3 at 72 ...):
//   - an item is (g, p, q): the 2^L x 2^L mirrored samples x[g][a*M + p~][b*M + q~]; one lane loads them
//     (buffer loads: one wave-uniform descriptor over the G maps, four lane offsets, the slot offset as the
//     scalar operand), runs the role network of split_roles.hpp along a, then along b, in registers, and holds
//     one input sample of each of the 4^L leaf blocks Z[g][ra][rb] of its map: G * M * M items fill the lanes;
//   - a wave pass transforms THE SAME leaf block (ra, rb) of the G maps, one map per lane group: lane =
//     column, M-point codelet (DCT-II or DCT-IV by the block's role types: wave-uniform, no divergence and no
//     pairing of unequal blocks), results back into the column; lane = row, second codelet, squares with the
//     block's weights. Image strides (G2Layout) keep both the column and the row accesses conflict-free.
// LDS traffic: 2 writes + 2 reads per point; workgroup barriers: 4 per G maps (2 where all 4^L blocks of the
// G maps fit the LDS at once). With L = 3 the blocks go through LDS in two sets, the second set's 32 values
// per lane parked in the producer's registers meanwhile (as in tile2d.hip); the next round's samples land in
// the registers a set has just vacated, trickled out between the codelets. L = 2 (72, 80: 16 samples per lane, 64 VGPRs,
// one set) runs TWO workgroups per CU. The loads are single-dword gathers, so a tensor needs no 16-byte alignment, and the
// cv2 path's odd front pad (71 / 79 / 143 / 159 -> 72 / 80 / 144 / 160) is the PAD instantiation of the same kernel.
//
// As in the split family the last add/sub layer of every DCT-IV node above the leaves is folded into the
// reduction ((a+b)^2 + (a-b)^2 = 2a^2 + 2b^2, SplitNode::wt): energy-path-only shortcut;
// dcts_dct2d_f32_ex(DCTS_ALGO_TILE2D) runs the STORE instantiation, which writes the leaf outputs, and
// k_assemble applies that layer explicitly - how the tests compare this kernel's coefficients with the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dctscore.h"
#include "split_roles.hpp"

namespace dctsi {
int dispatch_tile2g(int N, const void* tile_batch, hipStream_t st);
int dispatch_tile2g_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps,
                          hipStream_t st);
int has_tile2g(int N);
int has_tile2g_pad(int N);
int dispatch_tile2g_pad(int N, const void* tile_batch, hipStream_t st);
}  // namespace dctsi

namespace {

constexpr int kG2Waves = 16;

// LDS image of one leaf block: M rows of RS floats, BS floats per block. A pass holds PB blocks, block gl on
// lanes [gl * LWA, gl * LWA + M) while columns are accessed (axis A) and on [gl * LWB, gl * LWB + M) for
// rows (axis B); lanes outside repeat the address of the nearest active lane of their 32-lane LDS group
// (a broadcast). Found by exhaustive search (tools/g2_layout_search.py, which also re-checks the ones below):
// ds_read/write_b32 conflict-free for every row and column index.
template <int M, int PB>
struct G2Layout;
template <>
struct G2Layout<18, 3> {
  static constexpr int RS = 19, BS = 342, LWA = 22, LWB = 18;
};
template <>
struct G2Layout<16, 4> {
  static constexpr int RS = 17, BS = 272, LWA = 16, LWB = 16;
};
template <>
struct G2Layout<14, 4> {
  static constexpr int RS = 14, BS = 207, LWA = 15, LWB = 16;
};
template <>
struct G2Layout<20, 2> {
  static constexpr int RS = 21, BS = 420, LWA = 32, LWB = 20;
};
template <>
struct G2Layout<12, 5> {
  static constexpr int RS = 17, BS = 204, LWA = 12, LWB = 12;
};
template <>
struct G2Layout<20, 3> {
  static constexpr int RS = 25, BS = 500, LWA = 20, LWB = 20;
};

// lane -> (block of the pass, row / column index) with the idle lanes folded onto active ones
struct G2LaneMap {
  signed char g[64] = {}, j[64] = {}, act[64] = {};
  constexpr G2LaneMap(int M, int PB, int LW) {
    bool a[64] = {};
    for (int l = 0; l < 64; ++l) {
      const int gg = l / LW, jj = l - gg * LW;
      a[l] = gg < PB && jj < M;
    }
    for (int l = 0; l < 64; ++l) {
      int src = l;
      if (!a[l]) {
        const int lo = (l / 32) * 32;
        int best = -1;
        for (int d = 1; d < 32 && best < 0; ++d) {  // nearest active lane of the same 32-lane group, lower first
          if (l - d >= lo && a[l - d]) best = l - d;
          else if (l + d < lo + 32 && a[l + d]) best = l + d;
        }
        if (best < 0)
          for (int t = 63; t >= 0 && best < 0; --t)
            if (a[t]) best = t;
        src = best;
      }
      g[l] = (signed char)(src / LW);
      j[l] = (signed char)(src - (src / LW) * LW);
      act[l] = a[l] ? 1 : 0;
    }
  }
};

template <int L, int M, int G>
struct G2Cfg {
  static constexpr int S = 1 << L, N = M * S, NB = S * S, NW = kG2Waves;
  static constexpr int PB = G;  // a pass = one block id of the G maps of the round
  static_assert(PB * M <= 64, "lane groups of a pass");
  using Lay = G2Layout<M, PB>;
  static constexpr int RS = Lay::RS, BS = Lay::BS, LWA = Lay::LWA, LWB = Lay::LWB;
  static constexpr int ITEMS = G * M * M;
  static_assert(ITEMS <= 64 * NW, "one item per lane");
  // all 4^L blocks of the G maps at once where they fit beside the tables, else two sets (L = 3)
  static constexpr int NSETS = ((long long)NB * G * BS * 4 <= 150 * 1024) ? 1 : 2;
  static constexpr int NBS = NB / NSETS;  // blocks (= passes) per set
  static_assert(NBS % NW == 0, "passes per wave");
  static constexpr int PPW = NBS / NW;    // passes per wave and set
  static constexpr int ZSET = NBS * G * BS + 4;
  static_assert((long long)ZSET * 4 <= 156 * 1024, "LDS");
  static constexpr int NROT = RolePlan<L>{}.nrot;
  static constexpr int SLOTS = S * S, PER_SET = SLOTS / 2 / NSETS;  // samples per lane; slot pairs per lane and set
  // 8-byte loads shared by neighbouring lanes (g2_load_pairs, DCTS_G2_PAIR=1): half the load instructions, but
  // measured SLOWER on the same box for every shape (144 x 144: 45.3 -> 42.6 %, 128: 53.3 -> 50.0, 72: 49.6 -> 45.9 of
  // the HBM peak at 8192 / 8192 / 32768 maps): the DPP exchange and the select instructions cost more than the load
  // issue they save. Off.
#ifndef DCTS_G2_PAIR
  static constexpr bool PAIR = false;
#else
  static constexpr bool PAIR = (DCTS_G2_PAIR != 0);
#endif
};

// Which set a block is in, its position in the set and the order of the passes: blocks sorted by the cost of
// their codelets (DCT-IV dearer than DCT-II), dealt to the sets alternately, and inside a set to the waves in
// snake order (wave w takes positions w, 2*NW-1-w, 2*NW+w, ...: every wave's passes add up about evenly).
template <int L, int NSETS>
struct G2Sched {
  static constexpr int S = 1 << L, NB = S * S, NBS = NB / NSETS, NW = kG2Waves, PPW = NBS / NW;
  int blk[NSETS][NBS] = {};  // [set][li] -> ra * S + rb
  int set_of[NB] = {}, li_of[NB] = {};
  constexpr G2Sched() {
    constexpr RolePlan<L> plan{};
    int order[NB] = {}, cost[NB] = {};
    for (int b = 0; b < NB; ++b) {
      order[b] = b;
      cost[b] = plan.is4_of_role[b / S] + plan.is4_of_role[b % S];
    }
    for (int i = 1; i < NB; ++i)  // insertion sort by (cost, id): stable, constexpr-friendly
      for (int k = i; k > 0 && (cost[order[k]] < cost[order[k - 1]]); --k) {
        const int t = order[k];
        order[k] = order[k - 1];
        order[k - 1] = t;
      }
    // Two sets: by the parity of the block's row slot a. A lane loads the samples of the slots (a, b) and
    // (a + 2, b) with ONE 8-byte instruction shared with its neighbour lane (g2_load_pairs): both must belong
    // to the same set, so that their registers are vacated together.
    int fill[NSETS] = {};
    for (int i = 0; i < NB; ++i) {
      const int s = (NSETS == 1) ? 0 : (plan.slot_of_role[order[i] / S] & 1), rank = fill[s]++;  // rank-th cheapest block of set s
      const int row = rank / NW, col = rank % NW;
      const int li = row * NW + ((row % 2 == 0) ? col : NW - 1 - col);  // snake: pass `row` of wave li % NW
      blk[s][li] = order[i];
      set_of[order[i]] = s;
      li_of[order[i]] = li;
    }
  }
};

template <int L, int NSETS>
inline constexpr G2Sched<L, NSETS> kG2Sched{};

// rotation constants (c, s, sigma*c, sigma*s), sigma = (-1)^j of the pair index: [rot][p][4]
template <int L, int M>
struct G2RotTable {
  static constexpr int NROT = RolePlan<L>{}.nrot;
  float v[NROT > 0 ? NROT : 1][M][4] = {};
  constexpr G2RotTable() {
    constexpr RotTable<M, L> t{};
    for (int r = 0; r < NROT; ++r)
      for (int p = 0; p < M; ++p) {
        const float sg = RotTable<M, L>::sign0(r) * ((p & 1) ? -1.f : 1.f);
        v[r][p][0] = t.c[r][p];
        v[r][p][1] = t.s[r][p];
        v[r][p][2] = sg * t.c[r][p];
        v[r][p][3] = sg * t.s[r][p];
      }
  }
};
template <int L, int M>
__device__ const G2RotTable<L, M> kG2Rot{};

// per (set, li): leaf types and squared amplitude weights of the block (8 words)
struct G2BlockParam {
  int tA, tB, block, pad;
  float wA0, wA1, wB0, wB1;  // squared weights of output 0 / outputs > 0 per axis
};
template <int L, int M, int NSETS>
struct G2ParamTable {
  G2BlockParam v[NSETS][(1 << (2 * L)) / NSETS] = {};
};
template <int L, int M, int R>
constexpr void g2_role_weights(float& w0, float& w1) {
  using Leaf = typename RoleLeaf<(M << L), L, R>::type;
  const double a = Leaf::wt(true), b = Leaf::wt(false);
  w0 = float(a * a);
  w1 = float(b * b);
}
template <int L, int M, int NSETS, int... R>
constexpr G2ParamTable<L, M, NSETS> g2_make_params(std::integer_sequence<int, R...>) {
  constexpr int S = 1 << L;
  constexpr RolePlan<L> plan{};
  constexpr G2Sched<L, NSETS> sch{};
  float w0[S] = {}, w1[S] = {};
  (g2_role_weights<L, M, R>(w0[R], w1[R]), ...);
  G2ParamTable<L, M, NSETS> t{};
  for (int s = 0; s < NSETS; ++s)
    for (int li = 0; li < S * S / NSETS; ++li) {
      const int ra = sch.blk[s][li] / S, rb = sch.blk[s][li] % S;
      t.v[s][li].block = sch.blk[s][li];
      t.v[s][li].tA = plan.is4_of_role[ra];
      t.v[s][li].tB = plan.is4_of_role[rb];
      t.v[s][li].wA0 = w0[ra];
      t.v[s][li].wA1 = w1[ra];
      t.v[s][li].wB0 = w0[rb];
      t.v[s][li].wB1 = w1[rb];
    }
  return t;
}
template <int L, int M, int NSETS>
__device__ const G2ParamTable<L, M, NSETS> kG2Params = g2_make_params<L, M, NSETS>(std::make_integer_sequence<int, (1 << L)>{});

template <int M, int PB, int LW>
__device__ const G2LaneMap kG2Lanes{M, PB, LW};

// Per-map sums: every lane leaves its sum in LDS and wave 0 adds the sixteen waves' values per lane (fixed order) and runs ONE
// segmented reduction over the rows of a block after the next barrier - instead of a five-step shuffle reduction in each of the
// sixteen waves at the end of every round. Same box, % of the HBM peak: 72 x 72 38.7 -> 41.9 (5760 maps), 49.6 -> 51-52 (32768);
// 80: 41.7 -> 43.4; 144: 41.5 -> 42.4 (4992), 41 -> 43-44 (8192); 160: 39.8 -> 40.2. Bit-reproducible either way.
#ifndef DCTS_G2_LATE_REDUCE
#define DCTS_G2_LATE_REDUCE 1
#endif
// Waves without items skip the network and set-store phases, and the last of them takes the per-map sums: same box,
// 160: 39.5 -> 40.4 %, 80: 42.9 -> 43.6 %, 112: 35.5 -> 36.0 % of the HBM peak (144 / 72 have items on every wave: unchanged path).
#ifndef DCTS_G2_SKIP_IDLE
#define DCTS_G2_SKIP_IDLE 1
#endif
#ifndef DCTS_G2_SKEW
#define DCTS_G2_SKEW 0
#endif
#ifndef DCTS_G2_EXP
#define DCTS_G2_EXP 0  // timing experiments (wrong results): 1 no codelet arithmetic, 2 no LDS traffic behind axis A's reads, 3 no loads of the next round
#endif
#ifdef DCTS_G2_STAMPS
__device__ unsigned long long g_g2_stamps[16][16];
#define G2_STAMP(slot)                                                            \
  do {                                                                            \
    unsigned long long t_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
    __builtin_amdgcn_sched_barrier(0);                                            \
    acc_[slot] += t_ - last_;                                                     \
    last_ = t_;                                                                   \
  } while (0)
#else
#define G2_STAMP(slot) ((void)0)
#endif

__device__ __forceinline__ void g2_pin(float& x) { asm volatile("" : "+v"(x)); }

// the L-level role network on 2^L values held in registers: y[slot], constants by lane
template <int L, int NROT>
__device__ __forceinline__ void g2_network(float (&y)[1 << L], const float (&rc)[NROT > 0 ? NROT : 1][4]) {
  constexpr RolePlan<L> plan{};
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

// the group of G maps a round works on: all of one tensor (a TileBatch may hold several)
struct G2Group {
  const float* base;  // first map of the group (wave-uniform)
  float* out;         // its energy
  int count;          // maps in the group (<= G; the last group of a tensor may be short)
};
// The launch's groups: tensor t owns groups [gbegin[t], gbegin[t + 1]) (host-computed; a TileBatch holds up to 32 tensors).
// Which tensor a group belongs to is found WITHOUT memory: lane t keeps gbegin[t] in a register for the whole launch,
// and the tensor of group grp is the number of lanes with gbegin <= grp, minus one (one compare + ballot). The first
// version walked the table with dependent scalar loads every round: a third of a 72 x 72 round when sixteen small
// tensors share the launch (U2-Net-p's step: 61 us for what one tensor of the same size takes 38 us).
struct G2Batch {
  TileBatch tb;
  int gbegin[kTileItems + 1];  // gbegin[count] = all groups
};
template <int G>
__device__ __forceinline__ G2Group g2_group(const G2Batch& gb, int grp, int gbeg_lane) {
  const unsigned long long ge = __builtin_amdgcn_ballot_w64(gbeg_lane <= grp);
  const int t = __builtin_amdgcn_readfirstlane(__builtin_popcountll(ge) - 1);
  const TileBatch& tb = gb.tb;
  const long long nt = tb.begin[t + 1] - tb.begin[t];
  const long long m0 = (long long)(grp - gb.gbegin[t]) * G;
  G2Group r;
  const unsigned long long a = reinterpret_cast<unsigned long long>(tb.x[t] + m0 * tb.map_elems);
  r.base = reinterpret_cast<const float*>(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)a));
  r.out = tb.out[t] ? tb.out[t] + m0 : nullptr;
  const long long left = nt - m0;
  r.count = __builtin_amdgcn_readfirstlane((int)(left < G ? left : G));
  return r;
}

// Load order of the sample slot PAIRS {(a, b), (a + 2, b)}, a & 2 == 0, as a * S + b: set 0's (rows a even),
// then set 1's (a odd) - a pair's registers are free for the next round's samples once its set has gone to LDS.
// Inside a set a-major: consecutive loads of a lane read the SAME rows of the map (slots b, b + 1 ... of row
// a*M + p~ share cache lines: M floats are less than a line), so most of them hit the CU's vector cache
// (72 x 72: 45 -> 49 % of the HBM peak against b-major).
template <int L, int NSETS>
struct G2LoadOrder {
  static constexpr int S = 1 << L, NP = S * S / 2;
  int pair[NP] = {};
  constexpr G2LoadOrder() {
    int n = 0;
    for (int set = 0; set < NSETS; ++set)
      for (int a = 0; a < S; ++a)
        for (int b = 0; b < S; ++b)
          if ((a & 2) == 0 && (NSETS == 1 || (a & 1) == set)) pair[n++] = a * S + b;
  }
};
template <int L, int NSETS>
inline constexpr G2LoadOrder<L, NSETS> kG2LoadOrder{};

typedef float g2_v2f __attribute__((ext_vector_type(2)));
constexpr int g2_pi(int a) { return (a & 1) | ((a & 4) >> 1); }  // which register pair holds row slot a (.x: a & 2 == 0, .y: the slot two rows on)

// Loads of the item pair (g, p, 2k), (g, p, 2k + 1) - two neighbouring lanes - for the slot pairs [I0, I1) of the load
// order. Single dwords per lane cost the vector-memory pipeline ~10 cycles per wave instruction here (every
// instruction touches 4-7 cache lines: M-float row segments), and 64 of them per lane and round kept the waves
// blocked on load issue for a quarter of the time. The columns of the two items are adjacent in memory, so the EVEN
// lane fetches both items' samples of slot (a, b) with one 8-byte load and the ODD lane those of slot (a + 2, b) with
// the same instruction (its lane offset points two row slots further): half the instructions. g2_exchange() hands
// each lane its own two samples afterwards. Reads beyond the group's `bytes` return 0 (lanes without an item, maps
// beyond a short group, "no next group").
// PAD (the cv2 path of torch2dct, utils/common.py:235-236, for an odd H: one zero row AND one zero column in front): the map
// in memory is (N-1) x (N-1) with row pitch N-1; sample (r, c) of the padded tile is x[r-1][c-1], and row 0 / column 0 are
// zeros. The descriptor's base is moved N floats in front of the group's first map, so that offset (r * (N-1) + c) * 4
// addresses x[r-1][c-1]; row 0 is only touched by row slot a = 0 of the lanes with p = 0, column 0 by column slot b = 0 of
// the lanes with q = 0: those lanes get an out-of-range offset for those slots (zeros, no request) - G2Voffs::a0 / b0 / ab.
struct G2Voffs {
  int ee, eo, oe, oo;      // (row slot parity, column slot parity): ascending / descending p~, q~
  int ee_a0, eo_a0;        // PAD: row slot 0     (zero row for p == 0)
  int ee_b0, oe_b0;        // PAD: column slot 0  (zero column for q == 0)
  int ee_ab;               // PAD: slot (0, 0)
};
template <int L, int M, int G, int NSETS, bool PAIR, int PAD, int I0, int I1>
__device__ __forceinline__ void g2_load_pairs(const float* base, unsigned bytes, const G2Voffs& vo, g2_v2f (&vp)[(1 << L) / 2][1 << L]) {
  constexpr int S = 1 << L, N = M * S, NP = N - PAD;  // NP: row pitch in memory
  static_assert(!(PAIR && PAD), "the paired loads assume aligned column pairs");
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) - PAD * (NP + 1), 0,
                                                                      bytes ? bytes + PAD * (NP + 1) * 4 : 0u, 0x00020000);
  auto pick = [&](auto ia, auto ib) DCTS_LAMBDA_INLINE -> int {
    constexpr int a = decltype(ia)::value, b = decltype(ib)::value;
    if constexpr (PAD != 0 && a == 0 && b == 0) return vo.ee_ab;
    if constexpr (PAD != 0 && a == 0) return (b % 2 == 0) ? vo.ee_a0 : vo.eo_a0;
    if constexpr (PAD != 0 && b == 0) return (a % 2 == 0) ? vo.ee_b0 : vo.oe_b0;
    return (a % 2 == 0) ? ((b % 2 == 0) ? vo.ee : vo.eo) : ((b % 2 == 0) ? vo.oe : vo.oo);
  };
  dcts::static_for<(I1 > I0 ? I1 - I0 : 0)>([&](auto ii) DCTS_LAMBDA_INLINE {
    constexpr int sl = kG2LoadOrder<L, NSETS>.pair[I0 + decltype(ii)::value];
    constexpr int a = sl / S, b = sl % S;
    if constexpr (PAIR) {
      vp[g2_pi(a)][b] = __builtin_bit_cast(g2_v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, pick(std::integral_constant<int, a>{}, std::integral_constant<int, b>{}),
                                                                                         (a * M * NP + b * M) * 4, 0));
    } else {  // every lane loads its own two samples of the slot pair, one dword each (the offsets point at its own column)
      vp[g2_pi(a)][b].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          rs, pick(std::integral_constant<int, a>{}, std::integral_constant<int, b>{}), (a * M * NP + b * M) * 4, 0));
      vp[g2_pi(a)][b].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          rs, pick(std::integral_constant<int, a + 2>{}, std::integral_constant<int, b>{}), ((a + 2) * M * NP + b * M) * 4, 0));
    }
  });
}

// after the loads have landed: the even lane holds {its own, its neighbour's} sample of slot (a, b), the odd lane
// {its neighbour's, its own} of slot (a + 2, b) (mirrored column slots b: the other way round); one DPP swap per
// pair leaves both lanes with .x = their sample of (a, b) and .y = that of (a + 2, b)
template <int B>
__device__ __forceinline__ void g2_exchange(g2_v2f& r, bool odd) {
  const float X = r.x, Y = r.y;
  const float give = (B % 2 == 0) ? (odd ? X : Y) : (odd ? Y : X);
  const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
  if constexpr (B % 2 == 0) {
    r.x = odd ? got : X;
    r.y = odd ? Y : got;
  } else {
    r.x = odd ? got : Y;
    r.y = odd ? X : got;
  }
}

// one leaf pass: PB blocks (the same (ra, rb) of the PB maps), both axes; returns the lane's weighted energy
// (the lane maps are read ONCE per kernel and kept packed in a register each - g | j << 8 | act << 16: a table
// load inside the passes would be a vector-memory load, and its s_waitcnt vmcnt would also wait for every
// prefetch of the next round in flight)
template <int L, int M, int G, int TA, int TB, bool STORE>
__device__ __forceinline__ float g2_pass(lds_ptr zset, lds_cptr pp, int li, int map_a, int map_b, float* leaf_out, long long map0,
                                         int count) {
  using Cfg = G2Cfg<L, M, G>;
  constexpr int RS = Cfg::RS, BS = Cfg::BS, PB = Cfg::PB;
  // ---- axis A: lane = column q', transform along p', results back into the column --------------------------
  {
    const int gl = map_a & 0xff, j = (map_a >> 8) & 0xff;
    const bool act = (map_a >> 16) != 0;
    lds_ptr col = zset + (li * G + gl) * BS + j;
    float in[M], o[M];
    dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { in[decltype(i)::value] = col[decltype(i)::value * RS]; });
#if DCTS_G2_EXP == 1
    dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { o[decltype(i)::value] = in[decltype(i)::value] * 1.5f; });
#else
    if constexpr (TA)
      dcts::Dct4<M>::run(in, o);
    else
      dcts::Dct2<M>::run(in, o);
#endif
#if DCTS_G2_EXP == 2
    dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { g2_pin(o[decltype(i)::value]); });
#else
    if (act) dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { col[decltype(i)::value * RS] = o[decltype(i)::value]; });
#endif
  }
  // the wave's own LDS traffic is in order; only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- axis B: lane = row k1, transform along q', squares --------------------------------------------------
  const int gl = map_b & 0xff, j = (map_b >> 8) & 0xff;
  const bool act = (map_b >> 16) != 0;
  lds_cptr row = zset + (li * G + gl) * BS + j * RS;
  float z[M], w[M];
#if DCTS_G2_EXP == 2
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { z[decltype(i)::value] = __builtin_bit_cast(float, launder(map_b + decltype(i)::value)); });
#else
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { z[decltype(i)::value] = row[decltype(i)::value]; });
#endif
#if DCTS_G2_EXP == 1
  dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { w[decltype(i)::value] = z[decltype(i)::value] * 1.5f; });
#else
  if constexpr (TB)
    dcts::Dct4<M>::run(z, w);
  else
    dcts::Dct2<M>::run(z, w);
#endif
  if constexpr (STORE) {
    // debug / parity path: leaf outputs, unweighted, as [ra * M + k1][rb * M + k2] (k_assemble's layout)
    const int blk_id = __builtin_bit_cast(int, pp[2]);
    if (act && gl < count) {  // a short last group: the lanes of the missing maps hold zeros and have no tile to write
      float* o = leaf_out + ((map0 + gl) * (M << L) + (blk_id >> L) * M + j) * (long long)(M << L) + (blk_id & ((1 << L) - 1)) * M;
      dcts::static_for<M>([&](auto i) DCTS_LAMBDA_INLINE { o[decltype(i)::value] = w[decltype(i)::value]; });
    }
  }
  const float t0 = w[0] * w[0];
  float t1 = 0.f;
  dcts::static_for<M - 1>([&](auto i) DCTS_LAMBDA_INLINE { t1 = fmaf(w[decltype(i)::value + 1], w[decltype(i)::value + 1], t1); });
  const float e = (j == 0 ? pp[4] : pp[5]) * (pp[6] * t0 + pp[7] * t1);
  return act ? e : 0.f;
}

template <int L, int M, int G, bool STORE>
__device__ __forceinline__ float g2_pass_dispatch(int vid, lds_ptr zset, lds_cptr pp, int li, int map_a, int map_b, float* leaf_out,
                                                  long long map0, int count) {
  // wave-uniform: the G blocks of a pass share their role types
  if (vid == 0) return g2_pass<L, M, G, 0, 0, STORE>(zset, pp, li, map_a, map_b, leaf_out, map0, count);
  if (vid == 1) return g2_pass<L, M, G, 0, 1, STORE>(zset, pp, li, map_a, map_b, leaf_out, map0, count);
  if (vid == 2) return g2_pass<L, M, G, 1, 0, STORE>(zset, pp, li, map_a, map_b, leaf_out, map0, count);
  return g2_pass<L, M, G, 1, 1, STORE>(zset, pp, li, map_a, map_b, leaf_out, map0, count);
}

template <int L, int M, int G, bool STORE, int PAD>
__device__ __forceinline__ void g2_body(const G2Batch& gb, lds_ptr zbuf, lds_ptr rot, lds_ptr params, lds_ptr partials, float* leaf_out) {
  using Cfg = G2Cfg<L, M, G>;
  constexpr int S = Cfg::S, N = Cfg::N, RS = Cfg::RS, BS = Cfg::BS, NROT = Cfg::NROT, NSETS = Cfg::NSETS, NBS = Cfg::NBS,
                PPW = Cfg::PPW, NW = kG2Waves;
  constexpr RolePlan<L> plan{};
  const int lane_in = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DCTS_G2_STAMPS
  unsigned long long acc_[16] = {}, last_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
  // tables -> LDS
  if constexpr (NROT > 0)
    for (int i = threadIdx.x; i < NROT * M * 4; i += blockDim.x) rot[i] = (&kG2Rot<L, M>.v[0][0][0])[i];
  for (int i = threadIdx.x; i < NSETS * NBS; i += blockDim.x) {
    const G2BlockParam& bp = (&kG2Params<L, M, NSETS>.v[0][0])[i];
    params[i * 8 + 0] = __builtin_bit_cast(float, bp.tA);
    params[i * 8 + 1] = __builtin_bit_cast(float, bp.tB);
    params[i * 8 + 2] = __builtin_bit_cast(float, bp.block);
    params[i * 8 + 3] = 0.f;
    params[i * 8 + 4] = bp.wA0;
    params[i * 8 + 5] = bp.wA1;
    params[i * 8 + 6] = bp.wB0;
    params[i * 8 + 7] = bp.wB1;
  }
  // lane maps of the leaf passes (columns / rows), packed
  int map_a, map_b;
  {
    const G2LaneMap& la = kG2Lanes<M, Cfg::PB, Cfg::LWA>;
    const G2LaneMap& lb = kG2Lanes<M, Cfg::PB, Cfg::LWB>;
    map_a = (int)la.g[lane_in] | ((int)la.j[lane_in] << 8) | ((int)la.act[lane_in] << 16);
    map_b = (int)lb.g[lane_in] | ((int)lb.j[lane_in] << 8) | ((int)lb.act[lane_in] << 16);
  }
  // this lane's item (fixed for the whole launch)
  const int idx = wave * 64 + lane_in;
  const bool ok = idx < Cfg::ITEMS;
  const int ig = ok ? idx / (M * M) : 0;
  const int ir = ok ? idx - ig * (M * M) : 0;
  const int ip = ir / M, iq = ir - ip * M;
  // lane offsets of the four mirrored quadrant kinds (with DCTS_G2_PAIR, 8-byte pairs: the even column of the lane pair, odd
  // lanes two row slots further, see g2_load_pairs); lanes without an item read out of range (zeros, no traffic)
  static_assert(M % 2 == 0, "lane pairs share a row");
  static_assert(!(STORE && PAD), "the coefficient path takes unpadded tiles");
  constexpr int kOut = 0x7ffffff0;
  constexpr int NPITCH = N - PAD, MAPB = NPITCH * NPITCH * 4;  // row pitch (floats) and bytes of a map in memory
  // (`cnt`: maps in the group the offsets are for. Lanes of maps beyond a short group are out of range by their own
  // offset - with PAD the descriptor is N floats longer than the group's maps, so the range check alone would let the
  // first row of the missing map through, up to a map's length past the end of the tensor once the slot offset is added)
  auto voffs = [&](G2Voffs& vo, int cnt) DCTS_LAMBDA_INLINE {
    const int p = launder(ip), q = launder(iq), g = launder(ig);
    const bool ok = launder(idx) < Cfg::ITEMS && g < cnt;
    if constexpr (Cfg::PAIR) {
      const int q2 = q & ~1;
      const int gbo = g * MAPB + ((q & 1) ? 2 * M * NPITCH * 4 : 0);
      vo.ee = ok ? gbo + (p * NPITCH + q2) * 4 : kOut;
      vo.eo = ok ? gbo + (p * NPITCH + (M - 2 - q2)) * 4 : kOut;
      vo.oe = ok ? gbo + ((M - 1 - p) * NPITCH + q2) * 4 : kOut;
      vo.oo = ok ? gbo + ((M - 1 - p) * NPITCH + (M - 2 - q2)) * 4 : kOut;
    } else {
      const int gbo = g * MAPB;
      vo.ee = ok ? gbo + (p * NPITCH + q) * 4 : kOut;
      vo.eo = ok ? gbo + (p * NPITCH + (M - 1 - q)) * 4 : kOut;
      vo.oe = ok ? gbo + ((M - 1 - p) * NPITCH + q) * 4 : kOut;
      vo.oo = ok ? gbo + ((M - 1 - p) * NPITCH + (M - 1 - q)) * 4 : kOut;
    }
    if constexpr (PAD != 0) {
      vo.ee_a0 = p == 0 ? kOut : vo.ee;
      vo.eo_a0 = p == 0 ? kOut : vo.eo;
      vo.ee_b0 = q == 0 ? kOut : vo.ee;
      vo.oe_b0 = q == 0 ? kOut : vo.oe;
      vo.ee_ab = (p == 0 || q == 0) ? kOut : vo.ee;
    }
  };
  const int ngroups = gb.gbegin[gb.tb.count];
  // lane t: first group of tensor t (lanes beyond the tensors: never <= a group index)
  const int gbeg_lane = lane_in < gb.tb.count ? gb.gbegin[lane_in < kTileItems ? lane_in : 0] : 0x7fffffff;
  int grp = blockIdx.x;
  g2_v2f vp[S / 2][S];  // the lane's S x S samples: vp[g2_pi(a)][b].x = slot (a, b) for a & 2 == 0, .y = slot (a + 2, b)
  auto vget = [&](auto ia, auto ib) DCTS_LAMBDA_INLINE -> float {
    constexpr int a = decltype(ia)::value, b = decltype(ib)::value;
    if constexpr (a & 2)
      return vp[g2_pi(a)][b].y;
    else
      return vp[g2_pi(a)][b].x;
  };
  auto vset = [&](auto ia, auto ib, float val) DCTS_LAMBDA_INLINE {
    constexpr int a = decltype(ia)::value, b = decltype(ib)::value;
    if constexpr (a & 2)
      vp[g2_pi(a)][b].y = val;
    else
      vp[g2_pi(a)][b].x = val;
  };
  G2Group cur = g2_group<G>(gb, grp, gbeg_lane);  // grid <= ngroups
  {
    G2Voffs vo;
    voffs(vo, cur.count);
    g2_load_pairs<L, M, G, NSETS, Cfg::PAIR, PAD, 0, S * S / 2>(cur.base, (unsigned)(cur.count * MAPB), vo, vp);
  }
  __builtin_amdgcn_sched_barrier(0);
  long long pending = -1;  // a round whose partials wait for the workgroup sum
  float* pending_out = nullptr;
  int pending_count = 0, pslot = 0, pending_slot = 0;
  auto finish = [&]() DCTS_LAMBDA_INLINE {
    if (pending >= 0) {
#if DCTS_G2_LATE_REDUCE
      // ONE wave - the last, which holds the fewest items or none: per lane the sixteen waves' sums in fixed order, then ONE
      // segmented reduction over the rows of a block
      // (wave 0 where every wave has items: 144 / 72 lose 2-3 % with the duty on a wave that also has a full share of items)
      if (wave == ((DCTS_G2_SKIP_IDLE && Cfg::ITEMS <= (NW - 1) * 64) ? NW - 1 : 0)) {
        const int mb = launder(map_b);
        const int j = (mb >> 8) & 0xff;
        const bool act = (mb >> 16) != 0;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) t += partials[(pending_slot * NW + i) * 64 + launder(lane_in)];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          if (off < M) {
            const float u = __shfl_down(t, off, 64);
            if (act && j + off < M) t += u;
          }
        }
        constexpr float sc = float(4.0 / (double(N) * double(N)));
        if constexpr (!STORE)
          if (act && j == 0 && (mb & 0xff) < pending_count) pending_out[mb & 0xff] = t * sc;
      }
#else
      if (wave == 0 && lane_in < pending_count) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) t += partials[(pending_slot * NW + i) * G + lane_in];
        constexpr float sc = float(4.0 / (double(N) * double(N)));
        if constexpr (!STORE) pending_out[lane_in] = t * sc;
      }
#endif
      pending = -1;
    }
  };
  lds_barrier();  // tables are in LDS
  G2_STAMP(15);
  for (; grp < ngroups; grp += gridDim.x) {
    // ---- A: role networks of this lane's item, both axes, in registers -----------------------------------
    // (G * M * M items on 1024 lanes: the last waves may hold none at all - 800 items at 160 and 80, 784 at 112 - and skip
    // the networks and the stores of the sets: wave-uniform, and three waves fewer compete for the VALUs and the LDS there)
    if (DCTS_G2_SKIP_IDLE == 0 || wave * 64 < Cfg::ITEMS) {
      const int p = launder(ip), q = launder(iq);
      float rp[NROT > 0 ? NROT : 1][4];
      dcts::static_for<NROT>([&](auto ir_) DCTS_LAMBDA_INLINE {
        constexpr int r = decltype(ir_)::value;
        dcts::static_for<4>([&](auto ic) DCTS_LAMBDA_INLINE { rp[r][decltype(ic)::value] = rot[(r * M + p) * 4 + decltype(ic)::value]; });
      });
      const bool odd = (launder(lane_in) & 1) != 0;
      dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE {  // along a (the H axis) for every b: constants by p
        constexpr int b = decltype(ib)::value;
        if constexpr (Cfg::PAIR) dcts::static_for<S / 2>([&](auto ipair) DCTS_LAMBDA_INLINE { g2_exchange<b>(vp[decltype(ipair)::value][b], odd); });
        float y[S];
        dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE { y[decltype(ia)::value] = vget(ia, ib); });
        g2_network<L, NROT>(y, rp);
        dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE { vset(ia, ib, y[decltype(ia)::value]); });
        __builtin_amdgcn_sched_barrier(0);
      });
      float rq[NROT > 0 ? NROT : 1][4];
      dcts::static_for<NROT>([&](auto ir_) DCTS_LAMBDA_INLINE {
        constexpr int r = decltype(ir_)::value;
        dcts::static_for<4>([&](auto ic) DCTS_LAMBDA_INLINE { rq[r][decltype(ic)::value] = rot[(r * M + q) * 4 + decltype(ic)::value]; });
      });
      dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE {  // along b (the W axis) for every a: constants by q
        float y[S];
        dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE { y[decltype(ib)::value] = vget(ia, ib); });
        g2_network<L, NROT>(y, rq);
        // the outputs exist HERE (LLVM otherwise sinks the networks behind the barrier, down to the LDS stores)
        dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE {
          g2_pin(y[decltype(ib)::value]);
          vset(ia, ib, y[decltype(ib)::value]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    G2_STAMP(0);
    lds_barrier();  // every consumer is done with the previous round's last set
    G2_STAMP(1);
    finish();
    const bool more = grp + (int)gridDim.x < ngroups;
    const G2Group nxt = g2_group<G>(gb, more ? grp + (int)gridDim.x : grp, launder(gbeg_lane));
#if DCTS_G2_EXP == 3
    const unsigned nbytes = 0u;
#else
    const unsigned nbytes = more ? (unsigned)(nxt.count * MAPB) : 0u;  // no next group: every load reads "out of range"
#endif
    const long long map0 = STORE ? ((long long)grp * G) : 0;  // coefficient path: one tensor, groups are consecutive maps
    float e_acc = 0.f;
    dcts::static_for<NSETS>([&](auto iset) DCTS_LAMBDA_INLINE {
      constexpr int SET = decltype(iset)::value;
      // ---- this set's leaf-block samples -> LDS ------------------------------------------------------------
      if (DCTS_G2_SKIP_IDLE == 0 || wave * 64 < Cfg::ITEMS) {
        const int p = launder(ip), q = launder(iq), g = launder(ig);
        const int dump = NBS * G * BS + (launder(lane_in) & 3);  // lanes without an item store behind the set
        const int gofs = g * BS;
        const int o_aa = ok ? gofs + p * RS + q : dump, o_ad = ok ? gofs + p * RS + (M - 1 - q) : dump;
        const int o_da = ok ? gofs + (M - 1 - p) * RS + q : dump, o_dd = ok ? gofs + (M - 1 - p) * RS + (M - 1 - q) : dump;
        dcts::static_for<S>([&](auto ia) DCTS_LAMBDA_INLINE {
          constexpr int ra = decltype(ia)::value;
          dcts::static_for<S>([&](auto ib) DCTS_LAMBDA_INLINE {
            constexpr int rb = decltype(ib)::value;
            constexpr int bid = ra * S + rb;
            if constexpr (kG2Sched<L, NSETS>.set_of[bid] == SET) {
              constexpr int a = plan.slot_of_role[ra], b = plan.slot_of_role[rb];
              constexpr bool asc_a = plan.asc_of_role[ra] != 0, asc_b = plan.asc_of_role[rb] != 0;
              const int off = asc_a ? (asc_b ? o_aa : o_ad) : (asc_b ? o_da : o_dd);
              zbuf[(ok ? kG2Sched<L, NSETS>.li_of[bid] * (G * BS) : 0) + off] = vget(std::integral_constant<int, a>{}, std::integral_constant<int, b>{});
            }
          });
        });
      }
      G2_STAMP(2 + 4 * SET);
      lds_barrier();
      G2_STAMP(3 + 4 * SET);
#if DCTS_G2_SKEW > 0
      // Released together, the sixteen waves run their passes in lock step: all read LDS, then all compute, then
      // all write - the LDS pipe and the VALUs take turns. Half of the waves (two of the four on every SIMD)
      // start about one LDS phase late, so that one half's LDS traffic runs beside the other half's arithmetic.
      if (wave >= NW / 2) __builtin_amdgcn_s_sleep(DCTS_G2_SKEW);
#endif
      // ---- leaf passes of this wave, the next round's samples trickled into the vacated registers ---------
      G2Voffs vo;
      voffs(vo, nxt.count);
      dcts::static_for<PPW>([&](auto ipass) DCTS_LAMBDA_INLINE {
        constexpr int PASS = decltype(ipass)::value;
        const int li = PASS * NW + wave;  // G2Sched deals the blocks in snake order of their cost
        lds_cptr pp = params + (SET * NBS + li) * 8;
        const int vid = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pp[0]) * 2 + __builtin_bit_cast(int, pp[1]));
        // loads of this pass: an even share of the set's slots
        constexpr int i0 = SET * Cfg::PER_SET + (Cfg::PER_SET * PASS) / PPW, i1 = SET * Cfg::PER_SET + (Cfg::PER_SET * (PASS + 1)) / PPW;
        constexpr int ih = i0 + (i1 - i0) / 2;
        G2_STAMP(4 + 4 * SET);
        g2_load_pairs<L, M, G, NSETS, Cfg::PAIR, PAD, i0, ih>(nxt.base, nbytes, vo, vp);
        __builtin_amdgcn_sched_barrier(0);
        G2_STAMP(13);
        float e = g2_pass_dispatch<L, M, G, STORE>(vid, zbuf, pp, li, launder(map_a), launder(map_b), leaf_out, map0, cur.count);
        asm volatile("" : "+v"(e));
        __builtin_amdgcn_sched_barrier(0);
        G2_STAMP(4 + 4 * SET);
        g2_load_pairs<L, M, G, NSETS, Cfg::PAIR, PAD, ih, i1>(nxt.base, nbytes, vo, vp);
        __builtin_amdgcn_sched_barrier(0);
        G2_STAMP(13);
        e_acc += e;
      });
      G2_STAMP(4 + 4 * SET);
      if constexpr (SET + 1 < NSETS) {
        lds_barrier();  // every wave is done with this set: the next one may overwrite it
        G2_STAMP(5 + 4 * SET);
      }
    });
    // ---- per-map sums: over the rows of a block (lanes of a group), then over the waves ---------------------
#if DCTS_G2_LATE_REDUCE
    partials[(pslot * NW + wave) * 64 + launder(lane_in)] = e_acc;  // lanes outside a block hold 0; summed by wave 0 (finish)
#else
    {
      const int mb = launder(map_b);
      const int j = (mb >> 8) & 0xff;
      const bool act = (mb >> 16) != 0;
      float e = e_acc;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        if (off < M) {
          const float t = __shfl_down(e, off, 64);
          if (act && j + off < M) e += t;
        }
      }
      if (act && j == 0) partials[(pslot * NW + wave) * G + (mb & 0xff)] = e;
    }
#endif
    pending = grp;
    pending_out = cur.out;
    pending_count = cur.count;
    pending_slot = pslot;
    pslot ^= 1;
    cur = nxt;
    G2_STAMP(12);
  }
  lds_barrier();
  finish();
#ifdef DCTS_G2_STAMPS
  if (lane_in == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_g2_stamps[wave][i], acc_[i]);
#endif
}

// workgroups per CU: two where LDS and registers allow (L = 2: 16 samples per lane, one set) - two workgroups drift
// apart and fill each other's barrier waits and LDS / VALU / load-issue phases
template <int L, int M, int G>
constexpr int g2_wgs_per_cu() {
#ifdef DCTS_G2_ONE_WG
  return 1;
#else
  return (L == 2 && (long long)(G2Cfg<L, M, G>::ZSET + 2048) * 4 * 2 <= 160 * 1024) ? 2 : 1;
#endif
}

template <int L, int M, int G, bool STORE, int PAD = 0>
__global__ __launch_bounds__((64 * kG2Waves), (4 * g2_wgs_per_cu<L, M, G>())) void k_tile2g(G2Batch gb, float* leaf_out) {
  using Cfg = G2Cfg<L, M, G>;
  __shared__ __attribute__((aligned(16))) float zbuf[Cfg::ZSET];
  __shared__ __attribute__((aligned(16))) float rot[(Cfg::NROT > 0 ? Cfg::NROT : 1) * M * 4];
  __shared__ __attribute__((aligned(16))) float params[Cfg::NSETS * Cfg::NBS * 8];
  __shared__ float partials[2 * kG2Waves * (DCTS_G2_LATE_REDUCE ? 64 : G)];
  g2_body<L, M, G, STORE, PAD>(gb, (lds_ptr)zbuf, (lds_ptr)rot, (lds_ptr)params, (lds_ptr)partials, leaf_out);
}

// X(N, L, M, G)
#ifndef DCTS_TILE2G_TABLE
#define DCTS_TILE2G_TABLE(X) X(72, 2, 18, 3) X(80, 2, 20, 2) X(96, 3, 12, 5) X(112, 3, 14, 4) X(128, 3, 16, 4) X(144, 3, 18, 3) X(160, 3, 20, 2)
#endif
// ... and with the odd front pad (71, 79, 143, 159 -> 72, 80, 144, 160)
#ifndef DCTS_TILE2G_PAD_TABLE
#define DCTS_TILE2G_PAD_TABLE(X) X(72, 2, 18, 3) X(80, 2, 20, 2) X(144, 3, 18, 3) X(160, 3, 20, 2)
#endif

int g2_num_cus() {
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      n = 256;
    return n;
  }();
  return ncu;
}

// groups of G maps, never across tensors; DCTS_E_SHAPE if they do not fit an int (2^31 groups: not a real launch)
template <int G>
int g2_make_batch(const TileBatch& tb, G2Batch& gb) {
  gb.tb = tb;
  long long n = 0;
  for (int t = 0; t < kTileItems; ++t) {
    gb.gbegin[t] = (int)n;
    if (t < tb.count) n += (tb.begin[t + 1] - tb.begin[t] + G - 1) / G;
    if (n > 0x7fffff00LL) return DCTS_E_SHAPE;
  }
  gb.gbegin[kTileItems] = (int)n;
  for (int t = tb.count; t <= kTileItems; ++t) gb.gbegin[t] = (int)n;
  return DCTS_OK;
}

template <int L, int M, int G, int PAD = 0>
int launch_tile2g(const TileBatch& tb, hipStream_t st) {
  G2Batch gb;
  const int rc = g2_make_batch<G>(tb, gb);
  if (rc) return rc;
  const long long groups = gb.gbegin[tb.count];
  if (groups < 1) return DCTS_OK;
  const long long cap = (long long)g2_num_cus() * g2_wgs_per_cu<L, M, G>();  // one residency, persistent over rounds
  const long long grid = groups < cap ? groups : cap;
  hipLaunchKernelGGL((k_tile2g<L, M, G, false, PAD>), dim3((unsigned)grid), dim3(64 * kG2Waves), 0, st, gb, (float*)nullptr);
  return (int)hipGetLastError();
}

template <int L, int M, int G>
int coeff_tile2g(const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps, hipStream_t st) {
  constexpr int N = M << L;
  if (!scratch || scratch_maps < 1) return DCTS_E_WORKSPACE;
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
    G2Batch gb;
    const int rcb = g2_make_batch<G>(tb, gb);
    if (rcb) return rcb;
    const long long groups = gb.gbegin[1];
    const long long grid = groups < g2_num_cus() ? groups : g2_num_cus();
#ifndef DCTS_G2_NOSTORE
    hipLaunchKernelGGL((k_tile2g<L, M, G, true>), dim3((unsigned)grid), dim3(64 * kG2Waves), 0, st, gb, scratch);
#else
    (void)grid;
    return DCTS_E_UNSUPPORTED;  // development build without the coefficient instantiations
#endif
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    rc = launch_assemble<M, L, false>(scratch, nb, out + m0 * (long long)N * N, st);
    if (rc) return rc;
  }
  return DCTS_OK;
}

}  // namespace

#ifdef DCTS_G2_DEV
// development entry points (tools/g2_dev.py builds this file alone: seconds instead of minutes)
extern "C" int g2_dev_run(const float* x, long long nmaps, int edge, float* out, void* stream) {
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
  return dctsi::dispatch_tile2g(edge, &tb, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int g2_dev_run_pad(const float* x, long long nmaps, int edge_padded, float* out, void* stream) {
  TileBatch tb;
  for (int i = 0; i < kTileItems; ++i) {
    tb.x[i] = x;
    tb.out[i] = out;
    tb.begin[i] = 0;
  }
  tb.begin[1] = tb.begin[kTileItems] = nmaps;
  tb.map_elems = (long long)(edge_padded - 1) * (edge_padded - 1);
  tb.total = nmaps;
  tb.count = 1;
  return dctsi::dispatch_tile2g_pad(edge_padded, &tb, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int g2_dev_coeff(const float* x, long long nmaps, int edge, float* out, float* scratch, long long scratch_maps, void* stream) {
  return dctsi::dispatch_tile2g_coeff(edge, x, nmaps, out, scratch, scratch_maps, reinterpret_cast<hipStream_t>(stream));
}
#ifdef DCTS_G2_STAMPS
extern "C" int g2_dev_stamps(unsigned long long* host_out /*[16][16]*/, int reset) {
  if (reset) {
    static unsigned long long zeros[16][16] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_g2_stamps), zeros, sizeof(zeros));
  }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_g2_stamps), 16 * 16 * sizeof(unsigned long long));
}
#endif
#endif

namespace dctsi {
int has_tile2g(int N) {
#define DCTS_CASE(N_, L_, M_, G_) \
  if (N == N_) return 1;
  DCTS_TILE2G_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return 0;
}
int dispatch_tile2g(int N, const void* tile_batch, hipStream_t st) {
  const TileBatch& tb = *static_cast<const TileBatch*>(tile_batch);
#define DCTS_CASE(N_, L_, M_, G_) \
  case N_:                        \
    return launch_tile2g<L_, M_, G_>(tb, st);
  switch (N) {
    DCTS_TILE2G_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
// odd front pad (tile edge N = H + 1 after the pad; the maps in memory are H x H, tb.map_elems = H * H): the AUTO shapes only
int dispatch_tile2g_pad(int N, const void* tile_batch, hipStream_t st) {
  const TileBatch& tb = *static_cast<const TileBatch*>(tile_batch);
#define DCTS_CASE(N_, L_, M_, G_) \
  case N_:                        \
    return launch_tile2g<L_, M_, G_, 1>(tb, st);
  switch (N) {
    DCTS_TILE2G_PAD_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
int has_tile2g_pad(int N) {
#define DCTS_CASE(N_, L_, M_, G_) \
  if (N == N_) return 1;
  DCTS_TILE2G_PAD_TABLE(DCTS_CASE)
#undef DCTS_CASE
  return 0;
}
int dispatch_tile2g_coeff(int N, const float* x, long long nmaps, float* out, float* scratch, long long scratch_maps,
                          hipStream_t st) {
#define DCTS_CASE(N_, L_, M_, G_) \
  case N_:                        \
    return coeff_tile2g<L_, M_, G_>(x, nmaps, out, scratch, scratch_maps, st);
  switch (N) {
    DCTS_TILE2G_TABLE(DCTS_CASE)
    default:
      return DCTS_E_UNSUPPORTED;
  }
#undef DCTS_CASE
}
}  // namespace dctsi
