// split_roles.hpp - pieces shared by the large-tile kernel families (dct_kernels.hip, tile2d.hip):
// LDS pointer types, the multi-tensor map index space, the radix-2 role tree of the split DCT
// (which M-point problem each of the 2^L roles solves, the in-place butterfly network that
// produces the roles' inputs, its rotation constants) and a few wave-level helpers.
// Every translation unit gets its own copy (anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "dct_codelets.hpp"

namespace {

// LDS pointers stay in address space 3 end to end (see the split family for why)
using lds_ptr = __attribute__((address_space(3))) float*;
using lds_cptr = const __attribute__((address_space(3))) float*;

// Dense tensors of one large tile shape as ONE map index space (fused / pipelined kernels): map m of
// the batch is map m - begin[t] of tensor t. U2-Net-p hooks ten 288x288 tensors of 16 or 64 channels;
// launched one by one at batch 12 they give a CU 0.75 or 3 maps each, together 16.5.
constexpr int kTileItems = 32;
struct TileBatch {
  const float* x[kTileItems];
  float* out[kTileItems];
  long long begin[kTileItems + 1];  // begin[count] = total
  long long map_elems;              // floats per map (dense: maps of a tensor are adjacent)
  long long total;
  int count;
};
// `hint`: the caller's map indices ascend (a persistent workgroup walks m, m + grid, ...), so the scan resumes where
// the previous lookup of the same sequence ended instead of at tensor 0: every step is a dependent scalar load, and a
// seven-tensor batch of 288 x 288 maps cost 4.7 % of the launch in these scans (three lookups per map).
__device__ __forceinline__ int tile_item(const TileBatch& tb, long long m, int* hint = nullptr) {
  int t = hint ? *hint : 0;
  while (t + 1 < tb.count && m >= tb.begin[t + 1]) ++t;  // wave-uniform
  t = __builtin_amdgcn_readfirstlane(t);
  if (hint) *hint = t;
  return t;
}
__device__ __forceinline__ const float* tile_in(const TileBatch& tb, long long m, int* hint = nullptr) {
  const int t = tile_item(tb, m, hint);
  // explicitly wave-uniform (the raw direct-to-LDS loads take it as a scalar operand)
  const unsigned long long a = reinterpret_cast<unsigned long long>(tb.x[t] + (m - tb.begin[t]) * tb.map_elems);
  return reinterpret_cast<const float*>(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
__device__ __forceinline__ float* tile_out(const TileBatch& tb, long long m, int* hint = nullptr) {
  const int t = tile_item(tb, m, hint);
  return tb.out[t] + (m - tb.begin[t]);
}
// one tensor, no table: the pipelined 14 x 16 kernel has neither the SGPRs nor the VGPRs to spare
struct PlainMaps {
  const float* x;
  float* out;
  long long map_elems;
  long long total;
};
__device__ __forceinline__ const float* tile_in(const PlainMaps& pm, long long m, int* = nullptr) { return pm.x + m * pm.map_elems; }
__device__ __forceinline__ float* tile_out(const PlainMaps& pm, long long m, int* = nullptr) { return pm.out + m; }

template <int N>
struct SplitRoot {
  static constexpr int len = N;
  static constexpr bool is4 = false;
  template <int J>
  static __device__ __forceinline__ float in(const float* col, int rs) {
    return col[J * rs];
  }
  static constexpr double wt(bool zero) { return zero ? 0.70710678118654752440 : 1.0; }  // global DC
};

template <class P, int WHICH>
struct SplitNode {
  static constexpr int len = P::len / 2;
  static constexpr bool is4 = (!P::is4) && WHICH == 1;
  template <int J>
  static __device__ __forceinline__ float in(const float* col, int rs) {
    const float y0 = P::template in<J>(col, rs);
    const float y1 = P::template in<P::len - 1 - J>(col, rs);
    if constexpr (!P::is4) {
      return WHICH == 0 ? y0 + y1 : y0 - y1;
    } else {
      constexpr float c = float(dcts::cospi_frac(2 * J + 1, 4 * P::len));
      constexpr float sn = float(dcts::sinpi_frac(2 * J + 1, 4 * P::len));
      if constexpr (WHICH == 0) {
        return y0 * c + y1 * sn;
      } else {
        constexpr float sg = (J % 2 == 0) ? 1.f : -1.f;
        return y1 * (sg * c) - y0 * (sg * sn);
      }
    }
  }
  // amplitude weight of an output of this node; `zero`: its index in this node's output space is 0
  static constexpr double wt(bool zero) {
    const double f = (P::is4 && !zero) ? 1.41421356237309504880 : 1.0;
    return f * P::wt(zero && WHICH == 0);
  }
};

template <int N, int L, int R>
struct RoleLeaf {
  using type = SplitNode<typename RoleLeaf<N, L - 1, (R >> 1)>::type, (R & 1)>;
};
template <int N>
struct RoleLeaf<N, 0, 0> {
  using type = SplitRoot<N>;
};

// The role butterflies as an in-place network on the 2^L mirrored samples of one (p, line):
// slot s holds row s*M + p (s even) or s*M + M-1-p (s odd) of the strip. Level by level the
// samples of a node are paired (index j with n-1-j), the pair is replaced by the inputs of the
// node's two children, and after L levels every slot holds one input sample of one role. All
// structure (which slots pair up, which role ends where, whether a role sees its samples in
// ascending or descending p) is compile-time; only the rotation constants depend on p (table).
template <int L>
struct RolePlan {
  static constexpr int S = 1 << L;
  static constexpr int NOPS = L * (S / 2);
  int op_a[NOPS > 0 ? NOPS : 1] = {}, op_b[NOPS > 0 ? NOPS : 1] = {}, op_rot[NOPS > 0 ? NOPS : 1] = {};
  int nrot = 0;
  int rot_seg[NOPS > 0 ? NOPS : 1] = {}, rot_asc[NOPS > 0 ? NOPS : 1] = {}, rot_c[NOPS > 0 ? NOPS : 1] = {};
  int slot_of_role[S] = {}, asc_of_role[S] = {}, is4_of_role[S] = {};
  constexpr RolePlan() {
    int node[S] = {}, seg[S] = {}, asc[S] = {}, is4[S] = {};
    for (int s = 0; s < S; ++s) {
      seg[s] = s;
      asc[s] = (s % 2 == 0) ? 1 : 0;
    }
    int c = S, n = 0;
    for (int lvl = 0; lvl < L; ++lvl) {
      int nnode[S] = {}, nis4[S] = {}, nseg[S] = {}, nasc[S] = {};
      for (int i = 0; i < S; ++i) {
        if (seg[i] >= c / 2) continue;
        int k = -1;
        for (int t = 0; t < S; ++t)
          if (node[t] == node[i] && seg[t] == c - 1 - seg[i]) k = t;
        op_a[n] = i;
        op_b[n] = k;
        if (is4[i]) {
          op_rot[n] = nrot;
          rot_seg[nrot] = seg[i];
          rot_asc[nrot] = asc[i];
          rot_c[nrot] = c;
          ++nrot;
        } else {
          op_rot[n] = -1;
        }
        ++n;
        nnode[i] = node[i] * 2;      // child 0 keeps the lower sample's slot
        nnode[k] = node[i] * 2 + 1;  // child 1 takes the upper sample's slot
        nis4[i] = 0;
        nis4[k] = is4[i] ? 0 : 1;
        nseg[i] = nseg[k] = seg[i];
        nasc[i] = nasc[k] = asc[i];
      }
      for (int i = 0; i < S; ++i) {
        node[i] = nnode[i];
        is4[i] = nis4[i];
        seg[i] = nseg[i];
        asc[i] = nasc[i];
      }
      c /= 2;
    }
    for (int i = 0; i < S; ++i) {
      slot_of_role[node[i]] = i;
      asc_of_role[node[i]] = asc[i];
      is4_of_role[node[i]] = is4[i];
    }
  }
};

// rotation constants of the DCT-IV butterflies: for instance r and sample p the pair index is
// j = seg*M + (asc ? p : M-1-p) inside a node of n = c*M points: cos/sin((2j+1) pi / (4n)), (-1)^j
template <int M, int L>
struct RotTable {
  static constexpr int NR = (RolePlan<L>::NOPS > 0 ? RolePlan<L>::NOPS : 1);
  float c[NR][M] = {}, s[NR][M] = {};
  constexpr RotTable() {
    constexpr RolePlan<L> plan{};
    for (int r = 0; r < plan.nrot; ++r)
      for (int p = 0; p < M; ++p) {
        const int j = plan.rot_seg[r] * M + (plan.rot_asc[r] ? p : M - 1 - p);
        const int n = plan.rot_c[r] * M;
        c[r][p] = float(dcts::cospi_frac(2 * j + 1, 4 * n));
        s[r][p] = float(dcts::sinpi_frac(2 * j + 1, 4 * n));
      }
  }
  // (-1)^j = sign0(r) * (-1)^p: the sign of the second rotation output needs no table
  static constexpr float sign0(int r) {
    constexpr RolePlan<L> plan{};
    const int j0 = plan.rot_seg[r] * M + (plan.rot_asc[r] ? 0 : M - 1);
    return (j0 % 2 == 0) ? 1.f : -1.f;
  }
};
template <int M, int L>
__device__ const RotTable<M, L> kRotTable{};

// Workgroup barrier that orders LDS traffic only. __syncthreads() carries a workgroup-scope fence
// over ALL address spaces: the compiler puts s_waitcnt vmcnt(0) in front of every s_barrier, i.e.
// each barrier also waits for every direct-to-LDS load still in flight and the prefetch of the
// next strip is drained five times per step. The data those loads bring is published by the
// explicit s_waitcnt vmcnt(0) + barrier at the top of pass 1.
// (A fence restricted to the "local" address space still waits vmcnt(0): the loads in flight write
// LDS. Hence the raw instruction pair; LDS operations of a wave complete in order.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Opaque copy of a lane-dependent value: everything derived from the copy has to be recomputed where
// it is used. Without it LLVM hoists the address arithmetic of every phase (dump offsets per strip,
// column pointers of both buffers, shuffle indices, ...) out of the persistent loop and keeps some
// twenty loop-invariant VGPRs alive next to the parked tile: spills, and a scratch reload's
// s_waitcnt vmcnt(0) also waits for every direct-to-LDS load in flight.
__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// wave64 sum by DPP within rows of 16 lanes, then the four row totals in fixed order: no index
// registers (ds_bpermute needs one per offset), result uniform across the wave
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float a, auto ctrl) DCTS_LAMBDA_INLINE {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

// ---------------------------------------------------------------------------------------
// coefficient assembly (debug / parity path: dcts_dct2d_f32_ex with a large-tile family)
// ---------------------------------------------------------------------------------------
// The energy kernels stop at the leaf outputs of the 2^L roles and fold the last add/sub layer of
// every DCT-IV node above the leaves into the reduction ((a+b)^2 + (a-b)^2 = 2a^2 + 2b^2, the sqrt(2)
// in SplitNode::wt). For coefficient output that layer is applied explicitly: coefficient u of the
// length-N DCT-II is a signed sum of at most 2^ceil(L/2) leaf outputs (DCT-II nodes interleave their
// children's outputs; a DCT-IV node's are X[0] = A[0], X[n-1] = -B[0], X[2j] = A[j] + B[n/2-j],
// X[2j-1] = A[j] - B[n/2-j]; DCT-IV nodes only have DCT-II children, so a path holds at most
// ceil(L/2) of them). invw undoes the amplitude weight the energy kernels apply to a leaf output.
template <int M, int L>
struct AsmTable {
  static constexpr int N = M << L, MAXT = 1 << ((L + 1) / 2);
  short idx[N][MAXT] = {};  // role * M + k
  signed char sgn[N][MAXT] = {};
  unsigned char n[N] = {};
  float invw[N] = {};       // 1 / weight of leaf output role * M + k
  constexpr void walk(int level, int path, bool is4, int len, int i, int sign, int u) {
    if (level == L) {
      idx[u][n[u]] = short(path * M + i);
      sgn[u][n[u]] = (signed char)sign;
      ++n[u];
      return;
    }
    if (!is4) {
      walk(level + 1, path * 2 + (i & 1), (i & 1) != 0, len / 2, i >> 1, sign, u);
    } else {
      const int H = len / 2;
      if (i == 0) {
        walk(level + 1, path * 2, false, H, 0, sign, u);
      } else if (i == len - 1) {
        walk(level + 1, path * 2 + 1, false, H, 0, -sign, u);
      } else if (i % 2 == 0) {
        walk(level + 1, path * 2, false, H, i / 2, sign, u);
        walk(level + 1, path * 2 + 1, false, H, H - i / 2, sign, u);
      } else {
        const int j = (i + 1) / 2;
        walk(level + 1, path * 2, false, H, j, sign, u);
        walk(level + 1, path * 2 + 1, false, H, H - j, -sign, u);
      }
    }
  }
  template <int R>
  constexpr void weights() {
    using Leaf = typename RoleLeaf<N, L, R>::type;
    for (int k = 0; k < M; ++k) invw[R * M + k] = float(1.0 / Leaf::wt(k == 0));
  }
  template <int... R>
  constexpr void all_weights(std::integer_sequence<int, R...>) {
    (weights<R>(), ...);
  }
  constexpr AsmTable() {
    for (int u = 0; u < N; ++u) walk(0, 0, false, N, u, 1, u);
    all_weights(std::make_integer_sequence<int, (1 << L)>{});
  }
};
template <int M, int L>
__device__ const AsmTable<M, L> kAsmTable{};

// leaf[b][iH][iW] (iH = roleH * M + kH, iW likewise; weighted iff WEIGHTED) -> out[b][u][v], the
// orthonormal coefficients of torch_dct.dct_2d(norm='ortho') (utils/common.py:267)
template <int M, int L, bool WEIGHTED>
__global__ __launch_bounds__(256) void k_assemble(const float* __restrict__ leaf, long long nmaps, float* __restrict__ out) {
  constexpr int N = M << L;
  const AsmTable<M, L>& t = kAsmTable<M, L>;
  const long long total = nmaps * N * N;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / (N * N);
    const int uv = (int)(e - b * (N * N)), u = uv / N, v = uv - u * N;
    const float* lf = leaf + b * (long long)N * N;
    float acc = 0.f;
    for (int i = 0; i < t.n[u]; ++i)
      for (int j = 0; j < t.n[v]; ++j) {
        const int iH = t.idx[u][i], iW = t.idx[v][j];
        float val = lf[iH * N + iW];
        if (WEIGHTED) val *= t.invw[iH] * t.invw[iW];
        acc += float(t.sgn[u][i] * t.sgn[v][j]) * val;
      }
    constexpr float s0 = dcts::ortho_scale<N>(0), s1 = dcts::ortho_scale<N>(1);
    out[e] = acc * ((u == 0 ? s0 : s1) * (v == 0 ? s0 : s1));
  }
}
template <int M, int L, bool WEIGHTED>
inline int launch_assemble(const float* leaf, long long nmaps, float* out, hipStream_t st) {
  const long long total = nmaps * (long long)(M << L) * (M << L);
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL((k_assemble<M, L, WEIGHTED>), dim3((unsigned)blocks), dim3(256), 0, st, leaf, nmaps, out);
  return (int)hipGetLastError();
}

}  // namespace
