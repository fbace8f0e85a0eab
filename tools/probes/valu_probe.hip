// Issue-rate probes (not part of the product): how many cycles does one SIMD need per wave64 VALU /
// LDS instruction, as a function of the waves resident on it? Every DCT kernel in this repo is
// budgeted in VALU wave-instructions per map, so the price of one instruction decides the ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o build_dev/valu_probe tools/probes/valu_probe.hip
// One workgroup per CU (grid = CUs), 4*w waves per workgroup = w waves per SIMD. Every wave runs
// ITER iterations of 32 independent instructions of one kind and stamps s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %d (%s) line %d\n", (int)r_, hipGetErrorString(r_), __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048;

enum Kind { K_FMA = 0, K_ADD, K_MUL, K_PKFMA, K_PKADD, K_ADD_DPP, K_MOV, K_FMA_DEP2, K_LDS_R32, K_LDS_W32, K_LDS_R64, K_LDS_R128,
            K_MIX_FMA_LDSR, K_MIX_FMA_LDSW, K_COUNT };
const char* kKindName[K_COUNT] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_add_f32 dpp quad_perm",
                                  "v_mov_b32", "v_fma_f32 (2 chains)", "ds_read_b32", "ds_write_b32", "ds_read_b64", "ds_read_b128",
                                  "16 fma + 4 ds_read_b32", "16 fma + 4 ds_write_b32"};
// instructions per loop iteration (for the cycles-per-instruction quotient)
const int kPerIter[K_COUNT] = {32, 32, 32, 32, 32, 32, 32, 32, 16, 16, 16, 16, 20, 20};

template <int KIND>
__global__ __launch_bounds__(1024) void k_probe(float* sink, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63;
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = 1.0f + 0.001f * (i + lane);
  const float b = 0.999f, c = 0.0001f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  const f2 pb = {b, b}, pc = {c, c};
  // conflict-free addresses: lane-consecutive dwords (b32), 8-byte (b64), 16-byte (b128)
  const unsigned wave = threadIdx.x >> 6;
  const unsigned ad32 = (wave * 64 + lane) * 4u % 16384u;
  const unsigned ad64 = (wave * 64 + lane) * 8u % 32768u;
  const unsigned ad128 = (wave * 64 + lane) * 16u % 65536u;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < ITER; ++it) {
    if constexpr (KIND == K_FMA) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_ADD) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if constexpr (KIND == K_MUL) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
    } else if constexpr (KIND == K_PKADD) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
    } else if constexpr (KIND == K_ADD_DPP) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
        asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 16) & 31]));
    } else if constexpr (KIND == K_MOV) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
    } else if constexpr (KIND == K_FMA_DEP2) {
      // two dependent chains only: exposes the dependent-issue latency of one wave
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c));
      }
    } else if constexpr (KIND == K_LDS_R32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(ad32), "n"(i * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == K_LDS_W32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("ds_write_b32 %1, %0 offset:%2" ::"v"(a[i]), "v"(ad32), "n"(i * 256) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == K_LDS_R64) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[i]) : "v"(ad64), "n"(i * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == K_LDS_R128) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 q[8];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(ad128), "n"(i * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += q[i].x;
    } else if constexpr (KIND == K_MIX_FMA_LDSR) {
      // the codelet pattern: reads issued up front, arithmetic on other registers meanwhile
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[16 + i]) : "v"(ad32), "n"(i * 256));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == K_MIX_FMA_LDSW) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_write_b32 %1, %0 offset:%2" ::"v"(a[16 + i]), "v"(ad32), "n"(i * 256) : "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
  if (s == 123456.789f) sink[0] = s;
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KIND>
void run_kind(int ncu, float* sink, unsigned long long* dcyc) {
  for (int w : {1, 2, 3, 4}) {
    const int threads = 256 * w;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe<KIND>, dim3(ncu), dim3(threads), 0, 0, sink, dcyc);  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe<KIND>, dim3(ncu), dim3(threads), 0, 0, sink, dcyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(ncu * 16);
    CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> v;
    for (int b = 0; b < ncu; ++b)
      for (int k = 0; k < 4 * w; ++k) v.push_back(h[b * 16 + k]);
    std::sort(v.begin(), v.end());
    const double med = (double)v[v.size() / 2];
    const double per_wave_instr = (double)ITER * kPerIter[KIND];
    // s_memtime ticks at the shader clock on gfx950 (MI355X_MICROARCH.md): cycles per instruction
    // and SIMD = wave-lifetime cycles / (instructions per wave * waves per SIMD)
    printf("%-26s waves/SIMD %d: %8.0f cyc per wave  -> %.2f cyc per instr per SIMD  (%.3f ms wall, %.2f GHz-equiv)\n",
           kKindName[KIND], w, med, med / (per_wave_instr * w), ms, med / (ms * 1e6));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
  float* sink;
  unsigned long long* dcyc;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&dcyc, (size_t)ncu * 16 * 8));
  run_kind<K_FMA>(ncu, sink, dcyc);
  run_kind<K_ADD>(ncu, sink, dcyc);
  run_kind<K_MUL>(ncu, sink, dcyc);
  run_kind<K_PKFMA>(ncu, sink, dcyc);
  run_kind<K_PKADD>(ncu, sink, dcyc);
  run_kind<K_ADD_DPP>(ncu, sink, dcyc);
  run_kind<K_MOV>(ncu, sink, dcyc);
  run_kind<K_FMA_DEP2>(ncu, sink, dcyc);
  run_kind<K_LDS_R32>(ncu, sink, dcyc);
  run_kind<K_LDS_W32>(ncu, sink, dcyc);
  run_kind<K_LDS_R64>(ncu, sink, dcyc);
  run_kind<K_LDS_R128>(ncu, sink, dcyc);
  run_kind<K_MIX_FMA_LDSR>(ncu, sink, dcyc);
  run_kind<K_MIX_FMA_LDSW>(ncu, sink, dcyc);
  return 0;
}
