// Read-bandwidth ceiling probes (not part of the product): what can a kernel that does nothing but
// read reach on this box?  hipcc --offload-arch=gfx950 -O3 -o build_dev/bw_probe tools/probes/bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

// (a) float4 register loads, grid-stride, 4 independent loads in flight per lane
__global__ __launch_bounds__(256) void k_read_x4(const float4* __restrict__ x, long long n4, float* sink) {
  float s = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
    s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
  }
  for (; i < n4; i += stride) {
    const float4 a = x[i];
    s += a.x + a.y + a.z + a.w;
  }
  if (s == 123456.789f) sink[0] = s;
}

// (b) direct-to-LDS loads only: every wave streams consecutive 1 KiB pieces into its LDS slab,
// DEPTH pieces in flight, nothing is ever read back
template <int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_read_lds(const float* __restrict__ x, long long npieces) {
  __shared__ __attribute__((aligned(16))) float slab[WAVES][DEPTH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)slab[wave]);
  const long long wgid = (long long)blockIdx.x * WAVES + wave, nw = (long long)gridDim.x * WAVES;
  // wave w takes chunks of DEPTH pieces: chunk index c = w, w + nw, ...
  const long long nchunks = npieces / DEPTH;
  for (long long c = wgid; c < nchunks; c += nw) {
    const unsigned long long sa = reinterpret_cast<unsigned long long>(x + c * DEPTH * 256);
    const unsigned long long src = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)sa);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned dst = base + d * 1024;
      const unsigned off = (unsigned)(d * 1024 + lane * 16);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(off), "s"(src)
                   : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %d line %d\n", (int)r_, __LINE__); exit(1);} } while (0)

template <class F>
double time_us(F f, int reps = 10) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const long long bytes = 2LL << 30;  // 2 GiB: 8x the Infinity Cache
  float *x, *sink;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(x, 0, bytes));
  const long long n4 = bytes / 16, npieces = bytes / 1024;
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    double t = time_us([&] { hipLaunchKernelGGL(k_read_x4, dim3(blocks), dim3(256), 0, 0, (const float4*)x, n4, sink); });
    printf("float4 register loads, %5d blocks x 256: %8.1f us  %7.1f GB/s\n", blocks, t, bytes / t / 1e3);
  }
#define RUN_LDS(DEPTH, WAVES, BPC)                                                                              \
  {                                                                                                             \
    double t = time_us([&] { hipLaunchKernelGGL((k_read_lds<DEPTH, WAVES>), dim3(256 * BPC), dim3(64 * WAVES), 0, 0, x, npieces); }); \
    printf("direct-to-LDS, depth %2d KiB/wave, %d waves/WG, %d WG/CU (%3d KiB in flight per CU): %8.1f us  %7.1f GB/s\n", DEPTH, WAVES, BPC, \
           DEPTH * WAVES * BPC, t, bytes / t / 1e3);                                                             \
  }
  RUN_LDS(4, 4, 2) RUN_LDS(8, 4, 2) RUN_LDS(8, 4, 4) RUN_LDS(16, 4, 2) RUN_LDS(12, 4, 3) RUN_LDS(16, 2, 4) RUN_LDS(4, 4, 8) RUN_LDS(2, 4, 8)
  RUN_LDS(1, 4, 8) RUN_LDS(32, 1, 4)
  return 0;
}
