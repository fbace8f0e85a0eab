// Host-side sanitizer driver: links the ASan + UBSan build of libdctscore's
// C-ABI translation unit (make -C dct_pruning_amd/csrc san) and walks every entry point through its
// argument validation, size queries, descriptor packing and the host-side basis-table memo - the code
// that runs on the CPU whatever the GPU does. No GPU is needed: calls that pass validation reach the
// launch and come back with a positive hipError_t on a box without a device (or run, on a box with one).
// GPU AddressSanitizer is not available on this pool and is not attempted: device code is not instrumented.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dctscore.h"

static int g_fail = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
      ++g_fail;                                                          \
    }                                                                    \
  } while (0)

int main() {
  EXPECT(dcts_version() == DCTS_ABI_VERSION);
  for (int c = -12; c <= 2; ++c) EXPECT(dcts_strerror(c) != nullptr && std::strlen(dcts_strerror(c)) > 0);
  EXPECT(dcts_strerror(1000000) != nullptr);

  // size queries over the whole shape range (every table lookup and chunk computation)
  size_t biggest = 0;
  for (int64_t e = 1; e <= DCTS_MAX_EDGE + 8; e += (e < 80 ? 1 : 7)) {
    for (int64_t n : {1, 3, 256}) {
      const size_t a = dcts_workspace_bytes(n, 5, e, e), b = dcts_weighted_workspace_bytes(n, 5, e, e);
      EXPECT(b >= a);
      if (a > biggest) biggest = a;
    }
    (void)dcts_has_codelet(e, e);
    (void)dcts_has_codelet(e, e + 1);
  }
  EXPECT(dcts_workspace_bytes(0, 1, 8, 8) == 0 && dcts_workspace_bytes(1, 1, -3, 8) == 0);
  EXPECT(dcts_weighted_workspace_bytes(1, 0, 8, 8) == 0);
  EXPECT(dcts_has_codelet(56, 56) == 1 && dcts_has_codelet(57, 57) == 0 && dcts_has_codelet(0, 0) == 0);

  // fake "device" addresses: validation and the memo only do arithmetic on them
  std::vector<float> host(1 << 16, 1.0f);
  float* x = host.data();
  float* out = host.data() + 4096;
  alignas(16) static char wsbuf[1 << 20];
  void* ws = wsbuf;

  // argument validation, every entry point
  EXPECT(dcts_energy_f32(nullptr, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_energy_f32(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, nullptr, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_energy_f32(x, 0, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_SHAPE);
  EXPECT(dcts_energy_f32(x, 1, 1, 8, 513, 64, 64, 513, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_SHAPE);
  EXPECT(dcts_energy_f32(x, 1, 4, 8, 8, 256, 64, 8, 1, 3, 2, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_CHANNELS);
  EXPECT(dcts_energy_f32(x, 1, 4, 8, 8, 256, 64, 8, 1, -1, 2, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_CHANNELS);
  EXPECT(dcts_energy_f32(x, 1, 4, 8, 8, 256, 64, 8, 2, 0, 2, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_STRIDE);
  EXPECT(dcts_energy_f32(x, 1, 4, 8, 8, 256, 64, 7, 1, 0, 2, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_STRIDE);
  EXPECT(dcts_energy_f32((const float*)((char*)x + 2), 1, 4, 8, 8, 256, 64, 8, 1, 0, 2, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_ALIGN);
  EXPECT(dcts_energy_f32_ex(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, 99) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_energy_f32_ex(x, 1, 1, 23, 23, 529, 529, 23, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_CODELET) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_energy_f32_ex(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_LANE) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_energy_f32_ex(x, 1, 1, 23, 23, 529, 529, 23, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_FUSED) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_energy_f32_ex(x, 1, 1, 23, 23, 529, 529, 23, 1, 0, 1, 0, out, nullptr, 0, nullptr, DCTS_ALGO_DIRECT) == DCTS_E_WORKSPACE);
  EXPECT(dcts_energy_f32_ex(x, 1, 1, 23, 23, 529, 529, 23, 1, 0, 1, 0, out, ws, 16, nullptr, DCTS_ALGO_DIRECT) == DCTS_E_WORKSPACE);
  // a large tile whose base is only 4-byte aligned: refused by the families that stage with 16-byte direct-to-LDS loads (two
  // launches, pipelined); the fused family loads dwords into registers since round 3 and takes it (a launch: >= 0 here)
  EXPECT(dcts_energy_f32_ex(x + 1, 1, 1, 72, 72, 5184, 5184, 72, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_FUSED) >= 0);
  EXPECT(dcts_energy_f32_ex(x + 1, 1, 1, 72, 72, 5184, 5184, 72, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_SPLIT) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_energy_f32_ex(x + 1, 1, 1, 128, 128, 16384, 16384, 128, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_PIPE) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_dct2d_f32(nullptr, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_dct2d_f32_ex(x, 1, 1, 72, 72, 5184, 5184, 72, 1, 0, 1, 0, out, ws, sizeof wsbuf, nullptr, DCTS_ALGO_SPLIT) == DCTS_E_UNSUPPORTED);
  EXPECT(dcts_dct2d_f32_ex(x, 1, 1, 72, 72, 5184, 5184, 72, 1, 0, 1, 0, out, ws, 64, nullptr, DCTS_ALGO_FUSED) == DCTS_E_WORKSPACE);
  EXPECT(dcts_weighted_energy_f32(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, nullptr, out, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_weighted_energy_f32(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, x, out, nullptr, 0, nullptr) == DCTS_E_WORKSPACE);
  EXPECT(dcts_weighted_energy_f32(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, x, out, wsbuf + 4, 1024, nullptr) == DCTS_E_ALIGN);
  EXPECT(dcts_weighted_energy_f32(x, 1, 1, 8, 8, 64, 64, 8, 1, 0, 1, 0, x, out, ws, 8, nullptr) == DCTS_E_WORKSPACE);
  EXPECT(dcts_batch_sum_f32(nullptr, 1, 1, out, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_batch_sum_f32(x, 0, 1, out, nullptr) == DCTS_E_SHAPE);
  EXPECT(dcts_running_mean_update_f32(x, 1, 1, nullptr, 0.f, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_running_mean_update_f32(x, 1, 0, out, 0.f, nullptr) == DCTS_E_SHAPE);
  EXPECT(dcts_debug_stream_read_f32(nullptr, 4, out, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_debug_stream_read_f32(x, 0, out, nullptr) == DCTS_E_SHAPE);

  // descriptor arrays: validation walks every element; 70 items cross the 32 / 48 / 64-item chunking
  std::vector<dcts_tensor_item> items(70);
  for (size_t i = 0; i < items.size(); ++i) {
    items[i].x = x;
    items[i].out_nc = out;
    items[i].N = 2;
    items[i].C_total = 3;
    items[i].strideN = 3 * 64;
    items[i].strideC = 64;
    items[i].c_begin = 0;
    items[i].c_count = 3;
  }
  EXPECT(dcts_energy_multi_f32(nullptr, 1, 8, 8, 0, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  EXPECT(dcts_energy_multi_f32(items.data(), 0, 8, 8, 0, ws, sizeof wsbuf, nullptr) == DCTS_E_SHAPE);
  items[69].c_count = 4;
  EXPECT(dcts_energy_multi_f32(items.data(), 70, 8, 8, 0, ws, sizeof wsbuf, nullptr) == DCTS_E_CHANNELS);
  items[69].c_count = 3;
  items[33].x = nullptr;
  EXPECT(dcts_energy_multi_f32(items.data(), 70, 8, 8, 0, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  items[33].x = x;
  std::vector<dcts_shaped_item> shaped(70);
  for (size_t i = 0; i < shaped.size(); ++i) {
    shaped[i].t = items[i];
    shaped[i].H = shaped[i].W = (i % 3 == 0) ? 8 : ((i % 3 == 1) ? 4 : 23);
    shaped[i].t.strideC = shaped[i].H * shaped[i].W;
    shaped[i].t.strideN = 3 * shaped[i].t.strideC;
    shaped[i].pad_front_if_odd = 0;
    shaped[i].reserved = 0;
  }
  EXPECT(dcts_energy_mixed_f32(nullptr, 1, ws, sizeof wsbuf, nullptr) == DCTS_E_NULL);
  shaped[5].H = 0;
  EXPECT(dcts_energy_mixed_f32(shaped.data(), 70, ws, sizeof wsbuf, nullptr) == DCTS_E_SHAPE);
  shaped[5].H = 23;
  std::vector<dcts_update_desc> descs(130);
  for (auto& d : descs) {
    d.energy_nc = x;
    d.feature_result = out;
    d.N = 2;
    d.C_count = 3;
    d.total_before = 0.f;
    d.reserved = 0;
  }
  EXPECT(dcts_running_mean_update_multi_f32(nullptr, 1, nullptr) == DCTS_E_NULL);
  descs[100].C_count = 0;
  EXPECT(dcts_running_mean_update_multi_f32(descs.data(), 130, nullptr) == DCTS_E_SHAPE);
  descs[100].C_count = 3;

  // the memo: invalidations with pointers it has never seen, NULL, interior and overlapping ranges
  dcts_workspace_invalidate(nullptr);
  dcts_workspace_invalidate(ws);
  dcts_workspace_invalidate_range(nullptr, 0);
  dcts_workspace_invalidate_range(ws, 0);
  dcts_workspace_invalidate_range(wsbuf + 4096, 1 << 19);

  // calls that pass validation reach the launch. Without a device they return a positive hipError_t (never a
  // crash, never a negative code); with one they run. More than 16 distinct (workspace, shape) pairs wrap the
  // memo's ring; every other one is invalidated again through an overlapping range.
  int launched = 0;
  for (int i = 0; i < 40; ++i) {
    const int64_t e = 11 + 2 * (i % 12);  // odd edges 11..33 without a codelet: direct kernel, tables in the workspace
    char* wsi = wsbuf + (size_t)(i % 5) * 65536;
    const int rc = dcts_energy_f32_ex(x, 1, 1, e, e, e * e, e * e, e, 1, 0, 1, 0, out, wsi, sizeof wsbuf - (size_t)(i % 5) * 65536,
                                      nullptr, DCTS_ALGO_DIRECT);
    EXPECT(rc >= 0);
    launched += rc == 0;
    if (i % 2) dcts_workspace_invalidate_range(wsi + 128, 4096);
  }
  {
    const int rc = dcts_energy_multi_f32(items.data(), 70, 8, 8, 0, ws, sizeof wsbuf, nullptr);
    EXPECT(rc >= 0);
    const int rc2 = dcts_energy_mixed_f32(shaped.data(), 70, ws, sizeof wsbuf, nullptr);
    EXPECT(rc2 >= 0);
    const int rc3 = dcts_running_mean_update_multi_f32(descs.data(), 130, nullptr);
    EXPECT(rc3 >= 0);
    const int rc4 = dcts_weighted_energy_f32(x, 2, 3, 13, 13, 3 * 169, 169, 13, 1, 0, 3, 0, x, out, ws, sizeof wsbuf, nullptr);
    EXPECT(rc4 >= 0);
  }
  std::printf("san_host: %d failures (%d launches succeeded: %s)\n", g_fail, launched, launched ? "a GPU is present" : "no GPU, as expected here");
  return g_fail ? 1 : 0;
}
