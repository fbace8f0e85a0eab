/*
 * dctscore.h — C ABI of libdctscore.so, the MI355X (gfx950) implementation of the
 * DCT importance-score hot path of semchan/DCT_Pruning.
 *
 * The reference has no native code: the seam this library cuts at is the Python
 * block  utils/common.py:265-274  (get_feature_hook), :283-291 (densenet hook) and
 * :299-307 (u2net input hook):
 *
 *     c = [dct.dct_2d(output[i,j,:,:], norm='ortho') for i in range(a) for j in range(b)]
 *     c = cnt_score(c)            # per map: sum(coeff * coeff)      (utils/common.py:249-255)
 *     c = c.view(a, -1)           # [N, C] per-map energies
 *
 * i.e. "feature-map tensor on the device -> one fp32 energy per (sample, channel)".
 * Everything after that (sum over the batch, running mean, np.save) stays on the host in
 * Python exactly as the reference does it (utils/common.py:271-277).
 *
 * Conventions
 *   - All pointers are DEVICE pointers (hipMalloc'd / torch CUDA tensors). The caller owns
 *     every buffer; the library allocates no device memory. Its only state, all on the HOST:
 *       (1) a 16-entry memo of which byte range of which caller workspace holds the direct
 *           kernel's cosine-basis tables (per workspace pointer, stream and tile shape; see
 *           dcts_workspace_invalidate - a caller that writes into a workspace or frees it
 *           must say so);
 *       (2) values read ONCE per process: the device's CU count and per-kernel occupancy, and
 *           the environment variable DCTS_SPLIT_CHUNK_MB (size of the two-launch split path's
 *           intermediate buffer, default 256).
 *   - Strides are in ELEMENTS (floats), as torch.Tensor.stride() reports them.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Every entry
 *     point only ENQUEUES work on that stream; there is no implicit synchronisation.
 *   - Return value: 0 = ok; negative = DCTS_E_* (bad argument / unsupported shape);
 *     positive = a hipError_t raised by the launch. Nothing throws, aborts or prints.
 *   - Re-entrant and thread-safe; safe to call with the Python GIL released.
 */
#ifndef DCTSCORE_H_
#define DCTSCORE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: workspace contract (the library may leave basis tables in a workspace between calls; 16-byte
 *    alignment; dcts_workspace_invalidate[_range]), multi / mixed / weighted entry points. */
#define DCTS_ABI_VERSION 2

enum {
  DCTS_OK = 0,
  DCTS_E_NULL = -1,        /* a required pointer is NULL                                */
  DCTS_E_SHAPE = -2,       /* N, C, H or W <= 0, or H/W beyond DCTS_MAX_EDGE            */
  DCTS_E_CHANNELS = -3,    /* [c_begin, c_begin + c_count) not inside [0, C_total)      */
  DCTS_E_STRIDE = -4,      /* innermost stride != 1 or row stride < W (rows must be dense) */
  DCTS_E_WORKSPACE = -5,   /* workspace smaller than dcts_workspace_bytes() reports     */
  DCTS_E_UNSUPPORTED = -6, /* combination not implemented (see dcts_strerror)           */
  DCTS_E_ALIGN = -7        /* pointer not 4-byte aligned                                */
};

/* Largest tile edge (after the optional odd front pad) any kernel accepts. */
#define DCTS_MAX_EDGE 512

/* Kernel family selector for dcts_energy_f32_ex (testing / benchmarking). */
enum {
  DCTS_ALGO_AUTO = 0,     /* pick the fastest kernel that supports the shape                     */
  DCTS_ALGO_DIRECT = 1,   /* cosine-basis-in-LDS separable kernel, any (H, W) <= DCTS_MAX_EDGE   */
  DCTS_ALGO_CODELET = 2,  /* register-resident factorised DCT codelets (selected tile sizes)     */
  DCTS_ALGO_SPLIT = 3,    /* two-launch split codelet passes for edges 4*M / 8*M (68 ... 512)    */
  DCTS_ALGO_PREFETCH = 4, /* codelet kernel with direct-to-LDS prefetch of the next maps (dense,
                             even-edge square tiles; measured equal to ALGO_CODELET, opt-in)     */
  DCTS_ALGO_FUSED = 5,    /* single-launch split kernel, intermediate tile parked in VGPRs
                             (edges 72 ... 256 incl. 96 and 192; 288 and 320 with two roles per wave) */
  DCTS_ALGO_PIPE = 6,     /* the fused kernel software-pipelined: pass 2 of one map interleaved
                             with pass 1 of the next                                             */
  DCTS_ALGO_LANE = 7,     /* one lane per map, both passes in registers (7x7, 9x9)              */
  DCTS_ALGO_TILE2D = 8,   /* 2-D radix split: butterflies over both axes in registers, then 4^L independent
                             M x M leaf blocks - 224 (tile2d.hip); 72, 80, 96, 112, 128, 144, 160 with several maps
                             per round (tile2g.hip)                                              */
  DCTS_ALGO_RECT = 9      /* the 1-D codelets picked per axis at run time: any (H, W) with both edges (after the
                             odd pad) <= 64 - non-square maps, odd / prime edges, rows with strideH > W
                             (rect.hip); what AUTO takes for such shapes                           */
};

/* ABI version of the loaded library (== DCTS_ABI_VERSION it was built with). */
int dcts_version(void);

/* Human-readable text for a return code of this library (never NULL). */
const char* dcts_strerror(int code);

/*
 * Scratch bytes dcts_energy_f32 needs for a call with these sizes (may be 0). The caller
 * allocates it once and may reuse it across calls on the same stream.
 */
size_t dcts_workspace_bytes(int64_t N, int64_t C_count, int64_t H, int64_t W);

/*
 * Per-map DCT energy.  Replaces utils/common.py:267 + :249-255 (and :285 / :301 through
 * torch2dct, :230-239, when pad_front_if_odd != 0).
 *
 *   x            fp32 feature maps, logical shape [N, C_total, H, W]; element (n,c,h,w) is at
 *                x[n*strideN + c*strideC + h*strideH + w*strideW]; strideW must be 1, strideH >= W.
 *                Any (H, W) <= DCTS_MAX_EDGE is accepted; which kernel runs depends on the shape: square
 *                dense maps of the tabulated edges have their own kernels, non-square maps and maps with
 *                strideH > W with both edges <= 64 the run-time codelet pair (DCTS_ALGO_RECT: any
 *                edge 1 ... 64), everything else the cosine-matrix kernel (DCTS_ALGO_DIRECT).
 *   c_begin,
 *   c_count      channel slice to score (densenet hook: c_begin = C_total-12, c_count = 12).
 *   pad_front_if_odd
 *                0: transform the H x W map as is (torch_dct path).
 *                1: cv2 path of torch2dct — if H is odd, one zero row is put in front of the
 *                   rows AND one zero column in front of the columns before the transform
 *                   (np.pad(t,(1,0)) pads every axis; the test is on H only).
 *   out_nc       [N, c_count] fp32, row-major:
 *                out_nc[n*c_count + j] = sum_{u,v} DCT2_ortho(x[n, c_begin+j])[u,v]^2
 *   workspace    >= dcts_workspace_bytes(N, c_count, H, W) bytes of device memory, 16-byte aligned,
 *                or NULL when that is 0. (The size of the two-launch split path's intermediate
 *                buffer, and with it dcts_workspace_bytes for edges 72..320, follows the environment
 *                variable DCTS_SPLIT_CHUNK_MB, read once per process; default 256.)
 */
int dcts_energy_f32(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                    int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                    int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                    float* out_nc, void* workspace, size_t workspace_bytes, void* stream);

/* Same, with an explicit kernel family (DCTS_ALGO_*). DCTS_E_UNSUPPORTED if that family
 * has no kernel for the shape. */
int dcts_energy_f32_ex(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                       int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                       int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                       float* out_nc, void* workspace, size_t workspace_bytes, void* stream,
                       int32_t algo);

/* The library keeps immutable basis tables at the head of a workspace between calls (built once per
 * workspace, stream and tile shape; remembered on the host, nothing is read back). A caller that writes into a
 * workspace itself, or frees it and allocates another at the same address, says so here first. */
void dcts_workspace_invalidate(void* workspace);
/* The same for callers that wrote into (or are about to free) bytes [workspace, workspace + bytes): also
 * forgets tables cached under other pointers that overlap the range. */
void dcts_workspace_invalidate_range(void* workspace, size_t bytes);

/* 1 if DCTS_ALGO_CODELET has a kernel for an (H, W) tile (sizes AFTER the odd pad). */
int dcts_has_codelet(int64_t H, int64_t W);

/*
 * Full coefficient output (parity/debug and the coefficient-domain API the reference's
 * commented-out variants hint at, utils/common.py:268-269). Same addressing as above;
 *   out_coeff    [N, c_count, H', W'] fp32 dense, H' = H + (pad && H odd), W' likewise;
 *                out_coeff[n,j] = dct_2d(x[n, c_begin+j], norm='ortho').
 *   workspace    as for dcts_energy_f32 (same size query). With DCTS_ALGO_FUSED / DCTS_ALGO_TILE2D
 *                (dcts_dct2d_f32_ex) the large-tile energy kernels themselves produce the coefficients
 *                (leaf outputs into the workspace, then the DCT-IV add/sub layers the energy path folds
 *                into its reduction): the workspace must be 16-byte aligned and hold at least one
 *                H' x W' fp32 tile; more tiles mean fewer launches.
 */
int dcts_dct2d_f32(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                   int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                   int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                   float* out_coeff, void* workspace, size_t workspace_bytes, void* stream);

/* Same, with an explicit kernel family (DCTS_ALGO_*). */
int dcts_dct2d_f32_ex(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                      int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                      int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                      float* out_coeff, void* workspace, size_t workspace_bytes, void* stream,
                      int32_t algo);

/*
 * Score variant in the coefficient domain (the reference only hints at variants, utils/common.py:268-269;
 * SURVEY.md §8 f4): out_nc[n, j] = sum_{u,v} weights[u, v] * dct_2d(x[n, c_begin+j], norm='ortho')[u, v]^2,
 * `weights` a dense [H', W'] fp32 device array (all ones reproduces dcts_energy_f32 up to rounding).
 * Coefficients come from the same kernels as dcts_dct2d_f32_ex (the large-tile kernels' own coefficient path
 * for dense 72..320 tiles). Workspace: dcts_weighted_workspace_bytes(), 16-byte aligned.
 */
size_t dcts_weighted_workspace_bytes(int64_t N, int64_t C_count, int64_t H, int64_t W);
int dcts_weighted_energy_f32(const float* x, int64_t N, int64_t C_total, int64_t H, int64_t W,
                             int64_t strideN, int64_t strideC, int64_t strideH, int64_t strideW,
                             int32_t c_begin, int32_t c_count, int32_t pad_front_if_odd,
                             const float* weights, float* out_nc, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * Fused batch reduction for benchmarking and for the single-sweep harness:
 *   out_c[j] = sum_n energy[n, j]   (fp32; summation order: 16 interleaved slices, slice s adds
 *   n = s, s+16, s+32, ... in ascending order, then the 16 partial sums are added in slice order.
 *   Fixed, launch-independent and bit-reproducible, but not torch's c.view(a,-1).sum(0) order: the
 *   reference-exact path is dcts_energy_f32 + host-side sum(0), utils/common.py:271-274).
 */
int dcts_batch_sum_f32(const float* energy_nc, int64_t N, int64_t C_count, float* out_c,
                       void* stream);

/*
 * dcts_energy_f32 for `count` tensors of the SAME tile shape (H, W) in one launch (chunks of 32):
 * the single-sweep harness scores every hooked tensor of a shape at the end of the forward pass,
 * and CIFAR-sized layers are too small to fill the GPU one launch at a time. Rows must be dense
 * (strideH == W, strideW == 1). `items` is a HOST array, copied into the kernel arguments.
 * Shapes without a codelet kernel are processed tensor by tensor (same result, no batching);
 * `workspace` must then cover the largest item (dcts_workspace_bytes).
 */
typedef struct dcts_tensor_item {
  const float* x;   /* [N, C_total, H, W] view: element (n,c,h,w) at x[n*strideN + c*strideC + h*W + w] */
  float* out_nc;    /* [N, c_count] */
  int64_t N, C_total, strideN, strideC;
  int32_t c_begin, c_count;
} dcts_tensor_item;
int dcts_energy_multi_f32(const dcts_tensor_item* items, int32_t count, int64_t H, int64_t W,
                          int32_t pad_front_if_odd, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same for tensors of DIFFERENT tile shapes: every hooked tensor of a forward pass in one call.
 * Tensors with square tiles of edge 2, 4, 8, 16 or 32 (every hook point of the reference's CIFAR nets:
 * VGG-16-bn, ResNet-56/110, DenseNet-40, GoogLeNet) share ONE launch per 48 of them, whatever their
 * shapes; the others are grouped by shape as dcts_energy_multi_f32 does. Results are those of one
 * dcts_energy_f32 call per tensor, bit for bit. `workspace` must cover the largest item that needs one.
 */
typedef struct dcts_shaped_item {
  dcts_tensor_item t;
  int64_t H, W;
  int32_t pad_front_if_odd;
  int32_t reserved;
} dcts_shaped_item;
int dcts_energy_mixed_f32(const dcts_shaped_item* items, int32_t count, void* workspace, size_t workspace_bytes,
                          void* stream);

/*
 * Device-side form of the running-mean update of get_feature_hook, utils/common.py:273-277:
 *   c = sum_n energy_nc[n, :]
 *   feature_result = (feature_result * total_before + c) / (total_before + N)
 * feature_result is [C_count] fp32 in/out on the device (zeros before the first batch, like
 * the reference's torch.tensor(0.) broadcast); the caller keeps `total` on the host and
 * adds N after each call. Same three fp32 roundings as the reference (no FMA contraction);
 * the batch sum is the 16-slice fixed-order sum of dcts_batch_sum_f32.
 */
int dcts_running_mean_update_f32(const float* energy_nc, int64_t N, int64_t C_count,
                                 float* feature_result, float total_before, void* stream);

/*
 * The same update for `count` hook points in ONE launch (single-sweep harness: every layer's
 * energies of a batch are ready when the forward pass ends). `descs` is a HOST array; it is
 * copied into the kernel arguments, so it may be reused as soon as the call returns.
 */
typedef struct dcts_update_desc {
  const float* energy_nc; /* [N, C_count] device */
  float* feature_result;  /* [C_count] device, in/out */
  int64_t N;
  int64_t C_count;
  float total_before;     /* samples accumulated so far for this hook point */
  int32_t reserved;
} dcts_update_desc;
int dcts_running_mean_update_multi_f32(const dcts_update_desc* descs, int32_t count, void* stream);

/*
 * Measurement aid (not on the score path): reads n floats once with the kernels' own access
 * width (one dword per lane, coalesced) and discards them. Used to calibrate the FETCH_SIZE
 * performance counter against a known byte count (tools/pmc_traffic.py).
 */
int dcts_debug_stream_read_f32(const float* x, int64_t n, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCTSCORE_H_ */
