"""Operator-level seam: the tensor -> per-map energy call that replaces the reference's
Python list comprehension over maps (utils/common.py:265-270, :283-287, :299-303)."""
import torch

from . import _lib

ALGO_AUTO, ALGO_DIRECT, ALGO_CODELET, ALGO_SPLIT, ALGO_PREFETCH, ALGO_FUSED, ALGO_PIPE, ALGO_LANE, ALGO_TILE2D, ALGO_RECT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9

# one scratch buffer per (device, stream); grown on demand, reused across calls
_workspaces = {}


def _workspace(device, stream_ptr, nbytes):
    key = (device.index, stream_ptr)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:  # the allocator may hand these bytes out again
            _lib.load().dcts_workspace_invalidate_range(buf.data_ptr(), buf.numel())
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _lib.load().dcts_workspace_invalidate_range(buf.data_ptr(), buf.numel())
        _workspaces[key] = buf
    return buf


def _check_input(x):
    if not isinstance(x, torch.Tensor):
        raise TypeError("expected a torch.Tensor")
    if x.dim() != 4:
        raise ValueError("expected [N, C, H, W], got shape %s" % (tuple(x.shape),))
    if x.dtype != torch.float32:
        raise TypeError("feature maps must be float32 (the reference path is fp32), got %s" % x.dtype)
    if not x.is_cuda:
        raise RuntimeError(
            "dct_pruning_amd runs on the GPU only: got a %s tensor. There is no CPU fallback; "
            "the CPU restatement under oracle/ is test infrastructure." % x.device)


def _slice(x, c_begin, c_count):
    C = x.shape[1]
    if c_count is None:
        c_count = C - c_begin
    return int(c_begin), int(c_count)


def has_codelet(H, W):
    return bool(_lib.load().dcts_has_codelet(H, W))


def _call(fn_name, x, c_begin, c_count, pad_front_if_odd, out, algo):
    lib = _lib.load()
    N, C, H, W = x.shape
    if x.stride(3) != 1 or x.stride(2) < W:
        x = x.contiguous()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    nbytes = lib.dcts_workspace_bytes(N, c_count, H, W)
    ws = _workspace(x.device, stream, nbytes)
    with torch.cuda.device(x.device):
        code = getattr(lib, fn_name)(
            x.data_ptr(), N, C, H, W, x.stride(0), x.stride(1), x.stride(2), x.stride(3),
            c_begin, c_count, 1 if pad_front_if_odd else 0, out.data_ptr(),
            ws.data_ptr(), ws.numel(), stream, algo)
    _lib.check(code)
    return out


def energy_nc(x, c_begin=0, c_count=None, pad_front_if_odd=False, algo=ALGO_AUTO, out=None):
    """E[n, j] = sum_{u,v} dct_2d(x[n, c_begin+j], norm='ortho')[u,v]**2  -> [N, c_count] fp32.

    pad_front_if_odd=True reproduces torch2dct (utils/common.py:230-239): an odd-H map gets
    one zero row and one zero column in front before the transform.
    Enqueues on the current stream of x's device; no synchronisation.
    """
    _check_input(x)
    c_begin, c_count = _slice(x, c_begin, c_count)
    N = x.shape[0]
    if out is None:
        out = torch.empty((N, c_count), dtype=torch.float32, device=x.device)
    elif out.shape != (N, c_count) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError("out must be a contiguous float32 [N, c_count] tensor on x's device")
    return _call("dcts_energy_f32_ex", x, c_begin, c_count, pad_front_if_odd, out, algo)


def dct2d(x, c_begin=0, c_count=None, pad_front_if_odd=False, algo=ALGO_AUTO):
    """Orthonormal 2-D DCT-II coefficients of every map -> [N, c_count, H', W'] fp32."""
    _check_input(x)
    c_begin, c_count = _slice(x, c_begin, c_count)
    N, _, H, W = x.shape
    pad = 1 if (pad_front_if_odd and H % 2 == 1) else 0
    out = torch.empty((N, c_count, H + pad, W + pad), dtype=torch.float32, device=x.device)
    return _call("dcts_dct2d_f32_ex", x, c_begin, c_count, pad_front_if_odd, out, algo)


def batch_sum(energy):
    """out[j] = sum_n energy[n, j], n ascending (fused variant for the bench / single-sweep mode)."""
    if energy.dim() != 2 or energy.dtype != torch.float32 or not energy.is_cuda:
        raise ValueError("expected a float32 CUDA tensor [N, C]")
    energy = energy.contiguous()
    out = torch.empty((energy.shape[1],), dtype=torch.float32, device=energy.device)
    stream = torch.cuda.current_stream(energy.device).cuda_stream
    with torch.cuda.device(energy.device):
        _lib.check(_lib.load().dcts_batch_sum_f32(energy.data_ptr(), energy.shape[0], energy.shape[1],
                                                  out.data_ptr(), stream))
    return out


def energy_multi(items, pad_front_if_odd=False):
    """energy_nc for several tensors of the SAME (H, W) in one launch.

    items: list of (x, c_begin, c_count) with x [N, C, H, W] fp32 CUDA (c_count None = to the end).
    Returns the list of [N, c_count] outputs. Tensors must stay alive until the stream has run."""
    lib = _lib.load()
    if not items:
        return []
    H, W = items[0][0].shape[2], items[0][0].shape[3]
    dev = items[0][0].device
    arr = (_lib.TensorItem * len(items))()
    outs, keep = [], []
    need = 0
    for i, (x, c_begin, c_count) in enumerate(items):
        _check_input(x)
        if x.shape[2] != H or x.shape[3] != W or x.device != dev:
            raise ValueError("energy_multi needs tensors of one tile shape on one device")
        if x.stride(3) != 1 or x.stride(2) != W:
            x = x.contiguous()
        c_begin, c_count = _slice(x, c_begin, c_count)
        out = torch.empty((x.shape[0], c_count), dtype=torch.float32, device=dev)
        keep.append(x)
        outs.append(out)
        arr[i].x, arr[i].out_nc = x.data_ptr(), out.data_ptr()
        arr[i].N, arr[i].C_total = x.shape[0], x.shape[1]
        arr[i].strideN, arr[i].strideC = x.stride(0), x.stride(1)
        arr[i].c_begin, arr[i].c_count = c_begin, c_count
        need = max(need, lib.dcts_workspace_bytes(x.shape[0], c_count, H, W))
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = _workspace(dev, stream, need)
    with torch.cuda.device(dev):
        _lib.check(lib.dcts_energy_multi_f32(arr, len(items), H, W, 1 if pad_front_if_odd else 0,
                                             ws.data_ptr(), ws.numel(), stream))
    return outs


def energy_mixed(items):
    """energy_nc for tensors of ANY tile shapes in one call (dcts_energy_mixed_f32): small square tiles
    (edges 2..32) of all shapes share one launch, the rest go shape by shape.

    items: list of (x, c_begin, c_count, pad_front_if_odd). Returns the list of [N, c_count] outputs;
    the tensors must stay alive until the stream has run."""
    lib = _lib.load()
    if not items:
        return []
    dev = items[0][0].device
    arr = (_lib.ShapedItem * len(items))()
    outs, keep = [], []
    need = 0
    for i, (x, c_begin, c_count, pad) in enumerate(items):
        _check_input(x)
        if x.device != dev:
            raise ValueError("energy_mixed needs tensors on one device")
        H, W = x.shape[2], x.shape[3]
        if x.stride(3) != 1 or x.stride(2) != W:
            x = x.contiguous()
        c_begin, c_count = _slice(x, c_begin, c_count)
        out = torch.empty((x.shape[0], c_count), dtype=torch.float32, device=dev)
        keep.append(x)
        outs.append(out)
        t = arr[i].t
        t.x, t.out_nc = x.data_ptr(), out.data_ptr()
        t.N, t.C_total = x.shape[0], x.shape[1]
        t.strideN, t.strideC = x.stride(0), x.stride(1)
        t.c_begin, t.c_count = c_begin, c_count
        arr[i].H, arr[i].W, arr[i].pad_front_if_odd = H, W, 1 if pad else 0
        need = max(need, lib.dcts_workspace_bytes(x.shape[0], c_count, H, W))
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = _workspace(dev, stream, need)
    with torch.cuda.device(dev):
        _lib.check(lib.dcts_energy_mixed_f32(arr, len(items), ws.data_ptr(), ws.numel(), stream))
    return outs


def weighted_energy_nc(x, weights, c_begin=0, c_count=None, pad_front_if_odd=False):
    """Coefficient-domain score variant: E[n, j] = sum_{u,v} weights[u,v] * dct_2d(x[n, c_begin+j])[u,v]**2.
    `weights`: [H', W'] fp32 on x's device (H' = H + 1 for an odd H with pad_front_if_odd). All ones gives energy_nc."""
    _check_input(x)
    c_begin, c_count = _slice(x, c_begin, c_count)
    N, C, H, W = x.shape
    pad = 1 if (pad_front_if_odd and H % 2 == 1) else 0
    if weights.shape != (H + pad, W + pad) or weights.dtype != torch.float32 or weights.device != x.device:
        raise ValueError("weights must be a float32 [%d, %d] tensor on %s" % (H + pad, W + pad, x.device))
    if x.stride(3) != 1 or x.stride(2) < W:
        x = x.contiguous()
    weights = weights.contiguous()
    lib = _lib.load()
    out = torch.empty((N, c_count), dtype=torch.float32, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    ws = _workspace(x.device, stream, lib.dcts_weighted_workspace_bytes(N, c_count, H, W))
    with torch.cuda.device(x.device):
        _lib.check(lib.dcts_weighted_energy_f32(
            x.data_ptr(), N, C, H, W, x.stride(0), x.stride(1), x.stride(2), x.stride(3), c_begin, c_count,
            1 if pad_front_if_odd else 0, weights.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), stream))
    return out
