"""ctypes binding of libdctscore.so (include/dctscore.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, every
entry point raises. A CPU stand-in here would make parity claims meaningless.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdctscore.so")
ABI_VERSION = 2

_lock = threading.Lock()
_lib = None

_i64, _i32, _vp, _sz = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t

# symbol -> (restype, argtypes); mirrors include/dctscore.h one to one
SIGNATURES = {
    "dcts_version": (ctypes.c_int, []),
    "dcts_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "dcts_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "dcts_has_codelet": (ctypes.c_int, [_i64, _i64]),
    "dcts_workspace_invalidate": (None, [_vp]),
    "dcts_workspace_invalidate_range": (None, [_vp, _sz]),
    "dcts_energy_f32": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                       _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dcts_energy_f32_ex": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                          _i32, _i32, _i32, _vp, _vp, _sz, _vp, _i32]),
    "dcts_dct2d_f32": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                      _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dcts_dct2d_f32_ex": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                         _i32, _i32, _i32, _vp, _vp, _sz, _vp, _i32]),
    "dcts_weighted_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "dcts_weighted_energy_f32": (ctypes.c_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                                _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "dcts_batch_sum_f32": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "dcts_running_mean_update_f32": (ctypes.c_int, [_vp, _i64, _i64, _vp, ctypes.c_float, _vp]),
    "dcts_energy_multi_f32": (ctypes.c_int, [_vp, _i32, _i64, _i64, _i32, _vp, _sz, _vp]),
    "dcts_energy_mixed_f32": (ctypes.c_int, [_vp, _i32, _vp, _sz, _vp]),
    "dcts_running_mean_update_multi_f32": (ctypes.c_int, [_vp, _i32, _vp]),
    "dcts_debug_stream_read_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
}


class UpdateDesc(ctypes.Structure):
    """struct dcts_update_desc (include/dctscore.h)."""
    _fields_ = [("energy_nc", _vp), ("feature_result", _vp), ("N", _i64), ("C_count", _i64),
                ("total_before", ctypes.c_float), ("reserved", _i32)]


class TensorItem(ctypes.Structure):
    """struct dcts_tensor_item (include/dctscore.h)."""
    _fields_ = [("x", _vp), ("out_nc", _vp), ("N", _i64), ("C_total", _i64), ("strideN", _i64), ("strideC", _i64),
                ("c_begin", _i32), ("c_count", _i32)]


class ShapedItem(ctypes.Structure):
    """struct dcts_shaped_item (include/dctscore.h)."""
    _fields_ = [("t", TensorItem), ("H", _i64), ("W", _i64), ("pad_front_if_odd", _i32), ("reserved", _i32)]


class DctScoreError(RuntimeError):
    """A libdctscore entry point returned a non-zero code."""

    def __init__(self, code, text):
        super().__init__("libdctscore error %d: %s" % (code, text))
        self.code = code


def load():
    """Load libdctscore.so once and declare every prototype; raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                "libdctscore.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dct_pruning_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        got = lib.dcts_version()
        if got != ABI_VERSION:
            raise ImportError("libdctscore ABI %d != expected %d" % (got, ABI_VERSION))
        _lib = lib
        return lib


def check(code):
    if code != 0:
        raise DctScoreError(code, load().dcts_strerror(code).decode())
