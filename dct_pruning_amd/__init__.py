"""dct_pruning_amd — MI355X-native DCT importance-score path of semchan/DCT_Pruning.

Scope: forward-hooked feature maps [N,C,H,W] -> per-map orthonormal 2-D
DCT-II -> sum of squared coefficients -> per-channel running mean -> .npy score files
(reference: utils/common.py:230-309 and :367-977). The arithmetic runs in hand-written
gfx950 HIP kernels behind the C ABI of include/dctscore.h; this package is the host-side
mirror of the reference's hook / imp_score interface.
"""
from .ops import (  # noqa: F401
    ALGO_AUTO,
    ALGO_CODELET,
    ALGO_DIRECT,
    ALGO_FUSED,
    ALGO_PIPE,
    ALGO_LANE,
    ALGO_PREFETCH,
    ALGO_SPLIT,
    ALGO_TILE2D,
    ALGO_RECT,
    batch_sum,
    dct2d,
    energy_mixed,
    energy_multi,
    energy_nc,
    has_codelet,
    weighted_energy_nc,
)

__all__ = ["energy_nc", "energy_multi", "energy_mixed", "dct2d", "batch_sum", "has_codelet", "weighted_energy_nc", "ALGO_AUTO", "ALGO_DIRECT", "ALGO_CODELET", "ALGO_SPLIT", "ALGO_PREFETCH", "ALGO_FUSED", "ALGO_PIPE", "ALGO_LANE", "ALGO_TILE2D", "ALGO_RECT"]
