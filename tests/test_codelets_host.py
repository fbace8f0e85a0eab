"""dct_codelets.hpp compiled for the host (tests/native) against SciPy float64: the same
header the HIP kernels include, so the factorisation is checked without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.fft import dct, dctn

SIZES = [2, 4, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24, 28, 30, 32, 36, 40, 48, 56, 60, 64]
FP = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def lib(repo_root):
    d = os.path.join(repo_root, "tests", "native")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return ctypes.CDLL(os.path.join(d, "_build", "libcodelet_host.so"))


@pytest.mark.parametrize("n", SIZES)
def test_dct2_1d(lib, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    out = np.zeros(n, np.float32)
    assert lib.codelet_dct2_1d(n, x.ctypes.data_as(FP), out.ctypes.data_as(FP)) == 0
    ref = dct(x.astype(np.float64), type=2) / 2
    assert np.abs(out - ref).max() <= 4e-7 * max(np.abs(ref).max(), 1.0) * np.sqrt(n)


@pytest.mark.parametrize("n", [n for n in range(1, 65) if n not in SIZES])
def test_dct2_1d_lengths_without_a_square_kernel(lib, n):
    """rect.hip instantiates the codelet template for every length up to 64 (odd parts by the direct sum)."""
    rng = np.random.default_rng(3000 + n)
    x = rng.standard_normal(n).astype(np.float32)
    out = np.zeros(n, np.float32)
    assert lib.codelet_dct2_1d_any(n, x.ctypes.data_as(FP), out.ctypes.data_as(FP)) == 0
    ref = dct(x.astype(np.float64), type=2) / 2
    assert np.abs(out - ref).max() <= 4e-7 * max(np.abs(ref).max(), 1.0) * np.sqrt(n)


@pytest.mark.parametrize("n", SIZES)
def test_dct4_1d(lib, n):
    rng = np.random.default_rng(1000 + n)
    x = rng.standard_normal(n).astype(np.float32)
    out = np.zeros(n, np.float32)
    assert lib.codelet_dct4_1d(n, x.ctypes.data_as(FP), out.ctypes.data_as(FP)) == 0
    ref = dct(x.astype(np.float64), type=4) / 2
    assert np.abs(out - ref).max() <= 4e-7 * max(np.abs(ref).max(), 1.0) * np.sqrt(n)


@pytest.mark.parametrize("n", SIZES)
def test_energy_2d(lib, n):
    rng = np.random.default_rng(2000 + n)
    x = np.maximum(rng.standard_normal((n, n)), 0).astype(np.float32)
    coeff = np.zeros((n, n), np.float32)
    e = ctypes.c_float()
    assert lib.codelet_energy_2d(n, x.ctypes.data_as(FP), coeff.ctypes.data_as(FP), ctypes.byref(e)) == 0
    ref = dctn(x.astype(np.float64), type=2, norm="ortho")
    assert np.abs(coeff - ref).max() <= 1e-6 * np.abs(ref).max()
    assert abs(e.value - (ref ** 2).sum()) <= 2e-6 * (ref ** 2).sum()


def test_zero_input_gives_plus_zero(lib):
    n = 14
    x = np.zeros((n, n), np.float32)
    e = ctypes.c_float(1.0)
    assert lib.codelet_energy_2d(n, x.ctypes.data_as(FP), None, ctypes.byref(e)) == 0
    assert e.value == 0.0 and not np.signbit(np.float32(e.value))
