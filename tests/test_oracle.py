"""The CPU oracle against independent references: SciPy float64, Parseval, analytic vectors
(SURVEY.md §8c 'known-answer tests to adopt'). CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import dct_oracle as orc

SIZES = [2, 4, 7, 8, 9, 10, 14, 16, 18, 20, 28, 32, 36, 40, 56, 72, 80, 144, 224]


def _relu_maps(n, c, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.relu(torch.randn(n, c, h, w, generator=g))


@pytest.mark.parametrize("n", SIZES)
def test_coefficients_match_scipy_f64(n):
    x = _relu_maps(1, 2, n, n, 100 + n)
    got = orc.dct_2d(x).numpy().astype(np.float64)
    ref = orc.dct_2d_f64(x.numpy())
    assert np.abs(got - ref).max() <= 3e-6 * np.abs(ref).max()


@pytest.mark.parametrize("hw", [(9, 18), (7, 14), (32, 16), (5, 3)])
def test_non_square(hw):
    h, w = hw
    x = _relu_maps(1, 1, h, w, 7)
    got = orc.dct_2d(x).numpy().astype(np.float64)
    ref = orc.dct_2d_f64(x.numpy())
    assert np.abs(got - ref).max() <= 3e-6 * np.abs(ref).max()


@pytest.mark.parametrize("n", SIZES)
def test_parseval(n):
    x = _relu_maps(2, 3, n, n, 200 + n)
    e = orc.energy_nc(x).numpy().astype(np.float64)
    ref = (x.numpy().astype(np.float64) ** 2).sum(axis=(-2, -1))
    np.testing.assert_allclose(e, ref, rtol=2e-6)


def test_loop_and_batched_agree():
    x = _relu_maps(3, 5, 14, 14, 3)
    np.testing.assert_allclose(orc.energy_nc(x).numpy(), orc.energy_nc_batched(x).numpy(), rtol=1e-6)
    np.testing.assert_allclose(orc.energy_nc(x).numpy(), orc.energy_nc_f64(x), rtol=2e-6)


def test_zero_map_is_plus_zero():
    x = torch.zeros(1, 2, 8, 8)
    e = orc.energy_nc(x).numpy()
    assert (e == 0).all() and not np.signbit(e).any()


def test_constant_map():
    c, n = 1.5, 16
    x = torch.full((1, 1, n, n), c)
    d = orc.dct_2d(x)[0, 0].numpy()
    assert abs(d[0, 0] - c * n) < 1e-4
    d[0, 0] = 0
    assert np.abs(d).max() < 1e-4
    assert abs(orc.energy_nc(x).item() - c * c * n * n) / (c * c * n * n) < 1e-6


def test_single_basis_function():
    n, u, v = 8, 3, 5
    i = np.arange(n)
    cu = math.sqrt(2 / n) * np.cos(math.pi * (2 * i + 1) * u / (2 * n))
    cv = math.sqrt(2 / n) * np.cos(math.pi * (2 * i + 1) * v / (2 * n))
    x = torch.tensor(np.outer(cu, cv), dtype=torch.float32)[None, None]
    d = orc.dct_2d(x)[0, 0].numpy()
    assert abs(d[u, v] - 1) < 1e-5
    d[u, v] = 0
    assert np.abs(d).max() < 1e-5


def test_torch2dct_front_pad():
    # np.pad(t,(1,0)) pads BOTH axes when shape[0] is odd (utils/common.py:235-236)
    x = _relu_maps(1, 1, 9, 9, 11)[0, 0]
    d = orc.torch2dct(x)
    assert d.shape == (10, 10)
    assert abs((d * d).sum().item() - (x * x).sum().item()) <= 2e-6 * (x * x).sum().item()
    even = _relu_maps(1, 1, 8, 8, 12)[0, 0]
    assert orc.torch2dct(even).shape == (8, 8)
    e = orc.energy_nc(_relu_maps(2, 3, 9, 9, 5), pad_front_if_odd=True)
    assert e.shape == (2, 3)


def test_running_mean_rule():
    st = orc.HookState()
    a = _relu_maps(4, 6, 8, 8, 1)
    b = _relu_maps(2, 6, 8, 8, 2)
    orc.get_feature_hook(st, a)
    orc.get_feature_hook(st, b)
    ref = torch.cat([orc.energy_nc(a), orc.energy_nc(b)]).double().mean(0)
    np.testing.assert_allclose(st.feature_result.numpy(), ref.numpy(), rtol=1e-6)
    assert st.total.item() == 6 and st.feature_result.dtype == torch.float32


def test_densenet_hook_takes_last_12():
    st = orc.HookState()
    x = _relu_maps(2, 36, 8, 8, 9)
    orc.get_feature_hook_densenet(st, x)
    ref = orc.energy_nc(x, 24, 12, pad_front_if_odd=True).sum(0) / 2
    assert st.feature_result.shape == (12,)
    np.testing.assert_allclose(st.feature_result.numpy(), ref.numpy(), rtol=1e-6)


def test_u2net_input_hook():
    st = orc.HookState()
    x = _relu_maps(2, 4, 9, 9, 10)
    orc.get_feature_hook_u2net_input(st, (x,))
    ref = orc.energy_nc(x, pad_front_if_odd=True).sum(0) / 2
    np.testing.assert_allclose(st.feature_result.numpy(), ref.numpy(), rtol=1e-6)


def test_select_index_rule():
    imp = np.array([3., 0., 5., 0., 1., 4.], dtype=np.float32)
    np.testing.assert_array_equal(orc.select_index(imp, 6, 3), [0, 2, 5])
    assert orc.kept_filters(64, 0.5) == 32 and orc.kept_filters(512, 0.95) == 25


def test_published_jpeg_worked_example():
    """A vector that neither SciPy nor this repo produced: the 8x8 block of the JPEG literature's worked example and its
    forward DCT as published (to two decimals). The JPEG FDCT is the orthonormal 2-D DCT-II of the level-shifted samples,
    i.e. what torch_dct.dct_2d(norm='ortho') / cv2.dct compute at utils/common.py:267 / :237. It pins the oracle's
    transform DEFINITION and normalisation against a published result (to the publication's 5e-3, ~1e-5 of the DC term);
    the round-off of torch_dct / cv2 themselves stays unpinned (neither is installable here)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_example_8x8.json")))
    x = torch.tensor(g["block"], dtype=torch.float32) + g["level_shift"]
    want = np.array(g["published_dct_rows_0_to_2"])
    tol = g["published_precision"] + 1e-3
    for got in (orc.dct_2d(x).numpy(), orc.dct_2d_f64(x.numpy()[None])[0], orc.torch2dct(x).numpy()):
        assert np.abs(got[:3] - want).max() <= tol
    # the score of that map (Parseval: sum of the level-shifted samples squared) through every oracle entry point
    e = float((x.double() ** 2).sum())
    assert abs(float(orc.energy_nc(x[None, None])[0, 0]) - e) <= 1e-5 * e
    assert abs(float(orc.energy_nc_batched(x[None, None])[0, 0]) - e) <= 1e-5 * e
