"""The large-tile kernels compute the DCT, not just an isometry: coefficients through the split
kernels themselves (dcts_dct2d_f32_ex with DCTS_ALGO_FUSED / DCTS_ALGO_TILE2D: the energy kernel's
own butterflies, rotations and leaf codelets with the leaf outputs stored, then the DCT-IV add/sub
layers the energy path folds into its weights, k_assemble) against the oracle's float64 transform
(scipy dctn == torch_dct.dct_2d(norm='ortho'), utils/common.py:267). Energy-only tests cannot tell
a DCT from any other orthogonal transform (Parseval); these can."""
import numpy as np
import pytest
import torch

import dct_pruning_amd as dpa
from oracle import dct_oracle as orc
from helpers import synth

pytestmark = pytest.mark.gpu

TOL = 2e-6  # of the coefficient scale (max |coefficient| of the map batch)


@pytest.mark.parametrize("edge,algo", [(72, "FUSED"), (80, "FUSED"), (112, "FUSED"), (128, "FUSED"), (144, "FUSED"),
                                       (160, "FUSED"), (224, "FUSED"), (256, "FUSED"), (288, "FUSED"), (320, "FUSED"),
                                       (224, "TILE2D"), (72, "TILE2D"), (80, "TILE2D"), (112, "TILE2D"), (128, "TILE2D"),
                                       (144, "TILE2D"), (160, "TILE2D")])
def test_large_tile_coefficients_vs_float64(edge, algo):
    x = synth(1, 3, edge, edge, 400 + edge, dead=False)
    got = dpa.dct2d(x.cuda(), algo=getattr(dpa, "ALGO_" + algo)).cpu().numpy()
    ref = orc.dct_2d_f64(x.numpy())
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL * np.abs(ref).max()
    # and the energy path of the same family agrees with the sum of these coefficients squared
    e = dpa.energy_nc(x.cuda(), algo=getattr(dpa, "ALGO_" + algo)).cpu().numpy()
    assert np.allclose(e, (ref ** 2).sum(axis=(-2, -1)), rtol=1e-5, atol=0)


@pytest.mark.parametrize("algo", ["FUSED", "TILE2D"])
def test_single_basis_function_known_answer_224(algo):
    """x = outer(C[u, :], C[v, :]) has exactly one unit coefficient, at (u, v)."""
    n = 224
    k = np.arange(n)
    rows = []
    picks = [(0, 0), (1, 0), (0, 223), (5, 17), (111, 112), (223, 223), (28, 56), (27, 29), (113, 2)]
    for u, v in picks:
        cu = np.cos(np.pi * (2 * k + 1) * u / (2 * n)) * (np.sqrt(1.0 / n) if u == 0 else np.sqrt(2.0 / n))
        cv = np.cos(np.pi * (2 * k + 1) * v / (2 * n)) * (np.sqrt(1.0 / n) if v == 0 else np.sqrt(2.0 / n))
        rows.append(np.outer(cu, cv))
    x = torch.from_numpy(np.stack(rows)[None].astype(np.float32))
    got = dpa.dct2d(x.cuda(), algo=getattr(dpa, "ALGO_" + algo)).cpu().numpy()[0]
    for i, (u, v) in enumerate(picks):
        want = np.zeros((n, n))
        want[u, v] = 1.0
        assert np.abs(got[i] - want).max() <= 5e-6, (u, v)


@pytest.mark.parametrize("n", [72, 144])
def test_single_basis_function_known_answer_tile2g(n):
    """The same known-answer test for the several-maps-per-round kernel (tile2g.hip), 7 maps = two full rounds and a
    short one: x = outer(C[u, :], C[v, :]) has exactly one unit coefficient, at (u, v)."""
    k = np.arange(n)
    rows = []
    picks = [(0, 0), (1, 0), (0, n - 1), (5, 17), (n // 2 - 1, n // 2), (n - 1, n - 1), (n // 8, n // 4)]
    for u, v in picks:
        cu = np.cos(np.pi * (2 * k + 1) * u / (2 * n)) * (np.sqrt(1.0 / n) if u == 0 else np.sqrt(2.0 / n))
        cv = np.cos(np.pi * (2 * k + 1) * v / (2 * n)) * (np.sqrt(1.0 / n) if v == 0 else np.sqrt(2.0 / n))
        rows.append(np.outer(cu, cv))
    x = torch.from_numpy(np.stack(rows)[None].astype(np.float32))
    got = dpa.dct2d(x.cuda(), algo=dpa.ALGO_TILE2D).cpu().numpy()[0]
    for i, (u, v) in enumerate(picks):
        want = np.zeros((n, n))
        want[u, v] = 1.0
        assert np.abs(got[i] - want).max() <= 5e-6, (u, v)


def test_chunked_workspace_gives_the_same_coefficients():
    """More maps than the workspace holds tiles: the coefficient path runs in chunks."""
    x = synth(2, 40, 72, 72, 77, dead=False).cuda()
    ref = orc.dct_2d_f64(x.cpu().numpy())
    for algo in (dpa.ALGO_FUSED, dpa.ALGO_TILE2D):  # tile2g rounds of 3 maps against chunks of whatever the workspace holds
        a = dpa.dct2d(x, algo=algo)
        assert np.abs(a.cpu().numpy() - ref).max() <= TOL * np.abs(ref).max()


@pytest.mark.parametrize("edge", [8, 9, 32, 56, 72, 224, 288, 24])
def test_frequency_weighted_score_variant(edge):
    """dcts_weighted_energy_f32: sum of w[u,v] * coeff^2, against float64; all-ones weights
    give the plain energy; a low-pass weight is NOT Parseval-equivalent to anything in the spatial domain, so
    this is the test in which the coefficients themselves matter."""
    pad = edge % 2 == 1
    x = synth(2, 5, edge, edge, 600 + edge, dead=True).cuda()
    hp = edge + (1 if pad else 0)
    g = torch.Generator().manual_seed(edge)
    u = torch.arange(hp, dtype=torch.float32)
    for w in (torch.ones(hp, hp), torch.rand(hp, hp, generator=g), torch.exp(-(u[:, None] + u[None, :]) / (0.25 * hp))):
        got = dpa.weighted_energy_nc(x, w.cuda(), pad_front_if_odd=pad).cpu().numpy()
        ref = orc.weighted_energy_nc_f64(x, w.numpy(), pad_front_if_odd=pad)
        assert np.allclose(got, ref, rtol=2e-5, atol=1e-6 * ref.max())
        assert (got[:, 5 % x.shape[1]] >= 0).all()
    ones = dpa.weighted_energy_nc(x, torch.ones(hp, hp).cuda(), pad_front_if_odd=pad)
    assert torch.allclose(ones, dpa.energy_nc(x, pad_front_if_odd=pad), rtol=1e-5, atol=0)
    # a channel slice
    sl = dpa.weighted_energy_nc(x, torch.ones(hp, hp).cuda(), c_begin=2, c_count=2, pad_front_if_odd=pad)
    assert torch.equal(sl, ones[:, 2:4])
