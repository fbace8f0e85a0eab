"""Seeded differential fuzz of the C ABI against the oracle: random tile shapes (square sizes of every
kernel family, non-square and odd ones for the run-time codelet pair and the direct kernel), batch/channel counts, channel slices,
odd front pad, batch-strided and channel-strided views, single and multi-tensor entry points."""
import numpy as np
import pytest
import torch

import dct_pruning_amd as dpa
from oracle import dct_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-4
EDGES = [2, 4, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24, 28, 30, 32, 36, 40, 48, 56, 60, 64, 72, 80, 96, 112, 128, 144, 160, 192, 224, 256]


def _rel(got, ref):
    got, ref = got.double().cpu(), ref.double()
    nz = ref != 0
    if nz.any():
        assert ((got - ref).abs() / ref.abs().clamp_min(1e-30))[nz].max().item() <= RTOL
    g = got[~nz]
    assert (g == 0).all() and not torch.signbit(g).any()


def _case(rng):
    kind = rng.integers(0, 10)
    if kind < 7:
        h = w = int(rng.choice(EDGES))
    elif kind < 9:
        h, w = int(rng.integers(1, 65)), int(rng.integers(1, 65))  # any pair up to 64: the run-time codelet pair (rect.hip)
    else:
        h = w = int(rng.integers(65, 130))  # no codelet / split entry for most of these: direct kernel
    big = h * w >= 72 * 72
    n = int(rng.integers(1, 3 if big else 6))
    c = int(rng.integers(1, 24 if big else 90))
    return n, c, h, w


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_single_tensor(seed):
    rng = np.random.default_rng(1000 + seed)
    g = torch.Generator().manual_seed(2000 + seed)
    for _ in range(14):
        n, c, h, w = _case(rng)
        x = torch.relu(torch.randn(n, c, h, w, generator=g)) * torch.exp(0.5 * torch.randn(c, generator=g))[None, :, None, None]
        x[:, torch.arange(c) % 5 == 3] = 0
        cb = int(rng.integers(0, c))
        cc = int(rng.integers(1, c - cb + 1))
        pad = bool(rng.integers(0, 2))
        view = int(rng.integers(0, 3))
        xg = x.cuda()
        if view == 1 and n > 1:      # batch-strided view
            xg, x = xg[::2], x[::2]
        elif view == 2 and c > 2:    # channel-strided parent: slice of a wider tensor
            wide = torch.zeros(x.shape[0], c + 3, h, w, device="cuda")
            wide[:, 2:2 + c] = xg
            xg = wide[:, 2:2 + c]
        got = dpa.energy_nc(xg, c_begin=cb, c_count=cc, pad_front_if_odd=pad)
        ref = orc.energy_nc_batched(x, c_begin=cb, c_count=cc, pad_front_if_odd=pad)
        assert got.shape == ref.shape
        _rel(got, ref)
        assert torch.equal(got, dpa.energy_nc(xg, c_begin=cb, c_count=cc, pad_front_if_odd=pad))  # reproducible


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_multi_tensor(seed):
    rng = np.random.default_rng(3000 + seed)
    g = torch.Generator().manual_seed(4000 + seed)
    for _ in range(5):
        h = int(rng.choice([2, 4, 7, 8, 9, 14, 16, 28, 32, 56, 72, 128]))
        pad = bool(rng.integers(0, 2))
        items, hosts = [], []
        for _ in range(int(rng.integers(1, 40 if h <= 32 else 6))):
            n, c = int(rng.integers(1, 5)), int(rng.integers(1, 40 if h <= 32 else 8))
            x = torch.relu(torch.randn(n, c, h, h, generator=g))
            cb = int(rng.integers(0, c))
            cc = int(rng.integers(1, c - cb + 1))
            items.append((x.cuda(), cb, cc))
            hosts.append(x)
        outs = dpa.energy_multi(items, pad_front_if_odd=pad)
        for (xg, cb, cc), x, got in zip(items, hosts, outs):
            _rel(got, orc.energy_nc_batched(x, c_begin=cb, c_count=cc, pad_front_if_odd=pad))
            assert torch.equal(got, dpa.energy_nc(xg, c_begin=cb, c_count=cc, pad_front_if_odd=pad))
