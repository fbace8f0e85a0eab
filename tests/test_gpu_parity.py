"""GPU parity proper: libdctscore (through the C ABI) against the CPU oracle on the same
seeded inputs. Tolerance: 1e-4 relative on fp32 energies (BASELINE.json north_star); the
observed error is ~1e-6. Exact +0.0 for dead channels; prune masks identical."""
import numpy as np
import pytest
import torch

import dct_pruning_amd as dpa
from oracle import dct_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-4
CODELET = [2, 4, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24, 28, 30, 32, 36, 40, 48, 56, 60, 64]
DIRECT_ONLY = [3, 5, 13, 22, 72, 80, 144, 224]


def synth(n, c, h, w, seed, dead=True):
    """SURVEY.md §8(d) synthetic maps: relu(randn) * per-channel scale, every c % 8 == 5 dead."""
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, c, h, w, generator=g))
    s = torch.exp(0.5 * torch.randn(c, generator=g))
    if dead:
        s[torch.arange(c) % 8 == 5] = 0
    return x * s[None, :, None, None]


def rel_err(got, ref):
    got = got.double()
    ref = ref.double()
    return ((got - ref).abs() / ref.abs().clamp_min(1e-30))[ref != 0].max().item() if (ref != 0).any() else 0.0


def check(x, got, **kw):
    ref = orc.energy_nc_batched(x, **kw)
    assert got.shape == ref.shape
    assert rel_err(got.cpu(), ref) <= RTOL
    zero = ref == 0
    g = got.cpu()
    assert (g[zero] == 0).all() and not torch.signbit(g[zero]).any()


@pytest.mark.parametrize("n", CODELET)
def test_codelet_sizes(n):
    x = synth(3, 19, n, n, 10 + n)
    got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_CODELET)
    check(x, got)
    auto = dpa.energy_nc(x.cuda())
    if n in (7, 9):  # AUTO = the lane-per-map kernel there (different summation order)
        assert rel_err(auto.cpu(), got.cpu()) <= 1e-5
    else:
        assert torch.equal(auto, got)  # AUTO = the register-load codelet kernel
    if n % 2 == 0:
        # opt-in prefetching (direct-to-LDS) variant; many groups per wave exercise its loop
        check(x, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_PREFETCH))
        big = synth(40, 77, n, n, 11 + n).cuda()
        pre = dpa.energy_nc(big, algo=dpa.ALGO_PREFETCH)
        check(big.cpu(), pre)
        assert torch.equal(pre, dpa.energy_nc(big, algo=dpa.ALGO_PREFETCH))
    # per-map loop oracle on a subset (the reference's exact loop structure)
    ref = orc.energy_nc(x[:1, :4])
    assert rel_err(got[:1, :4].cpu(), ref) <= RTOL


@pytest.mark.parametrize("n", CODELET + DIRECT_ONLY)
def test_direct_sizes(n):
    c = 5 if n >= 144 else 11
    x = synth(2, c, n, n, 20 + n)
    got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_DIRECT)
    check(x, got)


SPLIT = [72, 80, 96, 112, 128, 144, 160, 192, 224, 256, 288, 320]
# round 3: the other multiples of 4 (<= 256) / 16 (<= 512) whose quarter / eighth is a codelet length: two-launch path only
SPLIT_MORE = [68, 76, 84, 88, 92, 100, 104, 108, 116, 120, 124, 136, 152, 168, 176, 184, 200, 208, 216, 232, 240, 248,
              272, 304, 336, 352, 368, 384, 400, 416, 432, 448, 464, 480, 496, 512]
FUSED = [72, 80, 96, 112, 128, 144, 160, 192, 224, 256, 288, 320]
TILE2G = [72, 80, 96, 112, 128, 144, 160]   # tile2g.hip: several maps per round (DCTS_ALGO_TILE2D selects it for these edges)
TILE2G_AUTO = [72, 80, 144, 160]        # ... and AUTO takes it for these


@pytest.mark.parametrize("n", SPLIT + SPLIT_MORE)
def test_split_sizes(n):
    """Large tiles: two-launch split-4 codelet passes (N = 4*M)."""
    x = synth(2, 11 if n <= 320 else 5, n, n, 30 + n)
    got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_SPLIT)
    check(x, got)
    if n not in FUSED:
        assert torch.equal(got, dpa.energy_nc(x.cuda()))  # AUTO picks the same kernels
    ref = orc.energy_nc(x[:1, :2])
    assert rel_err(got[:1, :2].cpu(), ref) <= RTOL


@pytest.mark.parametrize("n", FUSED)
def test_fused_sizes(n):
    """Single-launch split kernel (intermediate tile parked in VGPRs): more maps than workgroups so
    every workgroup loops and the double-buffered staging wraps around."""
    x = synth(2, 150 if n >= 200 else 700, n, n, 130 + n)
    got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_FUSED)
    check(x, got)
    if n not in PIPE and n not in TILE2G_AUTO:
        assert torch.equal(got, dpa.energy_nc(x.cuda()))  # AUTO picks it
    assert torch.equal(got, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_FUSED))  # bit-reproducible
    two = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_SPLIT)
    assert rel_err(got.cpu(), two.cpu()) <= 1e-5
    few = synth(1, 3, n, n, 131 + n)  # fewer maps than workgroups
    check(few, dpa.energy_nc(few.cuda(), algo=dpa.ALGO_FUSED))


PIPE = [128, 224]


@pytest.mark.parametrize("n", PIPE)
def test_pipelined_sizes(n):
    """Software-pipelined fused kernel (pass 2 of map m interleaved with pass 1 of map m+1). Map counts
    chosen so that workgroups run 1, 2, 3 and 4+ maps (prologue / odd and even steady iterations /
    epilogue with either parity) and so that fewer maps than workgroups also occurs."""
    for nmaps, seed in [(3, 1), (256, 2), (257, 3), (512, 4), (600, 5), (1100 if n < 200 else 800, 6)]:
        x = synth(1, nmaps, n, n, 170 + n + seed)
        got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_PIPE)
        check(x, got)
        assert torch.equal(got, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_PIPE))  # bit-reproducible
        if n not in TILE2D:
            assert torch.equal(got, dpa.energy_nc(x.cuda()))  # AUTO picks it
    dead = synth(1, 300, n, n, 99)
    dead[0, ::7] = 0.0  # dead channels: exactly +0.0
    got = dpa.energy_nc(dead.cuda(), algo=dpa.ALGO_PIPE).cpu()
    assert (got[0, ::7] == 0).all() and not torch.signbit(got[0, ::7]).any()
    check(dead, got)


TILE2D = [224]


@pytest.mark.parametrize("n", TILE2D)
def test_tile2d_sizes(n):
    """2-D radix-8 split kernel (tile2d.hip), AUTO for 224x224. Map counts so that workgroups run 1, 2, 3 and
    4+ maps (the next map's loads are issued one map ahead; the last map reloads itself) and fewer maps than
    workgroups; several tensors in one launch; dead channels; bit-reproducible; against the pipelined kernel."""
    for nmaps, seed in [(1, 0), (3, 1), (256, 2), (257, 3), (512, 4), (600, 5), (800, 6)]:
        x = synth(1, nmaps, n, n, 270 + n + seed)
        got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_TILE2D)
        check(x, got)
        assert torch.equal(got, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_TILE2D))  # bit-reproducible
        assert torch.equal(got, dpa.energy_nc(x.cuda()))  # AUTO picks it
        assert rel_err(got.cpu(), dpa.energy_nc(x.cuda(), algo=dpa.ALGO_PIPE).cpu()) <= 1e-5
    dead = synth(1, 300, n, n, 99)
    dead[0, ::7] = 0.0  # dead channels: exactly +0.0
    got = dpa.energy_nc(dead.cuda()).cpu()
    assert (got[0, ::7] == 0).all() and not torch.signbit(got[0, ::7]).any()
    check(dead, got)
    tensors = [synth(2, c, n, n, 500 + c).cuda() for c in (3, 16, 1, 64)]
    for x, e in zip(tensors, dpa.energy_multi([(x, 0, None) for x in tensors])):
        assert torch.equal(e, dpa.energy_nc(x))  # one launch for all of them == one call each


@pytest.mark.parametrize("n", TILE2G)
def test_tile2g_sizes(n):
    """Mid-size tiles as a 2-D radix split with G maps per round (tile2g.hip; G = 3 at 72 / 144, 4 at 112 / 128,
    2 at 80 / 160). Map counts around the round size (1 ... G + 1 maps: short last groups whose missing maps read
    as zeros and are not written), fewer groups than workgroups, exactly one residency, several rounds per
    workgroup with a short tail; dead channels; bit-reproducible; AUTO routing; against the fused kernel."""
    for nmaps, seed in [(1, 0), (2, 1), (3, 2), (4, 3), (5, 4), (7, 5), (255, 6), (770, 7), (1031, 8), (1800 if n <= 80 else 900, 9)]:
        x = synth(1, nmaps, n, n, 370 + n + seed)
        got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_TILE2D)
        check(x, got)
        assert torch.equal(got, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_TILE2D))  # bit-reproducible
        if n in TILE2G_AUTO:
            assert torch.equal(got, dpa.energy_nc(x.cuda()))  # AUTO picks it
        assert rel_err(got.cpu(), dpa.energy_nc(x.cuda(), algo=dpa.ALGO_FUSED).cpu()) <= 1e-5
    dead = synth(1, 300, n, n, 99)
    dead[0, ::7] = 0.0  # dead channels: exactly +0.0
    got = dpa.energy_nc(dead.cuda(), algo=dpa.ALGO_TILE2D).cpu()
    assert (got[0, ::7] == 0).all() and not torch.signbit(got[0, ::7]).any()
    check(dead, got)
    # the energies of a call land in [N, C] exactly: a guarded output buffer stays untouched behind them
    x = synth(2, 5, n, n, 41 + n)
    buf = torch.full((2 * 5 + 16,), -5.0, device="cuda")
    view = buf[:10].view(2, 5)
    dpa.energy_nc(x.cuda(), algo=dpa.ALGO_TILE2D, out=view)
    check(x, view.clone())
    assert (buf[10:] == -5.0).all()


@pytest.mark.parametrize("n", TILE2G_AUTO)
def test_tile2g_several_tensors_in_one_launch(n):
    """dcts_energy_multi_f32 hands tile2g up to 32 dense tensors as one index space of GROUPS (a round's maps all
    come from one tensor: the last group of every tensor may be short). Tensors of 1, 2, G, G + 1 ... maps, batch
    views, more than 32 tensors: the bits of one call per tensor."""
    counts = [1, 2, 3, 4, 5, 16, 64, 7, 1, 33] + [3] * 30
    tensors = [synth(1 + (i % 2), c, n, n, 800 + 3 * i + n).cuda() for i, c in enumerate(counts)]
    outs = dpa.energy_multi([(x, 0, None) for x in tensors])
    for x, e in zip(tensors, outs):
        assert torch.equal(e, dpa.energy_nc(x))
        check(x.cpu(), e)
    # channel slices of a wider tensor are not dense batches: they fall back to one call per tensor, same results
    wide = synth(2, 12, n, n, 5 + n).cuda()
    (e,) = dpa.energy_multi([(wide, 4, 5)])
    assert torch.equal(e, dpa.energy_nc(wide, c_begin=4, c_count=5))
    check(wide.cpu(), e, c_begin=4, c_count=5)


def test_split_chunking_many_maps():
    """More maps than one intermediate-buffer chunk (96 MiB): 160x160 x 1200 maps."""
    x = synth(4, 300, 160, 160, 77).cuda()
    got = dpa.energy_nc(x, algo=dpa.ALGO_SPLIT).double()
    ref = (x.double() ** 2).sum(dim=(-2, -1))
    assert rel_err(got.cpu(), ref.cpu()) <= RTOL
    assert (got[ref == 0] == 0).all()


@pytest.mark.parametrize("hw", [(9, 18), (7, 14), (32, 16), (5, 3), (1, 1), (1, 7), (288, 3)])
def test_non_square_direct(hw):
    h, w = hw
    x = synth(2, 3, h, w, 31, dead=False)
    check(x, dpa.energy_nc(x.cuda()))


RECT = [(56, 28), (28, 56), (14, 20), (20, 14), (7, 10), (64, 2), (2, 64), (9, 18), (32, 16), (60, 36), (48, 64), (8, 8), (56, 56), (7, 7)]


@pytest.mark.parametrize("hw", RECT)
def test_rect_codelet_pairs(hw):
    """Non-square maps whose two edges are codelet sizes: one kernel, the two 1-D codelets picked at run time
    (rect.hip). Energy against the oracle, AUTO = this kernel for non-square shapes, bit-reproducible over a launch with many
    groups per wave and a ragged tail, coefficients against float64."""
    h, w = hw
    x = synth(3, 19, h, w, 100 + 3 * h + w)
    got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_RECT)
    check(x, got)
    ref = orc.energy_nc(x[:1, :4])  # the reference's per-map loop
    assert rel_err(got[:1, :4].cpu(), ref) <= RTOL
    auto = dpa.energy_nc(x.cuda())
    if h != w:
        assert torch.equal(auto, got)
    else:  # square maps keep their own kernels; the same codelets in the same order
        assert rel_err(auto.cpu(), got.cpu()) <= 1e-5
    big = synth(5, 1201, h, w, 7 + h, dead=True).cuda()
    b = dpa.energy_nc(big, algo=dpa.ALGO_RECT)
    check(big.cpu(), b)
    assert torch.equal(b, dpa.energy_nc(big, algo=dpa.ALGO_RECT))
    xs = synth(2, 3, h, w, 9 + w, dead=False)
    co = dpa.dct2d(xs.cuda(), algo=dpa.ALGO_RECT).cpu().numpy()
    cref = orc.dct_2d_f64(xs.numpy())
    assert co.shape == cref.shape and np.abs(co - cref).max() <= 2e-6 * np.abs(cref).max()


@pytest.mark.parametrize("hw", [(56, 56), (28, 14), (14, 14), (9, 18), (36, 60)])
def test_rows_that_are_not_dense(hw):
    """A spatial crop of a wider tensor: strideH > W, 4-byte-aligned rows. Handed to the library as it is (no copy): equal,
    bit for bit, to the same kernel on a dense copy."""
    h, w = hw
    base = synth(2, 11, h + 5, w + 7, 300 + h).cuda()
    view = base[:, :, 2:2 + h, 3:3 + w]
    assert view.stride(2) == w + 7 and not view.is_contiguous()
    got = dpa.energy_nc(view)
    check(view.cpu().contiguous(), got)
    assert torch.equal(got, dpa.energy_nc(view.contiguous(), algo=dpa.ALGO_RECT))
    sl = dpa.energy_nc(view, c_begin=3, c_count=5)
    assert torch.equal(sl, got[:, 3:8])
    # batch-strided on top (samples 0 and 2 of four: maps are no longer one arithmetic sequence) and an odd front pad
    base4 = synth(4, 11, h + 5, w + 7, 301 + h).cuda()
    v2 = base4[::2, 2:9, 1:1 + h, 2:2 + w]
    check(v2.cpu().contiguous(), dpa.energy_nc(v2))
    if h % 2 == 0:
        v3 = base4[::2, 2:9, 1:h, 2:1 + w]  # (h - 1) x (w - 1), odd H: padded back to h x w
        check(v3.cpu().contiguous(), dpa.energy_nc(v3, pad_front_if_odd=True), pad_front_if_odd=True)


@pytest.mark.parametrize("hw", [(7, 9), (9, 7), (13, 19), (55, 27), (63, 63)])
def test_rect_odd_front_pad(hw):
    """cv2 path on a non-square odd-H map: one zero row and one zero column in front (np.pad(t, (1, 0)), utils/common.py:235-236)."""
    h, w = hw
    x = synth(2, 9, h, w, 500 + h)
    got = dpa.energy_nc(x.cuda(), pad_front_if_odd=True, algo=dpa.ALGO_RECT)
    check(x, got, pad_front_if_odd=True)
    assert torch.equal(got, dpa.energy_nc(x.cuda(), pad_front_if_odd=True)) or h == w
    co = dpa.dct2d(x[:1, :2].cuda(), pad_front_if_odd=True, algo=dpa.ALGO_RECT).cpu().numpy()
    cref = orc.dct_2d_f64(np.pad(x[:1, :2].numpy(), ((0, 0), (0, 0), (1, 0), (1, 0))))
    assert co.shape == cref.shape and np.abs(co - cref).max() <= 2e-6 * np.abs(cref).max()


def test_rect_is_refused_beyond_64():
    x = synth(1, 2, 72, 14, 1).cuda()
    with pytest.raises(Exception):
        dpa.energy_nc(x, algo=dpa.ALGO_RECT)
    check(x.cpu(), dpa.energy_nc(x))  # AUTO: the cosine-matrix kernel


@pytest.mark.parametrize("n", [n for n in range(1, 65) if n not in CODELET])
def test_every_edge_up_to_64_has_a_codelet_path(n):
    """Edges without a square kernel of their own (1, 3, 5, 11, 13, 15, 22, 26 ...: an --input_size such as 160, 176, 208, 240):
    AUTO = the run-time codelet pair, whose template factorises any length (odd parts by the direct sum)."""
    x = synth(2, 21, n, n, 700 + n)
    got = dpa.energy_nc(x.cuda())
    check(x, got)
    assert torch.equal(got, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_RECT))
    assert rel_err(dpa.energy_nc(x.cuda(), algo=dpa.ALGO_DIRECT).cpu(), got.cpu()) <= 1e-5
    xs = synth(1, 3, n, n, 5 + n, dead=False)
    co = dpa.dct2d(xs.cuda()).cpu().numpy()
    cref = orc.dct_2d_f64(xs.numpy())
    assert np.abs(co - cref).max() <= 2e-6 * np.abs(cref).max()


@pytest.mark.parametrize("hw", [(5, 3), (13, 22), (63, 1), (1, 7), (33, 64), (3, 64), (61, 59), (15, 8), (11, 44)])
def test_rect_odd_and_prime_edges(hw):
    h, w = hw
    x = synth(3, 50, h, w, 900 + h + w)
    got = dpa.energy_nc(x.cuda())
    check(x, got)
    assert torch.equal(got, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_RECT))


@pytest.mark.parametrize("n", [7, 9, 13, 15, 17, 19, 27, 31, 35, 39, 55, 63, 71])
def test_odd_front_pad(n):
    """cv2 path (torch2dct): odd H -> one zero row and column in front."""
    x = synth(2, 9, n, n, 40 + n)
    got = dpa.energy_nc(x.cuda(), pad_front_if_odd=True)
    check(x, got, pad_front_if_odd=True)
    ref = orc.energy_nc(x[:1, :3], pad_front_if_odd=True)
    assert rel_err(got[:1, :3].cpu(), ref) <= RTOL


@pytest.mark.parametrize("n", [71, 79, 143, 159])
def test_odd_front_pad_mid_size_tiles(n):
    """cv2 path on odd maps whose padded edge has a tile2g kernel (71 -> 72, 79 -> 80, 143 -> 144, 159 -> 160): the kernel
    gathers the (n x n) map with row pitch n and reads the zero row / column as out-of-range offsets. Map counts around the
    round size (short last groups), dead channels, a 4-byte-aligned base, several tensors in one launch, bit-reproducible;
    the direct kernel (which pads in LDS) as a second witness."""
    for nmaps, seed in [(1, 0), (2, 1), (3, 2), (4, 3), (7, 4), (260, 5), (771, 6)]:
        x = synth(1, nmaps, n, n, 470 + n + seed)
        got = dpa.energy_nc(x.cuda(), pad_front_if_odd=True)
        check(x, got, pad_front_if_odd=True)
        assert torch.equal(got, dpa.energy_nc(x.cuda(), pad_front_if_odd=True))
        assert torch.equal(got, dpa.energy_nc(x.cuda(), pad_front_if_odd=True, algo=dpa.ALGO_TILE2D))  # that is what AUTO ran
        assert rel_err(got.cpu(), dpa.energy_nc(x.cuda(), pad_front_if_odd=True, algo=dpa.ALGO_DIRECT).cpu()) <= 1e-5
    x = synth(2, 6, n, n, 31 + n)
    flat = torch.zeros(x.numel() + 1)
    flat[1:] = x.reshape(-1)
    view = flat.cuda()[1:].view(2, 6, n, n)  # base address only 4-byte aligned
    check(x, dpa.energy_nc(view, pad_front_if_odd=True), pad_front_if_odd=True)
    tensors = [synth(1 + (i % 2), c, n, n, 900 + i + n).cuda() for i, c in enumerate([1, 2, 3, 5, 16, 4])]
    outs = dpa.energy_multi([(t, 0, None) for t in tensors], pad_front_if_odd=True)
    for t, e in zip(tensors, outs):
        assert torch.equal(e, dpa.energy_nc(t, pad_front_if_odd=True))
        check(t.cpu(), e, pad_front_if_odd=True)
    # the last map of an allocation: nothing may be read behind it (the missing maps of a short group are out of range
    # by their own offsets, not by the descriptor's length) - a tensor that ends exactly at the end of its buffer
    buf = torch.empty(5 * n * n, device="cuda")
    t = synth(1, 5, n, n, 77)
    buf.copy_(t.reshape(-1))
    check(t, dpa.energy_nc(buf.view(1, 5, n, n), pad_front_if_odd=True), pad_front_if_odd=True)


@pytest.mark.parametrize("n", [8, 10, 36])
def test_pad_flag_is_noop_for_even(n):
    x = synth(2, 4, n, n, 50)
    a = dpa.energy_nc(x.cuda(), pad_front_if_odd=True)
    b = dpa.energy_nc(x.cuda(), pad_front_if_odd=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("n,algo", [(8, dpa.ALGO_CODELET), (16, dpa.ALGO_CODELET), (32, dpa.ALGO_DIRECT), (9, dpa.ALGO_AUTO)])
def test_channel_slice_densenet_style(n, algo):
    """get_feature_hook_densenet scores channels [C-12, C) of a wider tensor."""
    x = synth(3, 36, n, n, 60 + n)
    got = dpa.energy_nc(x.cuda(), c_begin=24, c_count=12, pad_front_if_odd=True, algo=algo)
    check(x, got, c_begin=24, c_count=12, pad_front_if_odd=True)


def test_non_contiguous_batch_and_channel_views():
    base = synth(4, 16, 14, 14, 70).cuda()
    view = base[::2, 3:11]  # strided in N, offset in C: still dense rows
    got = dpa.energy_nc(view)
    check(view.cpu(), got)
    tr = base.transpose(2, 3)  # rows not dense -> wrapper makes it contiguous
    check(tr.cpu().contiguous(), dpa.energy_nc(tr))


def test_coefficients_both_families():
    for n, algo in [(8, dpa.ALGO_CODELET), (56, dpa.ALGO_CODELET), (14, dpa.ALGO_DIRECT), (24, dpa.ALGO_DIRECT)]:
        x = synth(2, 3, n, n, 80 + n, dead=False)
        got = dpa.dct2d(x.cuda(), algo=algo).cpu().numpy()
        ref = orc.dct_2d_f64(x.numpy())
        assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    x = synth(1, 2, 9, 9, 5, dead=False)
    got = dpa.dct2d(x.cuda(), pad_front_if_odd=True).cpu().numpy()
    ref = orc.dct_2d_f64(np.pad(x.numpy(), ((0, 0), (0, 0), (1, 0), (1, 0))))
    assert got.shape == (1, 2, 10, 10) and np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_known_answers_on_gpu():
    n = 16
    x = torch.full((1, 1, n, n), 1.5)
    assert abs(dpa.energy_nc(x.cuda()).item() - 1.5 * 1.5 * n * n) <= 1e-6 * 1.5 * 1.5 * n * n
    z = torch.zeros(2, 3, 56, 56)
    e = dpa.energy_nc(z.cuda()).cpu()
    assert (e == 0).all() and not torch.signbit(e).any()


def test_parseval_large_batch_full_size():
    """Size-independent property at the bench's full sizes: energy == sum(x^2)."""
    for n, c, h in [(256, 64, 56), (64, 256, 14), (32, 512, 7), (8, 16, 224)]:
        x = synth(n, c, h, h, 90 + h).cuda()
        got = dpa.energy_nc(x).double()
        ref = (x.double() ** 2).sum(dim=(-2, -1))
        assert rel_err(got.cpu(), ref.cpu()) <= RTOL
        assert (got[ref == 0] == 0).all()


def test_linearity_scaling():
    x = synth(2, 8, 28, 28, 99).cuda()
    a = dpa.energy_nc(x)
    b = dpa.energy_nc(x * 2.0)
    assert torch.allclose(b, 4.0 * a, rtol=1e-6, atol=0)


def test_masks_identical_to_oracle():
    """utils/load_models.py:40-41: argsort(imp)[O-K:] then sort, for README compress rates."""
    for c, h, rate in [(64, 32, 0.5), (128, 16, 0.5), (512, 4, 0.95), (256, 56, 0.3), (512, 14, 0.7)]:
        x = synth(8, c, h, h, 7 * c + h)
        got = dpa.energy_nc(x.cuda()).cpu().sum(0) / 8
        ref = orc.energy_nc_batched(x).sum(0) / 8
        k = orc.kept_filters(c, rate)
        np.testing.assert_array_equal(orc.select_index(got.numpy(), c, k), orc.select_index(ref.numpy(), c, k))


def test_batch_sum_matches_host_sum():
    x = synth(16, 40, 8, 8, 123).cuda()
    e = dpa.energy_nc(x)
    got = dpa.batch_sum(e).cpu()
    ref = e.cpu().sum(0)
    assert torch.allclose(got, ref, rtol=1e-6)


def test_errors_are_loud():
    with pytest.raises(RuntimeError):
        dpa.energy_nc(torch.zeros(1, 1, 8, 8))  # CPU tensor: no fallback
    with pytest.raises(TypeError):
        dpa.energy_nc(torch.zeros(1, 1, 8, 8, dtype=torch.float64).cuda())
    from dct_pruning_amd._lib import DctScoreError
    with pytest.raises(DctScoreError):
        dpa.energy_nc(torch.zeros(1, 4, 8, 8).cuda(), c_begin=2, c_count=5)
    with pytest.raises(DctScoreError):
        dpa.energy_nc(torch.zeros(1, 1, 57, 57).cuda(), algo=dpa.ALGO_CODELET)


def test_device_accumulator_matches_reference_rule():
    """dcts_running_mean_update_f32 vs the reference's host-side update (utils/common.py:271-277)."""
    from dct_pruning_amd.accumulate import DeviceAccumulator, HostAccumulator
    host, dev = HostAccumulator(), DeviceAccumulator(37, "cuda:0")
    st = orc.HookState()
    for i, n in enumerate([5, 256, 3]):
        x = synth(n, 37, 8, 8, 300 + i)
        e = dpa.energy_nc(x.cuda())
        host.update(e)
        dev.update(e)
        orc.get_feature_hook(st, x)
    assert host.total.item() == dev.total == 264
    np.testing.assert_allclose(dev.scores(), host.scores(), rtol=2e-6)
    np.testing.assert_allclose(host.scores(), st.feature_result.numpy(), rtol=RTOL)
    dead = np.arange(37) % 8 == 5
    assert (dev.scores()[dead] == 0).all() and (host.scores()[dead] == 0).all()


def test_direct_kernel_workgroups_reuse_their_scratch_tile():
    """More maps than resident workgroups (grid cap 512): each workgroup loops and overwrites its
    intermediate tile in the workspace, so stale cache lines would show up here."""
    for n, c, h in [(3, 700, 24), (2, 600, 12), (1, 1100, 33)]:
        x = synth(n, c, h, h, 500 + h)
        got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_DIRECT)
        check(x, got)
        again = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_DIRECT)
        assert torch.equal(got, again)


@pytest.mark.parametrize("shape", [(1, 1, 2, 2), (1, 1, 7, 7), (1, 1, 56, 56), (1, 1, 224, 224), (2, 5000, 2, 2),
                                   (1, 10, 7, 7), (3, 1, 9, 9), (1, 65, 14, 14), (1, 1, 512, 512), (1, 2, 320, 320),
                                   (1, 3, 72, 72), (5, 7, 287, 287)])
def test_edge_shapes(shape):
    """Single maps, ragged last groups (maps not a multiple of the per-wave group), many tiny maps,
    the largest supported edge, odd edges next to split sizes."""
    x = synth(*shape, 900 + shape[2], dead=shape[1] > 7)
    check(x, dpa.energy_nc(x.cuda()))
    if shape[2] % 2 == 1:
        check(x, dpa.energy_nc(x.cuda(), pad_front_if_odd=True), pad_front_if_odd=True)


def test_empty_and_oversize_are_errors():
    from dct_pruning_amd._lib import DctScoreError
    with pytest.raises(DctScoreError):
        dpa.energy_nc(torch.zeros(0, 4, 8, 8).cuda())
    with pytest.raises(DctScoreError):
        dpa.energy_nc(torch.zeros(1, 1, 513, 513).cuda())
    with pytest.raises(DctScoreError):
        dpa.energy_nc(torch.zeros(1, 1, 224, 224).cuda(), algo=dpa.ALGO_CODELET)


@pytest.mark.parametrize("n", [7, 9])
def test_lane_per_map_kernel(n):
    """One lane per map (7x7, 9x9): dense tensors stream through LDS with direct-to-LDS loads, ragged
    tails (map counts that are not multiples of 64 or whose float count is not a multiple of 4),
    channel slices (per-lane loads), a misaligned base, dead channels."""
    for nmaps, seed in [(1, 1), (63, 2), (64, 3), (65, 4), (129, 5), (4099, 6), (70001, 7)]:
        x = synth(1, nmaps, n, n, 900 + n + seed)
        got = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_LANE)
        check(x, got)
        assert torch.equal(got, dpa.energy_nc(x.cuda()))  # AUTO picks it
        old = dpa.energy_nc(x.cuda(), algo=dpa.ALGO_CODELET)
        assert rel_err(got.cpu(), old.cpu()) <= 1e-5
    x = synth(3, 40, n, n, 950 + n)
    check(x, dpa.energy_nc(x.cuda(), c_begin=5, c_count=17, algo=dpa.ALGO_LANE), c_begin=5, c_count=17)
    flat = torch.zeros(3 * 40 * n * n + 1)
    flat[1:] = x.reshape(-1)
    view = flat.cuda()[1:].view(3, 40, n, n)  # base address only 4-byte aligned
    check(x, dpa.energy_nc(view, algo=dpa.ALGO_LANE))
    from dct_pruning_amd._lib import DctScoreError
    with pytest.raises(DctScoreError):
        dpa.energy_nc(torch.zeros(1, 4, 8, 8).cuda(), algo=dpa.ALGO_LANE)


@pytest.mark.parametrize("n,pad", [(8, False), (7, False), (9, False), (9, True), (14, False), (32, False), (56, False), (24, False),
                                   (72, False), (128, False), (224, False), (288, False),
                                   (13, False), (21, True), (88, False)])  # run-time codelet pair, its odd pad, a two-launch-only edge
def test_energy_multi_matches_single_calls(n, pad):
    """dcts_energy_multi_f32: many tensors of one tile shape in one launch == one call per tensor
    (bitwise), including channel slices, batch-strided views and more than 32 items (chunking)."""
    # large tiles: dense tensors are batched into one fused / pipelined launch (up to 32 per launch)
    nten = 37 if n < 56 else (35 if n == 72 else 5)
    base = [synth(2 + i % 3, 5 + 7 * (i % 4), n, n, 400 + i) for i in range(nten)]
    items = []
    for i, x in enumerate(base):
        xc = x.cuda()
        if i % 3 == 0:
            items.append((xc, 0, None))
        elif i % 3 == 1:
            items.append((xc, 2, 3))
        else:
            items.append((xc[::2], 1, None))
    outs = dpa.energy_multi(items, pad_front_if_odd=pad)
    assert len(outs) == len(items)
    for (x, cb, cc), got in zip(items, outs):
        ref = dpa.energy_nc(x, c_begin=cb, c_count=cc, pad_front_if_odd=pad)
        assert torch.equal(got, ref)
        check(x.cpu(), got, c_begin=cb, c_count=cc, pad_front_if_odd=pad)


def test_energy_multi_with_non_square_maps():
    """Several non-square tensors through dcts_energy_multi_f32: one call per tensor inside (the run-time codelet pair), same bytes."""
    base = [synth(2, 6 + i, 28, 14, 880 + i).cuda() for i in range(5)]
    outs = dpa.energy_multi([(x, 0, None) for x in base])
    for x, got in zip(base, outs):
        assert torch.equal(got, dpa.energy_nc(x))
        check(x.cpu(), got)


def test_mixed_shape_launch_is_bitwise_equal_to_per_tensor_calls():
    """dcts_energy_mixed_f32: tensors of different small tile shapes in one launch (the CIFAR nets'
    hook points), channel slices and odd-pad items included; results must be those of one
    dcts_energy_f32 call per tensor, bit for bit."""
    items = []
    for i, (n, c, h) in enumerate([(32, 64, 32), (32, 64, 16), (32, 128, 16), (32, 128, 8), (32, 256, 8), (32, 256, 4),
                                   (32, 512, 4), (32, 512, 2), (5, 3, 32), (1, 1, 2), (7, 36, 16), (3, 20, 9), (2, 8, 56),
                                   (2, 6, 72), (4, 16, 7)]):
        x = synth(n, c, h, h, 700 + i).cuda()
        if i == 10:
            items.append((x, c - 12, 12, True))      # DenseNet-style channel slice, cv2 flag on an even tile
        elif i == 11:
            items.append((x, 0, None, True))         # odd tile with the front pad: not a mixed-launch shape
        else:
            items.append((x, 0, None, False))
    outs = dpa.energy_mixed(items)
    for (x, cb, cc, pad), got in zip(items, outs):
        want = dpa.energy_nc(x, cb, cc, pad)
        assert got.shape == want.shape
        assert torch.equal(got, want)
    # more tensors than one launch takes (48)
    many = [(synth(4, 8 + (i % 5), 8, 8, 900 + i).cuda(), 0, None, False) for i in range(61)]
    for (x, _, _, _), got in zip(many, dpa.energy_mixed(many)):
        assert torch.equal(got, dpa.energy_nc(x))


def test_direct_kernel_basis_tables_are_cached_safely():
    """The direct kernel builds its basis tables once per (workspace, stream, shape); shape changes and other
    users of the same workspace (the two-launch split path) must not leave stale tables behind."""
    def check(shape, algo=dpa.ALGO_DIRECT):
        x = synth(2, 3, shape[0], shape[1], 31 + shape[0])
        got = dpa.energy_nc(x.cuda(), algo=algo).cpu()
        ref = torch.from_numpy(orc.energy_nc_f64(x)) if max(shape) > 64 else orc.energy_nc(x)
        assert rel_err(got, ref) <= RTOL, shape

    check((24, 24))
    check((24, 24))          # cached tables
    check((30, 20))          # other shape, same workspace
    check((24, 24))          # back: rebuilt
    check((72, 72), dpa.ALGO_SPLIT)   # the split path's intermediate overwrites the head of the workspace
    check((24, 24))
    x = synth(1, 2, 24, 24, 5, dead=False)
    got = dpa.dct2d(x.cuda(), algo=dpa.ALGO_DIRECT).cpu().numpy()
    assert np.abs(got - orc.dct_2d_f64(x.numpy())).max() <= 2e-6 * np.abs(got).max()
    check((24, 24))


def test_weighted_calls_of_several_shapes_share_one_workspace_safely():
    """the weighted path hands the coefficient path an INTERIOR pointer of the workspace whose offset
    depends on the tile shape; tables cached under such pointers were never dropped. Weighted 22x22, weighted
    13x13 (its tables land inside the 22x22 ones), weighted 22x22 again, then a large direct energy call - all on
    the ONE workspace ops.py keeps per (device, stream) - must each give the float64 definition."""
    from dct_pruning_amd import ops

    def weighted(n, seed):
        x = synth(2, 5, n, n, seed, dead=False)
        g = torch.Generator().manual_seed(seed + 1)
        w = torch.rand(n, n, generator=g)
        got = dpa.weighted_energy_nc(x.cuda(), w.cuda()).cpu()
        ref = torch.from_numpy(orc.weighted_energy_nc_f64(x, w.numpy()))
        assert rel_err(got, ref) <= 2e-5, n

    # size the shared workspace once, generously, so that every call below reuses the same buffer
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    base = ops._workspace(dev, stream, 64 << 20).data_ptr()
    assert not dpa.has_codelet(22, 22) and not dpa.has_codelet(13, 13) and not dpa.has_codelet(26, 26)  # direct kernel
    weighted(22, 1)
    weighted(13, 2)
    weighted(22, 3)
    x = synth(2, 64, 22, 22, 4)
    check(x, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_DIRECT))
    weighted(26, 5)
    check(x, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_DIRECT))  # its tables were overwritten by the weighted call
    x2 = synth(1, 600, 30, 20, 6)  # many maps: the T tiles reach far into the workspace
    got = dpa.energy_nc(x2.cuda(), algo=dpa.ALGO_DIRECT).cpu()
    assert rel_err(got, orc.energy_nc(x2)) <= RTOL
    weighted(13, 7)
    assert ops._workspace(dev, stream, 1).data_ptr() == base  # one buffer throughout


@pytest.mark.parametrize("n", [72, 256, 288])
def test_large_tile_with_a_4_byte_aligned_base(n):
    """The two-launch path stages with 16-byte direct-to-LDS loads and refuses a view whose base is only 4-byte aligned. The
    kernels that load single dwords take it: tile2g.hip (72) and, since round 3, the fused and the two-roles kernels (256, 288:
    samples straight into registers) - under AUTO the same kernel, aligned or not, and the same bits."""
    from dct_pruning_amd._lib import DctScoreError
    c = 3
    x = synth(1, c, n, n, 700 + n)
    flat = torch.zeros(c * n * n + 1)
    flat[1:] = x.reshape(-1)
    view = flat.cuda()[1:].view(1, c, n, n)
    assert view.data_ptr() % 16 == 4
    got = dpa.energy_nc(view).cpu()
    ref = torch.from_numpy(orc.energy_nc_f64(x)).float()
    assert rel_err(got, ref) <= RTOL
    with pytest.raises(DctScoreError) as ei:
        dpa.energy_nc(view, algo=dpa.ALGO_SPLIT)
    assert ei.value.code == -6  # DCTS_E_UNSUPPORTED
    fused = dpa.energy_nc(view, algo=dpa.ALGO_FUSED).cpu()
    assert torch.equal(fused, dpa.energy_nc(x.cuda(), algo=dpa.ALGO_FUSED).cpu())
    if n != 72:
        assert torch.equal(fused, got)  # AUTO is the fused family there
    # the aligned tensor itself goes through the same large-tile kernel: the same bits
    assert torch.equal(dpa.energy_nc(x.cuda()).cpu(), got)


@pytest.mark.parametrize("n", [96, 112, 128, 144, 192, 224, 320])
def test_4_byte_aligned_base_every_large_shape(n):
    """The other large shapes on a base that is only 4-byte aligned: a factorised kernel that loads dwords (the fused family, tile2g)
    instead of the cosine-matrix kernel; against the oracle on a few maps and Parseval on all."""
    c = 37
    x = synth(1, c, n, n, 900 + n)
    flat = torch.zeros(c * n * n + 1)
    flat[1:] = x.reshape(-1)
    view = flat.cuda()[1:].view(1, c, n, n)
    assert view.data_ptr() % 16 == 4
    got = dpa.energy_nc(view).cpu()
    par = (x.double() ** 2).sum(dim=(-2, -1))
    nz = par > 0
    assert ((got.double()[nz] - par[nz]).abs() / par[nz]).max().item() <= 1e-5 and (got[~nz] == 0).all()
    ref = torch.from_numpy(orc.energy_nc_f64(x[:, :6])).float()
    assert rel_err(got[:, :6], ref) <= RTOL
    assert rel_err(dpa.energy_nc(view, algo=dpa.ALGO_DIRECT).cpu(), got) <= 1e-5


def test_published_jpeg_worked_example_on_gpu():
    """The published 8x8 worked example of the JPEG literature (tests/golden/jpeg_example_8x8.json: data typed from the
    publication) through the product's coefficient path, codelet and direct kernels: a known answer that comes from
    neither SciPy, the oracle nor the reference."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_example_8x8.json")))
    x = (torch.tensor(g["block"], dtype=torch.float32) + g["level_shift"])[None, None]
    want = np.array(g["published_dct_rows_0_to_2"])
    for algo in (dpa.ALGO_AUTO, dpa.ALGO_CODELET, dpa.ALGO_DIRECT):
        got = dpa.dct2d(x.cuda(), algo=algo).cpu().numpy()[0, 0]
        assert np.abs(got[:3] - want).max() <= g["published_precision"] + 1e-3, algo
        e = dpa.energy_nc(x.cuda(), algo=algo).item()
        assert abs(e - float((x.double() ** 2).sum())) <= 1e-5 * e
