"""CPU oracle for the DCT importance-score path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module. The product path (dct_pruning_amd/) never does and has no CPU fallback.

It restates, function by function, what the reference does on the path of SURVEY.md §8(a):

    utils/common.py:230-239  torch2dct            -> torch2dct()
    utils/common.py:249-255  cnt_score            -> cnt_score()
    utils/common.py:258-259  feature_result/total -> HookState
    utils/common.py:262-277  get_feature_hook     -> get_feature_hook()
    utils/common.py:280-293  ..._densenet         -> get_feature_hook_densenet()
    utils/common.py:296-309  ..._u2net_input      -> get_feature_hook_u2net_input()
    utils/load_models.py:40-41 (and 8 siblings)   -> select_index()

The transform itself lives in two third-party packages that are NOT in /root/reference and
NOT installed in this image: `torch_dct` (PyPI torch-dct, version unpinned by the reference;
call site utils/common.py:267) and `cv2.dct` (OpenCV, unpinned; call site :237). Their
published algorithm is restated here:
    1-D, along the last axis:  v = [x0, x2, x4, ..., odd-indexed samples reversed];
    V = FFT_N(v);  X_k = Re(V_k) cos(pi k / 2N) + Im(V_k) sin(pi k / 2N);
    ortho: X_0 /= sqrt(N), X_k *= sqrt(2/N);   2-D = along W, transpose, along H, transpose.

PARITY PINNING. The reference ships no test, golden vector or fixture for this path
 and the two DCT packages are absent, so parity with their exact
round-off is UNPINNED. What pins this oracle instead:
  * the worked 8x8 example of the JPEG literature (block and forward DCT as published, two decimals;
    tests/golden/jpeg_example_8x8.json): the transform's definition and normalisation, from a source that is
    neither SciPy, this repo nor the reference (tests/test_oracle.py::test_published_jpeg_worked_example);
  * scipy.fft.dctn(type=2, norm='ortho') in float64 — the same mathematical transform
    (tests/test_oracle.py), agreement <= 2e-7 of the coefficient scale;
  * Parseval: sum(coeff^2) == sum(x^2) for the orthonormal transform;
  * analytic vectors (zero map, constant map, single basis function);
  * the harness semantics (hook order, slicing, running mean, file names) captured from
    the reference's own imp_score run in the build container (tests/golden/).
"""
import math

import numpy as np
import torch

try:  # scipy is the cv2.dct stand-in and the float64 cross-check
    from scipy.fft import dctn as _scipy_dctn
except Exception:  # pragma: no cover
    _scipy_dctn = None


# ----------------------------------------------------------------------------------------
# the transform (third-party torch_dct.dct / dct_2d restated)
# ----------------------------------------------------------------------------------------
def dct_1d(x, norm="ortho"):
    """DCT-II along the last axis by the even/odd reorder + length-N FFT route (fp32 in, fp32 out)."""
    shape = x.shape
    n = shape[-1]
    rows = x.contiguous().view(-1, n)
    v = torch.cat([rows[:, ::2], rows[:, 1::2].flip([1])], dim=1)
    vc = torch.fft.fft(v, dim=1)
    k = -torch.arange(n, dtype=x.dtype, device=x.device)[None, :] * math.pi / (2 * n)
    out = vc.real * torch.cos(k) - vc.imag * torch.sin(k)
    if norm == "ortho":
        out[:, 0] /= math.sqrt(n) * 2
        out[:, 1:] /= math.sqrt(n / 2) * 2
    return (2 * out).view(*shape)


def dct_2d(x, norm="ortho"):
    """2-D DCT-II of the last two axes: along W, transpose, along H, transpose back."""
    x1 = dct_1d(x, norm=norm)
    x2 = dct_1d(x1.transpose(-1, -2), norm=norm)
    return x2.transpose(-1, -2)


def dct_2d_f64(x):
    """Independent float64 reference (SciPy): same transform, different algorithm."""
    a = np.asarray(x, dtype=np.float64)
    return _scipy_dctn(a, type=2, norm="ortho", axes=(-2, -1))


def torch2dct(feature_map):
    """utils/common.py:230-239. cv2.dct (absent) is replaced by SciPy's fp32 dctn; the odd
    front pad is the reference's own np.pad(t, (1, 0)): one zero in front of EVERY axis when
    shape[0] is odd."""
    t = feature_map.cpu().numpy()
    t = np.float32(t)
    if t.shape[0] % 2 != 0:
        t = np.pad(t, (1, 0), "constant")
    d = _scipy_dctn(t, type=2, norm="ortho").astype(np.float32)
    return torch.from_numpy(d)


# ----------------------------------------------------------------------------------------
# score + hooks (utils/common.py:249-309)
# ----------------------------------------------------------------------------------------
def cnt_score(dct_list):
    """Per map: sum of squared coefficients, via .item() (fp32 -> Python float -> fp32)."""
    for idx, d in enumerate(dct_list):
        dct_list[idx] = torch.sum(d.mul(d)).item()
    return torch.tensor(dct_list)


class HookState:
    """The module globals feature_result / total of utils/common.py:258-259."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.feature_result = torch.tensor(0.)
        self.total = torch.tensor(0.)

    def update(self, c, a):
        """utils/common.py:275-277: weighted running mean over samples (three fp32 roundings)."""
        self.feature_result = self.feature_result * self.total + c
        self.total = self.total + a
        self.feature_result = self.feature_result / self.total


def get_feature_hook(state, output):
    """utils/common.py:262-277."""
    a, b = output.shape[0], output.shape[1]
    c = [dct_2d(output[i, j, :, :], norm="ortho") for i in range(a) for j in range(b)]
    c = cnt_score(c)
    c = c.view(a, -1)
    c = c.sum(0)
    state.update(c, a)


def get_feature_hook_densenet(state, output):
    """utils/common.py:280-293: last 12 channels, cv2 path."""
    a, b = output.shape[0], output.shape[1]
    c = [torch2dct(output[i, j, :, :]) for i in range(a) for j in range(b - 12, b)]
    c = cnt_score(c)
    c = c.view(a, -1).float()
    c = c.sum(0)
    state.update(c, a)


def get_feature_hook_u2net_input(state, inp):
    """utils/common.py:296-309: scores input[0] of the hooked module, cv2 path."""
    x = inp[0]
    a, b = x.shape[0], x.shape[1]
    c = [torch2dct(x[i, j, :, :]) for i in range(a) for j in range(b)]
    c = cnt_score(c)
    c = c.view(a, -1)
    c = c.sum(0)
    state.update(c, a)


# ----------------------------------------------------------------------------------------
# operator-level oracle: what dcts_energy_f32 must return
# ----------------------------------------------------------------------------------------
def energy_nc(x, c_begin=0, c_count=None, pad_front_if_odd=False):
    """[N, c_count] per-map energies with the reference's per-map loop and fp32 arithmetic."""
    x = x.detach().cpu().float()
    n, c_total = x.shape[0], x.shape[1]
    if c_count is None:
        c_count = c_total - c_begin
    if pad_front_if_odd:
        maps = [torch2dct(x[i, j]) for i in range(n) for j in range(c_begin, c_begin + c_count)]
    else:
        maps = [dct_2d(x[i, j], norm="ortho") for i in range(n) for j in range(c_begin, c_begin + c_count)]
    return cnt_score(maps).view(n, -1).float()


def energy_nc_batched(x, c_begin=0, c_count=None, pad_front_if_odd=False):
    """Same maths, all maps in one batched FFT call (fast path for bigger test inputs and the
    'best-effort CPU' leg of the baseline)."""
    x = x.detach().cpu().float()
    if c_count is None:
        c_count = x.shape[1] - c_begin
    xs = x[:, c_begin:c_begin + c_count]
    if pad_front_if_odd and xs.shape[2] % 2 != 0:
        xs = torch.nn.functional.pad(xs, (1, 0, 1, 0))
    d = dct_2d(xs.contiguous(), norm="ortho")
    return (d * d).sum(dim=(-2, -1))


def energy_nc_f64(x, c_begin=0, c_count=None, pad_front_if_odd=False):
    """float64 SciPy energies (numpy [N, c_count])."""
    a = x.detach().cpu().numpy().astype(np.float64)
    if c_count is None:
        c_count = a.shape[1] - c_begin
    a = a[:, c_begin:c_begin + c_count]
    if pad_front_if_odd and a.shape[2] % 2 != 0:
        a = np.pad(a, ((0, 0), (0, 0), (1, 0), (1, 0)))
    d = dct_2d_f64(a)
    return (d * d).sum(axis=(-2, -1))


# ----------------------------------------------------------------------------------------
# consumer rule (the "prune mask"), utils/load_models.py:40-41 and its 8 siblings
# ----------------------------------------------------------------------------------------
def select_index(imp, orifilter_num, currentfilter_num):
    select = np.argsort(imp)[orifilter_num - currentfilter_num:]
    select.sort()
    return select


def kept_filters(orifilter_num, rate):
    """Filter count the model constructors derive from a compress rate, e.g.
    models/cifar10/vgg.py:37: int(out_channels * (1 - rate))."""
    return int(orifilter_num * (1 - rate))


# ----------------------------------------------------------------------------------------
# score variant in the coefficient domain (the reference only hints at variants in
# comments, utils/common.py:268-269, so there is no reference code to restate: this is the definition
# dcts_weighted_energy_f32 is tested against)
# ----------------------------------------------------------------------------------------
def weighted_energy_nc_f64(x, weights, c_begin=0, c_count=None, pad_front_if_odd=False):
    """float64: E[n, j] = sum_{u,v} weights[u,v] * dct_2d(x[n, c_begin+j])[u,v]**2 (numpy [N, c_count])."""
    a = x.detach().cpu().numpy().astype(np.float64)
    if c_count is None:
        c_count = a.shape[1] - c_begin
    a = a[:, c_begin:c_begin + c_count]
    if pad_front_if_odd and a.shape[2] % 2 != 0:
        a = np.pad(a, ((0, 0), (0, 0), (1, 0), (1, 0)))
    d = dct_2d_f64(a)
    return (np.asarray(weights, dtype=np.float64)[None, None] * d * d).sum(axis=(-2, -1))
