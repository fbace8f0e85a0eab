"""dct_pruning_amd.transplant (vectorised index_select) against the element-by-element restatement of
the reference's loops (oracle/transplant_oracle.py; utils/load_models.py:17-64, :441-582): the
resulting state dicts must be EQUAL, bit for bit, on CPU and on the GPU. Miniature networks (same
keys and structure, narrow channels) keep the Python triple loops of the restatement to seconds;
the width tables are checked against the reference constructors' formulas at full size."""
import copy

import numpy as np
import pytest
import torch

from dct_pruning_amd import transplant as tp
from oracle import transplant_oracle as orc_t

VGG_RATES = [0.5] * 7 + [0.95] * 5                       # README.md:90
R50_RATES = [0.0] + [0.1] * 3 + [0.4] * 7 + [0.4] * 9    # README.md:211


def _rand(g, *shape):
    return torch.randn(*shape, generator=g)


def mini_vgg(g, widths, scale):
    """state dict with the keys / layout of the reference's VGG features (conv + norm per layer)."""
    sd, cin = {}, 3
    names = ["features.conv%d" % i for i, x in enumerate(tp.VGG_CFG) if x != "M"]
    for name, w in zip(names, widths):
        w = max(1, w // scale)
        sd[name + ".weight"] = _rand(g, w, cin, 3, 3)
        sd[name + ".bias"] = _rand(g, w)
        norm = name.replace("conv", "norm")
        for part in (".weight", ".bias", ".running_mean", ".running_var"):
            sd[norm + part] = _rand(g, w)
        cin = w
    return sd, names


def mini_resnet50(g, overall, mid, scale):
    sd = {}

    def conv_bn(conv, bn, cout, cin, k):
        sd[conv + ".weight"] = _rand(g, cout, cin, k, k)
        for part in (".weight", ".bias", ".running_mean", ".running_var"):
            sd[bn + part] = _rand(g, cout)
        sd[bn + ".num_batches_tracked"] = torch.tensor(int(torch.randint(0, 100, (1,), generator=g)))

    ov = [max(1, c // scale) for c in overall]
    md = [max(1, c // scale) for c in mid]
    conv_bn("conv1", "bn1", ov[0], 3, 3)
    blk, cin = 0, ov[0]
    for layer, num in enumerate(tp.RESNET50_STAGE_REPEAT):
        for k in range(num):
            base = "layer%d.%d." % (layer + 1, k)
            conv_bn(base + "conv1", base + "bn1", md[blk], cin, 1)
            conv_bn(base + "conv2", base + "bn2", md[blk], md[blk], 3)
            conv_bn(base + "conv3", base + "bn3", ov[blk + 1], md[blk], 1)
            if k == 0:
                conv_bn(base + "downsample.0", base + "downsample.1", ov[blk + 1], cin, 1)
            cin = ov[blk + 1]
            blk += 1
    sd["fc.weight"] = _rand(g, 10, cin)
    sd["fc.bias"] = _rand(g, 10)
    return sd


def scores_for(g, ori, stems_to_conv):
    """seeded scores with exact ties and dead channels."""
    imp = {}
    for stem, conv in stems_to_conv:
        c = ori[conv + ".weight"].size(0)
        s = torch.rand(c, generator=g)
        s[torch.arange(c) % 5 == 3] = 0.0
        if c > 4:
            s[1] = s[2]
        imp[stem] = s.numpy().astype(np.float32)
    return imp


def assert_same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].shape == b[k].shape, k
        assert torch.equal(a[k].cpu(), b[k].cpu()), k


def test_width_tables_follow_the_reference_constructors():
    assert tp.vgg_16_bn_widths(VGG_RATES) == [32, 32, 64, 64, 128, 128, 128, 25, 25, 25, 25, 25, 512]
    overall, mid = tp.resnet_50_widths(R50_RATES)
    assert overall[0] == 64 and overall[1:4] == [230] * 3 and overall[4:8] == [460] * 4 and overall[8:14] == [921] * 6
    assert overall[14:] == [2048] * 3
    assert mid == [38] * 3 + [76] * 4 + [153] * 3 + [153] * 3 + [307] * 3
    kept = tp.resnet_50_kept(R50_RATES)
    assert len(kept) == 53 and kept[0] == ("imp_conv1", 64, 64)
    assert kept[1] == ("imp_conv2", 64, 38) and kept[3] == ("imp_conv4", 256, 230) and kept[4] == ("imp_conv5", 256, 230)
    assert kept[-1] == ("imp_conv53", 2048, 2048)
    assert [k for _, _, k in tp.vgg_16_bn_kept(VGG_RATES)][-1] == 512


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_vgg_transplant_equals_reference_loops(device):
    g = torch.Generator().manual_seed(11)
    ori, names = mini_vgg(g, [x for x in tp.VGG_CFG if x != "M"], 8)
    slim, _ = mini_vgg(g, tp.vgg_16_bn_widths(VGG_RATES), 8)
    imp = scores_for(g, ori, [("imp_conv%d" % (i + 1), n) for i, n in enumerate(names)])
    want = orc_t.load_vgg_model(copy.deepcopy(slim), copy.deepcopy(ori), imp, names)
    got = tp.transplant_vgg({k: v.clone().to(device) for k, v in slim.items()},
                            {k: v.clone().to(device) for k, v in ori.items()}, imp)
    assert_same(got, want)


@pytest.mark.parametrize("rates", [R50_RATES, [0.25] + [0.0] * 3 + [0.4] * 16, [0.0] * 20])
@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_resnet50_transplant_equals_reference_loops(device, rates):
    g = torch.Generator().manual_seed(12)
    ori = mini_resnet50(g, *tp.resnet_50_widths([0.0] * 20), 16)
    slim = mini_resnet50(g, *tp.resnet_50_widths(rates), 16)
    imp = scores_for(g, ori, [("imp_conv%d" % (k + 1), c[0]) for k, c in enumerate(tp.resnet_50_convs())])
    want = orc_t.load_resnet_imagenet_model(copy.deepcopy(slim), copy.deepcopy(ori), imp)
    got = tp.transplant_resnet_50({k: v.clone().to(device) for k, v in slim.items()},
                                  {k: v.clone().to(device) for k, v in ori.items()}, imp)
    assert_same(got, want)
