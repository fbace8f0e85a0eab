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


R56_RATES = [0.0] + [0.18] * 29                                        # README.md:114
R110_RATES = [0.0] + [0.2] * 2 + [0.3] * 18 + [0.4] * 18 + [0.39] * 19  # README.md:138
D40_RATES = [0.0] + [0.2] * 12 + [0.0] + [0.2] * 12 + [0.0] + [0.2] * 12  # README.md:162


def mini_resnet_cifar(g, num_layers, overall, mid):
    """state dict with the keys of models/cifar10/resnet.py (full CIFAR widths are already small)."""
    sd = {}

    def conv_bn(conv, bn, cout, cin):
        sd[conv + ".weight"] = _rand(g, cout, cin, 3, 3)
        for part in (".weight", ".bias", ".running_mean", ".running_var"):
            sd[bn + part] = _rand(g, cout)

    conv_bn("conv1", "bn1", overall[0], 3)
    blk, cin = 0, overall[0]
    for layer, num in enumerate(tp.RESNET_CIFAR_REPEAT[num_layers]):
        for k in range(num):
            base = "layer%d.%d." % (layer + 1, k)
            conv_bn(base + "conv1", base + "bn1", mid[blk], cin)
            conv_bn(base + "conv2", base + "bn2", overall[blk + 1], mid[blk])
            cin = overall[blk + 1]
            blk += 1
    lin = "fc" if num_layers == 56 else "linear"  # models/cifar10/resnet.py:126-129
    sd[lin + ".weight"] = _rand(g, 10, 64)
    sd[lin + ".bias"] = _rand(g, 10)
    modules = [(k[:-len(".weight")], "conv") for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4]
    return sd, modules + [(lin, "linear")]


def mini_densenet40(g, widths):
    sd, names = {}, tp.densenet_40_conv_names()
    cin = 3
    inplanes = 0
    for cov_id, (name, w) in enumerate(zip(names, widths), start=1):
        k = 1 if name.startswith("trans") else 3
        if cov_id > 1:
            bn = name.replace("conv1", "bn1")
            for part in (".weight", ".bias", ".running_mean", ".running_var"):
                sd[bn + part] = _rand(g, cin)
        sd[name + ".weight"] = _rand(g, w, cin, k, k)
        if cov_id == 1 or name.startswith("trans"):
            inplanes = w
        else:
            inplanes += w
        cin = inplanes
    for part in (".weight", ".bias", ".running_mean", ".running_var"):
        sd["bn" + part] = _rand(g, cin)
    sd["fc.weight"] = _rand(g, 10, cin)
    sd["fc.bias"] = _rand(g, 10)
    return sd


def test_cifar_width_tables_follow_the_reference_constructors():
    overall, mid = tp.resnet_cifar_widths(R56_RATES, 56)
    assert overall == [16] + [13] * 9 + [26] * 9 + [64] * 9 and mid == [13] * 9 + [26] * 9 + [52] * 9
    kept = tp.resnet_cifar_kept(R56_RATES, 56)
    assert len(kept) == 54 and kept[0] == ("imp_conv2", 16, 13) and kept[-1] == ("imp_conv55", 64, 64)
    overall, mid = tp.resnet_cifar_widths(R110_RATES, 110)
    assert overall == [16] + [12] * 18 + [25] * 18 + [64] * 18
    assert mid == [11] * 18 + [19] * 18 + [39] * 18
    w = tp.densenet_40_widths(D40_RATES)
    assert len(w) == 39 and w[0] == 24 and w[1:13] == [9] * 12 and w[13] == 24 + 9 * 12
    assert w[26] == 24 + 2 * 9 * 12 and w[27:] == [9] * 12
    assert tp.densenet_40_widths([0.0] * 39) == [24] + [12] * 12 + [168] + [12] * 12 + [312] + [12] * 12
    # a pruned transition: floor(inplanes * (1 - r))
    assert tp.densenet_40_widths([0.0] * 13 + [0.5] + [0.0] * 25)[13] == 84


@pytest.mark.parametrize("num_layers,rates", [(56, R56_RATES), (110, R110_RATES), (56, [0.0] * 30),
                                              (56, [0.0, 0.5, 0.0] + [0.0, 0.3] * 14)])
@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_resnet_cifar_transplant_equals_reference_loops(device, num_layers, rates):
    g = torch.Generator().manual_seed(13)
    ori, modules = mini_resnet_cifar(g, num_layers, *tp.resnet_cifar_widths([0.0] * 60, num_layers))
    slim, _ = mini_resnet_cifar(g, num_layers, *tp.resnet_cifar_widths(rates, num_layers))
    imp = scores_for(g, ori, [(stem, conv) for conv, stem, _, _ in tp.resnet_cifar_convs(num_layers)])
    want = orc_t.load_resnet_model(copy.deepcopy(slim), copy.deepcopy(ori), num_layers, imp, modules)
    got = tp.transplant_resnet_cifar({k: v.clone().to(device) for k, v in slim.items()},
                                     {k: v.clone().to(device) for k, v in ori.items()}, imp, num_layers)
    assert_same(got, want)


@pytest.mark.parametrize("rates", [D40_RATES, [0.0] * 39, [0.0] * 13 + [0.5] + [0.3] * 12 + [0.25] + [0.6] * 12])
@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_densenet40_transplant_equals_reference_loops(device, rates):
    g = torch.Generator().manual_seed(14)
    ori = mini_densenet40(g, tp.densenet_40_widths([0.0] * 39))
    slim = mini_densenet40(g, tp.densenet_40_widths(rates))
    names = tp.densenet_40_conv_names()
    imp = scores_for(g, ori, [("imp_conv%d" % (k + 1), n) for k, n in enumerate(names)])
    want = orc_t.load_densenet_model(copy.deepcopy(slim), copy.deepcopy(ori), imp, names)
    # conv1 is never loaded by the reference (its previous-index list is empty): the slim model's own weights stay
    assert torch.equal(want["conv1.weight"], slim["conv1.weight"])
    got = tp.transplant_densenet_40({k: v.clone().to(device) for k, v in slim.items()},
                                    {k: v.clone().to(device) for k, v in ori.items()}, imp)
    assert_same(got, want)


G_RATES = [0.4] + [0.85] * 2 + [0.9] * 5 + [0.9] * 2   # README.md:186
MINI_FILTERS = [[f // 8 for f in row] for row in tp.GOOGLENET_FILTERS]
MINI_MID = [[12, 2], [16, 4], [12, 2], [14, 3], [16, 3], [18, 4], [20, 4], [20, 4], [24, 6]]  # models/cifar10/googlenet.py:155-165, / 8


def mini_googlenet(g, rates):
    """state dict + module list with the keys of models/cifar10/googlenet.py at 1/8 of the widths."""
    sd, modules = {}, [("pre_layers", "pre_layers")]

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = _rand(g, cout, cin, k, k)
        sd[name + ".bias"] = _rand(g, cout)
        modules.append((name, "conv"))

    def bn(name, c):
        for part in (".weight", ".bias", ".running_mean", ".running_var"):
            sd[name + part] = _rand(g, c)
        sd[name + ".num_batches_tracked"] = torch.tensor(int(torch.randint(0, 100, (1,), generator=g)))
        modules.append((name, "bn"))

    conv("pre_layers.0", 24, 3, 3)
    bn("pre_layers.1", 24)
    cin = 24
    for name, (n1, n3, n5mid, n5, pool), (red3, red5) in zip(tp.GOOGLENET_BLOCKS, tp.googlenet_widths(rates, MINI_FILTERS), MINI_MID):
        modules.append((name, "inception"))
        conv(name + ".branch1x1.0", n1, cin, 1), bn(name + ".branch1x1.1", n1)
        conv(name + ".branch3x3.0", red3, cin, 1), bn(name + ".branch3x3.1", red3)
        conv(name + ".branch3x3.3", n3, red3, 3), bn(name + ".branch3x3.4", n3)
        conv(name + ".branch5x5.0", red5, cin, 1), bn(name + ".branch5x5.1", red5)
        conv(name + ".branch5x5.3", n5mid, red5, 3), bn(name + ".branch5x5.4", n5mid)
        conv(name + ".branch5x5.6", n5, n5mid, 3), bn(name + ".branch5x5.7", n5)
        conv(name + ".branch_pool.1", pool, cin, 1), bn(name + ".branch_pool.2", pool)
        cin = n1 + n3 + n5 + pool
    sd["linear.weight"] = _rand(g, 10, cin)
    sd["linear.bias"] = _rand(g, 10)
    modules.append(("linear", "linear"))
    return sd, modules


def test_googlenet_width_table_follows_the_reference_constructor():
    w = tp.googlenet_widths(G_RATES)
    assert w[0] == (64, 19, 4, 4, 32)          # int(128*0.15) (float: 19.2 -> 19), int(32*0.15) = 4
    assert w[-1] == (384, 384, 12, 128, 128)   # last block: full branch outputs, pruned middle 5x5-branch conv
    kept = tp.googlenet_kept(G_RATES)
    assert kept[0] == ("imp_conv2_n3x3", 128, 19) and kept[1] == ("imp_conv2_n5x5", 32, 4) and len(kept) == 18
    assert tp.googlenet_widths([0.0] * 10)[3] == (160, 224, 64, 64, 64)


@pytest.mark.parametrize("rates", [G_RATES, [0.0] * 10, [0.0, 0.5, 0.0, 0.25, 0.0, 0.75, 0.5, 0.0, 0.5, 0.5]])
@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_googlenet_transplant_equals_reference_loops(device, rates):
    g = torch.Generator().manual_seed(15)
    ori, modules = mini_googlenet(g, [0.0] * 10)
    slim, _ = mini_googlenet(g, rates)
    imp = {}
    for b, name in enumerate(tp.GOOGLENET_BLOCKS):
        imp.update(scores_for(g, ori, [("imp_conv%d_n3x3" % (b + 2), name + ".branch3x3.3"),
                                       ("imp_conv%d_n5x5" % (b + 2), name + ".branch5x5.3")]))
    want = orc_t.load_google_model(copy.deepcopy(slim), copy.deepcopy(ori), imp, modules, MINI_FILTERS)
    # quirks of the reference kept: no conv bias and not the stem are transplanted
    assert torch.equal(want["pre_layers.0.weight"], slim["pre_layers.0.weight"])
    assert torch.equal(want["inception_a3.branch1x1.0.bias"], slim["inception_a3.branch1x1.0.bias"])
    got = tp.transplant_googlenet({k: v.clone().to(device) for k, v in slim.items()},
                                  {k: v.clone().to(device) for k, v in ori.items()}, imp, MINI_FILTERS)
    assert_same(got, want)


# ---- U^2-Net-p (utils/load_models.py:585-769) -------------------------------------------------------
U2_RATES = [0.40] * 40                                                  # prune_u2netp.py:99 (the default)
U2_MIXED = ([0.4, 0.0, 0.5, 0.0, 0.25] + [0.5, 0.0, 0.0, 0.4] + [0.0, 0.3, 0.3] + [0.5, 0.0] + [0.0, 0.0] + [0.25, 0.5]
            + [0.0] * 5 + [0.3, 0.3, 0.0, 0.6] + [0.5] * 3 + [0.0, 0.4] + [0.9, 0.1] + [0.3, 0.5, 0.2, 0.4, 0.6])


def mini_u2netp(g, rates, inner=8, outer=16):
    """conv weights of U2NETP(compress_rate) at reduced base widths + a bias and batch-norm tensors per
    unit, which the loader must leave alone (utils/load_models.py:609-610 touch '.weight' only)."""
    sd, names = {}, []
    for name, cout, cin in tp.u2netp_conv_shapes(rates, inner=inner, outer=outer):
        sd[name + ".weight"] = _rand(g, cout, cin, 3, 3)
        sd[name + ".bias"] = _rand(g, cout)
        if name.endswith("conv_s1"):
            bn = name[:-len("conv_s1")] + "bn_s1"
            for part in (".weight", ".bias", ".running_mean", ".running_var"):
                sd[bn + part] = _rand(g, cout)
        names.append(name)
    sd["outconv.weight"] = _rand(g, 1, 6, 1, 1)
    return sd, names + ["outconv"]


def u2_scores(g, ori, names):
    stems = []
    for n in names:
        if n == "outconv":
            continue
        parts = n.split(".")
        stems.append(("net." + n if len(parts) == 1 else "net.%s.%s.relu_s1" % (parts[0], parts[1]), n))
    return scores_for(g, ori, stems)


def test_u2netp_width_table_follows_the_reference_constructor():
    full = tp.u2netp_conv_shapes([0.0] * 39)
    assert len(full) == 118 and full[0] == ("stage1.rebnconvin.conv_s1", 64, 3)
    assert full[7] == ("stage1.rebnconv7.conv_s1", 16, 16) and full[8] == ("stage1.rebnconv6d.conv_s1", 16, 32)
    assert full[13] == ("stage1.rebnconv1d.conv_s1", 64, 32) and full[-1] == ("side6", 1, 64)
    slim = dict((n, (o, i)) for n, o, i in tp.u2netp_conv_shapes(U2_RATES))
    # models/DUTS/u2net.py:37-42: int(0.6 * 16) = 9 inside, the two deepest convs keep 16; :438-442: int(0.6 * 64) = 38
    assert slim["stage1.rebnconv1.conv_s1"] == (9, 38) and slim["stage1.rebnconv6.conv_s1"] == (16, 9)
    assert slim["stage1.rebnconv7.conv_s1"] == (16, 16) and slim["stage1.rebnconv6d.conv_s1"] == (9, 32)
    assert slim["stage5d.rebnconvin.conv_s1"] == (38, 76) and slim["stage6.rebnconv1d.conv_s1"] == (38, 18)
    assert slim["side3"] == (1, 38)
    # a rate of 1.0 leaves one channel (:43-52)
    assert dict((n, o) for n, o, _ in tp.u2netp_conv_shapes([1.0] * 39))["stage2.rebnconv1.conv_s1"] == 1
    # the module names are those of the network this repo hooks
    from dct_pruning_amd import nets
    convs = [n for n, m in nets.U2NETP().named_modules() if isinstance(m, torch.nn.Conv2d)]
    assert convs == [n for n, _, _ in full] + ["outconv"]


@pytest.mark.parametrize("rates", [U2_RATES, U2_MIXED])
@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_u2netp_transplant_equals_reference_loops(device, rates):
    g = torch.Generator().manual_seed(17)
    ori, names = mini_u2netp(g, [0.0] * 39)
    slim, _ = mini_u2netp(g, rates)
    imp = u2_scores(g, ori, names)
    want = orc_t.load_u2netp_model(copy.deepcopy(slim), copy.deepcopy(ori), imp, names)
    got = tp.transplant_u2netp({k: v.clone().to(device) for k, v in slim.items()},
                               {k: v.clone().to(device) for k, v in ori.items()}, imp, names)
    assert_same(got, want)
    # something was pruned and transplanted, and nothing but conv weights moved
    assert any(got[k].shape != ori[k].shape for k in got)
    assert all(torch.equal(got[k].cpu(), slim[k]) for k in got if not k.endswith("conv_s1.weight") and "side" not in k)


@pytest.mark.parametrize("rates,exc", [([0.0] * 39, TypeError),                                   # list(None), :621
                                       ([0.4] * 34 + [0.4, 0.4, 0.4, 0.0, 0.4], ValueError)])      # int('i'), :658
def test_u2netp_transplant_fails_where_the_reference_fails(rates, exc):
    g = torch.Generator().manual_seed(18)
    ori, names = mini_u2netp(g, [0.0] * 39)
    slim, _ = mini_u2netp(g, rates)
    imp = u2_scores(g, ori, names)
    with pytest.raises(exc):
        orc_t.load_u2netp_model(copy.deepcopy(slim), copy.deepcopy(ori), imp, names)
    with pytest.raises(exc):
        tp.transplant_u2netp(copy.deepcopy(slim), copy.deepcopy(ori), imp, names)
