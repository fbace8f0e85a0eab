"""Hook schedules of imp_score (utils/common.py:367-977) as data: for every net the ordered
hook points, the file stem(s) each one is saved under, the hook flavour and — for the
reference's default inputs and compress_rate = 0 — the hooked tensor's (C, H, W).

hook kinds:
  "full"      get_feature_hook              (utils/common.py:262-277)  all channels, torch_dct
  "last12"    get_feature_hook_densenet     (:280-293)  channels [C-12, C), cv2 path (odd pad)
  "input"     get_feature_hook_u2net_input  (:296-309)  input[0] of the module, cv2 path
"""
from collections import namedtuple

HookPoint = namedtuple("HookPoint", "module kind files C H W")
# module: attribute path below `net` (e.g. "layer1.0.relu1", "features.2")
# files:  list of (file_stem, c_lo, c_hi) — channel slice saved under that name (None = all)


def _hp(module, kind, stems, C, H, W=None):
    if isinstance(stems, str):
        stems = [stems]
    files = [(s, None, None) if isinstance(s, str) else s for s in stems]
    return HookPoint(module, kind, files, C, H, H if W is None else W)


def vgg_16_bn():
    """utils/common.py:384-397; net.relucfg = models/cifar10/vgg.py:6. Indices 6, 13, 23, 33
    are MaxPool2d modules (convs 2/4/7/10 are scored after pooling); conv 13 is never scored."""
    relucfg = [2, 6, 9, 13, 16, 19, 23, 26, 29, 33, 36, 39]
    shapes = [(64, 32), (64, 16), (128, 16), (128, 8), (256, 8), (256, 8), (256, 4), (512, 4), (512, 4),
              (512, 2), (512, 2), (512, 2)]
    return [_hp("features.%d" % idx, "full", "imp_conv%d" % (i + 1), c, h)
            for i, (idx, (c, h)) in enumerate(zip(relucfg, shapes))]


def _resnet_cifar(blocks_per_stage):
    """utils/common.py:400-437 (resnet_56) and :520-554 (resnet_110)."""
    pts = [_hp("relu", "full", "imp_conv1", 16, 32)]
    cnt = 1
    for i, (c, h) in enumerate([(16, 32), (32, 16), (64, 8)]):
        for j in range(blocks_per_stage):
            for r in ("relu1", "relu2"):
                cnt += 1
                pts.append(_hp("layer%d.%d.%s" % (i + 1, j, r), "full", "imp_conv%d" % cnt, c, h))
    return pts


def resnet_56():
    return _resnet_cifar(9)


def resnet_110():
    return _resnet_cifar(18)


def densenet_40():
    """utils/common.py:440-476: growth 12, 12 layers per block; first layer of a block uses the
    full hook, the others / transitions / final relu the last-12-channel cv2 hook."""
    pts = []
    c_in = 24
    for i, h in enumerate([32, 16, 8]):
        for j in range(12):
            c = c_in + 12 * j
            pts.append(_hp("dense%d.%d.relu" % (i + 1, j), "full" if j == 0 else "last12",
                           "imp_conv%d" % (13 * i + j + 1), c, h))
        c_in = c_in + 12 * 12
        if i < 2:
            pts.append(_hp("trans%d.relu" % (i + 1), "last12", "imp_conv%d" % (13 * (i + 1)), c_in, h))
    pts.append(_hp("relu", "last12", "imp_conv39", c_in, 8))
    return pts


GOOGLENET_FILTERS = [  # net.filters (models/cifar10/googlenet.py:132-146): n1x1, n3x3, n5x5, pool_planes
    [64, 128, 32, 32], [128, 192, 96, 64], [192, 208, 48, 64], [160, 224, 64, 64], [128, 256, 64, 64],
    [112, 288, 64, 64], [256, 320, 128, 128], [256, 320, 128, 128], [384, 384, 128, 128]]


def googlenet(filters_p=None):
    """utils/common.py:479-517: 10 hooked tensors, 37 files. idx 0 -> 'imp_conv1_' (trailing
    underscore as the reference writes it); idx >= 1 -> four branch slices by cumulative sums
    of net.filters_p[idx-1]."""
    fp = filters_p or GOOGLENET_FILTERS
    mods = ["pre_layers", "inception_a3", "maxpool1", "inception_a4", "inception_b4", "inception_c4",
            "inception_d4", "maxpool2", "inception_a5", "inception_b5"]
    hs = [32, 32, 16, 16, 16, 16, 16, 8, 8, 8]
    pts = [_hp(mods[0], "full", "imp_conv1_", 192, 32)]
    for idx in range(1, 10):
        f = fp[idx - 1]
        files, lo = [], 0
        for tp, n in zip(["n1x1", "n3x3", "n5x5", "pool_planes"], f):
            files.append(("imp_conv%d_%s" % (idx + 1, tp), lo, lo + n))
            lo += n
        pts.append(_hp(mods[idx], "full", files, sum(f), hs[idx]))
    return pts


def resnet_50(num_blocks=(3, 4, 6, 3)):
    """utils/common.py:557-607: 49 hooked tensors, 53 files; relu3 of block 0 of every stage is
    written twice (shortcut conv, then conv3)."""
    pts = [_hp("maxpool", "full", "imp_conv1", 64, 56)]
    cnt = 1
    planes = [64, 128, 256, 512]
    h_in = 56
    for i in range(4):
        h_out = h_in if i == 0 else h_in // 2
        for j in range(num_blocks[i]):
            p = planes[i]
            cnt += 1
            pts.append(_hp("layer%d.%d.relu1" % (i + 1, j), "full", "imp_conv%d" % cnt, p, h_in if j == 0 else h_out))
            cnt += 1
            pts.append(_hp("layer%d.%d.relu2" % (i + 1, j), "full", "imp_conv%d" % cnt, p, h_out))
            stems = []
            if j == 0:
                cnt += 1
                stems.append("imp_conv%d" % cnt)
            cnt += 1
            stems.append("imp_conv%d" % cnt)
            pts.append(_hp("layer%d.%d.relu3" % (i + 1, j), "full", stems, 4 * p, h_out))
        h_in = h_out
    return pts


def u2netp(size=288):
    """utils/common.py:610-977: 118 hooked tensors; file stem = 'net.' + attribute path.
    Spatial sizes for a `size` x `size` input (reference crop: 288, utils/common.py:154-155)."""
    def dims(stage):  # input resolution of encoder stage k (1..6)
        s = size
        for _ in range(min(stage, 6) - 1):
            s = (s + 1) // 2
        return s

    def rsu_sizes(stage, depth):
        """resolution of rebnconv{k} (k=1..depth) and rebnconv{k}d (k=1..depth-1) inside an RSU"""
        base = dims(stage)
        enc = []
        s = base
        for k in range(1, depth + 1):
            enc.append(s)
            if k < depth - 1:
                s = (s + 1) // 2
        return base, enc

    pts = []
    depth_of = {1: 7, 2: 6, 3: 5, 4: 4, 5: 4, 6: 4}

    def name(stage, dec, unit):
        return "stage%d%s.%s.relu_s1" % (stage, "d" if dec else "", unit)

    def add(stage, dec, unit, c, h):
        m = name(stage, dec, unit)
        pts.append(_hp(m, "full", "net." + m, c, h))

    def res(stage, k, depth, is_dec_unit):
        """spatial size of rebnconv{k} / rebnconv{k}d of an RSU with `depth` levels at `stage`"""
        base = dims(stage)
        if stage >= 5:  # RSU4F: dilated, no pooling
            return base
        lvl = min(k, depth - 1)  # rebnconv{depth} runs at the resolution of level depth-1
        s = base
        for _ in range(lvl - 1):
            s = (s + 1) // 2
        return s

    # 1) rebnconvin of every encoder / decoder stage (utils/common.py:610-635)
    for i in range(6):
        add(i + 1, False, "rebnconvin", 64, dims(i + 1))
        if i < 5:
            add(i + 1, True, "rebnconvin", 64, dims(i + 1))
    # 2) the interleaved per-unit order of utils/common.py:637-905
    for i in range(7):
        k = i + 1
        add(1, False, "rebnconv%d" % k, 16, res(1, k, 7, False))
        add(1, True, "rebnconv%d" % k, 16, res(1, k, 7, False))
        if i < 6:
            add(1, False, "rebnconv%dd" % k, 64 if k == 1 else 16, res(1, k, 7, True))
            add(1, True, "rebnconv%dd" % k, 64 if k == 1 else 16, res(1, k, 7, True))
            add(2, False, "rebnconv%d" % k, 16, res(2, k, 6, False))
            add(2, True, "rebnconv%d" % k, 16, res(2, k, 6, False))
        if i < 5:
            add(2, False, "rebnconv%dd" % k, 64 if k == 1 else 16, res(2, k, 6, True))
            add(2, True, "rebnconv%dd" % k, 64 if k == 1 else 16, res(2, k, 6, True))
            add(3, False, "rebnconv%d" % k, 16, res(3, k, 5, False))
            add(3, True, "rebnconv%d" % k, 16, res(3, k, 5, False))
        if i < 4:
            add(3, False, "rebnconv%dd" % k, 64 if k == 1 else 16, res(3, k, 5, True))
            add(3, True, "rebnconv%dd" % k, 64 if k == 1 else 16, res(3, k, 5, True))
            add(4, False, "rebnconv%d" % k, 16, res(4, k, 4, False))
            add(4, True, "rebnconv%d" % k, 16, res(4, k, 4, False))
            add(5, False, "rebnconv%d" % k, 16, res(5, k, 4, False))
            add(5, True, "rebnconv%d" % k, 16, res(5, k, 4, False))
            add(6, False, "rebnconv%d" % k, 16, res(6, k, 4, False))
        if i < 3:
            add(4, False, "rebnconv%dd" % k, 64 if k == 1 else 16, res(4, k, 4, True))
            add(4, True, "rebnconv%dd" % k, 64 if k == 1 else 16, res(4, k, 4, True))
            add(5, False, "rebnconv%dd" % k, 64 if k == 1 else 16, res(5, k, 4, True))
            add(5, True, "rebnconv%dd" % k, 64 if k == 1 else 16, res(5, k, 4, True))
            add(6, False, "rebnconv%dd" % k, 64 if k == 1 else 16, res(6, k, 4, True))
    # 3) side1..6: input of the side conv, 64 channels, cv2 hook (utils/common.py:907-975)
    for k in range(1, 7):
        pts.append(_hp("side%d" % k, "input", "net.side%d" % k, 64, dims(k)))
    return pts


SCHEDULES = {
    "vgg_16_bn": vgg_16_bn, "resnet_56": resnet_56, "resnet_110": resnet_110, "densenet_40": densenet_40,
    "googlenet": googlenet, "resnet_50": resnet_50, "u2netp": u2netp,
}


def scored_shape(pt):
    """(c_begin, c_count, pad_front_if_odd) of the operator call a hook point makes."""
    if pt.kind == "last12":
        return pt.C - 12, 12, True
    if pt.kind == "input":
        return 0, pt.C, True
    return 0, pt.C, False
