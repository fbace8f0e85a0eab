"""Full-width (compress_rate = 0) definitions of the seven architectures imp_score knows,
written only to produce activations of the right shapes at the right hook points.

They keep the reference's public surface for this path:
  * the attribute paths imp_score hooks (utils/common.py:384-977), e.g. net.features[idx],
    net.layer1[j].relu1, net.dense1[j].relu, net.inception_a3, net.stage1.rebnconvin.relu_s1;
  * the helper attributes it reads: net.relucfg (vgg), net.num_blocks (resnet_50),
    net.filters_p (googlenet);
  * state_dict keys and shapes of the reference's models/ (models/cifar10/vgg.py:27-46,
    resnet.py:52-139, densenet.py:12-101, googlenet.py:8-183, models/imagenet/resnet.py:40-130,
    models/DUTS/u2net.py:6-486) so the checkpoints importance_generation.py loads
    (importance_generation.py:25-53) fit these modules too.
Pruned widths (compress_rate != 0) are the consumer side (prune_*.py) and out of scope here;
imp_score itself is model-agnostic and accepts the reference's own model objects as well.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _relu():
    return nn.ReLU(inplace=True)


# ----------------------------------------------------------------------------------------
# VGG-16-bn (CIFAR)
# ----------------------------------------------------------------------------------------
class VGG16BN(nn.Module):
    CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)

    def __init__(self, num_classes=10):
        super().__init__()
        self.relucfg = [2, 6, 9, 13, 16, 19, 23, 26, 29, 33, 36, 39]
        mods, c_in = OrderedDict(), 3
        for i, v in enumerate(self.CFG):
            if v == "M":
                mods["pool%d" % i] = nn.MaxPool2d(2, 2)
                continue
            mods["conv%d" % i] = nn.Conv2d(c_in, v, 3, padding=1)
            mods["norm%d" % i] = nn.BatchNorm2d(v)
            mods["relu%d" % i] = _relu()
            c_in = v
        self.features = nn.Sequential(mods)
        self.classifier = nn.Sequential(OrderedDict(
            linear1=nn.Linear(512, 512), norm1=nn.BatchNorm1d(512), relu1=_relu(), linear2=nn.Linear(512, num_classes)))

    def forward(self, x):
        x = F.avg_pool2d(self.features(x), 2)
        return self.classifier(x.flatten(1))


# ----------------------------------------------------------------------------------------
# ResNet-56 / -110 (CIFAR): option-A shortcuts (strided slice + zero channel pad)
# ----------------------------------------------------------------------------------------
class _PadShortcut(nn.Module):
    def __init__(self, extra, stride):
        super().__init__()
        self.lo, self.hi, self.stride = extra // 2, extra - extra // 2, stride

    def forward(self, x):
        if self.stride != 1:
            x = x[:, :, ::self.stride, ::self.stride]
        return F.pad(x, (0, 0, 0, 0, self.lo, self.hi))


class _BasicBlock(nn.Module):
    def __init__(self, c_in, c_out, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, c_out, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c_out)
        self.relu1 = _relu()
        self.conv2 = nn.Conv2d(c_out, c_out, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c_out)
        self.relu2 = _relu()
        self.shortcut = nn.Sequential() if (stride == 1 and c_in == c_out) else _PadShortcut(c_out - c_in, stride)

    def forward(self, x):
        y = self.bn2(self.conv2(self.relu1(self.bn1(self.conv1(x)))))
        y = y + self.shortcut(x)
        return self.relu2(y)


class ResNetCifar(nn.Module):
    def __init__(self, depth, num_classes=10):
        super().__init__()
        n = (depth - 2) // 6
        self.conv1 = nn.Conv2d(3, 16, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        self.relu = _relu()
        c_in = 16
        for s, (c, stride) in enumerate([(16, 1), (32, 2), (64, 2)]):
            blocks = []
            for j in range(n):
                blocks.append(_BasicBlock(c_in, c, stride if j == 0 else 1))
                c_in = c
            setattr(self, "layer%d" % (s + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        # the reference names the classifier `fc` for depth 56 and `linear` for 110
        setattr(self, "fc" if depth == 56 else "linear", nn.Linear(64, num_classes))
        self._head = "fc" if depth == 56 else "linear"

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        return getattr(self, self._head)(self.avgpool(x).flatten(1))


# ----------------------------------------------------------------------------------------
# ResNet-50 (ImageNet)
# ----------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    def __init__(self, c_in, mid, c_out, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.relu1 = _relu()
        self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.relu2 = _relu()
        self.conv3 = nn.Conv2d(mid, c_out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(c_out)
        self.relu3 = _relu()
        self.is_downsample = project
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride, bias=False), nn.BatchNorm2d(c_out))

    def forward(self, x):
        y = self.relu1(self.bn1(self.conv1(x)))
        y = self.relu2(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y = y + (self.downsample(x) if self.is_downsample else x)
        return self.relu3(y)


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.num_blocks = [3, 4, 6, 3]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = _relu()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        c_in = 64
        for s, (mid, n) in enumerate(zip([64, 128, 256, 512], self.num_blocks)):
            stage = nn.ModuleList()
            for j in range(n):
                stage.append(_Bottleneck(c_in, mid, 4 * mid, (1 if s == 0 else 2) if j == 0 else 1, j == 0))
                c_in = 4 * mid
            setattr(self, "layer%d" % (s + 1), stage)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                x = blk(x)
        return self.fc(self.avgpool(x).flatten(1))


# ----------------------------------------------------------------------------------------
# DenseNet-40 (growth 12, no bottleneck, no compression)
# ----------------------------------------------------------------------------------------
class _DenseLayer(nn.Module):
    def __init__(self, c_in, growth):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(c_in)
        self.relu = _relu()
        self.conv1 = nn.Conv2d(c_in, growth, 3, padding=1, bias=False)

    def forward(self, x):
        return torch.cat((x, self.conv1(self.relu(self.bn1(x)))), 1)


class _Transition(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(c_in)
        self.relu = _relu()
        self.conv1 = nn.Conv2d(c_in, c_out, 1, bias=False)

    def forward(self, x):
        return F.avg_pool2d(self.conv1(self.relu(self.bn1(x))), 2)


class DenseNet40(nn.Module):
    def __init__(self, num_classes=10, growth=12, n=12):
        super().__init__()
        c = 2 * growth
        self.conv1 = nn.Conv2d(3, c, 3, padding=1, bias=False)
        for s in range(3):
            layers = []
            for _ in range(n):
                layers.append(_DenseLayer(c, growth))
                c += growth
            setattr(self, "dense%d" % (s + 1), nn.Sequential(*layers))
            if s < 2:
                setattr(self, "trans%d" % (s + 1), _Transition(c, c))
        self.bn = nn.BatchNorm2d(c)
        self.relu = _relu()
        self.avgpool = nn.AvgPool2d(8)
        self.fc = nn.Linear(c, num_classes)

    def forward(self, x):
        x = self.trans1(self.dense1(self.conv1(x)))
        x = self.trans2(self.dense2(x))
        x = self.relu(self.bn(self.dense3(x)))
        return self.fc(self.avgpool(x).flatten(1))


# ----------------------------------------------------------------------------------------
# GoogLeNet (CIFAR variant: 5x5 branch = two 3x3 convs)
# ----------------------------------------------------------------------------------------
def _cbr(*pairs):
    mods = []
    for c_in, c_out, k in pairs:
        mods += [nn.Conv2d(c_in, c_out, k, padding=k // 2), nn.BatchNorm2d(c_out), nn.ReLU(True)]
    return mods


class _Inception(nn.Module):
    def __init__(self, c_in, n1, r3, n3, r5, n5, npool):
        super().__init__()
        self.branch1x1 = nn.Sequential(*_cbr((c_in, n1, 1)))
        self.branch3x3 = nn.Sequential(*_cbr((c_in, r3, 1), (r3, n3, 3)))
        self.branch5x5 = nn.Sequential(*_cbr((c_in, r5, 1), (r5, n5, 3), (n5, n5, 3)))
        self.branch_pool = nn.Sequential(nn.MaxPool2d(3, 1, 1), *_cbr((c_in, npool, 1)))

    def forward(self, x):
        return torch.cat([self.branch1x1(x), self.branch3x3(x), self.branch5x5(x), self.branch_pool(x)], 1)


class GoogLeNet(nn.Module):
    FILTERS = ([64, 128, 32, 32], [128, 192, 96, 64], [192, 208, 48, 64], [160, 224, 64, 64], [128, 256, 64, 64],
               [112, 288, 64, 64], [256, 320, 128, 128], [256, 320, 128, 128], [384, 384, 128, 128])
    REDUCE = ([96, 16], [128, 32], [96, 16], [112, 24], [128, 24], [144, 32], [160, 32], [160, 32], [192, 48])
    NAMES = ("a3", "b3", "a4", "b4", "c4", "d4", "e4", "a5", "b5")

    def __init__(self, num_classes=10):
        super().__init__()
        self.filters = [list(f) for f in self.FILTERS]
        self.filters_p = [list(f) for f in self.FILTERS]  # compress_rate = 0: same widths
        self.pre_layers = nn.Sequential(*_cbr((3, 192, 3)))
        c_in = 192
        for name, f, r in zip(self.NAMES, self.FILTERS, self.REDUCE):
            setattr(self, "inception_" + name, _Inception(c_in, f[0], r[0], f[1], r[1], f[2], f[3]))
            c_in = sum(f)
        self.maxpool1 = nn.MaxPool2d(3, 2, 1)
        self.maxpool2 = nn.MaxPool2d(3, 2, 1)
        self.avgpool = nn.AvgPool2d(8, 1)
        self.linear = nn.Linear(c_in, num_classes)

    def forward(self, x):
        x = self.inception_b3(self.inception_a3(self.pre_layers(x)))
        x = self.maxpool1(x)
        for n in ("a4", "b4", "c4", "d4", "e4"):
            x = getattr(self, "inception_" + n)(x)
        x = self.maxpool2(x)
        x = self.inception_b5(self.inception_a5(x))
        return self.linear(self.avgpool(x).flatten(1))


# ----------------------------------------------------------------------------------------
# U^2-Net-p (DUTS): residual U-blocks of depth 7/6/5/4 and the dilated 4F variant
# ----------------------------------------------------------------------------------------
class _ReBnConv(nn.Module):
    def __init__(self, c_in, c_out, dirate=1):
        super().__init__()
        self.conv_s1 = nn.Conv2d(c_in, c_out, 3, padding=dirate, dilation=dirate)
        self.bn_s1 = nn.BatchNorm2d(c_out)
        self.relu_s1 = _relu()

    def forward(self, x):
        return self.relu_s1(self.bn_s1(self.conv_s1(x)))


def _up_like(src, ref):
    return F.interpolate(src, size=ref.shape[2:], mode="bilinear", align_corners=False)


class _RSU(nn.Module):
    """Residual U-block with `depth` encoder units. dilated=True is the RSU-4F form: no
    pooling, dilations 1/2/4/8 instead."""

    def __init__(self, depth, c_in, mid, c_out, dilated=False):
        super().__init__()
        self.depth, self.dilated = depth, dilated
        self.rebnconvin = _ReBnConv(c_in, c_out)
        for k in range(1, depth + 1):
            if dilated:
                d = 2 ** (k - 1)
            else:
                d = 2 if k == depth else 1
            setattr(self, "rebnconv%d" % k, _ReBnConv(c_out if k == 1 else mid, mid, d))
            if not dilated and k < depth - 1:
                setattr(self, "pool%d" % k, nn.MaxPool2d(2, 2, ceil_mode=True))
        for k in range(depth - 1, 0, -1):
            d = 2 ** (k - 1) if dilated else 1
            setattr(self, "rebnconv%dd" % k, _ReBnConv(2 * mid, c_out if k == 1 else mid, d))

    def forward(self, x):
        xin = self.rebnconvin(x)
        enc, h = [], xin
        for k in range(1, self.depth + 1):
            h = getattr(self, "rebnconv%d" % k)(h)
            enc.append(h)
            if not self.dilated and k < self.depth - 1:
                h = getattr(self, "pool%d" % k)(h)
        d = enc[-1]
        for k in range(self.depth - 1, 0, -1):
            d = getattr(self, "rebnconv%dd" % k)(torch.cat((d, enc[k - 1]), 1))
            if k > 1 and not self.dilated:
                d = _up_like(d, enc[k - 2])
        return d + xin


class U2NETP(nn.Module):
    def __init__(self, in_ch=3, out_ch=1):
        super().__init__()
        m = 64
        self.stage1 = _RSU(7, in_ch, 16, m)
        self.pool12 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.stage2 = _RSU(6, m, 16, m)
        self.pool23 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.stage3 = _RSU(5, m, 16, m)
        self.pool34 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.stage4 = _RSU(4, m, 16, m)
        self.pool45 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.stage5 = _RSU(4, m, 16, m, dilated=True)
        self.pool56 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.stage6 = _RSU(4, m, 16, m, dilated=True)
        self.stage5d = _RSU(4, 2 * m, 16, m, dilated=True)
        self.stage4d = _RSU(4, 2 * m, 16, m)
        self.stage3d = _RSU(5, 2 * m, 16, m)
        self.stage2d = _RSU(6, 2 * m, 16, m)
        self.stage1d = _RSU(7, 2 * m, 16, m)
        for k in range(1, 7):
            setattr(self, "side%d" % k, nn.Conv2d(m, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6, out_ch, 1)

    def forward(self, x):
        h1 = self.stage1(x)
        h2 = self.stage2(self.pool12(h1))
        h3 = self.stage3(self.pool23(h2))
        h4 = self.stage4(self.pool34(h3))
        h5 = self.stage5(self.pool45(h4))
        h6 = self.stage6(self.pool56(h5))
        d5 = self.stage5d(torch.cat((_up_like(h6, h5), h5), 1))
        d4 = self.stage4d(torch.cat((_up_like(d5, h4), h4), 1))
        d3 = self.stage3d(torch.cat((_up_like(d4, h3), h3), 1))
        d2 = self.stage2d(torch.cat((_up_like(d3, h2), h2), 1))
        d1 = self.stage1d(torch.cat((_up_like(d2, h1), h1), 1))
        s1 = self.side1(d1)
        sides = [s1] + [_up_like(getattr(self, "side%d" % k)(t), s1)
                        for k, t in zip(range(2, 7), (d2, d3, d4, d5, h6))]
        s0 = self.outconv(torch.cat(sides, 1))
        return tuple(torch.sigmoid(t) for t in [s0] + sides)


def get_network(name):
    """Counterpart of utils/common.py:31-54 for compress_rate = 0 (no .cuda(): the caller places it)."""
    table = {
        "vgg_16_bn": VGG16BN, "resnet_56": lambda: ResNetCifar(56), "resnet_110": lambda: ResNetCifar(110),
        "densenet_40": DenseNet40, "googlenet": GoogLeNet, "resnet_50": ResNet50, "u2netp": U2NETP,
    }
    if name not in table:
        raise ValueError("the network name you have entered is not supported yet: %r" % (name,))
    return table[name]()
