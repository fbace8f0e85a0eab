"""Weight transplant from the full-width network into the pruned one - the step right after the
score path, vectorised.

The reference walks the convolutions in order, derives the kept filters of each from its score file
    select_index = np.argsort(imp)[orifilter_num - currentfilter_num:]; select_index.sort()
and copies weights one scalar slice at a time in Python double / triple loops
(utils/load_models.py:17-64 for VGG-16-bn, :67-143 for the CIFAR ResNets, :385-438 for DenseNet-40,
:441-582 for ResNet-50: ~10^7 interpreter-level copies for ResNet-50). Here every such loop is one index_select on the output-filter axis, one on the
input-channel axis (the previous layer's kept filters) and one assignment: same result, bit for
bit, on whatever device the tensors live on.

Quirks kept because the result must equal the reference's state dict:
  * VGG: only conv `.weight` tensors are transplanted; conv biases and batch-norm tensors of the
    slim model stay as the caller's state dict has them (utils/load_models.py:43-61 never touches them);
  * VGG: once a layer keeps its full width and the previous layer did too, `last_select_index` is
    reset; a full-width layer after a pruned one gets its input channels sliced and the index
    stays as it was (:55-61);
  * ResNet-50: the downsample conv of a stage's first block does not update `last_select_index`
    (record_last = False, :506), batch-norm tensors follow the conv's kept filters, and every
    `num_batches_tracked` is copied (:560).

  * CIFAR ResNet-56/110: only `layerX.k.conv1/conv2` weights are transplanted (score files
    imp_conv2 ... - the counter starts at 1 and is incremented before use, :80, :89); a full-width conv
    after a pruned one gets its input channels sliced and then RESETS the index (:123); the stem
    conv and the linear layer come over whole, batch-norm tensors are never touched (:129-141);
  * DenseNet-40: `last_select_index` starts as an EMPTY LIST, not None (:388), so the "no previous
    index" branches are unreachable and the first conv copies nothing at all - conv1 keeps the slim
    model's own weights; after every conv the kept filters are appended at the position the
    concatenated feature map gives them in the ORIGINAL network (offset cov_id*12 - (cov_id-1)//13*12,
    :435), restarted after conv1 and after each transition (cov_id 1, 14, 27, :432); neither
    batch-norm tensors nor the classifier are copied;
  * GoogLeNet: inside an Inception block only the 3x3 conv and the two 5x5-branch convs lose filters
    (score files imp_conv<id>_n3x3 / _n5x5, the latter read for both 5x5-branch convs, :264, :313); the
    four entry convs get their input channels sliced by the previous block's kept list, which is
    assembled in the order [1x1, pool, 3x3, 5x5] (:233-242, :275-277, :320-322) - not the order
    torch.cat uses - with offsets from the ORIGINAL filter table; every conv of the net is on the
    "sketch" list, so no conv bias is ever copied and `pre_layers.0` (never pruned by the constructor)
    keeps the slim model's own weights; the batch-norms behind pruned convs and `pre_layers.1` are not
    copied either, the other batch-norms (four tensors, not num_batches_tracked) and the linear layer
    come over whole (:361-380).

Widths of the pruned networks (what `currentfilter_num` is) follow the model constructors:
models/cifar10/vgg.py:37, models/cifar10/resnet.py:5-30 and models/imagenet/resnet.py:8-27 (adapt_channel),
models/cifar10/densenet.py:71-75, :91-104, models/cifar10/googlenet.py:28-66, :147-149.
"""
import math
import os

import numpy as np
import torch

from .masks import select_index

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]  # models/cifar10/vgg.py:5
RESNET50_STAGE_REPEAT = [3, 4, 6, 3]                                                                # models/imagenet/resnet.py:3
RESNET50_STAGE_OUT = [64] + [256] * 3 + [512] * 4 + [1024] * 6 + [2048] * 3                         # :4
RESNET_CIFAR_REPEAT = {56: [9, 9, 9], 110: [18, 18, 18]}                                           # models/cifar10/resnet.py:7-12
DENSENET40_GROWTH, DENSENET40_BLOCK = 12, 12                                                        # models/cifar10/densenet.py:53, :58
GOOGLENET_FILTERS = [[64, 128, 32, 32], [128, 192, 96, 64], [192, 208, 48, 64], [160, 224, 64, 64], [128, 256, 64, 64],
                     [112, 288, 64, 64], [256, 320, 128, 128], [256, 320, 128, 128], [384, 384, 128, 128]]  # utils/load_models.py:149-159
GOOGLENET_BLOCKS = ["inception_a3", "inception_b3", "inception_a4", "inception_b4", "inception_c4", "inception_d4",
                    "inception_e4", "inception_a5", "inception_b5"]                                  # models/cifar10/googlenet.py:171-184
_BN_PARTS = [".weight", ".bias", ".running_mean", ".running_var"]


def vgg_16_bn_widths(compress_rate):
    """Output channels of the 13 convolutions (models/cifar10/vgg.py:18-19, :37: the rate list gets a
    trailing 0.0, so the 13th conv keeps its width)."""
    rates = list(compress_rate) + [0.0]
    widths, cnt = [], 0
    for x in VGG_CFG:
        if x == "M":
            continue
        widths.append(int(x * (1 - rates[cnt])))
        cnt += 1
    return widths


def resnet_50_widths(compress_rate):
    """(overall_channel, mid_channel) of models/imagenet/resnet.py:8-27."""
    rates = list(compress_rate)
    stage_oup = [rates[0]]
    for i in range(len(RESNET50_STAGE_REPEAT) - 1):
        stage_oup += [rates[i + 1]] * RESNET50_STAGE_REPEAT[i]
    stage_oup += [0.0] * RESNET50_STAGE_REPEAT[-1]
    mid_rates = rates[len(RESNET50_STAGE_REPEAT):]
    overall, mid = [], []
    for i, c in enumerate(RESNET50_STAGE_OUT):
        overall.append(int(c * (1 - stage_oup[i])))
        if i > 0:
            mid.append(int(c // 4 * (1 - mid_rates[i - 1])))
    return overall, mid


def resnet_50_convs():
    """The convolutions in the order load_resnet_imagenet_model visits them (utils/load_models.py:457-512):
    (conv name, bn name, record_last, block index or None, kind) with kind in
    {'stem', 'mid', 'downsample', 'out'}; score file k+1 belongs to entry k."""
    convs = [("conv1", "bn1", True, None, "stem")]
    blk = 0
    for layer, num in enumerate(RESNET50_STAGE_REPEAT):
        for k in range(num):
            base = "layer%d.%d." % (layer + 1, k)
            convs.append((base + "conv1", base + "bn1", True, blk, "mid"))
            convs.append((base + "conv2", base + "bn2", True, blk, "mid"))
            if k == 0:
                convs.append((base + "downsample.0", base + "downsample.1", False, blk, "downsample"))
            convs.append((base + "conv3", base + "bn3", True, blk, "out"))
            blk += 1
    return convs


def resnet_50_kept(compress_rate):
    """[(score file stem, original width, kept width)] for the 53 files, consumer order."""
    overall, mid = resnet_50_widths(compress_rate)
    out = []
    for k, (_, _, _, blk, kind) in enumerate(resnet_50_convs()):
        if kind == "stem":
            ori, cur = RESNET50_STAGE_OUT[0], overall[0]
        elif kind == "mid":
            ori, cur = RESNET50_STAGE_OUT[blk + 1] // 4, mid[blk]
        else:
            ori, cur = RESNET50_STAGE_OUT[blk + 1], overall[blk + 1]
        out.append(("imp_conv%d" % (k + 1), ori, cur))
    return out


def vgg_16_bn_kept(compress_rate):
    """[(score file stem, original width, kept width)] for the 13 convolutions (the 13th has no score
    file and never shrinks)."""
    widths = vgg_16_bn_widths(compress_rate)
    ori = [x for x in VGG_CFG if x != "M"]
    return [("imp_conv%d" % (k + 1), o, w) for k, (o, w) in enumerate(zip(ori, widths))]


def _resnet_cifar_stage_out(num_layers):
    rep = RESNET_CIFAR_REPEAT[num_layers]
    return [16] + [16] * rep[0] + [32] * rep[1] + [64] * rep[2]


def resnet_cifar_widths(compress_rate, num_layers):
    """(overall_channel, mid_channel) of models/cifar10/resnet.py:5-30 for resnet_56 / resnet_110."""
    rep = RESNET_CIFAR_REPEAT[num_layers]
    rates = list(compress_rate)
    stage_oup = [rates[0]]
    for i in range(len(rep) - 1):
        stage_oup += [rates[i + 1]] * rep[i]
    stage_oup += [0.0] * rep[-1]
    mid_rates = rates[len(rep):]
    stage_out = _resnet_cifar_stage_out(num_layers)
    overall = [int(c * (1 - r)) for c, r in zip(stage_out, stage_oup)]
    mid = [int(stage_out[i] * (1 - mid_rates[i - 1])) for i in range(1, len(stage_out))]
    return overall, mid


def resnet_cifar_convs(num_layers):
    """[(conv name, score file stem, block index, 'mid' | 'out')] in the order load_resnet_model visits
    them (utils/load_models.py:83-94): score file cnt with cnt = 2, 3, ... (imp_conv1 is the stem's)."""
    convs, cnt, blk = [], 1, 0
    for layer, num in enumerate(RESNET_CIFAR_REPEAT[num_layers]):
        for k in range(num):
            for l in range(2):
                cnt += 1
                convs.append(("layer%d.%d.conv%d" % (layer + 1, k, l + 1), "imp_conv%d" % cnt, blk, "mid" if l == 0 else "out"))
            blk += 1
    return convs


def resnet_cifar_kept(compress_rate, num_layers):
    """[(score file stem, original width, kept width)] for the files the consumer reads."""
    overall, mid = resnet_cifar_widths(compress_rate, num_layers)
    stage_out = _resnet_cifar_stage_out(num_layers)
    return [(stem, stage_out[blk + 1], mid[blk] if kind == "mid" else overall[blk + 1])
            for _, stem, blk, kind in resnet_cifar_convs(num_layers)]


def densenet_40_conv_names():
    """The 39 convolutions in named_modules() order (models/cifar10/densenet.py:68-75)."""
    names = ["conv1"]
    for b in (1, 2, 3):
        names += ["dense%d.%d.conv1" % (b, i) for i in range(DENSENET40_BLOCK)]
        if b < 3:
            names.append("trans%d.conv1" % b)
    return names


def densenet_40_widths(compress_rate):
    """Output channels of the 39 convolutions (models/cifar10/densenet.py:67-75, :91-104): conv1 is never
    pruned (compress_rate[0] is not read), a dense layer keeps int(12*(1-r)) filters, a transition
    int(floor(inplanes*(1-r)))."""
    n, rates = DENSENET40_BLOCK, list(compress_rate)
    inplanes = 2 * DENSENET40_GROWTH
    widths = [inplanes]
    for b in range(3):
        for r in rates[b * (n + 1) + 1:b * (n + 1) + 1 + n]:
            w = int(DENSENET40_GROWTH * (1 - r))
            widths.append(w)
            inplanes += w
        if b < 2:
            inplanes = int(math.floor(inplanes * (1 - rates[(b + 1) * (n + 1)]) // 1))
            widths.append(inplanes)
    return widths


def densenet_40_kept(compress_rate):
    """[(score file stem, original width, kept width)], consumer order (cov_id = 1 ... 39)."""
    ori = densenet_40_widths([0.0] * 39)
    return [("imp_conv%d" % (k + 1), o, w) for k, (o, w) in enumerate(zip(ori, densenet_40_widths(compress_rate)))]


def googlenet_widths(compress_rate, filters=None):
    """Per Inception block (n1x1, 3x3 conv out, 5x5-branch middle conv out, 5x5-branch last conv out,
    pool_planes) of the pruned net (models/cifar10/googlenet.py:28-66: int(n*(1-rate)), the last block
    keeps the full width on its two branch outputs but not on the middle 5x5-branch conv)."""
    filters = filters or GOOGLENET_FILTERS
    out = []
    for i, (n1, n3, n5, pool) in enumerate(filters):
        keep = 1 - compress_rate[i + 1]
        last = i == len(filters) - 1
        out.append((n1, n3 if last else int(n3 * keep), int(n5 * keep), n5 if last else int(n5 * keep), pool))
    return out


def googlenet_kept(compress_rate, filters=None):
    """[(score file stem, original width, kept width)] for the files the consumer reads (the _n5x5 file
    of a block serves both 5x5-branch convs; listed once, with the middle conv's width)."""
    filters = filters or GOOGLENET_FILTERS
    out = []
    for i, (w, f) in enumerate(zip(googlenet_widths(compress_rate, filters), filters)):
        out.append(("imp_conv%d_n3x3" % (i + 2), f[1], w[1]))
        out.append(("imp_conv%d_n5x5" % (i + 2), f[2], w[2]))
    return out


def _load_imp(imp_score, stem):
    if isinstance(imp_score, dict):
        return np.asarray(imp_score[stem])
    return np.load(os.path.join(imp_score, stem + ".npy"))


def _rows(t, idx):
    return t.index_select(0, torch.as_tensor(idx, dtype=torch.long, device=t.device))


def _rows_cols(t, rows, cols):
    if rows is not None:
        t = _rows(t, rows)
    if cols is not None:
        t = t.index_select(1, torch.as_tensor(cols, dtype=torch.long, device=t.device))
    return t


def transplant_vgg(state_dict, oristate_dict, imp_score, conv_names=None):
    """utils/load_models.py:17-64. `state_dict`: the pruned model's (updated in place and returned);
    `imp_score`: score directory or {stem: array}; conv_names: the Conv2d modules in named_modules()
    order (default: features.conv<i> of the reference's VGG)."""
    if conv_names is None:
        conv_names = ["features.conv%d" % i for i, x in enumerate(VGG_CFG) if x != "M"]
    last = None
    for cnt, name in enumerate(conv_names, start=1):
        key = name + ".weight"
        ori, cur = oristate_dict[key], state_dict[key]
        o, c = ori.size(0), cur.size(0)
        if o != c:
            sel = select_index(_load_imp(imp_score, "imp_conv%d" % cnt), o, c)
            if last is not None:
                # [index_i][index_j] = ori[i][j] for i in select_index, j in last_select_index (:43-47);
                # input channels beyond len(last) keep what the slim tensor had
                cur[:, :len(last)] = _rows_cols(ori, sel, last).to(cur.dtype)
            else:
                cur.copy_(_rows(ori, sel))
            last = sel
        elif last is not None:
            cur[:, :len(last)] = _rows_cols(ori, None, last).to(cur.dtype)  # :55-59
        else:
            state_dict[key] = ori  # :61 (the reference rebinds the entry to the original tensor)
            last = None
    return state_dict


def transplant_resnet_50(state_dict, oristate_dict, imp_score):
    """utils/load_models.py:441-582 for args.net == 'resnet_50'."""
    honey = set()
    last = None
    for k, (conv, bn, record_last, _, _) in enumerate(resnet_50_convs()):
        key = conv + ".weight"
        honey.add(key)
        ori, cur = oristate_dict[key], state_dict[key]
        o, c = ori.size(0), cur.size(0)
        stem = k == 0
        if o != c:
            sel = select_index(_load_imp(imp_score, "imp_conv%d" % (k + 1)), o, c)
            if last is not None and not stem:
                cur[:, :len(last)] = _rows_cols(ori, sel, last)
            else:
                cur.copy_(_rows(ori, sel))
            for part in _BN_PARTS:
                state_dict[bn + part].copy_(_rows(oristate_dict[bn + part], sel))
            if record_last or stem:
                last = sel
        elif last is not None and not stem:
            cur[:, :len(last)] = _rows_cols(ori, None, last)
            for part in _BN_PARTS:
                state_dict[bn + part] = oristate_dict[bn + part]
            if record_last:
                last = None
        else:
            state_dict[key] = ori
            for part in _BN_PARTS:
                state_dict[bn + part] = oristate_dict[bn + part]
            if record_last and not stem:
                last = None
        state_dict[bn + ".num_batches_tracked"] = oristate_dict[bn + ".num_batches_tracked"]
    # :570-580: every other conv (none for resnet_50) and the linear layer come over unchanged
    for key in oristate_dict:
        if key.endswith(".weight") and oristate_dict[key].dim() == 4 and key not in honey:
            state_dict[key] = oristate_dict[key]
    for key in ("fc.weight", "fc.bias"):
        if key in oristate_dict:
            state_dict[key] = oristate_dict[key]
    return state_dict


def transplant_resnet_cifar(state_dict, oristate_dict, imp_score, num_layers):
    """utils/load_models.py:67-143 (load_resnet_model, layer = 56 or 110)."""
    visited = set()
    last = None
    for conv, stem, _, _ in resnet_cifar_convs(num_layers):
        key = conv + ".weight"
        visited.add(key)
        ori, cur = oristate_dict[key], state_dict[key]
        o, c = ori.size(0), cur.size(0)
        if o != c:
            sel = select_index(_load_imp(imp_score, stem), o, c)
            if last is not None:
                cur[:, :len(last)] = _rows_cols(ori, sel, last)
            else:
                cur.copy_(_rows(ori, sel))
            last = sel
        elif last is not None:
            cur[:, :len(last)] = _rows_cols(ori, None, last)
            last = None  # :123 (the VGG loader keeps it here)
        else:
            state_dict[key] = ori
            last = None
    # :126-141: every other conv (the stem; shortcuts are parameter-free pads) and the linear layer whole
    for key, t in oristate_dict.items():
        if key.endswith(".weight") and t.dim() == 4 and key not in visited and "shortcut" not in key:
            state_dict[key] = t
        elif key.endswith(".weight") and t.dim() == 2:
            state_dict[key] = t
            state_dict[key[:-len("weight")] + "bias"] = oristate_dict[key[:-len("weight")] + "bias"]
    return state_dict


def transplant_densenet_40(state_dict, oristate_dict, imp_score, conv_names=None):
    """utils/load_models.py:385-438 (load_densenet_model)."""
    last = []  # :388 - a list from the start: the `last_select_index is not None` tests are always true
    for cov_id, name in enumerate(conv_names or densenet_40_conv_names(), start=1):
        key = name + ".weight"
        ori, cur = oristate_dict[key], state_dict[key]
        o, c = ori.size(0), cur.size(0)
        if o != c:
            sel = [int(i) for i in select_index(_load_imp(imp_score, "imp_conv%d" % cov_id), o, c)]
            if last:
                cur[:, :len(last)] = _rows_cols(ori, sel, last)
        else:
            if last:
                cur[:, :len(last)] = _rows_cols(ori, None, last)
            sel = list(range(o))
        if cov_id in (1, 14, 27):  # :432 conv1 and the transitions restart the concatenation
            last = list(sel)
        else:
            shift = cov_id * 12 - (cov_id - 1) // 13 * 12  # :435
            last = last + [x + shift for x in sel]
    return state_dict


def transplant_googlenet(state_dict, oristate_dict, imp_score, filters=None, blocks=None):
    """utils/load_models.py:146-382 (load_google_model with cpr=None, as load_model calls it). `filters`:
    the original filter table the offsets come from (default: the reference's; tests pass a miniature)."""
    filters = filters or GOOGLENET_FILTERS
    blocks = blocks or GOOGLENET_BLOCKS
    sketch_bn = {"pre_layers.1"}
    cur_last = []
    # pre_layers (cov_id 1, :332-358): transplanted only if it lost filters - the constructor never prunes it
    ori, cur = oristate_dict["pre_layers.0.weight"], state_dict["pre_layers.0.weight"]
    if ori.size(0) != cur.size(0):
        sel = select_index(_load_imp(imp_score, "imp_conv1"), ori.size(0), cur.size(0))
        cur_last = [int(i) for i in sel]
        cur.copy_(_rows(ori, sel))
    for b, name in enumerate(blocks):
        cov_id, f = b + 2, filters[b]
        sketch_bn |= {name + ".branch3x3.4", name + ".branch5x5.4", name + ".branch5x5.7"}
        last, cur_last = cur_last, []
        pool_part = []
        for entry in (".branch1x1.0", ".branch3x3.0", ".branch5x5.0", ".branch_pool.1"):  # :208-242 input channels only
            key = name + entry + ".weight"
            ori, cur = oristate_dict[key], state_dict[key]
            cols = last if ori.size(1) != cur.size(1) else None
            n = len(cols) if cols is not None else ori.size(1)
            if n:
                cur[:, :n] = _rows_cols(ori[:cur.size(0)], None, cols)
            if entry == ".branch1x1.0":
                cur_last += list(range(cur.size(0)))
            elif entry == ".branch_pool.1":
                pool_part = [x + f[0] + f[1] + f[2] for x in range(cur.size(0))]
        cur_last += pool_part  # appended before the 3x3 / 5x5 parts (:236-242 run before :275, :320)
        sel5 = None
        for entry, stem in ((".branch3x3.3", "_n3x3"), (".branch5x5.3", "_n5x5")):  # :244-279 filters only
            key = name + entry + ".weight"
            ori, cur = oristate_dict[key], state_dict[key]
            o, c = ori.size(0), cur.size(0)
            sel = [int(i) for i in select_index(_load_imp(imp_score, "imp_conv%d%s" % (cov_id, stem)), o, c)] if o != c else list(range(o))
            cur[:len(sel)] = _rows(ori, sel)
            if stem == "_n3x3":
                cur_last += [x + f[0] for x in sel]
            else:
                sel5 = sel
        key = name + ".branch5x5.6.weight"  # :281-328 filters and input channels
        ori, cur = oristate_dict[key], state_dict[key]
        cols = sel5 if ori.size(1) != cur.size(1) else list(range(ori.size(1)))
        o, c = ori.size(0), cur.size(0)
        sel = [int(i) for i in select_index(_load_imp(imp_score, "imp_conv%d_n5x5" % cov_id), o, c)] if o != c else list(range(o))
        cur_last += [x + f[0] + f[1] for x in sel]
        cur[:len(sel), :len(cols)] = _rows_cols(ori, sel, cols)
    # :361-380: every conv is on the sketch list (nothing to copy, biases included); batch-norms off the
    # list and the linear layer come over whole
    for key in oristate_dict:
        if key.endswith(".running_mean") and key[:-len(".running_mean")] not in sketch_bn:
            for part in _BN_PARTS:
                state_dict[key[:-len(".running_mean")] + part] = oristate_dict[key[:-len(".running_mean")] + part]
        elif key.endswith(".weight") and oristate_dict[key].dim() == 2:
            state_dict[key] = oristate_dict[key]
            state_dict[key[:-len("weight")] + "bias"] = oristate_dict[key[:-len("weight")] + "bias"]
    return state_dict


# ---------------------------------------------------------------------------------------------
# U^2-Net-p (models/DUTS/u2net.py, utils/load_models.py:585-769)
# ---------------------------------------------------------------------------------------------
U2NETP_DEPTH = {1: 7, 2: 6, 3: 5, 4: 4, 5: 4, 6: 4}   # RSU7, RSU6, RSU5, RSU4, RSU4F, RSU4F (models/DUTS/u2net.py:455-477)


def u2netp_conv_shapes(compress_rate, in_ch=3, out_ch=1, inner=16, outer=64):
    """[(conv module name, out channels, in channels)] of U2NETP(compress_rate) in named_modules() order,
    'outconv' excluded (models/DUTS/u2net.py:34-79, :137-175, :219-251, :288-313, :344-365, :386-486).
    Inside an RSU of depth D the first D-2 inner widths follow the rates (at least 1 channel), the two
    deepest convs keep `inner`; the five outer widths follow compress_rate[34:39]. `inner` / `outer`
    are 16 / 64 in the reference (smaller values give the tests a cheap network of the same shape)."""
    r = list(compress_rate)
    spans = {(1, False): (0, 5), (2, False): (5, 9), (3, False): (9, 12), (4, False): (12, 14), (5, False): (14, 16),
             (6, False): (16, 18), (1, True): (18, 23), (2, True): (23, 27), (3, True): (27, 30), (4, True): (30, 32),
             (5, True): (32, 34)}                                                     # adapt_channel, :386-401
    ext = [max(1, int((1 - x) * outer)) for x in r[34:39]]                           # :438-452
    m1, m2, m3, m4, m5 = ext

    def rsu(prefix, stage, rates, cin, cout):
        depth = U2NETP_DEPTH[stage]
        mid = [max(1, int((1 - rates[k]) * inner)) for k in range(depth - 2)] + [inner]  # mid_ch1 .. mid_ch{D-1}
        convs = [(prefix + ".rebnconvin.conv_s1", cout, cin), (prefix + ".rebnconv1.conv_s1", mid[0], cout)]
        for k in range(2, depth):
            convs.append((prefix + ".rebnconv%d.conv_s1" % k, mid[k - 1], mid[k - 2]))
        convs.append((prefix + ".rebnconv%d.conv_s1" % depth, mid[depth - 2], mid[depth - 2]))
        for k in range(depth - 1, 1, -1):
            convs.append((prefix + ".rebnconv%dd.conv_s1" % k, mid[k - 2], 2 * mid[k - 1]))
        convs.append((prefix + ".rebnconv1d.conv_s1", cout, 2 * mid[0]))
        return convs

    enc = [(1, in_ch, m1), (2, m1, m2), (3, m2, m3), (4, m3, m4), (5, m4, m5), (6, m5, m5)]
    dec = [(5, 2 * m5, m4), (4, 2 * m4, m3), (3, 2 * m3, m2), (2, 2 * m2, m1), (1, 2 * m1, m1)]
    out = []
    for stage, cin, cout in enc:
        lo, hi = spans[(stage, False)]
        out += rsu("stage%d" % stage, stage, r[lo:hi], cin, cout)
    for stage, cin, cout in dec:
        lo, hi = spans[(stage, True)]
        out += rsu("stage%dd" % stage, stage, r[lo:hi], cin, cout)
    for k, c in zip(range(1, 7), (m1, m1, m2, m3, m4, m5)):                          # :479-484
        out.append(("side%d" % k, out_ch, c))
    return out


def transplant_u2netp(state_dict, oristate_dict, imp_score, conv_names=None):
    """utils/load_models.py:585-769 (load_u2netp_model), the copy loops as index_select. Only conv
    `.weight` tensors are written (no conv bias, no batch-norm tensor, :609-610); the control flow -
    stage counter, the three lists of remembered indices, which branches read a score file - is the
    reference's, including where it raises (an un-pruned decoder `rebnconvin` behind a pruned layer
    evaluates int('i'), :658; a stage that starts with no live index evaluates list(None), :617/:621)."""
    if conv_names is None:
        conv_names = [n for n, _, _ in u2netp_conv_shapes([0.0] * 39)]
    last = None
    cnt, stage_id = 0, 1
    saved, saved_stage, saved_side = [], [], []

    def cols(dst, src_rows, col_idx, col_base, dst0):
        """dst[:, dst0 + k] = src_rows[:, col_idx[k] + col_base]"""
        idx = torch.as_tensor(np.asarray(col_idx, dtype=np.int64) + col_base, dtype=torch.long, device=src_rows.device)
        dst[:, dst0:dst0 + len(col_idx)] = src_rows.index_select(1, idx).to(dst.dtype)

    for name in conv_names:
        if name == "outconv":
            break
        side_name = name.split(".")[0]
        is_side = side_name[:4] == "side"
        decode = side_name[-1] == "d"
        flag = "d." if decode else "."
        midfix = None if is_side else name.split(".")[1]
        key = name + ".weight"
        ori, cur = oristate_dict[key], state_dict[key]
        o, c, oin = ori.size(0), cur.size(0), ori.size(1)
        cov_id = None if is_side else midfix[-2:]
        if decode and side_name[-2] != str(stage_id):                                   # :616-619
            stage_id -= 1
            saved_side.append(list(last))
            saved = []
        elif (not decode) and side_name[-1] != str(stage_id):                           # :620-623
            stage_id += 1
            saved_stage.append(list(last))
            saved = []

        def rank():
            return _load_imp(imp_score, "net.stage%d%s%s.relu_s1" % (stage_id, flag, midfix))

        if not is_side:
            if decode and cov_id == "in":
                second = "stage"          # second half of the input: the encoder stage's output (:638)
            elif cov_id[1] != "d":
                second = None
            else:
                second = "unit"           # the RSU's own rebnconv<k> (:717)
            if o != c:
                sel = select_index(rank(), o, c)
                rows = _rows(ori, sel)
                if second is None:
                    if last is not None:
                        cols(cur, rows, last, 0, 0)
                    else:
                        cur.copy_(rows)
                else:
                    other = saved_stage[stage_id - 1] if second == "stage" else saved[int(cov_id[0])]
                    cols(cur, rows, last, 0, 0)
                    cols(cur, rows, other, int(oin / 2), len(last))
                last = sel
                if second != "unit":
                    saved.append(list(sel))
            elif last is not None:
                sel = select_index(rank(), o, c)
                if second is not None:
                    other = saved[int(cov_id[0])]  # :658 reads cov_id[0] for 'in' too (ValueError there)
                cols(cur, ori, last, 0, 0)
                if second is not None:
                    cols(cur, ori, other, int(oin / 2), len(last))
                last = sel
                if second != "unit":
                    saved.append(list(sel))
            else:
                state_dict[key] = ori
                last = None
                if second != "unit":
                    saved.append(None)
        else:
            cnt += 1
            if o != c:
                sel = select_index(_load_imp(imp_score, "net.side%d" % cnt), o, c)
                rows = _rows(ori, sel)
                if last is not None:
                    cols(cur, rows, last, 0, 0)
                else:
                    cur.copy_(rows)
            elif last is not None:
                cols(cur, ori, last, 0, 0)
            else:
                state_dict[key] = ori
            last = saved_side[5 - cnt]                                                   # :767
    return state_dict
