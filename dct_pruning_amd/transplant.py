"""Weight transplant from the full-width network into the pruned one - the step right after the
score path, vectorised.

The reference walks the convolutions in order, derives the kept filters of each from its score file
    select_index = np.argsort(imp)[orifilter_num - currentfilter_num:]; select_index.sort()
and copies weights one scalar slice at a time in Python double / triple loops
(utils/load_models.py:17-64 for VGG-16-bn, :441-582 for ResNet-50: ~10^7 interpreter-level copies
for ResNet-50). Here every such loop is one index_select on the output-filter axis, one on the
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

Widths of the pruned networks (what `currentfilter_num` is) follow the model constructors:
models/cifar10/vgg.py:37 and models/imagenet/resnet.py:8-27 (adapt_channel).
"""
import os

import numpy as np
import torch

from .masks import select_index

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]  # models/cifar10/vgg.py:5
RESNET50_STAGE_REPEAT = [3, 4, 6, 3]                                                                # models/imagenet/resnet.py:3
RESNET50_STAGE_OUT = [64] + [256] * 3 + [512] * 4 + [1024] * 6 + [2048] * 3                         # :4
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
