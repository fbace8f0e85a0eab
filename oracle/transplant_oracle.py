"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's weight-transplant loops, element by
element as the reference writes them, for parity tests of dct_pruning_amd/transplant.py.
Only tests/ may import this module.

Restates utils/load_models.py:17-64 (load_vgg_model), :67-143 (load_resnet_model), :385-438
(load_densenet_model), :146-382 (load_google_model), :441-582 (load_resnet_imagenet_model, resnet_50 branch) and :585-769 (load_u2netp_model). The reference iterates model.named_modules(); here the ordered list of conv
module names is passed in (the state-dict keys are the same), and score files come from a dict
{stem: array} instead of np.load(args.imp_score + '/imp_conv%d.npy'). Everything else - the branch
structure, which tensors are copied scalar slice by scalar slice, which are rebound, when
last_select_index is kept or reset - follows the reference line by line.

Pinned by: tests/golden/transplant_<net>.json - digests of the state dicts the REFERENCE's own loaders
produce at full width on seeded weights and score files (tests/golden/make_transplant_goldens.py ran
utils/load_models.py in the build container). tests/test_transplant_goldens.py holds the restatements of
load_vgg_model, load_densenet_model, load_resnet_model (56) and load_u2netp_model to those digests; the
ResNet-50 and GoogLeNet restatements run for a minute at full width and are tied to the pinned ones only
through dct_pruning_amd/transplant.py (equal to them on miniature networks, equal to the reference's
digests at full width)."""
import copy

import numpy as np


def load_vgg_model(state_dict, oristate_dict, imp, conv_names):
    """utils/load_models.py:17-64."""
    last_select_index = None  # :19
    cnt = 0
    for name in conv_names:  # :23-26 (named_modules() filtered by isinstance(module, nn.Conv2d))
        cnt += 1  # :28
        oriweight = oristate_dict[name + '.weight']  # :29
        curweight = state_dict[name + '.weight']  # :30
        orifilter_num = oriweight.size(0)  # :31
        currentfilter_num = curweight.size(0)  # :32
        if orifilter_num != currentfilter_num:  # :34
            cov_id = cnt  # :36
            select_index = np.argsort(imp['imp_conv' + str(cov_id)])[orifilter_num - currentfilter_num:]  # :38-40
            select_index.sort()  # :41
            if last_select_index is not None:  # :43
                for index_i, i in enumerate(select_index):  # :44
                    for index_j, j in enumerate(last_select_index):  # :45
                        state_dict[name + '.weight'][index_i][index_j] = \
                            oristate_dict[name + '.weight'][i][j]  # :46-47
            else:
                for index_i, i in enumerate(select_index):  # :49
                    state_dict[name + '.weight'][index_i] = \
                        oristate_dict[name + '.weight'][i]  # :50-51
            last_select_index = select_index  # :53
        elif last_select_index is not None:  # :55
            for i in range(orifilter_num):  # :56
                for index_j, j in enumerate(last_select_index):  # :57
                    state_dict[name + '.weight'][i][index_j] = \
                        oristate_dict[name + '.weight'][i][j]  # :58-59
        else:
            state_dict[name + '.weight'] = oriweight  # :61
            last_select_index = None  # :62
    return state_dict


def load_resnet_imagenet_model(state_dict, oristate_dict, imp):
    """utils/load_models.py:441-582 with args.net == 'resnet_50'."""
    current_cfg = [3, 4, 6, 3]  # :442-449
    last_select_index = None  # :452
    all_honey_conv_weight = []  # :454
    bn_part_name = ['.weight', '.bias', '.running_mean', '.running_var']  # :456
    cnt = 1  # :459

    conv_weight_name = 'conv1.weight'  # :461
    all_honey_conv_weight.append(conv_weight_name)
    oriweight = oristate_dict[conv_weight_name]
    curweight = state_dict[conv_weight_name]
    orifilter_num = oriweight.size(0)
    currentfilter_num = curweight.size(0)
    if orifilter_num != currentfilter_num:  # :468
        select_index = np.argsort(imp['imp_conv' + str(cnt)])[orifilter_num - currentfilter_num:]  # :470-471
        select_index.sort()
        for index_i, i in enumerate(select_index):  # :474
            state_dict[conv_weight_name][index_i] = oristate_dict[conv_weight_name][i]
            for bn_part in bn_part_name:
                state_dict['bn1' + bn_part][index_i] = oristate_dict['bn1' + bn_part][i]
        last_select_index = select_index  # :481
    else:
        state_dict[conv_weight_name] = oriweight  # :483
        for bn_part in bn_part_name:
            state_dict['bn1' + bn_part] = oristate_dict['bn1' + bn_part]
    state_dict['bn1' + '.num_batches_tracked'] = oristate_dict['bn1' + '.num_batches_tracked']  # :487

    cnt += 1  # :489
    for layer, num in enumerate(current_cfg):  # :490
        layer_name = 'layer' + str(layer + 1) + '.'
        for k in range(num):  # :493
            iter = 3  # :497 (resnet_50)
            if k == 0:
                iter += 1  # :498-499
            for l in range(iter):  # :500
                record_last = True
                if k == 0 and l == 2:  # :502
                    conv_name = layer_name + str(k) + '.downsample.0'
                    bn_name = layer_name + str(k) + '.downsample.1'
                    record_last = False
                elif k == 0 and l == 3:  # :506
                    conv_name = layer_name + str(k) + '.conv' + str(l)
                    bn_name = layer_name + str(k) + '.bn' + str(l)
                else:
                    conv_name = layer_name + str(k) + '.conv' + str(l + 1)
                    bn_name = layer_name + str(k) + '.bn' + str(l + 1)

                conv_weight_name = conv_name + '.weight'  # :513
                all_honey_conv_weight.append(conv_weight_name)
                oriweight = oristate_dict[conv_weight_name]
                curweight = state_dict[conv_weight_name]
                orifilter_num = oriweight.size(0)
                currentfilter_num = curweight.size(0)

                if orifilter_num != currentfilter_num:  # :520
                    select_index = np.argsort(imp['imp_conv' + str(cnt)])[orifilter_num - currentfilter_num:]  # :522-523
                    select_index.sort()
                    if last_select_index is not None:  # :526
                        for index_i, i in enumerate(select_index):
                            for index_j, j in enumerate(last_select_index):
                                state_dict[conv_weight_name][index_i][index_j] = \
                                    oristate_dict[conv_weight_name][i][j]
                            for bn_part in bn_part_name:
                                state_dict[bn_name + bn_part][index_i] = \
                                    oristate_dict[bn_name + bn_part][i]
                    else:  # :536
                        for index_i, i in enumerate(select_index):
                            state_dict[conv_weight_name][index_i] = \
                                oristate_dict[conv_weight_name][i]
                            for bn_part in bn_part_name:
                                state_dict[bn_name + bn_part][index_i] = \
                                    oristate_dict[bn_name + bn_part][i]
                    if record_last:  # :545
                        last_select_index = select_index
                elif last_select_index is not None:  # :548
                    for index_i in range(orifilter_num):
                        for index_j, j in enumerate(last_select_index):
                            state_dict[conv_weight_name][index_i][index_j] = \
                                oristate_dict[conv_weight_name][index_i][j]
                    for bn_part in bn_part_name:
                        state_dict[bn_name + bn_part] = oristate_dict[bn_name + bn_part]
                    if record_last:
                        last_select_index = None
                else:  # :561
                    state_dict[conv_weight_name] = oriweight
                    for bn_part in bn_part_name:
                        state_dict[bn_name + bn_part] = oristate_dict[bn_name + bn_part]
                    if record_last:
                        last_select_index = None
                state_dict[bn_name + '.num_batches_tracked'] = oristate_dict[bn_name + '.num_batches_tracked']  # :568
                cnt += 1

    for key in oristate_dict:  # :571-580: convs not visited above, and the linear layer
        if key.endswith('.weight') and oristate_dict[key].dim() == 4 and key not in all_honey_conv_weight:
            state_dict[key] = oristate_dict[key]
    for key in ('fc.weight', 'fc.bias'):
        if key in oristate_dict:
            state_dict[key] = oristate_dict[key]
    return state_dict


def load_resnet_model(state_dict, oristate_dict, layer, imp, modules):
    """utils/load_models.py:67-143. `modules`: [(name, 'conv' | 'linear')] - what the reference gets from
    model.named_modules() filtered by isinstance(module, nn.Conv2d / nn.Linear) (:126-141)."""
    cfg = {56: [9, 9, 9], 110: [18, 18, 18]}  # :68-71
    current_cfg = cfg[layer]
    last_select_index = None  # :76
    all_conv_weight = []  # :78
    cnt = 1  # :83
    for layer, num in enumerate(current_cfg):  # :84
        layer_name = 'layer' + str(layer + 1) + '.'
        for k in range(num):  # :86
            for l in range(2):  # :87
                cnt += 1  # :89
                cov_id = cnt
                conv_name = layer_name + str(k) + '.conv' + str(l + 1)  # :92
                conv_weight_name = conv_name + '.weight'
                all_conv_weight.append(conv_weight_name)
                oriweight = oristate_dict[conv_weight_name]
                curweight = state_dict[conv_weight_name]
                orifilter_num = oriweight.size(0)
                currentfilter_num = curweight.size(0)
                if orifilter_num != currentfilter_num:  # :100
                    select_index = np.argsort(imp['imp_conv' + str(cov_id)])[orifilter_num - currentfilter_num:]  # :102-103
                    select_index.sort()
                    if last_select_index is not None:  # :106
                        for index_i, i in enumerate(select_index):
                            for index_j, j in enumerate(last_select_index):
                                state_dict[conv_weight_name][index_i][index_j] = \
                                    oristate_dict[conv_weight_name][i][j]
                    else:  # :111
                        for index_i, i in enumerate(select_index):
                            state_dict[conv_weight_name][index_i] = \
                                oristate_dict[conv_weight_name][i]
                    last_select_index = select_index  # :116
                elif last_select_index is not None:  # :118
                    for index_i in range(orifilter_num):
                        for index_j, j in enumerate(last_select_index):
                            state_dict[conv_weight_name][index_i][index_j] = \
                                oristate_dict[conv_weight_name][index_i][j]
                    last_select_index = None  # :123
                else:  # :125
                    state_dict[conv_weight_name] = oriweight
                    last_select_index = None
    for name, kind in modules:  # :129
        if kind == 'conv':
            conv_name = name + '.weight'
            if 'shortcut' in name:  # :134
                continue
            if conv_name not in all_conv_weight:
                state_dict[conv_name] = oristate_dict[conv_name]
        elif kind == 'linear':  # :139
            state_dict[name + '.weight'] = oristate_dict[name + '.weight']
            state_dict[name + '.bias'] = oristate_dict[name + '.bias']
    return state_dict


def load_densenet_model(state_dict, oristate_dict, imp, conv_names):
    """utils/load_models.py:385-438."""
    last_select_index = []  # :388 - an empty list, so every `is not None` below is true
    cnt = 0
    for name in conv_names:  # :393-396
        cnt += 1
        cov_id = cnt
        oriweight = oristate_dict[name + '.weight']
        curweight = state_dict[name + '.weight']
        orifilter_num = oriweight.size(0)
        currentfilter_num = curweight.size(0)
        if orifilter_num != currentfilter_num:  # :405
            select_index = list(np.argsort(imp['imp_conv' + str(cov_id)])[orifilter_num - currentfilter_num:])  # :407-408
            select_index.sort()
            if last_select_index is not None:  # :411
                for index_i, i in enumerate(select_index):
                    for index_j, j in enumerate(last_select_index):
                        state_dict[name + '.weight'][index_i][index_j] = \
                            oristate_dict[name + '.weight'][i][j]
            else:
                for index_i, i in enumerate(select_index):
                    state_dict[name + '.weight'][index_i] = \
                        oristate_dict[name + '.weight'][i]
        elif last_select_index is not None:  # :421
            for i in range(orifilter_num):
                for index_j, j in enumerate(last_select_index):
                    state_dict[name + '.weight'][i][index_j] = \
                        oristate_dict[name + '.weight'][i][j]
            select_index = list(range(0, orifilter_num))  # :426
        else:
            select_index = list(range(0, orifilter_num))  # :429
            state_dict[name + '.weight'] = oriweight
        if cov_id == 1 or cov_id == 14 or cov_id == 27:  # :432
            last_select_index = select_index
        else:
            tmp_select_index = [x + cov_id * 12 - (cov_id - 1) // 13 * 12 for x in select_index]  # :435
            last_select_index += tmp_select_index
    return state_dict


def load_google_model(state_dict, oristate_dict, imp, modules, filters=None):
    """utils/load_models.py:146-382 with cpr=None (how load_model calls it, :822). `modules`: the
    reference's model.named_modules() reduced to [(name, kind)] with kind in 'inception', 'pre_layers',
    'conv', 'bn', 'linear', in module order. `filters`: the table of :149-159 (argument only so that
    tests can run a miniature network; None = the reference's)."""
    if filters is None:
        filters = [
            [64, 128, 32, 32],
            [128, 192, 96, 64],
            [192, 208, 48, 64],
            [160, 224, 64, 64],
            [128, 256, 64, 64],
            [112, 288, 64, 64],
            [256, 320, 128, 128],
            [256, 320, 128, 128],
            [384, 384, 128, 128]
        ]
    all_honey_conv_name = []  # :166
    all_honey_bn_name = []
    cur_last_select_index = []  # :168

    def branch_of(weight_index):  # :211-218, :246-253, :283-290
        if '3x3' in weight_index:
            return '_n3x3'
        elif '5x5' in weight_index:
            return '_n5x5'
        elif '1x1' in weight_index:
            return '_n1x1'
        elif 'pool' in weight_index:
            return '_pool_planes'

    cnt = 0  # :170
    for name, kind in modules:  # :173
        if kind == 'inception':  # :176
            cnt += 1
            cov_id = cnt
            honey_filter_channel_index = ['.branch5x5.6']  # :181-183
            honey_channel_index = ['.branch1x1.0', '.branch3x3.0', '.branch5x5.0', '.branch_pool.1']  # :184-189
            honey_filter_index = ['.branch3x3.3', '.branch5x5.3']  # :190-193
            honey_bn_index = ['.branch3x3.4', '.branch5x5.4', '.branch5x5.7']  # :194-198
            for bn_index in honey_bn_index:
                all_honey_bn_name.append(name + bn_index)
            last_select_index = cur_last_select_index[:]  # :203
            cur_last_select_index = []

            for weight_index in honey_channel_index:  # :206
                branch_name = branch_of(weight_index)
                conv_name = name + weight_index + '.weight'
                all_honey_conv_name.append(name + weight_index)
                oriweight = oristate_dict[conv_name]
                curweight = state_dict[conv_name]
                orifilter_num = oriweight.size(1)  # :224 (input channels)
                currentfilter_num = curweight.size(1)
                if orifilter_num != currentfilter_num:
                    select_index = last_select_index
                else:
                    select_index = list(range(0, orifilter_num))
                for i in range(state_dict[conv_name].size(0)):  # :230
                    for index_j, j in enumerate(select_index):
                        state_dict[conv_name][i][index_j] = \
                            oristate_dict[conv_name][i][j]
                if branch_name == '_n1x1':  # :235
                    tmp_select_index = list(range(state_dict[conv_name].size(0)))
                    cur_last_select_index += tmp_select_index
                if branch_name == '_pool_planes':  # :238
                    tmp_select_index = list(range(state_dict[conv_name].size(0)))
                    tmp_select_index = [x + filters[cov_id - 2][0] + filters[cov_id - 2][1] + filters[cov_id - 2][2]
                                        for x in tmp_select_index]
                    cur_last_select_index += tmp_select_index

            for weight_index in honey_filter_index:  # :244
                branch_name = branch_of(weight_index)
                conv_name = name + weight_index + '.weight'
                all_honey_conv_name.append(name + weight_index)
                oriweight = oristate_dict[conv_name]
                curweight = state_dict[conv_name]
                orifilter_num = oriweight.size(0)
                currentfilter_num = curweight.size(0)
                if orifilter_num != currentfilter_num:  # :263
                    select_index = np.argsort(imp['imp_conv' + str(cov_id) + branch_name])[orifilter_num - currentfilter_num:]
                    select_index.sort()
                else:
                    select_index = list(range(0, orifilter_num))
                for index_i, i in enumerate(select_index):  # :271
                    state_dict[conv_name][index_i] = \
                        oristate_dict[conv_name][i]
                if branch_name == '_n3x3':  # :275
                    tmp_select_index = [x + filters[cov_id - 2][0] for x in select_index]
                    cur_last_select_index += tmp_select_index
                if branch_name == '_n5x5':  # :278
                    last_select_index = select_index

            for weight_index in honey_filter_channel_index:  # :281
                branch_name = branch_of(weight_index)
                conv_name = name + weight_index + '.weight'
                all_honey_conv_name.append(name + weight_index)
                oriweight = oristate_dict[conv_name]
                curweight = state_dict[conv_name]
                orifilter_num = oriweight.size(1)  # :298
                currentfilter_num = curweight.size(1)
                if orifilter_num != currentfilter_num:
                    select_index = last_select_index
                else:
                    select_index = range(0, orifilter_num)
                orifilter_num = oriweight.size(0)  # :306
                currentfilter_num = curweight.size(0)
                select_index_1 = copy.deepcopy(select_index)  # :309
                if orifilter_num != currentfilter_num:  # :311
                    select_index = np.argsort(imp['imp_conv' + str(cov_id) + branch_name])[orifilter_num - currentfilter_num:]
                    select_index.sort()
                else:
                    select_index = list(range(0, orifilter_num))
                if branch_name == '_n5x5':  # :320
                    tmp_select_index = [x + filters[cov_id - 2][0] + filters[cov_id - 2][1] for x in select_index]
                    cur_last_select_index += tmp_select_index
                for index_i, i in enumerate(select_index):  # :324
                    for index_j, j in enumerate(select_index_1):
                        state_dict[conv_name][index_i][index_j] = \
                            oristate_dict[conv_name][i][j]

        elif name == 'pre_layers':  # :329
            cnt += 1
            cov_id = cnt
            honey_filter_index = ['.0']
            honey_bn_index = ['.1']
            for bn_index in honey_bn_index:
                all_honey_bn_name.append(name + bn_index)
            for weight_index in honey_filter_index:
                conv_name = name + weight_index + '.weight'
                all_honey_conv_name.append(name + weight_index)
                oriweight = oristate_dict[conv_name]
                curweight = state_dict[conv_name]
                orifilter_num = oriweight.size(0)
                currentfilter_num = curweight.size(0)
                if orifilter_num != currentfilter_num:  # :350
                    select_index = np.argsort(imp['imp_conv' + str(cov_id)])[orifilter_num - currentfilter_num:]
                    select_index.sort()
                    cur_last_select_index = select_index[:]
                    for index_i, i in enumerate(select_index):
                        state_dict[conv_name][index_i] = \
                            oristate_dict[conv_name][i]

    for name, kind in modules:  # :361 "Reassign non sketch weights to the new network"
        if kind == 'conv':
            if name not in all_honey_conv_name:
                state_dict[name + '.weight'] = oristate_dict[name + '.weight']
                state_dict[name + '.bias'] = oristate_dict[name + '.bias']
        elif kind == 'bn':  # :369
            if name not in all_honey_bn_name:
                state_dict[name + '.weight'] = oristate_dict[name + '.weight']
                state_dict[name + '.bias'] = oristate_dict[name + '.bias']
                state_dict[name + '.running_mean'] = oristate_dict[name + '.running_mean']
                state_dict[name + '.running_var'] = oristate_dict[name + '.running_var']
        elif kind == 'linear':  # :377
            state_dict[name + '.weight'] = oristate_dict[name + '.weight']
            state_dict[name + '.bias'] = oristate_dict[name + '.bias']
    return state_dict


def load_u2netp_model(state_dict, oristate_dict, imp, conv_names):
    """utils/load_models.py:585-769, element by element. `conv_names`: the Conv2d modules in named_modules()
    order (the loop stops at 'outconv', :598-599); `imp`: {file stem without '.npy': array} for
    args.imp_score + '/net.stage<k>[d].<unit>.relu_s1.npy' and '/net.side<k>.npy'.

    The reference spells the same two scalar-slice loops out nine times; here they are the helpers `rows_cols`
    (dst[ri][cj + col0] = src[r][c + src_off]) and `whole_rows`, called from the same branch structure. The
    reference's own failure modes are kept: list(None) when a stage starts while no index is live (:617, :621),
    int('i') for an un-pruned decoder rebnconvin behind a pruned layer (:658), indexing a None entry of
    save_select_index."""
    last = None                      # last_select_index, :587
    side_cnt, stage_id = 0, 1        # cnt, stage_id, :589-590
    in_unit, per_stage, per_side = [], [], []   # save_select_index, save_stage_select_index, save_side_select_index

    def rank(key):
        return imp[key]

    def kept(key, o, c):             # :629-632 and its eight repeats
        idx = np.argsort(rank(key))[o - c:]
        idx.sort()
        return idx

    def rows_cols(dst, src, row_pairs, cols, col0=0, src_off=0):
        for ri, r in row_pairs:
            for cj, cc in enumerate(cols):
                dst[ri][cj + col0] = src[r][cc + src_off]

    def whole_rows(dst, src, row_pairs):
        for ri, r in row_pairs:
            dst[ri] = src[r]

    for name in conv_names:
        if name == 'outconv':        # :598-599
            break
        top = name.split('.')[0]     # side_name, :602
        decode = top[-1] == 'd'      # :603
        is_side = top[:4] == 'side'
        unit = None if is_side else name.split('.')[1]   # midfix, :607
        tag = None if is_side else unit[-2:]             # cov_id, :615
        key = name + '.weight'
        src, dst = oristate_dict[key], state_dict[key]
        o, c, half = src.size(0), dst.size(0), int(src.size(1) / 2)   # :611-613, int(oriin_num/2)
        if decode and top[-2] != str(stage_id):          # :616-619
            stage_id -= 1
            per_side.append(list(last))
            in_unit = []
        elif (not decode) and top[-1] != str(stage_id):  # :620-623
            stage_id += 1
            per_stage.append(list(last))
            in_unit = []
        score_key = None if is_side else 'net.stage%d%s%s.relu_s1' % (stage_id, 'd.' if decode else '.', unit)

        if not is_side:
            # which second input half the conv has: the encoder stage's output (decoder rebnconvin, :627),
            # none (rebnconvin of an encoder stage and rebnconv<k>, :667), the unit's own rebnconv<k> (rebnconv<k>d, :706)
            kind = 'stage' if (decode and tag == 'in') else ('none' if tag[1] != 'd' else 'unit')
            if o != c:
                sel = kept(score_key, o, c)
                pairs = list(enumerate(sel))
                if kind == 'none':
                    if last is not None:                 # :674-678
                        rows_cols(dst, src, pairs, last)
                    else:                                # :679-682
                        whole_rows(dst, src, pairs)
                else:                                    # :634-640, :713-719
                    rows_cols(dst, src, pairs, last)
                    other = per_stage[stage_id - 1] if kind == 'stage' else in_unit[int(tag[0])]
                    rows_cols(dst, src, pairs, other, col0=len(last), src_off=half)
                last = sel
                if kind != 'unit':
                    in_unit.append(list(sel))            # :643, :685
            elif last is not None:
                sel = kept(score_key, o, c)              # read although nothing is pruned here (:646-649, :688-691, :724-727)
                pairs = [(r, r) for r in range(o)]
                rows_cols(dst, src, pairs, last)
                if kind != 'none':
                    other = in_unit[int(tag[0])]         # :658 uses cov_id[0] for 'in' as well -> ValueError
                    rows_cols(dst, src, pairs, other, col0=len(last), src_off=half)
                last = sel
                if kind != 'unit':
                    in_unit.append(list(sel))
            else:                                        # :662-665, :701-704, :738-740
                state_dict[key] = src
                last = None
                if kind != 'unit':
                    in_unit.append(None)
        else:                                            # :742-767
            side_cnt += 1
            if o != c:
                sel = kept('net.side%d' % side_cnt, o, c)
                pairs = list(enumerate(sel))
                if last is not None:
                    rows_cols(dst, src, pairs, last)
                else:
                    whole_rows(dst, src, pairs)
            elif last is not None:
                rows_cols(dst, src, [(r, r) for r in range(o)], last)
            else:
                state_dict[key] = src
            last = per_side[5 - side_cnt]                # :767
    return state_dict
