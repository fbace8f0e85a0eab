"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's weight-transplant loops, element by
element as the reference writes them, for parity tests of dct_pruning_amd/transplant.py.
Only tests/ may import this module.

Restates utils/load_models.py:17-64 (load_vgg_model) and :441-582 (load_resnet_imagenet_model,
resnet_50 branch). The reference iterates model.named_modules(); here the ordered list of conv
module names is passed in (the state-dict keys are the same), and score files come from a dict
{stem: array} instead of np.load(args.imp_score + '/imp_conv%d.npy'). Everything else - the branch
structure, which tensors are copied scalar slice by scalar slice, which are rebound, when
last_select_index is kept or reset - follows the reference line by line.

Pinned by: nothing in the reference (it has no tests); the restatement is checked by review against
the cited lines. "parity unpinned" in the sense of the task statement."""
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
